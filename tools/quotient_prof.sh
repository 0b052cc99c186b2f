#!/bin/bash
# PMC passes over tools/quotient_ab.py (FinalExp, both evaluators): SQ instruction / wait counters, LDS, FETCH_SIZE.
# Run on the GPU box from the repo root; CSVs land in gpurun_out/ with the given tag.
set -e
TAG=${1:-q}
CH=${2:-0}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/quotient_ab.py --chunks $CH --reps 2"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_BUSY_CYCLES -d $OUT/prof_sq -o sq -- $CMD > $OUT/${TAG}_sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVES -d $OUT/prof_sq2 -o sq2 -- $CMD > $OUT/${TAG}_sq2.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_fetch -o fetch -- $CMD > $OUT/${TAG}_fetch.log 2>&1
cd $R
db() { find $OUT/$1 -name "*results.db" | head -1; }
python3 tools/rocprof_export.py pmc $(db prof_sq) $OUT/${TAG}_pmc_sq.csv
[ -n "$(db prof_sq2)" ] && python3 tools/rocprof_export.py pmc $(db prof_sq2) $OUT/${TAG}_pmc_sq2.csv || true
python3 tools/rocprof_export.py pmc $(db prof_fetch) $OUT/${TAG}_pmc_fetch.csv
rm -rf $OUT/prof_sq $OUT/prof_sq2 $OUT/prof_fetch
grep -h "quotient" $OUT/${TAG}_pmc_sq.csv $OUT/${TAG}_pmc_sq2.csv $OUT/${TAG}_pmc_fetch.csv | grep -v "tables\|combine\|powers\|weights"
