#!/bin/bash
# Issue-side counters of the quotient kernel (round 6): which instruction classes are active, what the LDS pipe does, branches,
# instruction fetch.  Four PMC passes over tools/quotient_ab.py (FinalExp, tiled evaluator), each in a run of its own with no tracing.
# Run on the GPU box from the repo root: bash tools/quotient_pmc.sh TAG  ->  gpurun_out/TAG_quotient_pmc.csv
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/quotient_ab.py --chunks 0 --reps 2"
pass() {  # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" -d $OUT/prof_$name -o $name -- $CMD > $OUT/${TAG}_$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/${TAG}_$name.log; return 0; }
  python3 $R/tools/rocprof_export.py pmc $(find $OUT/prof_$name -name "*results.db" | head -1) $OUT/${TAG}_pmc_$name.csv
  rm -rf $OUT/prof_$name
}
pass a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
pass b SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass c SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM
pass d SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_SMEM SQ_LEVEL_WAVES SQ_CYCLES
cd $R
head -1 $OUT/${TAG}_pmc_a.csv > $OUT/${TAG}_quotient_pmc.csv
for p in a b c d; do [ -f $OUT/${TAG}_pmc_$p.csv ] && grep "quotient_tiles" $OUT/${TAG}_pmc_$p.csv >> $OUT/${TAG}_quotient_pmc.csv; done
cat $OUT/${TAG}_quotient_pmc.csv
