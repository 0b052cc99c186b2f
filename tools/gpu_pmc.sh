#!/bin/bash
# SQ counters of the heavy kernels with one proof in flight (two passes of eight counters); prints the rows of the kernels named in $2 (regex)
TAG=${1:-pmc}
PAT=${2:-quotient_tiles|lde_columns_v2|leaf_hash_kernel}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ONE="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-boundary --inflight 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $OUT/pmc1_$TAG -o p -- $ONE > $OUT/pmc1_$TAG.log 2>&1 || { tail -5 $OUT/pmc1_$TAG.log; exit 1; }
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC -d $OUT/pmc2_$TAG -o p -- $ONE > $OUT/pmc2_$TAG.log 2>&1 || { tail -5 $OUT/pmc2_$TAG.log; exit 1; }
cd $R
python3 tools/rocprof_export.py pmc $(find $OUT/pmc1_$TAG -name "*results.db" | head -1) $OUT/${TAG}_sq1.csv
python3 tools/rocprof_export.py pmc $(find $OUT/pmc2_$TAG -name "*results.db" | head -1) $OUT/${TAG}_sq2.csv
rm -rf $OUT/pmc1_$TAG $OUT/pmc2_$TAG
grep -E "$PAT" $OUT/${TAG}_sq1.csv $OUT/${TAG}_sq2.csv | grep -v ",4096,"
