#!/bin/bash
# SQ counters of the leaf-hash kernels a lone FinalExp commitment can take (tools/experiments/pair_check.py proves with the quad and the
# pair form in turn).  Run on the GPU box from the repo root; CSV rows land in gpurun_out/${TAG}_hash_sq{1,2}.csv
TAG=${1:-hash}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/tools/experiments/pair_check.py"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $OUT/pmc1_$TAG -o p -- $CMD > $OUT/pmc1_$TAG.log 2>&1 || { tail -5 $OUT/pmc1_$TAG.log; exit 1; }
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC -d $OUT/pmc2_$TAG -o p -- $CMD > $OUT/pmc2_$TAG.log 2>&1 || { tail -5 $OUT/pmc2_$TAG.log; exit 1; }
cd $R
for k in 1 2; do
  db=$(find $OUT/pmc${k}_$TAG -name "*results.db" | head -1)
  [ -n "$db" ] && python3 tools/rocprof_export.py pmc $db $OUT/${TAG}_hash_sq$k.csv
  rm -rf $OUT/pmc${k}_$TAG
done
head -1 $OUT/${TAG}_hash_sq1.csv
grep -hE "leaf_hash_(pair_)?kernel" $OUT/${TAG}_hash_sq1.csv $OUT/${TAG}_hash_sq2.csv
