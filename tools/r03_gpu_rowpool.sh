#!/bin/bash
# one signature / a batch of 8 with the pool's small commitments in the row form up to a leaf count (0 = never, the default)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
OPS=tests/golden/signature_operands_8.bin
for rl in 0 64 4096 2048 0; do
  STARKHIP_POOL_ROW_LEAVES=$rl build/signature_demo --batch 1 --steps 12 --warmup 2 > $OUT/rowpool.json 2> /dev/null
  python3 -c "
import json;d=json.load(open('$OUT/rowpool.json'));print('batch1 row_leaves',$rl,round(d['ms_per_step'],1),d['best_ms'],[round(x) for x in d['step_ms']])"
done
for rl in 0 64 4096; do
  STARKHIP_POOL_ROW_LEAVES=$rl build/signature_demo --batch 8 --operands $OPS --steps 5 --warmup 1 > $OUT/rowpool.json 2> /dev/null
  python3 -c "
import json;d=json.load(open('$OUT/rowpool.json'));print('batch8 row_leaves',$rl,d['value'],d['best_ms'],[round(x) for x in d['step_ms']])"
done
