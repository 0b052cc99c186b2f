#!/bin/bash
# batch of 8 signatures: small jobs longest-first (default) against arrival order (STARKHIP_POOL_FIFO=1), alternating
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
OPS=tests/golden/signature_operands_8.bin
for rep in 1 2 3; do
  for fifo in 1 0; do
    STARKHIP_POOL_FIFO=$fifo build/signature_demo --batch 8 --operands $OPS --steps 6 --warmup 1 > $OUT/lpt_fifo${fifo}_rep${rep}.json 2> /dev/null
    python3 -c "
import json;d=json.load(open('$OUT/lpt_fifo${fifo}_rep${rep}.json'));print('fifo',$fifo,'rep',$rep,d['value'],d['best_ms'],[round(x) for x in d['step_ms']])"
  done
done
build/signature_demo --batch 1 --steps 10 --warmup 2 > $OUT/lpt_batch1.json 2> /dev/null
python3 -c "
import json;d=json.load(open('$OUT/lpt_batch1.json'));print('batch1',d['value'],d['best_ms'],[round(x) for x in d['step_ms']])"
