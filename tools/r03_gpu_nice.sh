#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
OPS=tests/golden/signature_operands_8.bin
for rep in 1 2 3; do
  for nice in 0 10; do
    STARKHIP_GEN_NICE=$nice build/signature_demo --batch 8 --operands $OPS --steps 6 --warmup 1 > $OUT/nice${nice}_rep${rep}.json 2> /dev/null
    python3 -c "
import json;d=json.load(open('$OUT/nice${nice}_rep${rep}.json'));print('nice',$nice,'rep',$rep,d['value'],d['best_ms'],[round(x) for x in d['step_ms']])"
  done
done
for nice in 0 10; do
STARKHIP_GEN_NICE=$nice build/signature_demo --batch 1 --steps 10 --warmup 2 > $OUT/nice${nice}_batch1.json 2> /dev/null
python3 -c "
import json;d=json.load(open('$OUT/nice${nice}_batch1.json'));print('batch1 nice',$nice,d['value'],d['best_ms'],[round(x) for x in d['step_ms']])"
done
STARKHIP_GEN_NICE=10 build/signature_demo --batch 8 --operands $OPS --steps 3 --warmup 1 --timeline > $OUT/nice10_tl.json 2> $OUT/nice10_timeline.txt
grep final_exp $OUT/nice10_timeline.txt | cut -c1-140
