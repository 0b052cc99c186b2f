#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
OPS=tests/golden/signature_operands_8.bin
for cfg in "--big 4 --small 16" "--big 6 --small 16" "--big 4 --small 12" "--big 4 --small 16 --gen 6" "--big 4 --small 16"; do
  build/signature_demo --batch 8 --operands $OPS --steps 6 --warmup 1 $cfg > $OUT/sweep.json 2> /dev/null
  python3 -c "
import json;d=json.load(open('$OUT/sweep.json'));print('$cfg',d['value'],d['best_ms'],[round(x) for x in d['step_ms']])"
done
