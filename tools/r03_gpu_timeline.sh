#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
build/signature_demo --batch 8 --operands tests/golden/signature_operands_8.bin --steps 4 --warmup 1 --timeline "$@" > $OUT/tl_batch8.json 2> $OUT/tl_batch8_timeline.txt
python3 -c "
import json;d=json.load(open('$OUT/tl_batch8.json'));print(d['value'],d['best_ms'],[round(x) for x in d['step_ms']])"
