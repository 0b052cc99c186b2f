#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "both_leaf_hash_forms and 3" 2>&1 | tail -2
for lane in 1 0; do
  for inf in 4 8; do
    STARKHIP_POOL_BIG_LANE=$lane python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-boundary --inflight $inf > $OUT/lane${lane}_inf${inf}.json 2> $OUT/lane.err || true
    python3 -c "
import json;d=json.loads(open('$OUT/lane${lane}_inf${inf}.json').read().strip().splitlines()[-1]);print('lane',$lane,'inflight',$inf,round(d['value'],3),round(d['ms_per_step'],1),d['timed_proofs_verified'],d['oracle_digest_match'], round(d['kernels']['leaf_hash']['avg_ms'],1))"
  done
done
