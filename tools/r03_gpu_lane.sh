#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_pool.py -x -q -m gpu -k "both_leaf_hash_forms and 3 or lane_form_groups" 2>&1 | tail -2
for rep in 1 2; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-boundary > $OUT/lane_waits_$rep.json 2> $OUT/lane.err || true
  python3 -c "
import json;d=json.loads(open('$OUT/lane_waits_$rep.json').read().strip().splitlines()[-1]);print('steps 20 warmup 5 inflight',d['config']['proofs_in_flight_per_gpu'],round(d['value'],3),round(d['ms_per_step'],1),d['timed_proofs_verified'],d['oracle_digest_match'])"
done
