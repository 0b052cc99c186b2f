#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
for g in 3 5 4; do
  STARKHIP_POOL_LANE_GROUP=$g python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-boundary > $OUT/lane_group_$g.json 2> $OUT/lane.err || true
  python3 -c "
import json;d=json.loads(open('$OUT/lane_group_$g.json').read().strip().splitlines()[-1]);print('group',$g,'inflight',d['config']['proofs_in_flight_per_gpu'],round(d['value'],3),round(d['ms_per_step'],1),d['timed_proofs_verified'],d['oracle_digest_match'])"
done
