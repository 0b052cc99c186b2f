#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
for inf in 6 7 8 4 6; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-boundary --inflight $inf > $OUT/laneauto_inf$inf.json 2> $OUT/lane.err || true
  python3 -c "
import json;d=json.loads(open('$OUT/laneauto_inf$inf.json').read().strip().splitlines()[-1]);print('steps 20 warmup 5 inflight',$inf,round(d['value'],3),round(d['ms_per_step'],1),d['timed_proofs_verified'],d['oracle_digest_match'])"
done
