#!/bin/bash
# alternating bench.py runs of the default library and a variant: usage ab_lib.sh <variant.so> [pairs] [steps]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
V=$1; PAIRS=${2:-3}; STEPS=${3:-48}
for k in $(seq $PAIRS); do
  for lib in "" "$V"; do
    if [ -z "$lib" ]; then unset STARKHIP_LIBRARY; name=default; else export STARKHIP_LIBRARY=$R/$lib; name=$lib; fi
    python3 bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-boundary --no-solo 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$name', 'value', round(d['value'],3), 'steady', round(d['value_steady_state'] or 0,3), 'groups', r.get('group_sizes'), 'lane ms', round(r['avg_launch_ms'],1), 'gen ms', round(d['generate_trace_ms_timed_region'],1))"
  done
done
