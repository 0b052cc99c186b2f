#!/bin/bash
# the untimed hand-over legs of bench.py (host rows / recorded trace / device-resident) with the default library and a variant, alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
V=$1; PAIRS=${2:-2}
for k in $(seq $PAIRS); do
  for lib in "" "$V"; do
    if [ -z "$lib" ]; then unset STARKHIP_LIBRARY; name=default; else export STARKHIP_LIBRARY=$R/$lib; name=variant; fi
    python3 bench.py --steps 16 --warmup 1 --no-cpu-baseline --no-solo 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', 'value', round(d['value'],3), {k: round(d[k]['value'],2) for k in ('value_host_rows','value_compact','value_device_resident') if k in d and d[k].get('value')})"
  done
done
