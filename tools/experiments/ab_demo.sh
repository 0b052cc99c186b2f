#!/bin/bash
# alternating signature_demo --batch 8 runs with the default library and a variant (LD_LIBRARY_PATH override): usage ab_demo.sh <variant dir> [pairs]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
V=$1; PAIRS=${2:-3}
mkdir -p /tmp/vlib && cp $R/$V/*.so /tmp/vlib/libstarkhip.so
for k in $(seq $PAIRS); do
  for which in default variant; do
    if [ $which = variant ]; then export LD_LIBRARY_PATH=/tmp/vlib; else unset LD_LIBRARY_PATH; fi
    build/signature_demo --batch 8 --operands tests/golden/signature_operands_8.bin --steps 4 --warmup 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', 'sigs/s', d['value'], 'step ms', d['step_ms'], 'launches', d['commit_launches'])"
  done
done
