#!/bin/bash
# one-proof-in-flight kernel durations (uncontended) of the current build: LDE, leaf hash (quad form), quotient
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
env $1 timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --inflight 1 --no-boundary --no-cpu-baseline > gpurun_out/solo.json 2> gpurun_out/solo.err || { tail -3 gpurun_out/solo.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/solo.json").read().strip().splitlines()[-1])
print("latency", round(d["latency_ms_one_in_flight"], 1), {k: round(v["avg_ms"], 2) for k, v in d["kernels"].items()}, {k: round(v, 1) for k, v in d["phase_ms_one_in_flight"].items()})
PY
