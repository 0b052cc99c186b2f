#!/bin/bash
# bench.py value and host CPU-seconds per proof under affinity masks (plain taskset; one GPU)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cpus=$(python3 -c "import os; print(','.join(str(c) for c in sorted(os.sched_getaffinity(0))))")
for n in ${CPUS:-16 8 4 2}; do
  mask=$(echo $cpus | cut -d, -f1-$n)
  taskset -c $mask python3 bench.py --steps 24 --warmup 1 --no-cpu-baseline --no-boundary --no-solo > gpurun_out/cpu_$n.json 2> gpurun_out/cpu_$n.err || { tail -3 gpurun_out/cpu_$n.err; continue; }
  python3 - $n gpurun_out/cpu_$n.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
h = d["host"]
print("cpus", sys.argv[1], "granted", h["cpus_granted"], "value", round(d["value"], 2), "cpu-s/proof", round(h["cpu_seconds_per_proof"], 3), {k: round(v, 3) for k, v in h["cpu_seconds_per_proof_by_role"].items()},
      "gen threads", h["pools"][0]["generator_threads"], "rec threads", h["pools"][0]["trace_threads_big"], "gen ms", round(d["generate_trace_ms_timed_region"], 1))
PY
done
