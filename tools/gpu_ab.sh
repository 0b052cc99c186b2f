#!/bin/bash
# A/B runs of bench.py under environment knobs: one line per configuration (value, lane kernel, LDE / quotient durations in flight).
# usage: bash tools/gpu_ab.sh TAG "ENV1=.. ENV2=.." "ENV.." ...   (an empty string = the defaults)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout -k 10 300 python3 bench.py --steps ${STEPS:-24} --warmup 1 --no-cpu-baseline --no-boundary --no-solo ${BENCH_ARGS} > $OUT/${TAG}_ab$i.json 2> $OUT/${TAG}_ab$i.err || { echo "cfg [$cfg] failed"; tail -3 $OUT/${TAG}_ab$i.err; exit 1; }
  python3 - "$cfg" $OUT/${TAG}_ab$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d["roofline"]
sh = r["share_of_timed_kernel_time"]
n = d["steps"]
print("[%s] value %.3f  dominant %s %.1f ms x %.2f  per-proof kernel ms: %s  phases: lde %.0f merkle %.0f quot %.0f" % (
    sys.argv[1], d["value"], r["kernel"], r["avg_launch_ms"], r["launches_side_by_side"], {k: round(v / n, 1) for k, v in sh.items()},
    d["phase_ms_timed_region"]["ifft_lde"], d["phase_ms_timed_region"]["trace_merkle"], d["phase_ms_timed_region"]["quotient"]))
PY
done
