#!/bin/bash
# One gpurun job: the GPU test suite, the default bench line, the two-rank rehearsal of `bench.py --gpus 2` started by bench.py itself.
# usage (from the repo root on the GPU box): bash tools/gpu_job.sh TAG [tests|bench|all]
TAG=${1:-job}
PART=${2:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
rc=0
if [ "$PART" != "bench" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest.log 2>&1 || rc=$?
  tail -15 $OUT/${TAG}_pytest.log
  [ $rc -ne 0 ] && exit $rc
fi
if [ "$PART" != "tests" ]; then
  timeout -k 10 600 python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || rc=$?
  tail -3 $OUT/${TAG}_bench.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$OUT/${TAG}_bench.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("value", round(d["value"], 3), "ms/step", round(d["ms_per_step"], 1), "inflight", d["config"]["proofs_in_flight_per_gpu"], "verified", d["timed_proofs_verified"], "oracle", d["oracle_digest_match"])
    print("roofline", r["kernel"], round(r["avg_launch_ms"], 1), "ms x", round(r["launches_side_by_side"], 2), "->", round(r["achieved"], 1), "GB/s frac", round(r["frac"], 4))
    print("legs", {k: round(d[k]["value"], 3) for k in ("value_host_rows", "value_compact", "value_device_resident") if k in d and d[k].get("value")})
    print("one in flight", d.get("latency_ms_one_in_flight"), {k: round(v["avg_ms"], 2) for k, v in d["kernels"].items()})
    print("reservation", d["config"]["pool_reservation_GB"], "gen ms", d.get("generate_trace_ms_timed_region"))
    print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
except Exception as e:
    print("bench line unreadable:", e)
PY
  [ $rc -ne 0 ] && exit $rc
  STARKHIP_BENCH_REHEARSE=1 timeout -k 10 600 python3 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_bench_rehearse2.json 2> $OUT/${TAG}_bench_rehearse2.err || rc=$?
  cut -c1-300 $OUT/${TAG}_bench_rehearse2.json
fi
exit $rc
