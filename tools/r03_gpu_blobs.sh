#!/bin/bash
# page-locked recycled proof blobs: tests, then A/B of single-proof latencies and the bench line (STARKHIP_PINNED_PROOFS=0 is the old path)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_pool.py -x -q -m gpu -v 2>&1 | tee $OUT/y_pool_tests.txt | tail -12
for pin in 0 1; do
  STARKHIP_PINNED_PROOFS=$pin python3 tools/air_latency.py > $OUT/y_air_latency_pin$pin.json 2> $OUT/y_air_latency_pin$pin.err
  STARKHIP_PINNED_PROOFS=$pin python3 bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-boundary > $OUT/y_bench_pin$pin.json 2> $OUT/y_bench_pin$pin.err
done
python3 - <<'PY'
import json
for pin in (0, 1):
    a = json.load(open("gpurun_out/y_air_latency_pin%d.json" % pin))
    print("pin", pin, {k: (round(v["wall_ms"], 1), round(v["wall_with_numpy_copy_ms"], 1), round(v["phase_ms"]["queries"], 2), round(v["host_ms"]["fiat_shamir"], 1)) for k, v in a.items()})
    b = json.loads(open("gpurun_out/y_bench_pin%d.json" % pin).read().strip().splitlines()[-1])
    print("   bench", round(b["value"], 3), round(b["latency_ms_one_in_flight"], 1), round(b["phase_ms_one_in_flight"]["queries"], 2), b["host_ms_one_in_flight"]["fiat_shamir"])
PY
