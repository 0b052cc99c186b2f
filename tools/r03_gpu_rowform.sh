#!/bin/bash
# row form of the leaf hash: parity tests, then single-proof latencies with it (auto) and without (leaf_hash_form = 1 via env)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "merkle_cap" 2>&1 | tee $OUT/z_rowform_tests.txt | tail -5
timeout -k 10 600 python3 tools/air_latency.py > $OUT/z_air_latency_row.json 2> $OUT/z_air_latency_row.err
python3 tools/host_perm_rate.py > $OUT/z_host_perm_rate.txt 2>&1 || true
python3 - <<'PY'
import json
a = json.load(open("gpurun_out/z_air_latency_row.json"))
print({k: (round(v["wall_ms"], 1), round(v["phase_ms"]["trace_merkle"], 1), round(v["host_ms"]["fiat_shamir"], 1)) for k, v in a.items()})
print(open("gpurun_out/z_host_perm_rate.txt").read()[:300])
PY
