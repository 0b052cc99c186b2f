#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tee $OUT/final_gpu_tests.txt | tail -4
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/final_bench_driver_cfg.json 2> $OUT/final_bench_driver_cfg.err
python3 -c "
import json;d=json.loads(open('$OUT/final_bench_driver_cfg.json').read().strip().splitlines()[-1]);print('bench',round(d['value'],3),round(d['ms_per_step'],1),d['timed_proofs_verified'],d['oracle_digest_match'],round(d['latency_ms_one_in_flight'],1),d['roofline']['traffic'],d['value_host_boundary']['value'],d['value_compact']['value'])"
