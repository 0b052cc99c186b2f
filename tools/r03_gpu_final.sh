#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
( while true; do rocm-smi --showmeminfo vram 2>/dev/null | grep "Used" | awk '{print $NF}' >> $OUT/vram_used.txt; sleep 2; done ) &
MON=$!
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/final_bench_driver_cfg.json 2> $OUT/final_bench_driver_cfg.err
python3 bench.py > $OUT/final_bench_default.json 2> $OUT/final_bench_default.err
kill $MON
python3 -c "
import json
for f in ('final_bench_driver_cfg','final_bench_default'):
    d=json.loads(open('$OUT/'+f+'.json').read().strip().splitlines()[-1]);print(f,round(d['value'],3),round(d['ms_per_step'],1),d['config']['proofs_in_flight_per_gpu'],d['timed_proofs_verified'],d['oracle_digest_match'],round(d['latency_ms_one_in_flight'],1),d['value_host_boundary']['value'],d['value_compact']['value'],d['timed_region_commitments'].get('ms_per_commitment_four_side_by_side'))
print('peak vram GB', max(int(x) for x in open('$OUT/vram_used.txt').read().split())/1e9)
"
tail -2 $OUT/final_bench_driver_cfg.err
