cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -k "merkle or poseidon" 2>&1 | tail -1
STARKHIP_LEAF_HASH_W3=3 python -m pytest tests/test_gpu_kernels.py -x -q -k "merkle or poseidon" 2>&1 | tail -1
for w in 0 1; do
  STARKHIP_LEAF_HASH_W3=$w python bench.py --no-cpu-baseline --no-boundary > $O/r03_v_bench_w$w.json 2>/dev/null; python -c "
import json;d=json.load(open('$O/r03_v_bench_w$w.json'));print('W3=$w', d['value'],d['ms_per_step'],d['latency_ms_one_in_flight'],{k:round(v['avg_ms'],2) for k,v in d['kernels'].items()}, d['timed_proofs_verified'], d['oracle_digest_match'])"
done
OPS=tests/golden/signature_operands_8.bin
for w in 0 2 3 0 2 3; do
  STARKHIP_LEAF_HASH_W3=$w build/signature_demo --batch 8 --operands $OPS --steps 6 --warmup 1 --timeline > $O/r03_v_demo8_w$w.json 2> $O/r03_v_demo8_w$w.err; echo "W3=$w"; cut -c140-400 $O/r03_v_demo8_w$w.json
done
for w in 0 2 3; do
  STARKHIP_LEAF_HASH_W3=$w build/signature_demo --batch 1 --steps 8 --warmup 2 > $O/r03_v_demo1_w$w.json 2> /dev/null; echo "W3=$w"; cut -c140-400 $O/r03_v_demo1_w$w.json
done
