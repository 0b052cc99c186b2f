set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_pool.py tests/test_gpu_rccl.py "tests/test_gpu_airs.py::test_handoff_round_trip_of_gpu_proofs" tests/test_gpu_kernels.py -x -q > $O/r03_d_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03_d_pytest.log
tail -5 $O/r03_d_pytest.log
OPS=tests/golden/signature_operands_8.bin
for pol in 0 1 2; do
  timeout -k 10 300 build/signature_demo --batch 8 --operands $OPS --steps 3 --warmup 1 --policy $pol --timeline > $O/r03_d_demo_batch8_p$pol.json 2> $O/r03_d_demo_batch8_p$pol.err; echo "demo8 p$pol rc=$?"; cut -c140-330 $O/r03_d_demo_batch8_p$pol.json
  timeout -k 10 200 build/signature_demo --batch 1 --steps 5 --warmup 2 --policy $pol --timeline > $O/r03_d_demo_batch1_p$pol.json 2> $O/r03_d_demo_batch1_p$pol.err; echo "demo1 p$pol rc=$?"; cut -c140-330 $O/r03_d_demo_batch1_p$pol.json
done
timeout -k 10 300 build/signature_demo --batch 8 --operands $OPS --steps 3 --warmup 1 --policy 0 --small 24 --big 4 > $O/r03_d_demo_batch8_p0_s24b4.json 2>/dev/null; echo "demo8 p0 s24 b4 rc=$?"; cut -c140-330 $O/r03_d_demo_batch8_p0_s24b4.json
timeout -k 10 300 python bench.py --no-cpu-baseline --no-boundary > $O/r03_d_bench.json 2> $O/r03_d_bench.err; echo "bench rc=$?"; python -c "
import json;d=json.load(open('$O/r03_d_bench.json'));print(d['value'],d['ms_per_step'],d['latency_ms_one_in_flight'],{k:round(v['avg_ms'],2) for k,v in d['kernels'].items()}, d['timed_proofs_verified'], d['oracle_digest_match'])"
