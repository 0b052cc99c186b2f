set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_pool.py tests/test_gpu_rccl.py "tests/test_gpu_airs.py::test_handoff_round_trip_of_gpu_proofs" -x -q > $O/r03_b_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03_b_pytest.log
tail -5 $O/r03_b_pytest.log
OPS=tests/golden/signature_operands_8.bin
for pol in 0 1; do
  timeout -k 10 300 build/signature_demo --batch 8 --operands $OPS --steps 2 --warmup 1 --policy $pol > $O/r03_b_demo_batch8_p$pol.json 2> $O/r03_b_demo_batch8_p$pol.err; echo "demo8 p$pol rc=$?"; cat $O/r03_b_demo_batch8_p$pol.json
  timeout -k 10 200 build/signature_demo --batch 1 --steps 4 --warmup 1 --policy $pol > $O/r03_b_demo_batch1_p$pol.json 2> $O/r03_b_demo_batch1_p$pol.err; echo "demo1 p$pol rc=$?"; cat $O/r03_b_demo_batch1_p$pol.json
done
timeout -k 10 300 python tools/bench_signature.py --batch 8 --steps 2 > $O/r03_b_sig8_pool.json 2> $O/r03_b_sig8_pool.err; echo "sig8 pool rc=$?"; cut -c1-400 $O/r03_b_sig8_pool.json
timeout -k 10 300 python tools/bench_signature.py --batch 8 --steps 2 --driver python > $O/r03_b_sig8_python.json 2> $O/r03_b_sig8_python.err; echo "sig8 python rc=$?"; cut -c1-400 $O/r03_b_sig8_python.json
timeout -k 10 300 python bench.py --no-cpu-baseline --no-boundary > $O/r03_b_bench.json 2> $O/r03_b_bench.err; echo "bench rc=$?"; cut -c1-300 $O/r03_b_bench.json
