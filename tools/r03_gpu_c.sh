set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_pool.py tests/test_gpu_rccl.py "tests/test_gpu_airs.py::test_handoff_round_trip_of_gpu_proofs" -x -q > $O/r03_c_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03_c_pytest.log
tail -5 $O/r03_c_pytest.log
OPS=tests/golden/signature_operands_8.bin
for pol in 0 1 2; do
  timeout -k 10 300 build/signature_demo --batch 8 --operands $OPS --steps 3 --warmup 1 --policy $pol --timeline > $O/r03_c_demo_batch8_p$pol.json 2> $O/r03_c_demo_batch8_p$pol.err; echo "demo8 p$pol rc=$?"; cut -c140-420 $O/r03_c_demo_batch8_p$pol.json
  timeout -k 10 200 build/signature_demo --batch 1 --steps 4 --warmup 1 --policy $pol --timeline > $O/r03_c_demo_batch1_p$pol.json 2> $O/r03_c_demo_batch1_p$pol.err; echo "demo1 p$pol rc=$?"; cut -c140-420 $O/r03_c_demo_batch1_p$pol.json
done
GPU_MAX_HW_QUEUES=16 timeout -k 10 300 build/signature_demo --batch 8 --operands $OPS --steps 3 --warmup 1 --policy 0 > $O/r03_c_demo_batch8_p0_env.json 2>/dev/null; echo "demo8 p0 env rc=$?"; cut -c140-420 $O/r03_c_demo_batch8_p0_env.json
timeout -k 10 300 build/signature_demo --batch 8 --operands $OPS --steps 3 --warmup 1 --policy 0 --small 40 > $O/r03_c_demo_batch8_p0_s40.json 2>/dev/null; echo "demo8 p0 small40 rc=$?"; cut -c140-420 $O/r03_c_demo_batch8_p0_s40.json
