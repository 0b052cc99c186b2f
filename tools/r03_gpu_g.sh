set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
OPS=tests/golden/signature_operands_8.bin
timeout -k 10 900 python -m pytest tests/test_gpu_pool.py tests/test_gpu_airs.py -x -q > $O/r03_g_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03_g_pytest.log; tail -3 $O/r03_g_pytest.log
for cfg in "1 16 3 0" "1 16 3 1" "1 16 4 0" "1 16 4 1" "2 16 3 1"; do
  set -- $cfg
  timeout -k 10 300 build/signature_demo --batch 8 --operands $OPS --steps 3 --warmup 1 --policy $1 --small $2 --big $3 --priority $4 --timeline > $O/r03_g_demo_batch8_p$1_s$2_b$3_pr$4.json 2> $O/r03_g_demo_batch8_p$1_s$2_b$3_pr$4.err; echo "demo8 p$1 s$2 b$3 prio$4 rc=$?"; cut -c140-330 $O/r03_g_demo_batch8_p$1_s$2_b$3_pr$4.json
done
for cfg in "1 0" "1 1" "2 1" "1 2"; do
  set -- $cfg
  timeout -k 10 200 build/signature_demo --batch 1 --steps 6 --warmup 2 --policy $1 --priority $2 --timeline > $O/r03_g_demo_batch1_p$1_pr$2.json 2> $O/r03_g_demo_batch1_p$1_pr$2.err; echo "demo1 p$1 prio$2 rc=$?"; cut -c140-330 $O/r03_g_demo_batch1_p$1_pr$2.json
done
