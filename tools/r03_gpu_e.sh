set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
nproc; grep -m1 "model name" /proc/cpuinfo
timeout -k 10 900 python -m pytest tests/test_gpu_pool.py -x -q > $O/r03_e_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03_e_pytest.log
tail -3 $O/r03_e_pytest.log
OPS=tests/golden/signature_operands_8.bin
for cfg in "0 16" "2 16" "0 40" "1 40" "2 40"; do
  set -- $cfg
  timeout -k 10 300 build/signature_demo --batch 8 --operands $OPS --steps 3 --warmup 1 --policy $1 --small $2 --timeline > $O/r03_e_demo_batch8_p$1_s$2.json 2> $O/r03_e_demo_batch8_p$1_s$2.err; echo "demo8 p$1 s$2 rc=$?"; cut -c140-330 $O/r03_e_demo_batch8_p$1_s$2.json
done
for pol in 0 1 2; do
  timeout -k 10 200 build/signature_demo --batch 1 --steps 5 --warmup 2 --policy $pol --timeline > $O/r03_e_demo_batch1_p$pol.json 2> $O/r03_e_demo_batch1_p$pol.err; echo "demo1 p$pol rc=$?"; cut -c140-330 $O/r03_e_demo_batch1_p$pol.json
done
