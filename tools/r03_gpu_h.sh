set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
OPS=tests/golden/signature_operands_8.bin
for cfg in "6 0 0" "4 0 0" "3 0 0" "4 8 0" "4 0 2" "6 16 0"; do
  set -- $cfg
  timeout -k 10 200 build/signature_demo --batch 1 --steps 10 --warmup 2 --gen $1 --trace-threads $2 --priority $3 --timeline > $O/r03_h_demo1_g$1_t$2_pr$3.json 2> $O/r03_h_demo1_g$1_t$2_pr$3.err; echo "demo1 gen$1 tt$2 prio$3 rc=$?"; cut -c140-400 $O/r03_h_demo1_g$1_t$2_pr$3.json
done
for cfg in "4 16 0" "4 16 1" "3 16 0" "4 20 1" "4 12 1"; do
  set -- $cfg
  timeout -k 10 300 build/signature_demo --batch 8 --operands $OPS --steps 4 --warmup 1 --big $1 --small $2 --priority $3 > $O/r03_h_demo8_b$1_s$2_pr$3.json 2> /dev/null; echo "demo8 b$1 s$2 prio$3 rc=$?"; cut -c140-400 $O/r03_h_demo8_b$1_s$2_pr$3.json
done
