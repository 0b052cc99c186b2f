set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
OPS=tests/golden/signature_operands_8.bin
for cfg in "0 16" "1 16" "2 16" "2 24"; do
  set -- $cfg
  timeout -k 10 300 build/signature_demo --batch 8 --operands $OPS --steps 3 --warmup 1 --policy $1 --small $2 --timeline > $O/r03_f_demo_batch8_p$1_s$2.json 2> $O/r03_f_demo_batch8_p$1_s$2.err; echo "demo8 p$1 s$2 rc=$?"; cut -c140-330 $O/r03_f_demo_batch8_p$1_s$2.json
done
for pol in 0 1 2; do
  timeout -k 10 200 build/signature_demo --batch 1 --steps 5 --warmup 2 --policy $pol --timeline > $O/r03_f_demo_batch1_p$pol.json 2> $O/r03_f_demo_batch1_p$pol.err; echo "demo1 p$pol rc=$?"; cut -c140-330 $O/r03_f_demo_batch1_p$pol.json
done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-boundary > $O/r03_f_bench.json 2> $O/r03_f_bench.err; echo "bench rc=$?"; python -c "
import json;d=json.load(open('$O/r03_f_bench.json'));print(d['value'],d['ms_per_step'],d['latency_ms_one_in_flight'],{k:round(v['avg_ms'],2) for k,v in d['kernels'].items()}, d['timed_proofs_verified'], d['oracle_digest_match'])"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-boundary --inflight 3 > $O/r03_f_bench_if3.json 2> $O/r03_f_bench_if3.err; echo "bench rc=$?"; python -c "
import json;d=json.load(open('$O/r03_f_bench_if3.json'));print(d['value'],d['ms_per_step'])"
