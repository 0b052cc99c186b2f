set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
bash tools/profile_round.sh r03_k sig 2>&1 | tail -8
for inf in 3 4 3 4 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-boundary --inflight $inf --steps 12 > $O/r03_l_bench_if$inf.json 2> /dev/null; python -c "
import json;d=json.load(open('$O/r03_l_bench_if$inf.json'));print('inflight $inf', d['value'],d['ms_per_step'])"
done
STARKHIP_BENCH_REHEARSE=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 1 --inflight 2 > $O/r03_l_bench_rehearse2.json 2> $O/r03_l_bench_rehearse2.err; echo "rehearse rc=$?"; cut -c1-400 $O/r03_l_bench_rehearse2.json; tail -3 $O/r03_l_bench_rehearse2.err
