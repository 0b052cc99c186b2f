# default build (kernels beside the lane hash at a raised issue priority) against build/noprio (make variant NAME=noprio DEFS=-DSTARKHIP_NO_PRIO):
# bench.py from operands, and the signature path from compiled host code
cd ${GRAFT_REPO_ROOT:-.}
NP=STARKHIP_LIBRARY=$PWD/build/noprio/libstarkhip_noprio.so
STEPS=48 bash tools/gpu_ab.sh prio_ab "" "$NP" "" "$NP" || exit 1
PL=LD_PRELOAD=$PWD/build/noprio/libstarkhip_noprio.so   # the demo is linked against the default library: the preloaded one's symbols win
for v in "X=1" "$PL" "X=1" "$PL"; do
  echo "== $v"
  env $v build/signature_demo --batch 8 --operands tests/golden/signature_operands_8.bin --steps 4 --warmup 1 2>/dev/null | cut -c150-260
  env $v build/signature_demo --batch 1 --steps 8 --warmup 2 2>/dev/null | cut -c150-260
done
