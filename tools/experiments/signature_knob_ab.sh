# One signature alone (build/signature_demo --batch 1) with and without an experiment knob, alternating on one box:
#   bash tools/experiments/signature_knob_ab.sh KNOB VALUE_A VALUE_B      (e.g. STARKHIP_POOL_RESERVE_CUS 0 32)
cd ${GRAFT_REPO_ROOT:-.}
K=${1:-STARKHIP_POOL_RESERVE_CUS}; A=${2:-0}; B=${3:-32}
O=gpurun_out
mkdir -p $O
for v in $A $B $A $B; do
  echo "== $K=$v"
  env $K=$v timeout -k 10 200 build/signature_demo --batch 1 --steps 10 --warmup 2 --timeline > $O/sig_${K}_$v.json 2> $O/sig_${K}_${v}_timeline.txt || exit 1
  cut -c150-330 $O/sig_${K}_$v.json
done
tail -8 $O/sig_${K}_${B}_timeline.txt | cut -c1-230
