# One signature alone (build/signature_demo --batch 1) with and without an experiment knob: bash tools/experiments/signature_knob_ab.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out
for v in 0 76000 0 76000; do
  echo "== QUAD_BIG_LDS=$v"
  STARKHIP_QUAD_BIG_LDS=$v timeout -k 10 200 build/signature_demo --batch 1 --steps 10 --warmup 2 --timeline > $O/exp14_$v.json 2> $O/exp14_${v}_timeline.txt || exit 1
  cut -c1-400 $O/exp14_$v.json
done
