#!/bin/bash
# times the default build and every build/abl_*/ variant of the LDE kernel (GPU)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for lib in "" $(ls -d build/abl_* build/var_* 2>/dev/null); do
  if [ -z "$lib" ]; then name=default; unset STARKHIP_LIBRARY; else name=$lib; export STARKHIP_LIBRARY=$R/$(ls $lib/*.so | head -1); fi
  python3 - <<PY
import starky_bls12_381_amd as S
pv = S.Prover(0)
out = []
for impl in (0, 1):
    pv.set_option("lde_impl", impl)
    out.append(round(pv.lde_bench(60000, 13, 2, 4), 2))
print("$name", "wave", out[0], "v2", out[1])
PY
done
