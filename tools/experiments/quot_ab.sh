#!/bin/bash
# quotient kernel time (one FinalExp proof in flight, HIP events) and proof digest with the default library and a variant, alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
V=$1; PAIRS=${2:-3}
for k in $(seq $PAIRS); do
  for lib in "" "$V"; do
    if [ -z "$lib" ]; then unset STARKHIP_LIBRARY; name=default; else export STARKHIP_LIBRARY=$R/$lib; name=variant; fi
    python3 - <<PY
import sys, os, hashlib
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import numpy as np, starky_bls12_381_amd as S
from bls_util import random_fp12
pv = S.Prover(0)
air = S.AIR_FINAL_EXP; cfg = S.StarkConfig.for_air(air)
compact, pis = S.trace_final_exp(random_fp12(0x5EED0001), compact=True)
ts = []
for rep in range(4):
    proof = pv.prove(air, cfg, compact, pis)
    if rep: ts.append(round(float(pv.last_kernel_timings()[2] if not isinstance(pv.last_kernel_timings(), dict) else pv.last_kernel_timings()["quotient_eval"]), 2))
print("$name", "quotient ms", ts, hashlib.sha256(proof.tobytes()).hexdigest()[:16])
PY
  done
done
