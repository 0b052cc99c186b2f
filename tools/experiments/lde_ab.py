#!/usr/bin/env python3
"""A/B of the two LDE kernels on the FinalExp trace, one proof in flight (development aid; GPU): alternating proofs with
"lde_impl" 0 (wave-resident) and 1 (lde_columns_v2_kernel), HIP-event durations of the LDE launches (starkhip_last_kernel_timings)."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import starky_bls12_381_amd as S
from bls_util import random_fp12

pv = S.Prover(0)
air = S.AIR_FINAL_EXP
cfg = S.StarkConfig.for_air(air)
compact, pis = S.trace_final_exp(random_fp12(0x5EED0001), compact=True)
ref = None
res = {0: [], 1: []}
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    for impl in (0, 1):
        pv.set_option("lde_impl", impl)
        proof = pv.prove(air, cfg, compact, pis)
        if ref is None:
            ref = proof
        assert np.array_equal(proof, ref)
        if rep:
            res[impl].append(pv.last_kernel_timings()["lde_columns"] if isinstance(pv.last_kernel_timings(), dict) else pv.last_kernel_timings()[0])
for impl in (0, 1):
    print("lde_impl", impl, "LDE ms:", [round(float(x), 2) for x in res[impl]])
import hashlib
print("proof sha256", hashlib.sha256(ref.tobytes()).hexdigest())
print("golden      ", open(os.path.join(ROOT, "tests", "golden", "final_exp_seed_5eed0001_proof.sha256")).read().split()[0])
