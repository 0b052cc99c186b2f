#!/usr/bin/env python3
"""The pair form of the leaf hash against the oracle on small shapes and against the quad form on one FinalExp commitment (same proof
bytes; kernel durations from the library's events)."""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_lib as O  # noqa: E402
import starky_bls12_381_amd as S  # noqa: E402
from bls_util import random_fp12  # noqa: E402

pv = S.Prover(0)
out = {"shapes": []}
bad = 0
TIMING = "--timing" in sys.argv   # a variant build whose digests may be wrong (ablations): the pair form's duration only
for log_N, ncols, cap_h in [] if TIMING else [(5, 8, 2), (5, 16, 2), (6, 9, 2), (7, 40, 2), (10, 13, 4), (10, 24, 4), (12, 200, 4), (15, 19, 4), (5, 3767, 4)]:
    rng = np.random.default_rng(log_N + ncols)
    mat = rng.integers(0, S.P, size=(ncols, 1 << log_N), dtype=np.uint64)
    mat[:, 0] = 0
    mat[:, -1] = np.uint64(S.P - 1)
    pv.set_option("leaf_hash_form", 4)
    cap = pv.merkle_cap(mat, cap_h)
    pv.set_option("leaf_hash_form", 0)
    ok = bool(np.array_equal(cap, O.merkle_cap(np.ascontiguousarray(mat.T), cap_h)))
    out["shapes"].append((log_N, ncols, ok))
    bad += not ok
print(json.dumps(out), flush=True)
if bad == 0 and "--no-proof" not in sys.argv:
    air = S.AIR_FINAL_EXP
    cfg = S.StarkConfig.for_air(air)
    trace, pis = S.trace_final_exp(random_fp12(0x5EED0001), compact=True)
    res = {}
    proofs = {}
    for form, name in ((4, "pair"),) if TIMING else ((1, "quad"), (4, "pair"), (0, "auto")):
        pv.set_option("leaf_hash_form", form)
        ms = []
        for rep in range(3):
            proofs[name] = pv.prove(air, cfg, trace, pis)
            ms.append(round(pv.last_kernel_timings()["leaf_hash"], 2))
        res[name] = ms
    res["same_bytes"] = None if TIMING else bool(np.array_equal(proofs["quad"], proofs["pair"]) and np.array_equal(proofs["quad"], proofs["auto"]))
    print(json.dumps(res))
pv.close()
sys.exit(1 if bad else 0)
