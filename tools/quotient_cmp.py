#!/usr/bin/env python3
"""Development aid: prove PairingPrecomp (or --air) with the interpreter and let the prover compare the tiled evaluator's
values with it point by point (ctx option quotient_debug = 9; mismatches are printed on stderr)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import starky_bls12_381_amd as S  # noqa: E402
from bls_util import fp_arr, native_vectors, random_fp12  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "precomp"
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 0
v = native_vectors()
b = {k: int(x) for k, x in v["bls_signature"].items()} if "bls_signature" in v else None
if which == "precomp":
    air = S.AIR_PAIRING_PRECOMP
    t, pis = S.trace_pairing_precomp(fp_arr(b["hm_x1"], b["hm_x2"]), fp_arr(b["hm_y1"], b["hm_y2"]), fp_arr(b["hm_z1"], b["hm_z2"]))
elif which == "ecc":
    air = None
else:
    air = S.AIR_FINAL_EXP
    t, pis = S.trace_final_exp(random_fp12(0x5EED0001))
cfg = S.StarkConfig.for_air(air)
pv = S.Prover(0)
pv.set_option("quotient_impl", 1)
pv.set_option("quotient_chunks", chunks)
pv.set_option("quotient_debug", 9)
pv.prove(air, cfg, t, pis)
