#!/usr/bin/env python3
"""Run the CPU oracle on the full FinalExponentiateStark proof of the reference's `aa` vector
(src/native.rs:1546-1557) and store the SHA-256 of the proof bytes as a golden fixture.
Needs ~35 GB of host RAM and several minutes; run on the GPU box:  python tests/make_final_exp_golden.py
With `--seed 0x5EED0001 OUT` the input is bench.py's seeded synthetic Fp12 instead (the first input the
benchmark times), e.g.  python tests/make_final_exp_golden.py --seed 0x5EED0001 tests/golden/final_exp_seed_5eed0001_proof.sha256
(kept under tests/: it executes the CPU oracle, which only tests, smoke() and bench.py's cpu_baseline leg may do)"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_lib as O  # noqa: E402
import starky_bls12_381_amd as S  # noqa: E402
from bls_util import fp_arr, native_vectors, random_fp12  # noqa: E402

air = S.AIR_FINAL_EXP
argv = sys.argv[1:]
label = "final_exp_aa_proof"
if argv and argv[0] == "--seed":
    seed = int(argv[1], 0)
    aa = random_fp12(seed)
    label = "final_exp_seed_%x_proof" % seed
    argv = argv[2:]
else:
    aa = fp_arr(*[int(s) for s in native_vectors()["final_exp_input_aa"]])
t, pis = S.trace_final_exp(aa)
cols = S.trace_rows_to_poly_values(t)
del t
cfg = S.StarkConfig.for_air(air)
t0 = time.time()
proof = O.prove(S.air_program(air), cfg, cols, pis)
print("oracle FinalExp prove: %.1f s on %d threads" % (time.time() - t0, O.lib.oracle_num_threads()), flush=True)
S.verify_stark_proof(air, cfg, proof)
d = hashlib.sha256(proof.tobytes()).hexdigest()
out = argv[0] if argv else os.path.join(ROOT, "tests", "golden", label + ".sha256")
open(out, "w").write(d + "  %s (CPU oracle, %d u64 words)\n" % (label, proof.size))
print(d)
