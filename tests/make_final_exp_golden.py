#!/usr/bin/env python3
"""Run the CPU oracle on the full FinalExponentiateStark proof of the reference's `aa` vector
(src/native.rs:1546-1557) and store the SHA-256 of the proof bytes as a golden fixture.
Needs ~35 GB of host RAM and several minutes; run on the GPU box:  python tests/make_final_exp_golden.py
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
from bls_util import fp_arr, native_vectors  # noqa: E402

air = S.AIR_FINAL_EXP
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
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "final_exp_aa_proof.sha256")
open(out, "w").write(d + "  final_exp_aa_proof (CPU oracle, %d u64 words)\n" % proof.size)
print(d)
