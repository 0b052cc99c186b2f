#!/usr/bin/env python3
"""Operand fixture for the C++ signature demo (tools/signature_demo.cpp --batch B --operands FILE): B different VALID BLS
signatures derived from the reference's own test vector (src/native.rs:1480-1498, tests/golden/native_vectors.json) by
`signature.synthetic_signatures` (H_i = t_i H(m), sig_i = s_i H_i, pk_i = s_i G1), written as B records of 120
little-endian u32 limbs: pk x, y (12 each), H(m) x, y (24 each), signature x, y (24 each); Z = (1, 0) implied.

    python tools/make_signature_operands.py 8 tests/golden/signature_operands_8.bin
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from starky_bls12_381_amd import signature as G  # noqa: E402

count, out = int(sys.argv[1]), sys.argv[2]
seed = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0x2000
vec = json.load(open(os.path.join(ROOT, "tests", "golden", "native_vectors.json")))["bls_signature"]
sigs = G.synthetic_signatures(count, vec, seed)
recs = np.concatenate([np.concatenate([pk[0], pk[1], hm[0], hm[1], sig[0], sig[1]]).astype("<u4") for pk, hm, sig in sigs])
assert recs.size == 120 * count
recs.tofile(out)
print(f"{out}: {count} signatures, seed {seed:#x}, {recs.nbytes} bytes")
