#!/usr/bin/env python3
"""Where the wave-resident LDE kernel disagrees with lde_columns_v2_kernel (development aid; GPU)."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import starky_bls12_381_amd as S
import lde_wave_model as M

pv = S.Prover(0)
n = 1 << 13
rng = np.random.default_rng(7)
for rate in (0, 2):
    vals = rng.integers(0, S.P, size=(2, n), dtype=np.uint64)
    pv.set_option("lde_impl", 1)
    _, want = pv.lde_batch(vals, rate)
    pv.set_option("lde_impl", 0)
    _, got = pv.lde_batch(vals, rate)
    R = 1 << rate
    for col in range(2):
        w = want[col].reshape(n, R).T  # [s][k]
        g = got[col].reshape(n, R).T
        for s in range(R):
            bad = np.flatnonzero(w[s] != g[s])
            print("rate", rate, "col", col, "coset", s, "mismatches", bad.size, "first", bad[:8])
            if bad.size and bad.size < n:
                print("   bit pattern of bad indices: OR %x AND %x" % (np.bitwise_or.reduce(bad), np.bitwise_and.reduce(bad)))
    if rate == 0:
        # the model's forward transform output on the same input, to see which stage departs: compare got with a permutation of want
        col = 0
        wv = {int(v): i for i, v in enumerate(want[col])}
        hits = [(k, wv[int(v)]) for k, v in enumerate(got[col][:4096]) if int(v) in wv]
        print("values of `got` that occur in `want` at another index:", len(hits), hits[:16])
