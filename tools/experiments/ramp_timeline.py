#!/usr/bin/env python3
"""Start and tail of a 20-proof run through the eight-context pool: per proof submit / generation / prove start / done (ms from the first
submit) and how its trace commitment went out."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import starky_bls12_381_amd as S  # noqa: E402
from bls_util import random_fp12  # noqa: E402

air = S.AIR_FINAL_EXP
x = [random_fp12(0x5EED0001 + i) for i in range(8)]
pool = S.ProofPool(0, big_contexts=8, small_contexts=1, warm_up=1)
try:
    for rep in range(2):
        for t in [pool.submit_witness(air, x[i % 8]) for i in range(8)]:
            pool.wait(t, keep=False)
    t0 = time.perf_counter()
    tickets = [pool.submit_witness(air, x[i % 8]) for i in range(20)]
    rows = []
    for t in tickets:
        _, info = pool.wait(t, keep=False)
        rows.append(info)
    wall = time.perf_counter() - t0
    base = min(r["timeline_s"][0] for r in rows)
    print("wall %.1f ms = %.2f proofs/s" % (wall * 1e3, 20 / wall))
    for i, r in enumerate(rows):
        tl = [(v - base) * 1e3 for v in r["timeline_s"]]
        print("%2d submit %6.1f gen %6.1f..%6.1f prove %6.1f done %7.1f | %s x%d hash %.0f ms | lde %.1f quot %.1f" %
              (i, tl[0], tl[1], tl[2], tl[3], tl[4], r["leaf_hash_form"], r["leaf_hash_group"], r["kernel_ms"]["leaf_hash"], r["kernel_ms"]["lde_columns"], r["kernel_ms"]["quotient_eval"]))
finally:
    pool.close()
