#!/usr/bin/env python3
"""Recording time of the three threaded trace generators against starkhip_trace_set_threads (host only, no GPU work)."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import starky_bls12_381_amd as S  # noqa: E402
from starky_bls12_381_amd import aggregate as A, signature as G  # noqa: E402
from bls_util import native_vectors  # noqa: E402

G.tune_host_allocator()
sigs = G.synthetic_signatures(1, native_vectors()["bls_signature"], 5)
t0 = time.perf_counter()
jobs, nat = A.signature_jobs(*sigs[0])
print("natives of one signature: %.3f s;  os.cpu_count() = %d, affinity = %d" % (time.perf_counter() - t0, os.cpu_count(), len(os.sched_getaffinity(0))))
for name in ("pp1", "ml1", "final_exp"):
    args = jobs[name][1]
    G.GENERATORS[name](*args, compact=True)
    row = []
    for n in (1, 2, 4, 8, 16, 32):
        S.set_trace_threads(n)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            G.GENERATORS[name](*args, compact=True)
            best = min(best, time.perf_counter() - t0)
        row.append("%d: %.3f" % (n, best))
    print(name, "  ".join(row))
