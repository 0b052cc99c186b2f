#!/usr/bin/env python3
"""FinalExp proofs/s when the trace is handed over as a HOST buffer (the reference's boundary: generate_trace returns rows
in host memory): pageable row-major rows -> H2D -> transpose on the device -> prove.  Never the bench's `value`."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import starky_bls12_381_amd as S  # noqa: E402
from bls_util import random_fp12  # noqa: E402

air = S.AIR_FINAL_EXP
cfg = S.StarkConfig.for_air(air)
t0 = time.perf_counter()
trace, pis = S.trace_final_exp(random_fp12(0x5EED0001))
t_gen = time.perf_counter() - t0
pv = S.Prover(0)
pv.prove(air, cfg, trace, pis)  # warm-up: tables, program, buffers
for layout, data, name in ((0, trace, "row-major host rows"), (1, S.trace_rows_to_poly_values(trace), "column-major host columns")):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        pv.prove(air, cfg, data, pis, layout=layout)
        ts.append(time.perf_counter() - t0)
    dev = pv.last_timings()
    print(f"{name}: {min(ts) * 1e3:.0f} ms per proof end to end ({1 / min(ts):.2f} proofs/s); device phases total {dev['total']:.0f} ms, upload {dev['upload']:.0f} ms")
print(f"host trace generation: {t_gen * 1e3:.0f} ms (one core)")
