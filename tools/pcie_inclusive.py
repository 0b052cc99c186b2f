#!/usr/bin/env python3
"""FinalExp proofs/s when the trace is handed over as a HOST buffer (the reference's boundary: generate_trace returns rows
in host memory): host rows -> H2D -> transpose on the device -> prove.  Three hand-over forms: pageable row-major rows
(what a plain caller has), pageable column-major columns, and rows generated straight into a page-locked buffer from
starkhip_host_alloc (same upload rate - the link is the limit either way - but a reused buffer halves host trace
generation, which otherwise page-faults 4.8 GB of fresh memory per trace).  Never the bench's `value`."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import starky_bls12_381_amd as S  # noqa: E402
from bls_util import random_fp12  # noqa: E402

air = S.AIR_FINAL_EXP
cfg = S.StarkConfig.for_air(air)
x = random_fp12(0x5EED0001)
t0 = time.perf_counter()
trace, pis = S.trace_final_exp(x)
t_gen = time.perf_counter() - t0
pv = S.Prover(0)
ref = pv.prove(air, cfg, trace, pis)  # warm-up: tables, program, buffers
t0 = time.perf_counter()
pinned = pv.host_array(trace.shape)
t_pin = time.perf_counter() - t0
t0 = time.perf_counter()
S.trace_final_exp(x, out=pinned)
t_gen_pinned = time.perf_counter() - t0
assert np.array_equal(pinned, trace)
cases = ((0, trace, "pageable row-major rows"), (1, S.trace_rows_to_poly_values(trace), "pageable column-major columns"),
         (0, pinned, "page-locked row-major rows"))
for layout, data, name in cases:
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        proof = pv.prove(air, cfg, data, pis, layout=layout)
        ts.append(time.perf_counter() - t0)
    assert np.array_equal(proof, ref)
    dev = pv.last_timings()
    print(f"{name}: {min(ts) * 1e3:.0f} ms per proof end to end ({1 / min(ts):.2f} proofs/s); device phases total {dev['total']:.0f} ms, upload {dev['upload']:.0f} ms")
print(f"host trace generation: {t_gen * 1e3:.0f} ms into pageable memory, {t_gen_pinned * 1e3:.0f} ms into the page-locked buffer (one core); "
      f"allocating the {trace.nbytes / 1e6:.0f} MB page-locked buffer: {t_pin * 1e3:.0f} ms (once)")

# two proofs in flight from host rows (two contexts, two host threads), as bench.py does with resident traces:
# one proof's upload and transpose overlap the other's kernels
import threading  # noqa: E402

pv2 = S.Prover(0)
pv2.prove(air, cfg, pinned, pis)  # warm-up of the second context
steps, todo, lock = 6, [], threading.Lock()


def worker(p):
    while True:
        with lock:
            if not todo:
                return
            todo.pop()
        p.prove(air, cfg, pinned, pis)


todo.extend(range(steps))
t0 = time.perf_counter()
th = [threading.Thread(target=worker, args=(p,)) for p in (pv, pv2)]
for w_ in th:
    w_.start()
for w_ in th:
    w_.join()
el = (time.perf_counter() - t0) / steps
print(f"two proofs in flight from page-locked host rows: {el * 1e3:.0f} ms per proof ({1 / el:.2f} proofs/s), upload included")

# compact trace: the generator records runs, the device expands them -- no 4.8 GB of host rows at all
t0 = time.perf_counter()
compact, cpis = S.trace_final_exp(x, compact=True)
t_rec = time.perf_counter() - t0
assert np.array_equal(pv.prove(air, cfg, compact, cpis), ref)
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    pv.prove(air, cfg, compact, cpis)
    ts.append(time.perf_counter() - t0)
dev = pv.last_timings()
print(f"compact trace: {compact.nbytes / 1e6:.0f} MB of records ({compact.n_records} runs) generated in {t_rec * 1e3:.0f} ms (one core); "
      f"{min(ts) * 1e3:.0f} ms per proof end to end ({1 / min(ts):.2f} proofs/s), upload + expansion {dev['upload']:.1f} ms")
todo.extend(range(steps))
t0 = time.perf_counter()
th = [threading.Thread(target=lambda p=p: [p.prove(air, cfg, compact, cpis) for _ in range(steps // 2)]) for p in (pv, pv2)]
for x_ in th:
    x_.start()
for x_ in th:
    x_.join()
todo.clear()
el = (time.perf_counter() - t0) / steps
print(f"two proofs in flight from compact traces: {el * 1e3:.0f} ms per proof ({1 / el:.2f} proofs/s), upload + expansion included")
