#!/usr/bin/env python3
"""Where a FinalExp proof's time goes when the trace arrives as page-locked host ROWS with eight proofs in flight (bench.py's
`value_host_rows` leg): per proof the start and end of proving and the upload / LDE / commitment phases."""
import os
import sys
import time

if "preload" in sys.argv:  # the system's HIP runtime first, so that torch (which brings its own, older one) and the library share it
    import ctypes
    for lib in ("libhsa-runtime64.so.1", "libamdhip64.so.7"):
        ctypes.CDLL(os.path.join("/opt/rocm/lib", lib), mode=ctypes.RTLD_GLOBAL)
if "torchfirst" in sys.argv or "preload" in sys.argv:
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
    print("torch", torch.__version__, torch.version.hip, sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l or "hsa-runtime" in l}))
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import starky_bls12_381_amd as S  # noqa: E402
from bls_util import random_fp12  # noqa: E402

air = S.AIR_FINAL_EXP
cfg = S.StarkConfig.for_air(air)
n, C_ = 8192, S.air_columns(air)
inflight = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if len(sys.argv) > 2 and sys.argv[2] == "torch":  # as bench.py does before it creates its pool
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
pool = S.ProofPool(0, big_contexts=inflight, small_contexts=1, warm_up=1)
helper = S.Prover(0)
rows = helper.host_array((n, C_))
_, pis = S.trace_final_exp(random_fp12(0x5EED0001), out=rows)
witness = "witness" in sys.argv  # the bench's headline path instead: operand -> recording -> upload -> proof
xs = [random_fp12(0x5EED0001 + i) for i in range(inflight)]
sub = (lambda i: pool.submit_witness(air, xs[i % inflight])) if witness else (lambda i: pool.submit(air, cfg, rows, pis))
for t in [sub(i) for i in range(inflight)]:
    pool.wait(t, keep=False)
t0 = time.perf_counter()
tickets = [sub(i) for i in range(3 * inflight)]
infos = [pool.wait(t, keep=False)[1] for t in tickets]
dt = time.perf_counter() - t0
base = min(i["timeline_s"][0] for i in infos)
for k, i in enumerate(infos if len(sys.argv) < 4 else []):
    tl, ph = i["timeline_s"], i["phase_ms"]
    print("%2d prove %7.1f .. %7.1f ms | upload %6.1f lde %6.1f merkle %6.1f quotient %6.1f rest %6.1f | %s x %d" % (
        k, (tl[3] - base) * 1e3, (tl[4] - base) * 1e3, ph["upload"], ph["ifft_lde"], ph["trace_merkle"], ph["quotient"],
        ph["total"] - ph["upload"] - ph["ifft_lde"] - ph["trace_merkle"] - ph["quotient"], i["leaf_hash_form"], i["leaf_hash_group"]))
print("%.3f proofs/s" % (len(tickets) / dt))
pool.close()
