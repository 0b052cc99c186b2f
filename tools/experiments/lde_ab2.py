#!/usr/bin/env python3
"""LDE kernel time inside prove() for the real FinalExp trace: from a recorded trace (parked in the LDE buffer, four launches) and from
a column-major trace in the caller's device memory (one launch); both kernels."""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import starky_bls12_381_amd as S
from bls_util import random_fp12
pv = S.Prover(0)
air = S.AIR_FINAL_EXP
cfg = S.StarkConfig.for_air(air)
x = random_fp12(0x5EED0001)
compact, pis = S.trace_final_exp(x, compact=True)
dense, _ = S.trace_final_exp(x)
d = torch.from_numpy(dense.view(np.int64)).cuda().t().contiguous()
del dense
n = d.shape[1]
for what in ("compact", "device"):
    for impl in (0, 1):
        pv.set_option("lde_impl", impl)
        ts = []
        for rep in range(3):
            if what == "compact":
                pv.prove(air, cfg, compact, pis)
            else:
                pv.prove_device(air, cfg, d.data_ptr(), n, pis, layout=1, keep=False)
            ts.append((round(float(pv.last_kernel_timings()[0] if not isinstance(pv.last_kernel_timings(), dict) else pv.last_kernel_timings()["lde_columns"]), 2), round(float(list(pv.last_timings().values())[1] if isinstance(pv.last_timings(), dict) else pv.last_timings()[1]), 2)))
        print(what, "impl", impl, "(kernel ms, ifft_lde phase ms):", ts)
