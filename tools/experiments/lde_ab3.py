#!/usr/bin/env python3
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import starky_bls12_381_amd as S
from bls_util import random_fp12
pv = S.Prover(0)
dense, pis = S.trace_final_exp(random_fp12(0x5EED0001))
d = torch.from_numpy(dense.view(np.int64)).cuda().t().contiguous()
C, n = d.shape
b = lambda k=0, ptr=None: round(pv.lde_bench(C, 13, 2, 3, k, ptr), 2)
print("synthetic all transformed", b(0), "| 11/64 const", b(11))
print("real trace", b(0, d.data_ptr()))
pv.set_option("lde_closed_forms", 0)
print("closed forms off: synthetic", b(0), "real", b(0, d.data_ptr()))
g = torch.Generator(device="cuda"); g.manual_seed(1)
r = torch.randint(0, 2**62, d.shape, dtype=torch.int64, device="cuda", generator=g)
print("closed forms off: torch random", b(0, r.data_ptr()))
r2 = torch.where(d != 0, r, torch.zeros_like(r))
print("closed forms off: random where the trace is non-zero", b(0, r2.data_ptr()))
r3 = r & 0xFFFFFFFF
print("closed forms off: random 32-bit values", b(0, r3.data_ptr()))
r4 = torch.where(d != 0, r3, torch.zeros_like(r))
print("closed forms off: random 32-bit values where the trace is non-zero", b(0, r4.data_ptr()))
