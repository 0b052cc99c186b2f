import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import starky_bls12_381_amd as S
pv = S.Prover(0)
for impl in (0, 1):
    pv.set_option("lde_impl", impl)
    print("impl", impl, {(cols, k): round(pv.lde_bench(cols, 13, 2, 3, k), 2) for cols, k in ((60000, 0), (73527, 11), (8192, 64), (8192, 64 + 256), (73527, 7 + 256), (73527, 4))})
