import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import starky_bls12_381_amd as S
pv = S.Prover(0)
for impl in (0, 1):
    pv.set_option("lde_impl", impl)
    print("impl", impl, "eight launches after a warm-up launch:", pv.lde_bench(73527, 13, 2, 8, 11, each=True))
    print("impl", impl, "eight launches, no warm-up:           ", [pv.lde_bench(73527, 13, 2, 0, 11, each=True)[0] for _ in range(2)])
