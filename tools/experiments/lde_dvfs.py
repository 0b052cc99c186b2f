import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import starky_bls12_381_amd as S
pv = S.Prover(0)
print("eight launches after a warm-up launch:", pv.lde_bench(73527, 13, 2, 8, 11, each=True))
print("one launch, nothing in front:         ", [pv.lde_bench(73527, 13, 2, 0, 11, each=True)[0] for _ in range(3)])
print("one launch after a memset of the whole output buffer:", [pv.lde_bench(73527, 13, 2, 0, 11 + 2048, each=True)[0] for _ in range(3)])
