"""Host permutation rate (the challenger's sequential sponge): best of several trials, microseconds per permutation.
which 0 = the tuned AVX-512 permutation (poseidon_host.cpp), 1 = the portable loop (poseidon.h)."""
import ctypes as C
import sys
import time

sys.path.insert(0, '.')
import numpy as np
import starky_bls12_381_amd as S

f = S.lib.starkhip_poseidon_permute_host_many
f.argtypes = [C.POINTER(C.c_uint64), C.c_size_t, C.c_int]
f.restype = None
for which in (0, 1):
    best = 1e9
    for trial in range(7):
        s = np.arange(12, dtype=np.uint64)
        t0 = time.perf_counter()
        f(s.ctypes.data_as(C.POINTER(C.c_uint64)), 100000, which)
        best = min(best, time.perf_counter() - t0)
    print("which", which, "%.3f us/perm (best of 7 x 100000)" % (best / 1e5 * 1e6), hex(int(s[0])))
