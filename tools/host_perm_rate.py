import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
import starky_bls12_381_amd as S
f = S.lib.starkhip_poseidon_permute_host_many
f.argtypes = [C.POINTER(C.c_uint64), C.c_size_t, C.c_int]; f.restype = None
for which in (0, 1):
    s = np.arange(12, dtype=np.uint64)
    t0 = time.time(); f(s.ctypes.data_as(C.POINTER(C.c_uint64)), 100000, which); dt = time.time() - t0
    print("which", which, "%.3f us/perm" % (dt / 1e5 * 1e6), hex(int(s[0])))
