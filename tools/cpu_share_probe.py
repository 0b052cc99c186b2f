import os, time, threading, ctypes, sys
print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a")
for f in ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us","/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f): print(f, open(f).read().strip())
print("affinity:", len(os.sched_getaffinity(0)), "nproc:", os.cpu_count())
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import starky_bls12_381_amd as S
import numpy as np
lib = S.lib
def work(n, out, i):
    st = np.arange(12, dtype=np.uint64)
    t0 = time.perf_counter()
    lib.starkhip_poseidon_permute_host_many(st.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), n, 0)
    out[i] = time.perf_counter() - t0
lib.starkhip_poseidon_permute_host_many.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_size_t, ctypes.c_int]
for nt in (1, 4, 8, 16, 32, 64, 128):
    out = [0] * nt
    th = [threading.Thread(target=work, args=(400000, out, i)) for i in range(nt)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    wall = time.perf_counter() - t0
    print(f"{nt:4d} threads: wall {wall:.3f}s  perms/s total {nt*400000/wall/1e6:.2f} M  (per-thread {400000/max(out)/1e6:.2f} M)")
