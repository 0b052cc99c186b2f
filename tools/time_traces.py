import time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import starky_bls12_381_amd as S
from bls_util import random_fp12
from test_aggregate_cpu import _bls_points
from starky_bls12_381_amd import aggregate as A
_, pk, hm, sig = _bls_points()
for rep in range(2):
    t0=time.perf_counter(); t,p=S.trace_final_exp(random_fp12(5)); t1=time.perf_counter(); print("final_exp trace", round(t1-t0,3), "s", t.nbytes/1e9, "GB"); del t
    t0=time.perf_counter(); t,p=S.trace_miller_loop(pk[0],pk[1],hm[0],hm[1],hm[2]); t1=time.perf_counter(); print("miller trace", round(t1-t0,3)); del t
    t0=time.perf_counter(); t,p=S.trace_pairing_precomp(hm[0],hm[1],hm[2]); t1=time.perf_counter(); print("precomp trace", round(t1-t0,3)); del t
    t0=time.perf_counter(); a=np.zeros((8192,73527),dtype=np.uint64); a[:]=0; t1=time.perf_counter(); print("np.zeros+touch 4.8GB", round(t1-t0,3)); del a
