#!/usr/bin/env python3
"""The lane-form leaf hash by itself: one FinalExp commitment alone on the chip (one wave per SIMD) and four side by side (two waves
per SIMD, what a pool's lane groups run).  Kernel durations from the HIP events the library records around the launch.

    python tools/lane_bench.py            # the library in the tree;  STARKHIP_LIBRARY=... for a variant build
"""
import json
import os
import sys

os.environ.setdefault("STARKHIP_POOL_BIG_LANE", "1")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import starky_bls12_381_amd as S  # noqa: E402
from bls_util import random_fp12  # noqa: E402

air = S.AIR_FINAL_EXP
x = [random_fp12(0x5EED0001 + i) for i in range(4)]
out = {}
pool = S.ProofPool(0, big_contexts=4, small_contexts=1, warm_up=1)
try:
    for t in [pool.submit_witness(air, x[i]) for i in range(4)]:
        pool.wait(t, keep=False)
    groups = []
    for rep in range(4):
        infos = [pool.wait(t, keep=False)[1] for t in [pool.submit_witness(air, x[i]) for i in range(4)]]
        groups.append([(i["leaf_hash_form"], i["leaf_hash_group"], round(i["kernel_ms"]["leaf_hash"], 1)) for i in infos])
    out["groups_of_four"] = groups
    four = [ms for g in groups for (form, n, ms) in g if form == "lane" and n == 4]
    out["lane_ms_in_groups_of_four"] = sum(four) / len(four) if four else None
finally:
    pool.close()
pv = S.Prover(0)
try:
    pv.set_option("leaf_hash_form", 3)
    cfg = S.StarkConfig.for_air(air)
    trace, pis = S.trace_final_exp(x[0], compact=True)
    alone = []
    for rep in range(3):
        pv.prove(air, cfg, trace, pis)
        alone.append(round(pv.last_kernel_timings()["leaf_hash"], 1))
    out["lane_ms_alone"] = alone
finally:
    pv.close()
print(json.dumps(out))
