#!/usr/bin/env python3
"""What one proof of each AIR costs the chip when the pool is full of them: N proofs of ONE AIR through the library's pool (its default
contexts, trace generation inside), milliseconds per proof = elapsed / N.  These are the weights of the longest-job-first placement
(csrc/scheduler.cpp air_cost, parallel.AIR_COST): what a job adds to a device's queue is its share of the device's time, not its latency.

    python tools/air_pool_cost.py [--proofs 48] > profiles/rNN_air_pool_cost.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--proofs", type=int, default=48)
    args = ap.parse_args()
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import starky_bls12_381_amd as S
    from starky_bls12_381_amd import aggregate as A
    from starky_bls12_381_amd import signature as G
    from bls_util import native_vectors
    from test_ecc_aggregate_cpu import pack, reference_vector

    sig = G.synthetic_signatures(1, native_vectors()["bls_signature"], 7)[0]
    jobs, _ = A.signature_jobs(*sig)
    pts, bits, _ = reference_vector()
    work = {"FinalExponentiateStark": (S.AIR_FINAL_EXP, jobs["final_exp"][1], max(16, args.proofs // 2)),
            "MillerLoopStark": (S.AIR_MILLER_LOOP, jobs["ml1"][1], args.proofs),
            "PairingPrecompStark": (S.AIR_PAIRING_PRECOMP, jobs["pp1"][1], args.proofs),
            "ECCAggStark": (S.AIR_ECC_AGGREGATE, pack(pts, bits), args.proofs),
            "FP12MulStark": (S.AIR_FP12_MUL, jobs["fp12_mul"][1], args.proofs)}
    pool = S.ProofPool(0, big_contexts=8, small_contexts=16, warm_up=1)
    out = {}
    try:
        for name, (air, operands, n) in work.items():
            for t in [pool.submit_witness(air, *operands) for _ in range(8)]:  # warm: plans, tables, every context once
                pool.wait(t, keep=False)
            t0 = time.perf_counter()
            tickets = [pool.submit_witness(air, *operands) for _ in range(n)]
            for t in tickets:
                pool.wait(t, keep=False)
            ms = (time.perf_counter() - t0) * 1e3 / n
            out[name] = {"proofs": n, "ms_per_proof_in_a_full_pool": round(ms, 2)}
            print(name, out[name], file=sys.stderr, flush=True)
    finally:
        pool.close()
    fe = out["FinalExponentiateStark"]["ms_per_proof_in_a_full_pool"]
    for v in out.values():
        v["relative_to_final_exp_at_92"] = round(92.0 * v["ms_per_proof_in_a_full_pool"] / fe, 2)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
