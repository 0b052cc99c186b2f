#!/usr/bin/env python3
"""Signatures/s on one MI355X: the six STARK proofs of one BLS signature check (2 x PairingPrecomp, 2 x MillerLoop,
FP12Mul, FinalExp; BASELINE.json configs[3]/[4] in single-GPU form).  Traces are generated on the host and moved to
HBM (column-major) before the timed region, like bench.py.

--batch 1 (default): one signature at a time, its six proofs on --inflight contexts (latency of one check).
--batch B: B signatures scheduled by proof TYPE.  The small AIRs are latency chains (MillerLoop: 2048 leaves of 12 167
  sequential permutations = 128 waves for ~0.2 s), so a single one leaves 7/8 of the SIMDs idle while sixteen of them
  side by side fill the chip; FinalExp's leaf hash is a one-shot grid of exactly two waves per SIMD that any foreign
  wave stretches.  Hence two phases: all small proofs with --small-inflight contexts, then the B FinalExp proofs
  --big-inflight at a time.  The same (synthetic) signature is used B times, so the traces are resident once.
Prints one JSON line."""
import argparse
import json
import os
import sys
import threading
import time

# one hardware queue per in-flight proof: with the HIP default of 4, more streams share queues and their kernels serialise
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run_pool(S, A, provers, jobs, resident, record=None):
    """jobs: list of names; each free context takes the next one."""
    todo = list(jobs)
    lock = threading.Lock()

    def worker(pv):
        while True:
            with lock:
                if not todo:
                    return
                name = todo.pop(0)
            d, n, pis = resident[name]
            air = A.JOB_AIR[name]
            t = time.perf_counter()
            pv.prove_device(air, S.StarkConfig.for_air(air), d.data_ptr(), n, pis, layout=1, keep=False)
            if record is not None:
                record.setdefault(name, []).append((time.perf_counter() - t) * 1e3)
    th = [threading.Thread(target=worker, args=(pv,)) for pv in provers]
    for x in th:
        x.start()
    for x in th:
        x.join()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--inflight", type=int, default=6)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--small-inflight", type=int, default=24)
    ap.add_argument("--big-inflight", type=int, default=4)
    args = ap.parse_args()
    import numpy as np
    import torch
    import starky_bls12_381_amd as S
    from starky_bls12_381_amd import aggregate as A
    from starky_bls12_381_amd import parallel
    from test_aggregate_cpu import _bls_points

    _, pk, hm, sig = _bls_points()
    jobs, natives = A.signature_jobs(pk, hm, sig)
    gens = {"pp1": S.trace_pairing_precomp, "pp2": S.trace_pairing_precomp, "ml1": S.trace_miller_loop, "ml2": S.trace_miller_loop,
            "fp12_mul": S.trace_fp12_mul, "final_exp": S.trace_final_exp}
    resident = {}
    t0 = time.perf_counter()
    for name in A.JOB_ORDER:
        trace, pis = gens[name](*jobs[name][1])
        d = torch.from_numpy(trace.view(np.int64)).to("cuda:0").t().contiguous()
        resident[name] = (d, trace.shape[0], pis)
        del trace
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    by_cost = sorted(A.JOB_ORDER, key=lambda n: -parallel.AIR_COST[A.JOB_AIR[n]])
    out = {"metric": "BLS signature checks/s (6 STARK proofs each) on 1 MI355X", "unit": "signatures/s", "batch": args.batch,
           "host_trace_generation_and_upload_s": t_gen, "signature_valid": A.signature_is_valid(natives),
           "data": "reference test vector src/native.rs:1480-1498" + (f", used {args.batch} times" if args.batch > 1 else "")}
    per_proof = {}
    if args.batch <= 1:
        provers = [S.Prover(0) for _ in range(max(1, args.inflight))]
        run_pool(S, A, provers, by_cost, resident)  # warm-up (tables, programs, buffers)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run_pool(S, A, provers, by_cost, resident, per_proof)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / args.steps
        out.update({"value": 1.0 / el, "ms_per_signature": el * 1e3, "proofs_in_flight": len(provers)})
    else:
        small = [n for n in by_cost if n != "final_exp"] * args.batch
        small.sort(key=lambda n: -parallel.AIR_COST[A.JOB_AIR[n]])  # all MillerLoop first, then PairingPrecomp, then FP12Mul
        big = ["final_exp"] * args.batch
        small_pv = [S.Prover(0) for _ in range(max(1, args.small_inflight))]
        big_pv = [S.Prover(0) for _ in range(max(1, args.big_inflight))]
        run_pool(S, A, small_pv, [n for n in by_cost if n != "final_exp"] * len(small_pv), resident)  # warm every context on every small AIR
        run_pool(S, A, big_pv, ["final_exp"] * len(big_pv), resident)
        torch.cuda.synchronize()
        t_small = t_big = 0.0
        for _ in range(args.steps):
            t0 = time.perf_counter()
            run_pool(S, A, small_pv, small, resident, per_proof)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_pool(S, A, big_pv, big, resident, per_proof)
            torch.cuda.synchronize()
            t_small += t1 - t0
            t_big += time.perf_counter() - t1
        el = (t_small + t_big) / args.steps
        out.update({"value": args.batch / el, "ms_per_signature": el * 1e3 / args.batch, "ms_small_phase": t_small / args.steps * 1e3,
                    "ms_final_exp_phase": t_big / args.steps * 1e3, "small_in_flight": len(small_pv), "final_exp_in_flight": len(big_pv)})
        provers = small_pv + big_pv
    out["per_proof_wall_ms"] = {k: round(sum(v) / len(v), 1) for k, v in per_proof.items()}
    print(json.dumps(out))
    for pv in provers:
        pv.close()


if __name__ == "__main__":
    main()
