#!/usr/bin/env python3
"""Signatures/s on one MI355X: the six STARK proofs of one BLS signature check (2 x PairingPrecomp, 2 x MillerLoop,
FP12Mul, FinalExp; BASELINE.json configs[3]/[4] in single-GPU form).  Traces are generated on the host and moved to
HBM (column-major) before the timed region, like bench.py; proofs are proven two at a time on two contexts.
Prints one JSON line."""
import argparse
import json
import os
import sys
import threading
import time

# one hardware queue per in-flight proof: with the HIP default of 4, six streams share queues and their kernels serialise
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--inflight", type=int, default=6)
    args = ap.parse_args()
    import numpy as np
    import torch
    import starky_bls12_381_amd as S
    from starky_bls12_381_amd import aggregate as A
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
    provers = [S.Prover(0) for _ in range(max(1, args.inflight))]
    order = sorted(A.JOB_ORDER, key=lambda n: -__import__("starky_bls12_381_amd").parallel.AIR_COST[A.JOB_AIR[n]])
    per_air, phases = {}, {}

    def run_all(record):
        todo = list(order)
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
                if record:
                    per_air[name] = (time.perf_counter() - t) * 1e3
                    phases[name] = {k: round(v, 1) for k, v in pv.last_timings().items() if v >= 1.0}
        th = [threading.Thread(target=worker, args=(pv,)) for pv in provers]
        for x in th:
            x.start()
        for x in th:
            x.join()

    run_all(False)  # warm-up (tables, programs, buffers)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_all(True)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"metric": "BLS signature checks/s (6 STARK proofs each) on 1 MI355X", "value": 1.0 / el, "unit": "signatures/s",
                      "ms_per_signature": el * 1e3, "proofs_in_flight": len(provers), "per_proof_wall_ms": per_air, "per_proof_phase_ms": phases,
                      "host_trace_generation_and_upload_s": t_gen, "signature_valid": A.signature_is_valid(natives), "data": "reference test vector src/native.rs:1480-1498"}))
    for pv in provers:
        pv.close()


if __name__ == "__main__":
    main()
