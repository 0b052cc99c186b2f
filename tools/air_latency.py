#!/usr/bin/env python3
"""Single-proof latency of every AIR on one MI355X, one proof in flight: wall time and the prover's phase times.

    python tools/air_latency.py [--reps 3] > profiles/rNN_air_latency.json

The 1024-row AIRs are latency chains (MillerLoop: 2048 leaves of 12 167 sequential permutations), so their numbers say how
well ONE proof uses the chip, not the throughput of many side by side (tools/bench_signature.py)."""
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
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import starky_bls12_381_amd as S
    from starky_bls12_381_amd import aggregate as A
    from starky_bls12_381_amd import signature as G
    from bls_util import native_vectors

    sig = G.synthetic_signatures(1, native_vectors()["bls_signature"], 7)[0]
    jobs, _ = A.signature_jobs(*sig)
    pv = S.Prover(0)
    out = {}
    for name in ("fp12_mul", "pp1", "ml1", "final_exp"):
        air = A.JOB_AIR[name]
        cfg = S.StarkConfig.for_air(air)
        trace, pis = G.GENERATORS[name](*jobs[name][1], compact=True)
        pv.prove(air, cfg, trace, pis)  # warm: buffers, plan, tables
        wall, wall_py, phases, kernels = [], [], [], []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            proof = pv.prove(air, cfg, trace, pis)
            wall_py.append((time.perf_counter() - t0) * 1e3)
            wall.append(pv.last_call_s * 1e3)  # starkhip_prove_compact alone: recording in, proof blob out
            phases.append(pv.last_timings())
            kernels.append(pv.last_kernel_timings())
        S.verify_stark_proof(air, cfg, proof)
        best = min(range(args.reps), key=lambda i: wall[i])
        out[S.AIR_NAMES[air]] = {"shape": list(trace.shape), "wall_ms": wall[best], "wall_with_numpy_copy_ms": wall_py[best],
                                 "phase_ms": phases[best], "kernel_ms": kernels[best], "host_ms": pv.last_host_timings(),
                                 "proof_bytes": int(proof.size) * 8}
    pv.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
