#!/usr/bin/env python3
"""A/B of the two constraint evaluators on one GPU: FinalExp (or --air) proven with the op-stream interpreter and with the
tiled evaluator (chunk sweep); proofs must be identical; prints the quotient kernel's HIP-event time for each."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--air", default="final_exp", choices=["final_exp", "miller", "precomp", "fp12_mul"])
    ap.add_argument("--chunks", default="0,2,4,8,16")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--debug", type=int, default=0, help="quotient_debug option (1: no tile loads, 2: no arithmetic) -- timing only")
    args = ap.parse_args()
    import torch
    import starky_bls12_381_amd as S
    from bls_util import random_fp12, native_vectors
    air = {"final_exp": S.AIR_FINAL_EXP, "miller": S.AIR_MILLER_LOOP, "precomp": S.AIR_PAIRING_PRECOMP, "fp12_mul": S.AIR_FP12_MUL}[args.air]
    cfg = S.StarkConfig.for_air(air)
    if air == S.AIR_FINAL_EXP:
        trace, pis = S.trace_final_exp(random_fp12(0x5EED0001))
    elif air == S.AIR_FP12_MUL:
        trace, pis = S.trace_fp12_mul(random_fp12(1), random_fp12(2))
    else:
        v = native_vectors()
        from bls_util import fp_arr
        g2 = [np.concatenate([fp_arr(*[int(x) for x in [c]]) for c in pt]) for pt in v["hm"]] if False else None
        raise SystemExit("use tests for the small AIRs")
    n = trace.shape[0]
    d = torch.from_numpy(trace.view(np.int64)).cuda().t().contiguous()
    del trace
    pv = S.Prover(0)
    out = {}
    ref = None
    for impl, chunks in [(1, 0)] + [(0, int(c)) for c in args.chunks.split(",")]:
        pv.set_option("quotient_impl", impl)
        if impl == 0:
            pv.set_option("quotient_chunks", chunks)
            if args.debug:  # the option exists in `make DEBUG_KNOBS=1` builds only
                pv.set_option("quotient_debug", args.debug)
        ts = []
        for r in range(args.reps):
            try:
                proof = pv.prove_device(air, cfg, d.data_ptr(), n, pis, layout=1, keep=(r == 0))
            except S.StarkhipError:
                proof = None  # debug modes compute garbage: the quotient is not divisible
            if r == 0:
                if ref is None:
                    ref = proof
                    S.verify_stark_proof(air, cfg, proof)
                same = proof is not None and bool(np.array_equal(ref, proof))
            ts.append(pv.last_kernel_timings()["quotient_eval"])
        out[f"impl{impl}_chunks{chunks}"] = {"quotient_ms": [round(t, 2) for t in ts], "identical_to_interpreter": same,
                                             "phase_ms": {k: round(v, 2) for k, v in pv.last_timings().items()}}
        print(f"impl {impl} chunks {chunks}: quotient {ts} identical {same}", flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
