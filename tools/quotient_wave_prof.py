#!/usr/bin/env python3
"""Per-wave clocks of the tiled quotient evaluator (round 6): how long the eight evaluating waves of a workgroup sit at the tile barrier, from s_memtime stamps in the profiling variant of the kernel.

    make variant NAME=qprof DEFS=-DSTARKHIP_QT_PROF
    STARKHIP_LIBRARY=build/qprof/libstarkhip_qprof.so python3 tools/quotient_wave_prof.py > gpurun_out/quotient_wave_prof.json
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import starky_bls12_381_amd as S
    from starky_bls12_381_amd import api
    from bls_util import random_fp12
    air = S.AIR_FINAL_EXP
    cfg = S.StarkConfig.for_air(air)
    trace, pis = S.trace_final_exp(random_fp12(0x5EED0001))
    n = trace.shape[0]
    d = torch.from_numpy(trace.view(np.int64)).cuda().t().contiguous()
    del trace
    pv = S.Prover(0)
    for r in range(2):
        pv.prove_device(air, cfg, d.data_ptr(), n, pis, layout=1)
    ms = pv.last_kernel_timings()["quotient_eval"]
    waves = 9  # the profile's row per workgroup: eight evaluating waves (+ an unused row; the SMALL_N kernels' producer)
    chunks = 16
    blocks = (n << 2) // 64
    wgs = chunks * blocks
    buf = np.zeros(wgs * waves * 4, dtype=np.uint64)
    fn = api.lib.starkhip_debug_qt_prof
    fn.argtypes = [C.c_void_p, C.c_size_t]
    rc = fn(buf.ctypes.data, buf.size)
    assert rc == 0, rc
    p = buf.reshape(wgs, waves, 4).astype(np.float64)
    ev = p[:, :8, :]
    tot = ev[:, :, 0]
    bar = ev[:, :, 1]
    out = {
        "quotient_ms": ms,
        "workgroups": wgs,
        "evaluator_total_cycles_mean": tot.mean(),
        "evaluator_barrier_fraction_mean": (bar / tot).mean(),
        "evaluator_barrier_fraction_by_wave": (bar.sum(0) / tot.sum(0)).round(4).tolist(),
        "evaluator_barrier_fraction_best_wave_per_wg": (bar / tot).min(1).mean(),
        "evaluator_barrier_fraction_worst_wave_per_wg": (bar / tot).max(1).mean(),
        "barriers_per_wave": ev[:, :, 2].mean(),
        "by_chunk_barrier_fraction": (bar.reshape(chunks, blocks, 8).sum((1, 2)) / tot.reshape(chunks, blocks, 8).sum((1, 2))).round(4).tolist(),
        "by_chunk_total_cycles": tot.reshape(chunks, blocks, 8).mean((1, 2)).round(0).tolist(),
    }
    print(json.dumps(out, indent=1))
    # per-tile busy cycles of the workgroups with blockIdx.x == 0 (one per chunk): gpurun_out/quotient_tile_prof.npy [64][8][192]
    tb = np.zeros(64 * 8 * 192, dtype=np.uint64)
    fn2 = api.lib.starkhip_debug_qt_tile_prof
    fn2.argtypes = [C.c_void_p, C.c_size_t]
    assert fn2(tb.ctypes.data, tb.size) == 0
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", "quotient_tile_prof.npy"), tb.reshape(64, 8, 192))


if __name__ == "__main__":
    main()
