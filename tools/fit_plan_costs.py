#!/usr/bin/env python3
"""Least-squares fit of the planner's cost model (csrc/quotient_plan.h: REC_COST, PAIR_COST, ... PIECE_COST) to the per-tile clocks of
the quotient kernel's profiling variant.

    build/plan_streams 3 16 > /tmp/streams.csv                                   # census of the record streams (tools/plan_streams.cpp)
    STARKHIP_LIBRARY=build/qprof/libstarkhip_qprof.so python3 tools/quotient_wave_prof.py   # gpurun_out/quotient_tile_prof.npy
    python3 tools/fit_plan_costs.py /tmp/streams.csv gpurun_out/quotient_tile_prof.npy
"""
import sys

import numpy as np
import pandas as pd


def main():
    d = pd.read_csv(sys.argv[1])
    tp = np.load(sys.argv[2]).astype(float)  # [chunk < 64][wave][tile < 192]
    d["cyc"] = [tp[c, w, t] if (c < 64 and t < 192) else np.nan for c, w, t in zip(d.chunk, d.wave, d.tile_in_chunk)]
    d = d.dropna()
    dd = d[d.tile_in_chunk > 0].copy()  # a chunk's first tile includes the wave's start-up
    feat = [f for f in ["plain4", "plain_tail", "fast_pairs2", "fast_pair_odd", "dpairs2", "special", "piece_ends", "direct", "slot_cells", "noop"] if dd[f].sum() > 0]
    X = np.c_[dd[feat].values.astype(float), np.ones(len(dd))]
    y = dd.cyc.values
    coef = np.linalg.lstsq(X, y, rcond=None)[0]
    pred = X @ coef
    print(f"{len(dd)} (tile, wave) phases; cycles of the wave's clock per item:")
    for f, c in zip(feat + ["per tile"], coef):
        n = dd[f].sum() if f in dd else len(dd)
        print(f"  {f:14s} {c:9.1f}   x {int(n):8d} = {c * n / 1e6:7.2f} M")
    print(f"  R^2 = {1 - ((y - pred) ** 2).sum() / ((y - y.mean()) ** 2).sum():.4f}; busy cycles in all {y.sum() / 1e6:.1f} M")
    g = dd.groupby(["chunk", "tile_in_chunk"]).cyc.agg(["max", "mean"])
    print(f"  per tile: sum of the busiest wave {g['max'].sum() / 1e6:.2f} M, sum of the mean {g['mean'].sum() / 1e6:.2f} M, ratio {g['max'].sum() / g['mean'].sum():.3f}")
    bw = dd.groupby("wave").cyc.sum() / 1e6
    print("  busy by wave (M cycles):", bw.round(2).tolist())


if __name__ == "__main__":
    main()
