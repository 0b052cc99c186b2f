#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace results.db: how much of the wall time of the run's busiest window had at least one
kernel executing, and how much had two or more (two proofs in flight).  usage: kernel_overlap.py <results.db> [from_fraction to_fraction]   (default: the window is found automatically)"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    cand = [t for t in tabs if "kernel_dispatch" in t]
    rows = None
    for t in cand:
        cols = [r[1] for r in cur.execute(f"pragma table_info({t})")]
        if "start" in cols and "end" in cols:
            rows = list(cur.execute(f"select start, end from {t}"))
            break
    if not rows:
        raise SystemExit(f"no kernel dispatch table with start/end in {tabs}")
    rows.sort()
    t0, t1 = rows[0][0], max(e for _, e in rows)
    # sum of kernel durations per 250 ms bin
    bins = {}
    for s_, e_ in rows:
        b = int((s_ - t0) // 250e6)
        while s_ < e_:
            edge = t0 + (b + 1) * 250e6
            bins[b] = bins.get(b, 0) + min(e_, edge) - s_
            s_ = min(e_, edge)
            b += 1
    dens = [bins.get(b, 0) / 250e6 for b in range(max(bins) + 1)]
    if len(sys.argv) > 2:
        lo = t0 + (t1 - t0) * float(sys.argv[2])
        hi = t0 + (t1 - t0) * (float(sys.argv[3]) if len(sys.argv) > 3 else 1.0)
    else:
        # the two-in-flight window: the longest run of bins with more than 1.3 kernels running on average, minus one
        # bin at each end (ramp-up, and bench.py's trailing one-proof-at-a-time pass)
        best, cur = (0, 0), None
        for i, d in enumerate(dens + [0.0]):
            if d > 1.3:
                cur = i if cur is None else cur
            elif cur is not None:
                if i - cur > best[1] - best[0]:
                    best = (cur, i)
                cur = None
        if best[1] - best[0] < 3:
            raise SystemExit(f"no two-in-flight window found; densities: {dens}")
        lo, hi = t0 + (best[0] + 1) * 250e6, t0 + (best[1] - 1) * 250e6
    ev = []
    for s, e in rows:
        if e <= lo or s >= hi:
            continue
        ev.append((max(s, lo), 1))
        ev.append((min(e, hi), -1))
    ev.sort()
    depth, last, busy1, busy2 = 0, lo, 0, 0
    for t, d in ev:
        if depth >= 1:
            busy1 += t - last
        if depth >= 2:
            busy2 += t - last
        depth += d
        last = t
    span = hi - lo
    print(f"window {span / 1e6:.1f} ms: at least one kernel running {busy1 / span:.3f}, two or more {busy2 / span:.3f}, idle {1 - busy1 / span:.3f}")
    print("sum of kernel durations / wall per 250 ms bin:", " ".join(f"{d:.2f}" for d in dens))


if __name__ == "__main__":
    main()
