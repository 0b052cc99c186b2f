#!/usr/bin/env python3
"""Turn rocprofv3's sqlite output (ROCm 7.2 default format) into the small CSV summaries kept under profiles/.

  rocprof_export.py stats  <results.db> <out.csv>          # what --kernel-trace --stats prints: per-kernel calls / total / average ns
  rocprof_export.py pmc    <results.db> <out.csv>          # per kernel: launches, average counter value, average duration
  rocprof_export.py bygrid <results.db> <out.csv>          # --kernel-trace: calls / average per (kernel, grid size) -- the leaf hash
                                                           # runs once per proof over the trace (2 048 waves) and once over the two
                                                           # quotient polynomials; `stats` lumps the two
"""
import csv
import sqlite3
import sys


def short(name):
    name = name.replace("starkhip::", "")
    cut = name.find("(")
    name = name if cut < 0 else name[:cut]
    return name[:90]


def main():
    mode, db, out = sys.argv[1:4]
    con = sqlite3.connect(db)
    cur = con.cursor()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        if mode == "stats":
            # rocprofv3's top_kernels view reports microseconds (6 leaf-hash launches: 3 x ~180 ms + 3 small trees)
            w.writerow(["kernel", "calls", "total_us", "average_us", "percent"])
            for name, calls, total, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
                w.writerow([short(name), calls, f"{total:.0f}", f"{avg:.0f}", f"{pct:.3f}"])
        elif mode == "bygrid":
            # the leaf hash runs with the same grid over the trace (73 527 columns) and over the two quotient polynomials: the
            # long launches -- at least half the longest -- are listed on their own
            w.writerow(["kernel", "grid_x", "workgroup_x", "calls", "average_us", "total_us", "long_calls", "long_average_us", "max_us"])
            groups = {}
            for name, gx, wx, dur in cur.execute("select name, grid_x, workgroup_x, end - start from kernels"):
                groups.setdefault((short(name), gx, wx), []).append(dur)
            for (name, gx, wx), d in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
                long_ = [x for x in d if 2 * x >= max(d)]
                w.writerow([name, gx, wx, len(d), f"{sum(d) / len(d) / 1e3:.0f}", f"{sum(d) / 1e3:.0f}", len(long_),
                            f"{sum(long_) / len(long_) / 1e3:.0f}", f"{max(d) / 1e3:.0f}"])
        else:
            w.writerow(["kernel", "counter", "launches", "average_value", "average_duration_ns"])
            q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                 "group by kernel_name, counter_name order by sum(duration) desc")
            for name, ctr, n, val, dur in cur.execute(q):
                w.writerow([short(name), ctr, n, f"{val:.1f}", f"{dur:.0f}"])


if __name__ == "__main__":
    main()
