#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace results.db: the kernels longer than a threshold in the last `window` seconds of the run, one
line each (start and end in ms relative to the window, duration, queue, name).
usage: kernel_timeline.py <results.db> [window_s=0.7] [min_ms=1.0]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    window = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
    min_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    cur = sqlite3.connect(db).cursor()
    views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    src = "kernels" if "kernels" in views else next(v for v in views if "kernel_dispatch" in v)
    cols = [r[1] for r in cur.execute(f"pragma table_info({src})")]
    name_col = "name" if "name" in cols else ("kernel_name" if "kernel_name" in cols else None)
    q_col = next((c for c in ("stream_id", "queue_id", "queue") if c in cols), None)
    sel = f"select start, end, {name_col or 'kernel_id'}, {q_col or '0'} from {src}"
    rows = sorted(cur.execute(sel))
    t_end = max(r[1] for r in rows)
    lo = t_end - window * 1e9
    print(f"# source {src}; columns {cols}")
    for s, e, name, q in rows:
        if e < lo or (e - s) < min_ms * 1e6:
            continue
        name = str(name).replace("starkhip::", "")
        cut = name.find("(")
        print(f"{(s - lo) / 1e6:9.1f} {(e - lo) / 1e6:9.1f} {(e - s) / 1e6:8.1f} ms  q{q}  {name[:cut] if cut > 0 else name[:60]}")


if __name__ == "__main__":
    main()
