#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace of the default bench run (several proofs in flight): how the trace commitments actually ran --
launches of leaf_hash_lane_kernel (groups, one lane per leaf) and of leaf_hash_pair_kernel (a big commitment on its own, two lanes per
leaf; leaf_hash_kernel, the quad form, in libraries before it), their durations and how many ran
side by side.  bench.py quotes the result next to its one-proof-in-flight roofline figures (profiles/lane_group_latest.json, with
the SHA-256 of the kernel sources like pmc_traffic_latest.json).

    python tools/lane_group_stats.py <results.db> <out.json>
"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tools.kernel_fingerprint import kernel_fingerprint  # noqa: E402


def main():
    db, out = sys.argv[1:3]
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    res = {}
    for key, pat in (("leaf_hash_lane_kernel", "leaf_hash_lane_kernel"), ("leaf_hash_pair_kernel", "leaf_hash_pair_kernel"), ("leaf_hash_kernel", "leaf_hash_kernel")):
        ls = [(s, e) for n, s, e in rows if pat in n and "multi" not in n]
        if not ls:
            continue
        longest = max(e - s for s, e in ls)
        big = [(s, e) for s, e in ls if 2 * (e - s) >= longest]   # the trace commitments (the quotient's are short launches of the same kernels)
        conc = [sum(1 for s2, e2 in big if s2 <= (s + e) / 2 <= e2) for s, e in big]
        full = [(e - s) / 1e6 for (s, e), c in zip(big, conc) if c >= 4]
        res[key] = {"trace_commitment_launches": len(big), "average_ms": sum(e - s for s, e in big) / len(big) / 1e6,
                    "side_by_side_average": sum(conc) / len(conc), "launches_in_groups_of_four": len(full),
                    "average_ms_in_groups_of_four": (sum(full) / len(full)) if full else None}
    res["source_sha256"] = kernel_fingerprint("leaf_hash_lane_kernel")
    res["_source"] = "rocprofv3 --kernel-trace of `python3 bench.py --steps 24 --warmup 2 --no-cpu-baseline --no-boundary` (default proofs in flight)"
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
