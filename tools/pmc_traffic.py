#!/usr/bin/env python3
"""profiles/pmc_traffic_latest.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of
`python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --inflight 1`): per-launch HBM-side bytes of the three heavy
kernels, trace-commitment launches only.  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950.
usage: pmc_traffic.py <out.json> <fetch_results.db> <write_results.db> [<fetch2.db> <write2.db> ...]"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_fingerprint import KERNEL_SOURCES, kernel_fingerprint  # noqa: E402


def main():
    out_path, dbs = sys.argv[1], sys.argv[2:]
    res = {}
    # pairs of (FETCH_SIZE db, WRITE_SIZE db); a kernel is taken from the first pair that saw it
    for pair in range(0, len(dbs) - 1, 2):
        seen = {}
        for f, ctr in ((dbs[pair], "FETCH_SIZE"), (dbs[pair + 1], "WRITE_SIZE")):
            cur = sqlite3.connect(f).cursor()
            for k in ("leaf_hash_lane_kernel", "leaf_hash_pair_kernel", "leaf_hash_kernel", "quotient_tiles_kernel", "lde_columns_wave_kernel", "lde_columns_v2_kernel"):
                rows = list(cur.execute("select value, duration from counters_collection where kernel_name like ? and counter_name = ? order by duration desc",
                                        ("%" + k + "%", ctr)))
                if not rows:
                    continue
                big = [r for r in rows if r[1] > 0.5 * rows[0][1]]
                # a trace's LDE is SEVERAL launches (3/4, 3/16, 3/64 and the last 1/64 of the columns: prover.hip run_lde_trace): the bytes
                # of all of them per proof (the quotient commitment's four-column launches are in the sum too: 0.005 % of it)
                # (lde_columns_wave_kernel: every launch of a trace has the same persistent grid; the first launch of a proof -- 3/4 of the
                # columns -- is the `big` one, so len(big) = proofs)
                part = rows if k.startswith("lde_columns") else big
                seen.setdefault(k, {})[ctr] = (sum(r[0] for r in part) / len(big), len(big), sum(r[1] for r in part) / len(big) / 1e6)
        for k, v in seen.items():
            if k not in res and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                res[k] = v
    out = {}
    for k, v in res.items():
        fetch = v["FETCH_SIZE"][0] * 1024 * 2  # gfx950: 128-byte requests are tallied as 64 bytes
        write = v["WRITE_SIZE"][0] * 1024
        out[k] = {"fetch_bytes_corrected": fetch, "write_bytes": write, "traffic_bytes": fetch + write, "fetch_size_raw_KB": v["FETCH_SIZE"][0],
                  "write_size_raw_KB": v["WRITE_SIZE"][0], "launches_averaged": v["FETCH_SIZE"][1], "avg_ms_under_pmc": v["FETCH_SIZE"][2],
                  "source_sha256": kernel_fingerprint(k) if k in KERNEL_SOURCES else None}
    out["_source"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline "
                      "--no-boundary --inflight 1` (quad-form leaf hash, LDE, quotient) and of `python3 bench.py --steps 8 --warmup 1 --no-cpu-baseline "
                      "--no-boundary --no-solo` (default proofs in flight: leaf_hash_lane_kernel; counter passes serialise the launches, the bytes per "
                      "launch are the same), trace-commitment launches only; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-byte "
                      "requests as 64 bytes); Infinity-Cache hits are included in these memory-side counters")
    json.dump(out, open(out_path, "w"), indent=1)
    for k, v in out.items():
        if k != "_source":
            print(k, round(v["traffic_bytes"] / 1e9, 2), "GB", round(v["avg_ms_under_pmc"], 1), "ms")


if __name__ == "__main__":
    main()
