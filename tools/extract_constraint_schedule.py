#!/usr/bin/env python3
"""Extract the ordered constraint schedule of the five AIRs from the reference's Rust source (text) and write
tests/golden/constraint_schedule.json.

Runs in the build container only (needs /root/reference).  The product is not involved: the source text of
`eval_packed_generic` and of every `add_*_constraints` function it reaches is interpreted symbolically by
tools/rust_subset.py, which yields, in the reference's own order, every
`yield_constr.constraint{,_transition,_first_row,_last_row}(poly)` as (kind, polynomial over local/next/public-input
variables with Goldilocks coefficients).  The fixture holds, per AIR:

  * columns, public inputs, constraint degree, number of rows (the reference's constants), K = number of constraints;
  * `segments`: the schedule cut at every change of gadget call stack — for each segment [index into `stacks`,
    number of constraints, run-length-encoded kinds, first 16 hex digits of the SHA-256 of its constraints];
    `stacks` = lists of indices into `frames`, a frame = "function@definition file:line<-call site file:line";
  * `gadgets`: per gadget function {constraints yielded by one invocation, nested gadgets included: number of such
    invocations} (SURVEY App. B.2's table, recomputed);
  * `sha256`: digest over all K constraints (kind byte + canonical polynomial).

tests/test_constraint_schedule_cpu.py decodes the product's flat constraint program (starkhip_air_program), expands
every constraint to the same canonical form and must reproduce every segment digest: a dropped, swapped, mis-kinded
or mis-transcribed constraint changes a digest and the failing segment names the reference call site.
"""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rust_subset as R  # noqa: E402

SRC = "/root/reference/src"
MODULES = ["native", "big_arithmetic", "utils", "fp", "fp2", "fp6", "fp12", "g1", "fp12_mul", "miller_loop", "calc_pairing_precomp",
           "final_exponentiate", "ecc_aggregate"]
# AIR name, module, Stark type, num_rows passed to `new` by the drivers (src/aggregate_proof.rs:35,78,124,158,189)
AIRS = [
    ("FP12MulStark", "fp12_mul", "FP12MulStark", 16),
    ("PairingPrecompStark", "calc_pairing_precomp", "PairingPrecompStark", 1024),
    ("MillerLoopStark", "miller_loop", "MillerLoopStark", 1024),
    ("FinalExponentiateStark", "final_exponentiate", "FinalExponentiateStark", 8192),
    ("ECCAggStark", "ecc_aggregate", "ECCAggStark", 8192),
]
KIND_BYTE = {"plain": 0, "transition": 1, "first": 2, "last": 3}


def rle(kinds):
    out = []
    for k in kinds:
        if out and out[-1][0] == k:
            out[-1][1] += 1
        else:
            out.append([k, 1])
    return out


def extract(interp, name, mod, ty, num_rows):
    m = interp.mods[mod]
    cols = interp.const_value(m, "COLUMNS") if "COLUMNS" in m.consts else interp.const_value(m, "TOTAL_COLUMNS")
    pis = interp.const_value(m, "PUBLIC_INPUTS")
    item = m.methods[(ty, "eval_packed_generic")]
    interp.records.clear()
    interp.stack.clear()
    interp.fn_counts.clear()
    selfv = {"num_rows": num_rows}
    t0 = time.time()
    interp.call_fn(item, [selfv, R.Vars(cols, pis), R.Consumer()], f"{mod}.rs")
    gadgets = {f"{fn}@{where}": {str(n): c for n, c in sorted(cnt.items())} for (fn, where), cnt in interp.fn_counts.items()}
    degree = interp.call_fn(m.methods[(ty, "constraint_degree")], [selfv], f"{mod}.rs")
    recs = interp.records
    frames, frame_idx, stacks, stack_idx, segments = [], {}, [], {}, []
    whole = hashlib.sha256()
    i = 0
    max_deg = 0
    while i < len(recs):
        j = i
        sid = recs[i][2]
        h = hashlib.sha256()
        kinds = []
        while j < len(recs) and recs[j][2] == sid:
            kind, sym, _ = recs[j]
            c = sym.canonical()
            d = max((sum(1 for v in mm if not v >> 31) for mm in sym.t), default=0) + (1 if kind in ("first", "last") else 0)
            max_deg = max(max_deg, d)
            for hh in (h, whole):
                hh.update(bytes([KIND_BYTE[kind]]))
                hh.update(len(c).to_bytes(4, "little"))
                hh.update(c)
            kinds.append(KIND_BYTE[kind])
            j += 1
        st = []
        for fn, where, site in interp.stack_list[sid]:
            fr = f"{fn}@{where}<-{site}"
            if fr not in frame_idx:
                frame_idx[fr] = len(frames)
                frames.append(fr)
            st.append(frame_idx[fr])
        st = tuple(st)
        if st not in stack_idx:
            stack_idx[st] = len(stacks)
            stacks.append(list(st))
        segments.append([stack_idx[st], j - i, rle(kinds), h.hexdigest()[:16]])
        i = j
    print(f"{name}: {len(recs)} constraints, {len(segments)} segments, degree {degree} (max seen {max_deg}), "
          f"{time.time() - t0:.1f} s", file=sys.stderr)
    return {
        "air": name, "source": f"src/{mod}.rs:{item.line}", "columns": cols, "public_inputs": pis, "degree": degree,
        "max_constraint_degree_seen": max_deg, "rows": num_rows, "n_constraints": len(recs), "sha256": whole.hexdigest(),
        "gadgets": gadgets, "frames": frames, "stacks": stacks, "segments": segments,
    }


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden",
                                                                  "constraint_schedule.json")
    only = set(sys.argv[2:])
    interp = R.Interp(SRC, MODULES)
    airs = []
    for name, mod, ty, rows in AIRS:
        if only and name not in only:
            continue
        airs.append(extract(interp, name, mod, ty, rows))
    doc = {
        "generated_by": "tools/extract_constraint_schedule.py (symbolic interpretation of the reference's Rust source text)",
        "encoding": "variable codes: local column c = c, next-row column c = c | 1<<30, public input i = i | 1<<31; a polynomial = "
                    "monomials sorted as tuples of sorted codes, each as u32 degree, u32 codes, u64 coefficient (little endian, "
                    "canonical mod 2^64-2^32+1); a segment digest = SHA-256 over (kind byte 0 plain/1 transition/2 first/3 last, "
                    "u32 length, polynomial) of its constraints in order",
        "airs": airs,
    }
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=0, separators=(",", ":"))
        f.write("\n")
    print("wrote", out_path, os.path.getsize(out_path), "bytes", file=sys.stderr)


if __name__ == "__main__":
    main()
