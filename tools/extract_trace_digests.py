#!/usr/bin/env python3
"""Run the reference's trace generators -- the TEXT of `generate_trace` and of every `fill_*` function and native they reach
(/root/reference/src/*.rs) -- on the reference's own test vectors with tools/rust_subset.py, and write
tests/golden/trace_digests.json: per AIR the inputs (u32 limbs), the trace's SHA-256 and one digest per block of 256 columns.

Runs in the build container only (needs /root/reference).  The product is not involved.  tests/test_trace_digests_cpu.py
feeds the same inputs to the product's generators (starkhip_trace_*) and must reproduce every digest, i.e. the product's trace
equals the reference's cell for cell -- the precondition for byte-identical proofs that "every constraint vanishes" does not
give (cells no constraint reads would still enter the Merkle leaves).
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rust_subset as R  # noqa: E402

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SRC = "/root/reference/src"
MODULES = ["native", "big_arithmetic", "utils", "fp", "fp2", "fp6", "fp12", "g1", "fp12_mul", "miller_loop", "calc_pairing_precomp",
           "final_exponentiate", "ecc_aggregate"]
BLOCK = 256
P_BLS = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


def limbs(v, n=12):
    return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def fp(v):
    return R.TupleStruct("Fp", (limbs(v % P_BLS),))


def fp2(a, b):
    return R.TupleStruct("Fp2", ([fp(a), fp(b)],))


def fp12(vals):
    return R.TupleStruct("Fp12", ([fp(v) for v in vals],))


def digests(m):
    a = np.ascontiguousarray(m, dtype="<u8")
    out = {"sha256": hashlib.sha256(a.tobytes()).hexdigest(), "block_columns": BLOCK, "blocks": []}
    for c in range(0, a.shape[1], BLOCK):
        out["blocks"].append(hashlib.sha256(np.ascontiguousarray(a[:, c:c + BLOCK]).tobytes()).hexdigest()[:12])
    return out


def splitmix(seed):
    state = seed & 0xFFFFFFFFFFFFFFFF

    def nxt():
        nonlocal state
        state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)
    return nxt


def random_fp(nxt):
    v = 0
    for _ in range(6):
        v = (v << 64) | nxt()
    return v % P_BLS


def run(interp, mod, ty, rows, args):
    m = interp.mods[mod]
    item = m.methods[(ty, "generate_trace")]
    t0 = time.time()
    tr = interp.call_fn(item, [{"num_rows": rows}] + args, f"{mod}.rs")
    print(f"{ty}: {tr.m.shape} in {time.time() - t0:.1f} s", file=sys.stderr)
    return tr.m


def main():
    out_path = os.path.join(ROOT, "tests", "golden", "trace_digests.json")
    argv = sys.argv[1:]
    if "--out" in argv:  # a partial result (one AIR per process); merge with --merge
        out_path = argv[argv.index("--out") + 1]
        del argv[argv.index("--out"):argv.index("--out") + 2]
    if argv and argv[0] == "--merge":
        doc = json.load(open(out_path))
        for part in argv[1:]:
            doc["airs"].update(json.load(open(part))["airs"])
        with open(out_path, "w") as f:
            json.dump(doc, f, indent=0, separators=(",", ":"))
            f.write("\n")
        print("merged", argv[1:], "into", out_path, file=sys.stderr)
        return
    only = set(argv)
    vec = json.load(open(os.path.join(ROOT, "tests", "golden", "native_vectors.json")))
    interp = R.Interp(SRC, MODULES)
    # The 1024- and 8192-row AIRs write 10^8 cells through utils.rs:3-19 (`assign_u32_in_series`, a loop of single stores): that
    # helper alone is replaced by a slice store (rust_subset.Interp.fast_assign).  FP12MulStark and PairingPrecompStark were
    # extracted with the interpreted loop; FP12MulStark gives the same digest both ways.
    interp.fast_assign = bool({n.split("#")[0] for n in only} & {"MillerLoopStark", "FinalExponentiateStark", "ECCAggStark"}) or any("#" in n for n in only)
    doc = {"generated_by": "tools/extract_trace_digests.py (the reference's generate_trace / fill_* source text run by tools/rust_subset.py)",
           "encoding": "trace as rows x columns of canonical Goldilocks values, little-endian u64, row-major; `blocks` = first 12 hex digits "
                       "of the SHA-256 of each block of 256 columns (all rows)", "airs": {}}
    if os.path.exists(out_path):
        doc["airs"] = json.load(open(out_path)).get("airs", {})
    nxt = splitmix(0x7ACE0001)
    cases = {}
    # FP12MulStark (src/aggregate_proof.rs:120-133): two random Fp12
    x12 = [random_fp(nxt) for _ in range(12)]
    y12 = [random_fp(nxt) for _ in range(12)]
    cases["FP12MulStark"] = ("fp12_mul", "FP12MulStark", 16, {"x": x12, "y": y12}, lambda: [fp12(x12), fp12(y12)])
    # the reference's BLS vector (src/native.rs:1480-1498): P = pk, Q = H(m)
    b = {k: int(v) for k, v in vec["bls_signature"].items()}
    q = {"qx": [b["hm_x1"], b["hm_x2"]], "qy": [b["hm_y1"], b["hm_y2"]], "qz": [b["hm_z1"], b["hm_z2"]]}
    # PairingPrecompStark (src/aggregate_proof.rs:23-54): generate_trace(x, y, z: [[u32; 12]; 2])
    cases["PairingPrecompStark"] = ("calc_pairing_precomp", "PairingPrecompStark", 1024, q,
                                    lambda: [[limbs(q["qx"][0]), limbs(q["qx"][1])], [limbs(q["qy"][0]), limbs(q["qy"][1])],
                                             [limbs(q["qz"][0]), limbs(q["qz"][1])]])

    # MillerLoopStark (src/aggregate_proof.rs:78-101): generate_trace(x, y: Fp, ell_coeffs) with the native precompute of Q
    def miller_args():
        ell = interp.call_fn(interp.mods["native"].fns["calc_pairing_precomp"], [fp2(*q["qx"]), fp2(*q["qy"]), fp2(*q["qz"])], "native.rs")
        return [fp(b["pk_x"]), fp(b["pk_y"]), ell]
    cases["MillerLoopStark"] = ("miller_loop", "MillerLoopStark", 1024, {"px": [b["pk_x"]], "py": [b["pk_y"]], **q}, miller_args)
    # FinalExponentiateStark (src/aggregate_proof.rs:150-165): the reference's own input `aa` (src/native.rs:1546-1563)
    aa = [int(v) for v in vec["final_exp_input_aa"]]
    cases["FinalExponentiateStark"] = ("final_exponentiate", "FinalExponentiateStark", 8192, {"x": aa}, lambda: [fp12(aa)])
    # ECCAggStark (src/aggregate_proof.rs:181-221): the reference's own aggregation vector (src/ecc_aggregate.rs:489-523), padded to
    # 512 operands the way tests/test_ecc_aggregate_cpu.py pads it (the last point repeated with its bit cleared)
    ev = json.load(open(os.path.join(ROOT, "tests", "golden", "ecc_aggregate_vector.json")))
    pts = [(int(x), int(y)) for x, y in ev["points"]]
    bits = [bool(v) for v in ev["bits"]]
    n_vec = len(pts)
    pts += [pts[-1]] * (512 - len(pts))
    bits += [False] * (512 - len(bits))
    cases["ECCAggStark"] = ("ecc_aggregate", "ECCAggStark", 8192,
                            {"points_x": [p_[0] for p_ in pts[:n_vec]], "points_y": [p_[1] for p_ in pts[:n_vec]],
                             "bits": [int(v) for v in bits[:n_vec]], "padding": ["the last point repeated with a cleared bit, up to 512 operands"]},
                            lambda: [[[fp(x), fp(y)] for x, y in pts], bits])
    # second inputs: the signature point as Q, another generator-independent P, a seeded (invertible) Fp12 for FinalExp
    nxt2 = splitmix(0x7ACE0002)
    q2 = {"qx": [b["s_x1"], b["s_x2"]], "qy": [b["s_y1"], b["s_y2"]], "qz": [b["s_z1"], b["s_z2"]]}
    cases["PairingPrecompStark#2"] = ("calc_pairing_precomp", "PairingPrecompStark", 1024, q2,
                                      lambda: [[limbs(q2["qx"][0]), limbs(q2["qx"][1])], [limbs(q2["qy"][0]), limbs(q2["qy"][1])],
                                               [limbs(q2["qz"][0]), limbs(q2["qz"][1])]])

    def miller_args2():
        ell = interp.call_fn(interp.mods["native"].fns["calc_pairing_precomp"], [fp2(*q2["qx"]), fp2(*q2["qy"]), fp2(*q2["qz"])], "native.rs")
        return [fp(b["gx"]), fp(P_BLS - b["gy"]), ell]
    cases["MillerLoopStark#2"] = ("miller_loop", "MillerLoopStark", 1024, {"px": [b["gx"]], "py": [P_BLS - b["gy"]], **q2}, miller_args2)
    x12b = [random_fp(nxt2) for _ in range(12)]
    cases["FinalExponentiateStark#2"] = ("final_exponentiate", "FinalExponentiateStark", 8192, {"x": x12b}, lambda: [fp12(x12b)])
    for name, (mod, ty, rows, inputs, mk) in cases.items():
        if only and name not in only:
            continue
        try:
            m = run(interp, mod, ty, rows, mk())
        except Exception as ex:
            for fr in getattr(ex, "rust_stack", []):
                print("   in", fr, file=sys.stderr)
            raise
        doc["airs"][name] = {"rows": rows, "columns": int(m.shape[1]), "inputs": {k: [str(v) for v in vals] for k, vals in inputs.items()},
                             **digests(m)}
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=0, separators=(",", ":"))
        f.write("\n")
    print("wrote", out_path, os.path.getsize(out_path), "bytes", file=sys.stderr)


if __name__ == "__main__":
    main()
