"""The product's constraint programs against the schedule extracted from the reference's Rust source.

tests/golden/constraint_schedule.json is produced by tools/extract_constraint_schedule.py, which interprets the text of
the reference's `eval_packed_generic` / `add_*_constraints` functions symbolically (tools/rust_subset.py) — the product
plays no part in it.  Here every constraint of the product's flat program (starkhip_air_program: kind, gates, terms) is
expanded to the same canonical polynomial and the digests must agree segment by segment, so constraint ORDER, KIND and
BODY of all 722 581 constraints are pinned to /root/reference/src/{fp12_mul.rs:58-99, calc_pairing_precomp.rs:376-2123,
miller_loop.rs:191-411,644-677, final_exponentiate.rs:283-827,907-1136, ecc_aggregate.rs:92-268} and the gadget files."""
import hashlib
import json
import os

import numpy as np
import pytest

import air_blob as B
import oracle_lib as O
import starky_bls12_381_amd as S

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "constraint_schedule.json")) as f:
    SCHEDULE = {a["air"]: a for a in json.load(f)["airs"]}

AIR_IDS = {
    "FP12MulStark": S.AIR_FP12_MUL,
    "PairingPrecompStark": S.AIR_PAIRING_PRECOMP,
    "MillerLoopStark": S.AIR_MILLER_LOOP,
    "FinalExponentiateStark": S.AIR_FINAL_EXP,
    "ECCAggStark": S.AIR_ECC_AGGREGATE,
}


def _stack(sched, seg):
    return " / ".join(sched["frames"][i] for i in sched["stacks"][seg[0]])


@pytest.mark.parametrize("name", sorted(AIR_IDS))
def test_program_reproduces_reference_schedule(name):
    sched = SCHEDULE[name]
    air = AIR_IDS[name]
    prog = B.parse_blob(S.air_program(air))
    assert prog["n_cols"] == sched["columns"] == S.air_columns(air)
    assert prog["n_pis"] == sched["public_inputs"]
    assert prog["degree"] == sched["degree"]
    assert prog["n_constraints"] == sched["n_constraints"]
    assert S.air_default_rows(air) == sched["rows"]
    it = B.constraints(prog)
    whole = hashlib.sha256()
    k = 0
    for seg in sched["segments"]:
        _, n, kinds_rle, digest = seg
        want_kinds = [kk for kk, c in kinds_rle for _ in range(c)]
        h = hashlib.sha256()
        for j in range(n):
            kind, gates, terms = next(it)
            assert kind == want_kinds[j], f"constraint {k + j}: kind {B.KINDS[kind]} != {B.KINDS[want_kinds[j]]} in {_stack(sched, seg)}"
            c = B.canonical(B.expand(gates, terms))
            rec = bytes([kind]) + len(c).to_bytes(4, "little") + c
            h.update(rec)
            whole.update(rec)
        assert h.hexdigest()[:16] == digest, f"constraints {k}..{k + n - 1} differ from the reference in {_stack(sched, seg)}"
        k += n
    assert next(it, None) is None
    assert k == sched["n_constraints"]
    assert whole.hexdigest() == sched["sha256"]


# constraints yielded by one invocation, nested gadgets included
GADGET_COUNTS = {
    "add_multiplication_constraints": 397, "add_addition_constraints": 24, "add_subtraction_constraints": 24,
    "add_reduce_constraints": 541, "add_range_check_constraints": 36, "add_addition_fp_constraints": 12,
    "add_subtraction_fp_constraints": 12, "add_negate_fp2_constraints": 48, "add_g1_addition_constraints": 5530, "add_fp_single_multiply_constraints": 12,
    "add_fp_reduce_single_constraints": 72, "add_fp2_mul_constraints": 3150, "add_fp2_fp_mul_constraints": 2080,
    "add_multiply_by_b_constraints": 2284, "add_addition_with_reduction_constranints": 264,
    "add_subtraction_with_reduction_constranints": 336, "add_non_residue_multiplication_constraints": 348,
    "add_fp4_sq_constraints": 11406, "add_fp2_forbenius_map_constraints": 1048, "add_fp6_multiplication_constraints": 25188,
    "add_multiply_by_1_constraints": 10062, "add_multiply_by_01_constraints": 18498, "add_fp6_forbenius_map_constraints": 9690,
    "add_negate_fp6_constraints": 144, "add_multiply_by_014_constraints": 52830, "add_fp12_multiplication_constraints": 82128,
    "add_cyclotomic_sq_constraints": 51534, "add_cyclotomic_exp_constraints": 134814, "add_fp12_forbenius_map_constraints": 29267,
    "add_fp12_conjugate_constraints": 288, "add_miller_loop_constraints": 140510,
}


def test_gadget_counts_match_survey():
    seen = {}
    for sched in SCHEDULE.values():
        for key, counts in sched["gadgets"].items():
            seen.setdefault(key.split("@")[0], set()).update(int(n) for n in counts)
    for fn, n in GADGET_COUNTS.items():
        assert seen[fn] == {n}, (fn, seen[fn])


def test_schedule_catches_a_swapped_constraint():
    """Sanity of the method: swapping two neighbouring constraints of equal kind changes the segment digest."""
    sched = SCHEDULE["FP12MulStark"]
    prog = B.parse_blob(S.air_program(S.AIR_FP12_MUL))
    recs = []
    n = sched["segments"][0][1]
    it = B.constraints(prog)
    for _ in range(n):
        kind, gates, terms = next(it)
        c = B.canonical(B.expand(gates, terms))
        recs.append(bytes([kind]) + len(c).to_bytes(4, "little") + c)
    assert hashlib.sha256(b"".join(recs)).hexdigest()[:16] == sched["segments"][0][3]
    recs[0], recs[1] = recs[1], recs[0]
    assert hashlib.sha256(b"".join(recs)).hexdigest()[:16] != sched["segments"][0][3]


# ---- evaluators on random (non-trace) frames against the reference-pinned polynomials
def _rand(rng, n):
    return np.array([int(rng.integers(0, B.P, dtype=np.uint64)) for _ in range(n)], dtype=np.uint64)


def _ext_mul(a, b):
    return ((a[0] * b[0] + 7 * a[1] * b[1]) % B.P, (a[0] * b[1] + a[1] * b[0]) % B.P)


def _ext_eval(poly, local, nxt, pis):
    acc = (0, 0)
    for m, c in poly.items():
        t = (c, 0)
        for v in m:
            if v & B.PI_FLAG:
                p = int(pis[v & 0x7FFFFFFF])
                t = (t[0] * p % B.P, t[1] * p % B.P)
            else:
                t = _ext_mul(t, (nxt if v & B.REF_NEXT else local)[v & B.COL_MASK])
        acc = ((acc[0] + t[0]) % B.P, (acc[1] + t[1]) % B.P)
    return acc


@pytest.mark.parametrize("name", ["FP12MulStark", "ECCAggStark"])
def test_evaluators_on_a_random_frame(name):
    """Not a trace: random local / next rows, random public inputs, random masks.  The polynomials come from the product's
    program, which test_program_reproduces_reference_schedule pins to the reference text; evaluated here with Python integers
    they must equal (i) the oracle's per-constraint values and (ii) the product's host evaluator (grouped Horner fold over the
    quadratic extension, csrc/air_eval.h) folded with two random alphas exactly as ConstraintConsumer folds."""
    air = AIR_IDS[name]
    prog = B.parse_blob(S.air_program(air))
    rng = np.random.default_rng(0xA1B2 + air)
    C_, npis = prog["n_cols"], prog["n_pis"]
    local, nxt, pis, masks = _rand(rng, C_), _rand(rng, C_), _rand(rng, npis), _rand(rng, 4)
    each = O.eval_frame(S.air_program(air), local, nxt, pis, masks)
    polys = [(kind, B.expand(g, t)) for kind, g, t in B.constraints(prog)]
    for k, (kind, poly) in enumerate(polys):
        want = B.evaluate(poly, local, nxt, pis) * int(masks[kind]) % B.P
        assert int(each[k]) == want, f"oracle evaluator differs at constraint {k}"
    # extension-field frame for the product's evaluator
    le = [(int(a), int(b)) for a, b in zip(local, _rand(rng, C_))]
    ne = [(int(a), int(b)) for a, b in zip(nxt, _rand(rng, C_))]
    me = [(int(a), int(b)) for a, b in zip(masks, _rand(rng, 4))]
    al = [(int(a), int(b)) for a, b in zip(_rand(rng, 2), _rand(rng, 2))]
    acc = [(0, 0), (0, 0)]
    for kind, poly in polys:
        v = _ext_mul(_ext_eval(poly, le, ne, pis), me[kind])
        for j in range(2):
            a = _ext_mul(acc[j], al[j])
            acc[j] = ((a[0] + v[0]) % B.P, (a[1] + v[1]) % B.P)
    got = S.air_eval_frame(air, np.array(le, dtype=np.uint64), np.array(ne, dtype=np.uint64), pis, np.array(me, dtype=np.uint64),
                           np.array(al, dtype=np.uint64))
    assert [tuple(int(x) for x in r) for r in got] == acc
