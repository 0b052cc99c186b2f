"""The product's trace generators against the reference's, cell for cell.

tests/golden/trace_digests.json comes from tools/extract_trace_digests.py, which runs the TEXT of the reference's
`generate_trace` functions (and every `fill_*` function and native they reach) in the Rust-subset interpreter on fixed inputs
and records SHA-256 digests of the resulting matrices.  Here the same inputs go through starkhip_trace_*; every digest must be
reproduced.  "Every constraint vanishes on the trace" (test_airs_cpu.py) does not imply this: cells no constraint reads
would still enter the Merkle leaves and change the proof bytes."""
import hashlib
import json
import os

import numpy as np
import pytest

import starky_bls12_381_amd as S

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "trace_digests.json")))


def limbs(v, n=12):
    return [(int(v) >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def fp12_arr(vals):
    return np.array([w for v in vals for w in limbs(v)], dtype=np.uint32)


def fp2_arr(vals):
    return np.array([w for v in vals for w in limbs(v)], dtype=np.uint32)


def check(name, trace):
    g = GOLD["airs"][name]
    a = np.ascontiguousarray(trace, dtype="<u8")
    assert a.shape == (g["rows"], g["columns"])
    bc = g["block_columns"]
    bad = [c // bc for c in range(0, a.shape[1], bc)
           if hashlib.sha256(np.ascontiguousarray(a[:, c:c + bc]).tobytes()).hexdigest()[:12] != g["blocks"][c // bc]]
    assert not bad, f"{name}: column blocks {bad[:10]} (of {len(bad)}) differ from the reference's trace"
    assert hashlib.sha256(a.tobytes()).hexdigest() == g["sha256"]


def test_fp12_mul_trace_equals_the_reference():
    g = GOLD["airs"]["FP12MulStark"]["inputs"]
    trace, _ = S.trace_fp12_mul(fp12_arr(g["x"]), fp12_arr(g["y"]))
    check("FP12MulStark", trace)


def test_pairing_precomp_trace_equals_the_reference():
    g = GOLD["airs"]["PairingPrecompStark"]["inputs"]
    trace, _ = S.trace_pairing_precomp(fp2_arr(g["qx"]), fp2_arr(g["qy"]), fp2_arr(g["qz"]))
    check("PairingPrecompStark", trace)


def _need(name):
    if name not in GOLD["airs"]:
        pytest.skip(f"no reference digest of {name} in tests/golden/trace_digests.json")
    return GOLD["airs"][name]["inputs"]


@pytest.mark.parametrize("name", ["MillerLoopStark", "MillerLoopStark#2"])
def test_miller_loop_trace_equals_the_reference(name):
    g = _need(name)
    trace, _ = S.trace_miller_loop(np.array(limbs(g["px"][0]), dtype=np.uint32), np.array(limbs(g["py"][0]), dtype=np.uint32),
                                   fp2_arr(g["qx"]), fp2_arr(g["qy"]), fp2_arr(g["qz"]))
    check(name, trace)


@pytest.mark.parametrize("name", ["FinalExponentiateStark", "FinalExponentiateStark#2"])
def test_final_exp_trace_equals_the_reference(name):
    g = _need(name)
    trace, _ = S.trace_final_exp(fp12_arr(g["x"]))
    check(name, trace)


def test_pairing_precomp_second_trace_equals_the_reference():
    g = _need("PairingPrecompStark#2")
    trace, _ = S.trace_pairing_precomp(fp2_arr(g["qx"]), fp2_arr(g["qy"]), fp2_arr(g["qz"]))
    check("PairingPrecompStark#2", trace)


def test_ecc_aggregate_trace_equals_the_reference():
    g = _need("ECCAggStark")
    n = S.ECC_NUM_POINTS
    pts = list(zip(g["points_x"], g["points_y"]))
    bits = [bool(int(b)) for b in g["bits"]]
    pts += [pts[-1]] * (n - len(pts))
    bits += [False] * (n - len(bits))
    arr = np.array([limbs(x) + limbs(y) for x, y in pts], dtype=np.uint32)
    trace, _ = S.trace_ecc_aggregate(arr, np.array(bits, dtype=bool))
    check("ECCAggStark", trace)


def test_every_public_input_is_bound_to_the_trace_by_a_constraint():
    """The reference builds the public inputs in its drivers (src/aggregate_proof.rs:37-56,80-99,126-131,160-165,191-217), next to
    plonky2 calls that cannot be run here.  They need no pin of their own: every public input of every AIR occurs in a constraint
    (the constraints are pinned to the reference's text by test_constraint_schedule_cpu.py, and vanish on the product's traces with
    the product's public inputs, test_airs_cpu.py), so with the trace fixed cell for cell the public inputs are determined."""
    from air_blob import constraints, parse_blob
    for air in (S.AIR_FP12_MUL, S.AIR_PAIRING_PRECOMP, S.AIR_MILLER_LOOP, S.AIR_FINAL_EXP, S.AIR_ECC_AGGREGATE):
        prog = parse_blob(S.air_program(air))
        used = {pi for _, _, terms in constraints(prog) for _, pi, _ in terms if pi >= 0}
        assert used == set(range(prog["n_pis"])), S.AIR_NAMES[air]
