"""CPU tests of the restated AIRs: shapes and constraint counts, trace <-> constraints
consistency on every row, oracle prove -> product verify."""
import numpy as np
import pytest

import oracle_lib as O
import starky_bls12_381_amd as S
from bls_util import native_vectors, fp_arr, random_fp12

# columns, public inputs, constraint degree, number of constraints, rows   (README.md:36-39 of the reference)
EXPECTED = {
    S.AIR_FP12_MUL: (60285, 432, 3, 82560, 16),
    S.AIR_FINAL_EXP: (73527, 288, 5, 360800, 8192),
    S.AIR_MILLER_LOOP: (97330, 5064, 3, 145574, 1024),
    S.AIR_PAIRING_PRECOMP: (29376, 4968, 4, 113634, 1024),
}


def _available(air):
    try:
        S.air_columns(air)
        return True
    except S.StarkhipError:
        return False


@pytest.mark.parametrize("air", sorted(EXPECTED))
def test_air_shape_and_constraint_count(air):
    if not _available(air):
        pytest.skip(f"{S.AIR_NAMES[air]} not restated yet")
    cols, pis, deg, k, rows = EXPECTED[air]
    assert S.air_columns(air) == cols
    assert S.air_public_inputs(air) == pis
    assert S.air_constraint_degree(air) == deg
    assert S.air_num_constraints(air) == k
    assert S.air_default_rows(air) == rows


def test_fp12_mul_trace_satisfies_all_constraints_and_pis_match_native():
    x, y = random_fp12(0x5EED2000), random_fp12(0x5EED2001)
    t, pis = S.trace_fp12_mul(x, y)
    assert t.shape == (16, 60285)
    assert np.array_equal(pis[:144], x) and np.array_equal(pis[144:288], y)
    assert np.array_equal(pis[288:], S.native_fp12_mul(x, y))
    blob = S.air_program(S.AIR_FP12_MUL)
    assert O.check_trace(blob, t, pis)[0] == 0
    # rows 12..15 are zero padding (src/fp12_mul.rs:44-48)
    assert not t[12:].any()
    bad = t.copy()
    bad[5, 40000] = (int(bad[5, 40000]) + 1) % S.P
    assert O.check_trace(blob, bad, pis)[0] > 0
    wrong_pis = pis.copy()
    wrong_pis[300] = (int(wrong_pis[300]) + 1) % S.P
    assert O.check_trace(blob, t, wrong_pis)[0] > 0


def test_fp12_mul_edge_operands():
    one = fp_arr(1, *([0] * 11))
    zero = fp_arr(*([0] * 12))
    pm1 = fp_arr(*([S_BLS_P - 1] * 12))
    blob = S.air_program(S.AIR_FP12_MUL)
    for x, y in ((one, one), (zero, pm1), (pm1, pm1)):
        t, pis = S.trace_fp12_mul(x, y)
        assert O.check_trace(blob, t, pis)[0] == 0


S_BLS_P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB


def test_trace_into_a_caller_buffer_ignores_what_was_in_it():
    """The page-locked hand-over (Prover.host_array) reuses one buffer for many traces: stale contents must not leak."""
    import ctypes as C
    x, y = random_fp12(0x5EED0301), random_fp12(0x5EED0302)
    want, want_pis = S.trace_fp12_mul(x, y)
    buf = np.full(want.shape, 0xDEADBEEFDEADBEEF, dtype=np.uint64)
    got, got_pis = S.trace_fp12_mul(x, y, out=buf)
    assert got is buf and np.array_equal(got, want) and np.array_equal(got_pis, want_pis)
    with pytest.raises(ValueError):
        S.trace_fp12_mul(x, y, out=np.zeros((want.shape[0], want.shape[1] + 1), dtype=np.uint64))
    with pytest.raises(ValueError):
        S.trace_fp12_mul(x, y, out=np.zeros(want.shape, dtype=np.int64))
    # no context, no page-locked memory (and no crash)
    p = C.c_void_p(1)
    assert S.lib.starkhip_host_alloc(None, 4096, C.byref(p)) == S.ERR_NO_DEVICE and not p.value
    S.lib.starkhip_host_free(None)


def test_fp12_mul_oracle_proof_verifies():
    air = S.AIR_FP12_MUL
    x, y = random_fp12(0x5EED2002), random_fp12(0x5EED2003)
    t, pis = S.trace_fp12_mul(x, y)
    cfg = S.StarkConfig.for_air(air)
    proof = O.prove(S.air_program(air), cfg, S.trace_rows_to_poly_values(t), pis)
    S.verify_stark_proof(air, cfg, proof)
    bad = proof.copy()
    bad[16 + 64 + 64 + 5] = (int(bad[16 + 64 + 64 + 5]) + 1) % S.P  # an opening
    with pytest.raises(S.StarkhipError):
        S.verify_stark_proof(air, cfg, bad)


def test_final_exp_trace_satisfies_all_constraints_on_the_reference_vector():
    """src/native.rs:1546-1563 (`aa`, final exponentiation == 1) through generate_trace: every one of the
    360800 constraints vanishes on every one of the 8192 rows, and the public output is Fp12::one()."""
    if not _available(S.AIR_FINAL_EXP):
        pytest.skip("FinalExponentiateStark not restated yet")
    aa = fp_arr(*[int(s) for s in native_vectors()["final_exp_input_aa"]])
    t, pis = S.trace_final_exp(aa)
    assert t.shape == (8192, 73527)
    assert np.array_equal(pis[:144], aa)
    assert np.array_equal(pis[144:], fp_arr(1, *([0] * 11)))
    blob = S.air_program(S.AIR_FINAL_EXP)
    assert O.check_trace(blob, t, pis)[0] == 0
    # the same generator recording its writes as runs stands for exactly this matrix, at 1/30 of the bytes
    c, cpis = S.trace_final_exp(aa, compact=True)
    assert c.shape == t.shape and np.array_equal(cpis, pis) and c.nbytes * 20 < t.nbytes
    e, conflicts = c.expand()
    assert conflicts == 0 and np.array_equal(e, t)
    # recorded on several host threads (starkhip_trace_set_threads: the 32 ops become tasks once the native chain is known): the
    # same matrix, and the same records for any thread count > 1
    del e
    counts = []
    try:
        for threads in (3, 8):
            S.set_trace_threads(threads)
            cp, ppis = S.trace_final_exp(aa, compact=True)
            e, conflicts = cp.expand()
            assert conflicts == 0 and np.array_equal(e, t) and np.array_equal(ppis, pis)
            counts.append((cp.n_records, cp.nbytes))
            del e
    finally:
        assert S.set_trace_threads(1) == 8
    assert counts[0] == counts[1] and counts[0][0] < c.n_records * 1.02
    # 4441 rows carry operations (TOTAL_ROW), the rest of the op window is zero
    assert not t[4441:, 12949:].any()
    t[100, 20000] = (int(t[100, 20000]) + 1) % S.P
    bad, first = O.check_trace(blob, t, pis)
    assert bad > 0 and first[1] in (99, 100)


def _bls():
    return {k: int(s) for k, s in native_vectors()["bls_signature"].items()}


def test_miller_loop_trace_satisfies_all_constraints_on_the_reference_signature():
    """G1 generator x signature point of src/native.rs:1490-1498: 68 steps x 12 rows, result == native miller_loop."""
    if not _available(S.AIR_MILLER_LOOP):
        pytest.skip("MillerLoopStark not restated yet")
    b = _bls()
    args = (fp_arr(b["gx"]), fp_arr(b["gy"]), fp_arr(b["s_x1"], b["s_x2"]), fp_arr(b["s_y1"], b["s_y2"]), fp_arr(b["s_z1"], b["s_z2"]))
    t, pis = S.trace_miller_loop(*args)
    assert t.shape == (1024, 97330)
    assert np.array_equal(pis[-144:], S.native_miller_loop(*args))
    assert np.array_equal(pis[24:24 + 68 * 72], S.native_pairing_precomp(*args[2:]))
    blob = S.air_program(S.AIR_MILLER_LOOP)
    assert O.check_trace(blob, t, pis)[0] == 0
    t[30, 50000] = (int(t[30, 50000]) + 1) % S.P
    assert O.check_trace(blob, t, pis)[0] > 0


def test_pairing_precomp_trace_satisfies_all_constraints_on_the_reference_signature():
    if not _available(S.AIR_PAIRING_PRECOMP):
        pytest.skip("PairingPrecompStark not restated yet")
    b = _bls()
    q = (fp_arr(b["hm_x1"], b["hm_x2"]), fp_arr(b["hm_y1"], b["hm_y2"]), fp_arr(b["hm_z1"], b["hm_z2"]))
    t, pis = S.trace_pairing_precomp(*q)
    assert t.shape == (1024, 29376)
    assert np.array_equal(pis[:24], q[0]) and np.array_equal(pis[48:72], q[2])
    assert np.array_equal(pis[72:], S.native_pairing_precomp(*q))
    blob = S.air_program(S.AIR_PAIRING_PRECOMP)
    assert O.check_trace(blob, t, pis)[0] == 0
    # a point with z != 1 exercises the z * z^-1 multiplications
    g = __import__("bls_util").splitmix64(0x5EED1000)
    from bls_util import random_fp
    q2 = tuple(fp_arr(random_fp(g), random_fp(g)) for _ in range(3))
    t2, pis2 = S.trace_pairing_precomp(*q2)
    assert O.check_trace(blob, t2, pis2)[0] == 0
    t2[200, 15000] = (int(t2[200, 15000]) + 1) % S.P
    assert O.check_trace(blob, t2, pis2)[0] > 0
