"""GPU parity tests (through the C ABI) against the CPU oracle: bit-exact."""
import numpy as np
import pytest

import oracle_lib as O
import starky_bls12_381_amd as S

pytestmark = pytest.mark.gpu
P = S.P


def _rand(rng, shape):
    return rng.integers(0, P, size=shape, dtype=np.uint64)


def test_poseidon_permutation_batch(prover):
    rng = np.random.default_rng(1)
    states = _rand(rng, (1000, 12))
    states[0] = 0
    states[1] = np.arange(12)
    states[2] = P - 1
    out = prover.poseidon_permute_batch(states)
    for i in range(0, 1000, 37):
        assert np.array_equal(out[i], O.poseidon_permute(states[i]))
    assert np.array_equal(out[2], O.poseidon_permute(states[2]))


@pytest.mark.parametrize("log_n,rate_bits,ncols", [(1, 1, 5), (4, 1, 301), (6, 2, 130), (7, 1, 33), (8, 1, 37), (8, 2, 16), (9, 2, 21), (10, 2, 257),
                                                    (10, 1, 64), (11, 1, 10), (13, 2, 9), (12, 3, 3), (13, 1, 2)])
def test_lde_matches_oracle(prover, log_n, rate_bits, ncols):
    rng = np.random.default_rng(log_n * 10 + rate_bits)
    vals = _rand(rng, (ncols, 1 << log_n))
    vals[0] = 0
    vals[-1] = P - 1
    coeffs, lde = prover.lde_batch(vals, rate_bits)
    ocoeffs, olde_rows = O.lde_rows(vals, rate_bits)
    assert np.array_equal(coeffs, ocoeffs)
    assert np.array_equal(lde, olde_rows.T)


@pytest.mark.parametrize("log_N,ncols,cap_h", [(5, 60285 // 16, 4), (4, 3, 4), (8, 4, 4), (8, 5, 2), (12, 200, 4), (6, 8, 0),
                                               # many leaves, widths around the 8-element sponge blocks
                                               (14, 21, 4), (14, 8, 4), (14, 6, 4), (14, 15, 4), (15, 9, 4), (14, 3, 4)])
def test_merkle_cap_matches_oracle(prover, log_N, ncols, cap_h):
    rng = np.random.default_rng(log_N + ncols)
    mat = _rand(rng, (ncols, 1 << log_N))
    cap = prover.merkle_cap(mat, cap_h)
    assert np.array_equal(cap, O.merkle_cap(np.ascontiguousarray(mat.T), cap_h))


@pytest.mark.parametrize("n,rate_bits", [(16, 1), (64, 1), (64, 2), (1024, 2), (1024, 1), (8192, 2)])
def test_toy_air_proof_is_bit_identical_to_oracle(prover, n, rate_bits):
    air = S.AIR_TEST_FIBONACCI
    cfg = S.StarkConfig.standard_fast_config()
    cfg.rate_bits = rate_bits
    t, pis = S.trace_fibonacci(3, 5, n)
    proof = prover.prove(air, cfg, t, pis)
    S.verify_stark_proof(air, cfg, proof)
    oproof = O.prove(S.air_program(air), cfg, S.trace_rows_to_poly_values(t), pis)
    assert proof.size == oproof.size
    assert np.array_equal(proof, oproof)
    # column-major input path gives the same bytes
    proof2 = prover.prove(air, cfg, S.trace_rows_to_poly_values(t), pis, layout=1)
    assert np.array_equal(proof, proof2)


def test_invalid_witness_never_yields_an_accepted_proof(prover):
    cfg = S.StarkConfig.standard_fast_config()
    t, pis = S.trace_fibonacci(3, 5, 64)
    t[9, 1] = (int(t[9, 1]) + 1) % P
    proof = prover.prove(S.AIR_TEST_FIBONACCI, cfg, t, pis)  # factor == blow-up: trim_to_len cannot fail
    with pytest.raises(S.StarkhipError) as e:
        S.verify_stark_proof(S.AIR_TEST_FIBONACCI, cfg, proof)
    assert e.value.code == S.ERR_VERIFY


def test_bad_shapes_are_refused(prover):
    cfg = S.StarkConfig.standard_fast_config()
    t, pis = S.trace_fibonacci(3, 5, 64)
    with pytest.raises(S.StarkhipError) as e:
        prover.prove(S.AIR_TEST_FIBONACCI, cfg, t[:48], pis)
    assert e.value.code == S.ERR_BAD_SHAPE
    with pytest.raises(S.StarkhipError):
        prover.prove(S.AIR_TEST_FIBONACCI, cfg, t, pis[:2])


@pytest.mark.parametrize("slots", [0, 7, 24])
def test_quotient_cell_cache_does_not_change_the_proof(slots, monkeypatch):
    """The optional per-wave LDS cell cache of the quotient kernel (STARKHIP_QUOTIENT_SLOTS, off by default) is a pure
    re-scheduling of cell loads: FP12Mul proofs are identical to the oracle's for every slot count."""
    import oracle_lib as O
    from bls_util import random_fp12
    monkeypatch.setenv("STARKHIP_QUOTIENT_SLOTS", str(slots))
    pv = S.Prover(0)  # the slot count is read when a context compiles an AIR's op stream
    try:
        air = S.AIR_FP12_MUL
        t, pis = S.trace_fp12_mul(random_fp12(0x5EED3000), random_fp12(0x5EED3001))
        cfg = S.StarkConfig.for_air(air)
        proof = pv.prove(air, cfg, t, pis)
        ref = O.prove(S.air_program(air), cfg, S.trace_rows_to_poly_values(t), pis)
        assert np.array_equal(proof, ref)
    finally:
        pv.close()


def test_lazy_reduction_arithmetic_on_boundary_operands(prover):
    """The kernels keep field values as arbitrary 64-bit representatives and correct wraps lazily; the wrap paths fire with
    probability ~2^-32 per operation on proof data, so they are driven here with boundary operands, all pairs."""
    EPS = (1 << 32) - 1
    M = (1 << 64) - 1
    edge = [0, 1, 2, EPS - 1, EPS, EPS + 1, 1 << 32, (1 << 63) - 1, 1 << 63, P - 2, P - 1, P, P + 1, M - EPS, M - 1, M, 0xFFFFFFFE00000001,
            0x00000001FFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF00000000]
    rng = np.random.default_rng(5)
    vals = edge + [int(x) for x in rng.integers(0, 1 << 63, size=12, dtype=np.uint64) * 2 + rng.integers(0, 2, size=12, dtype=np.uint64)]
    a = np.array([x for x in vals for _ in vals], dtype=np.uint64)
    b = np.array([y for _ in vals for y in vals], dtype=np.uint64)
    ai, bi = [int(x) for x in a], [int(y) for y in b]
    m44 = (1 << 44) - 1
    want = {
        0: [x * y % P for x, y in zip(ai, bi)],
        1: [(x * y + (x ^ y)) % P for x, y in zip(ai, bi)],
        2: [(x + y) % P for x, y in zip(ai, bi)],
        3: [(x - y) % P for x, y in zip(ai, bi)],
        4: [(x + y) % P for x, y in zip(ai, bi)],
        5: [(x - y) % P for x, y in zip(ai, bi)],
        6: [((x & m44) + ((y & m44) << 32)) % P for x, y in zip(ai, bi)],
        7: [((x << 64) + y) % P for x, y in zip(ai, bi)],
        8: [x % P for x in ai],
    }
    for e in range(96):
        want[100 + e] = [(x << e) % P for x in ai]
    for op, w in want.items():
        got = prover.field_ops(op, a, b)
        assert [int(g) for g in got] == w, f"op {op}"
