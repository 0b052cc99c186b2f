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


@pytest.mark.parametrize("rate_bits,ncols", [(0, 3), (1, 5), (2, 17), (3, 4)])
def test_lde_8192_rows_in_both_kernels(prover, rate_bits, ncols):
    """8192-row columns take the wave-resident kernel (lde_columns_wave_kernel: one exchange across waves per transform, the others
    inside a wave, the last index bit by v_permlane32_swap; tools/lde_wave_model.py is its index model) when nothing but the LDE is
    asked for; "lde_impl" = 1 sends them through lde_columns_v2_kernel like every other shape.  Both give the oracle's values for
    random columns, columns of boundary values, and with the closed forms off (every column transformed)."""
    n = 1 << 13
    rng = np.random.default_rng(4000 + rate_bits)
    vals = _rand(rng, (ncols, n))
    vals[0] = 0
    vals[1] = P - 1
    vals[2, ::2] = P - 1
    vals[2, 1::2] = 1
    ocoeffs, olde_rows = O.lde_rows(vals, rate_bits)
    for impl in (0, 1):
        for closed in (1, 0):
            prover.set_option("lde_impl", impl)
            prover.set_option("lde_closed_forms", closed)
            try:
                coeffs, lde = prover.lde_batch(vals, rate_bits)
            finally:
                prover.set_option("lde_impl", 0)
                prover.set_option("lde_closed_forms", 1)
            assert np.array_equal(lde, olde_rows.T), (impl, closed)
            assert np.array_equal(coeffs, ocoeffs), (impl, closed)


@pytest.mark.parametrize("log_n,rate_bits", [(13, 2), (12, 2), (12, 1), (13, 1)])
def test_lde_closed_forms_match_oracle(prover, log_n, rate_bits):
    """Columns the LDE kernel does NOT transform (kernels_lde.hip: one workgroup per column, 2^12 and 2^13 rows): constant
    columns and unit vectors -- FinalExp's replicated Fp12 blocks and its one-hot row selectors -- take closed forms; next to them
    the near misses that must still go through the transforms (two ones, a single 2, a one among other values, all ones but one).
    Coefficients and LDE equal the oracle's for every column, with the closed forms on and off."""
    n = 1 << log_n
    rng = np.random.default_rng(99 + log_n + rate_bits)
    cols = []
    for c in (0, 1, 7, P - 1, 1 << 32, (1 << 32) - 1):                       # constants
        cols.append(np.full(n, c, dtype=np.uint64))
    for r in (0, 1, 5, n // 2, n - 2, n - 1, int(rng.integers(0, n))):       # unit vectors e_r
        v = np.zeros(n, dtype=np.uint64)
        v[r] = 1
        cols.append(v)
    two = np.zeros(n, dtype=np.uint64); two[3] = two[n - 7] = 1; cols.append(two)          # two ones
    big = np.zeros(n, dtype=np.uint64); big[9] = 2; cols.append(big)                        # a single 2
    mix = _rand(rng, n); mix[11] = 1; cols.append(mix)                                      # a one among other values
    hole = np.ones(n, dtype=np.uint64); hole[n - 1] = 0; cols.append(hole)                  # all ones but one
    same16 = np.zeros(n, dtype=np.uint64); same16[::n // 16] = 1; cols.append(same16)       # sixteen ones, one per thread slot
    cols.append(_rand(rng, n))
    vals = np.stack(cols)
    ocoeffs, olde_rows = O.lde_rows(vals, rate_bits)
    for on in (1, 0):
        prover.set_option("lde_closed_forms", on)
        try:
            coeffs, lde = prover.lde_batch(vals, rate_bits)
        finally:
            prover.set_option("lde_closed_forms", 1)
        assert np.array_equal(coeffs, ocoeffs), on
        assert np.array_equal(lde, olde_rows.T), on


@pytest.mark.parametrize("log_N,ncols,cap_h", [(5, 60285 // 16, 4), (4, 3, 4), (8, 4, 4), (8, 5, 2), (12, 200, 4), (6, 8, 0),
                                               # many leaves, widths around the 8-element sponge blocks
                                               (14, 21, 4), (14, 8, 4), (14, 6, 4), (14, 15, 4), (15, 9, 4), (14, 3, 4),
                                               # the quad layout's paths: capacity-only layers before further full blocks, tails of 1..7
                                               (10, 16, 4), (10, 24, 4), (10, 12, 4), (10, 13, 4), (10, 7, 4), (10, 17, 4), (10, 20, 4), (10, 11, 4),
                                               (10, 10, 4), (10, 14, 4)])
def test_merkle_cap_matches_oracle(prover, log_N, ncols, cap_h):
    rng = np.random.default_rng(log_N + ncols)
    mat = _rand(rng, (ncols, 1 << log_N))
    cap = prover.merkle_cap(mat, cap_h)
    assert np.array_equal(cap, O.merkle_cap(np.ascontiguousarray(mat.T), cap_h))


@pytest.mark.parametrize("log_N,ncols,cap_h", [(5, 60285 // 16, 4), (4, 3, 4), (8, 5, 2), (12, 200, 4), (6, 8, 0), (13, 21, 4), (2, 9, 0), (3, 100, 1),
                                               # tails of 0 .. 7 after one and after two full blocks
                                               (10, 16, 4), (10, 24, 4), (10, 12, 4), (10, 13, 4), (10, 9, 4), (10, 17, 4), (10, 20, 4), (10, 11, 4),
                                               (10, 10, 4), (10, 14, 4), (10, 15, 4), (10, 23, 4),
                                               # the pair form's own range (>= 32 768 leaves), and a last wave that is not full
                                               (15, 19, 4), (16, 9, 4), (7, 40, 2)])
@pytest.mark.parametrize("form", [2, 1, 3, 4])
def test_merkle_cap_in_both_leaf_hash_forms(prover, log_N, ncols, cap_h, form):
    """The row form (16 lanes per leaf, what a lone context uses for <= 4096 leaves), the quad form (4 lanes per leaf), the lane form
    (one lane per leaf) and the pair form (two lanes per leaf, a lone commitment of >= 32 768 leaves) give the oracle's cap for every
    shape, whichever the automatic choice would have been."""
    rng = np.random.default_rng(1000 + log_N + ncols)
    mat = _rand(rng, (ncols, 1 << log_N))
    mat[:, 0] = 0                      # an all-zero leaf
    mat[:, -1] = np.uint64(P - 1)      # and one of p - 1
    prover.set_option("leaf_hash_form", form)
    try:
        cap = prover.merkle_cap(mat, cap_h)
    finally:
        prover.set_option("leaf_hash_form", 0)
    assert np.array_equal(cap, O.merkle_cap(np.ascontiguousarray(mat.T), cap_h))


def test_leaf_hash_form_option_knows_its_five_values(prover):
    """0 = automatic, 1 quad, 2 row, 3 lane, 4 pair; anything else is refused and the setting stays."""
    for bad in (5, -1):
        with pytest.raises(S.StarkhipError):
            prover.set_option("leaf_hash_form", bad)
    rng = np.random.default_rng(5)
    mat = _rand(rng, (9, 1 << 15))   # 32 768 leaves: the automatic choice is the pair form
    assert np.array_equal(prover.merkle_cap(mat, 4), O.merkle_cap(np.ascontiguousarray(mat.T), 4))


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
    # one row, more rows than the reference's largest trace, a public input that is not a canonical field element,
    # and a blow-up too small for the constraint degree of a real AIR: all refused before any GPU work
    for rows in (1, 16384):
        tt, pp = S.trace_fibonacci(3, 5, 64)
        big = np.zeros((rows, tt.shape[1]), dtype=np.uint64)
        with pytest.raises(S.StarkhipError) as e:
            prover.prove(S.AIR_TEST_FIBONACCI, cfg, big, pp)
        assert e.value.code == S.ERR_BAD_SHAPE
    bad_pis = pis.copy()
    bad_pis[0] = np.uint64(P)
    with pytest.raises(S.StarkhipError) as e:
        prover.prove(S.AIR_TEST_FIBONACCI, cfg, t, bad_pis)
    assert e.value.code == S.ERR_BAD_SHAPE
    from bls_util import random_fp12
    t12, pis12 = S.trace_fp12_mul(random_fp12(0x5EED3100), random_fp12(0x5EED3101))
    low = S.StarkConfig.for_air(S.AIR_FP12_MUL)
    low.rate_bits = 0
    with pytest.raises(S.StarkhipError) as e:
        prover.prove(S.AIR_FP12_MUL, low, t12, pis12)
    assert e.value.code == S.ERR_BAD_SHAPE
    with pytest.raises(S.StarkhipError) as e:
        prover.prove(99, cfg, t, pis)
    assert e.value.code == S.ERR_BAD_AIR


@pytest.mark.parametrize("slots", [0, 7, 24])
def test_quotient_cell_cache_does_not_change_the_proof(slots):
    """The optional per-wave LDS cell cache of the op-stream interpreter (options quotient_impl = 1, quotient_slots; off by
    default) is a pure re-scheduling of cell loads: FP12Mul proofs are identical to the oracle's for every slot count."""
    import oracle_lib as O
    from bls_util import random_fp12
    pv = S.Prover(0)
    pv.set_option("quotient_impl", 1)
    pv.set_option("quotient_slots", slots)
    try:
        air = S.AIR_FP12_MUL
        t, pis = S.trace_fp12_mul(random_fp12(0x5EED3000), random_fp12(0x5EED3001))
        cfg = S.StarkConfig.for_air(air)
        proof = pv.prove(air, cfg, t, pis)
        ref = O.prove(S.air_program(air), cfg, S.trace_rows_to_poly_values(t), pis)
        assert np.array_equal(proof, ref)
    finally:
        pv.close()


def _multiply_reduce_paths(a, b, c=0):
    """Word-level model of gl_mul_nc / gl_mad_nc (csrc/gl_dev.h): which internal flags an operand pair raises.
    (c0: carry of a0*b0 + c, cm: carry of the middle multiply-add, t_ov: m2_hi + c0 reaches 2^32, borrow and carry of the
    reduction, h0 == 0).  Also checks the header's claim that the single correction never wraps."""
    M32, M64 = (1 << 32) - 1, (1 << 64) - 1
    a0, a1, b0, b1 = a & M32, a >> 32, b & M32, b >> 32
    p0 = a0 * b0 + c
    c0, p0 = p0 >> 64, p0 & M64
    m2 = a1 * b0 + a0 * b1 + (p0 >> 32)
    cm, m2 = m2 >> 64, m2 & M64
    t = (m2 >> 32) + c0
    p3 = a1 * b1 + t
    assert p3 < 1 << 64
    l, h0, h1 = ((m2 & M32) << 32) | (p0 & M32), p3 & M32, p3 >> 32
    d = l - h1 - cm
    borrow, d = int(d < 0), d & M64
    r = d + h0 * M32
    carry, r = r >> 64, r & M64
    x = r + (carry - borrow) * M32
    assert 0 <= x < 1 << 64 and x % P == (a * b + c) % P
    return c0, cm, t >> 32, borrow, carry, int(h0 == 0)


def test_multiply_reduce_flag_paths(prover):
    """The 13-instruction multiply keeps its carries in scalar flag registers and settles borrow and carry with one
    correction; each combination of those flags that operands can produce is driven here (found by search with the
    model above), for a*b, a*b + (a xor b) and the plain 128-bit reduction."""
    pairs = [
        (0x80000000, 0x6ec9d28663ca828d), (0x2, 0x4ba927c3ecf45ccb), (0x8a05a6c4647159, 0xb8b6d8fe442e3d43),
        (0x8000000000000000, 0x8000000000000000), (0xfffffffe00000001, 0x7fffffff00000000), (0xa6cecc1b78e51061, 0xfffffffffffffffe),
        (0xfffffffeffffffff, 0x8000000080000001), (0xd66b829e6a8ac4ba, 0xa46d6753ec148cb4), (0xfffffffefffffffe, 0xffffffff00000003),
        (0xfffffffe, 0x7ffffffffffffffe), (0xfffffffeffffffff, 0xffff0000ffff), (0x1ffffffff, 0xffff0000fffffffd),
        (0x8000000080000000, 0x7fffffff00000001), (0x7fffffff80000001, 0xffffffffffffffff), (0x7fffffffffffffff, 0x80000001ffffffff),
        (0xfffffffd80000001, 0x74b31bfbf8449560), (0xfffffffdffffffff, 0xfffffffeffffffff), (0xffffffff7fffffff, 0x4a2f20aaf3c64af7),
        (0x7fffffffffffffff, 0xffffffffffffffff), (0x2, 0xffffffff00000000), (0x0, 0x0), (0x2, 0xffffffffffffffff),
        (0x200000000, 0x8000000000000000), (0x180000000, 0xfffffffe00000000), (0x3ffffffff, 0xffffffff80000000),
        (0xfffffffeffffffff, 0xfffffffdffffffff), (0x2ffffffff, 0xffffffffffffffff), (0xffffffff00000002, 0xffffffffffffffff),
        (0x300000000, 0xaaaaaaab00000000),
    ]
    pairs += [(y, x) for x, y in pairs]
    mul_paths = {_multiply_reduce_paths(x, y) for x, y in pairs}
    mad_paths = {_multiply_reduce_paths(x, y, x ^ y) for x, y in pairs}
    # every flag is seen both ways, and the interesting combinations are present
    for paths, flags in ((mul_paths, (1, 3, 4, 5)), (mad_paths, (0, 1, 2, 3, 4, 5))):
        for f in flags:
            assert {p[f] for p in paths} == {0, 1}, (f, sorted(paths))
    assert any(p[3] and p[4] for p in mul_paths) and any(p[3] and not p[4] for p in mul_paths) and any(p[4] and not p[3] for p in mul_paths)
    assert any(p[0] and p[2] for p in mad_paths)  # the carry of a0*b0 + c ripples past m2's high word
    a = np.array([x for x, _ in pairs], dtype=np.uint64)
    b = np.array([y for _, y in pairs], dtype=np.uint64)
    assert [int(g) for g in prover.field_ops(0, a, b)] == [x * y % P for x, y in pairs]
    assert [int(g) for g in prover.field_ops(1, a, b)] == [(x * y + (x ^ y)) % P for x, y in pairs]
    assert [int(g) for g in prover.field_ops(7, a, b)] == [((x << 64) + y) % P for x, y in pairs]
    # the variant with a wave-uniform multiplicand in scalar registers (op 9: a[i] * b[0] + b[i]), every b as b[0] in turn
    for k in range(len(pairs)):
        bk = np.roll(b, -k)
        k0 = int(bk[0])
        assert [int(g) for g in prover.field_ops(9, a, bk)] == [(x * k0 + int(y)) % P for (x, _), y in zip(pairs, bk)], k


def test_multiply_reduce_on_a_million_structured_operands(prover):
    """Operands whose 32-bit words are drawn from {0, 1, 2, 2^31, 2^32 - 2, 2^32 - 1, random}: every word pattern that can
    raise or suppress a carry in the product or the reduction, 2^20 pairs, against Python integers."""
    rng = np.random.default_rng(11)
    special = np.array([0, 1, 2, 1 << 31, (1 << 32) - 2, (1 << 32) - 1], dtype=np.uint64)

    def words(n):
        w = rng.integers(0, 1 << 32, size=n, dtype=np.uint64)
        pick = rng.integers(0, 12, size=n)
        return np.where(pick < 6, special[np.minimum(pick, 5)], w)
    n = 1 << 20
    a = (words(n) << np.uint64(32)) | words(n)
    b = (words(n) << np.uint64(32)) | words(n)
    ai, bi = [int(v) for v in a], [int(v) for v in b]
    assert [int(g) for g in prover.field_ops(0, a, b)] == [x * y % P for x, y in zip(ai, bi)]
    assert [int(g) for g in prover.field_ops(1, a, b)] == [(x * y + (x ^ y)) % P for x, y in zip(ai, bi)]
    assert [int(g) for g in prover.field_ops(7, a, b)] == [((x << 64) + y) % P for x, y in zip(ai, bi)]
    k0 = bi[0]
    assert [int(g) for g in prover.field_ops(9, a, b)] == [(x * k0 + y) % P for x, y in zip(ai, bi)]


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
