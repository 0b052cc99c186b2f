"""GPU parity tests for the real AIRs: proof bytes identical to the CPU oracle, accepted by the verifier."""
import numpy as np
import pytest

import oracle_lib as O
import starky_bls12_381_amd as S
from bls_util import random_fp12

pytestmark = pytest.mark.gpu


def _available(air):
    try:
        S.air_columns(air)
        return True
    except S.StarkhipError:
        return False


@pytest.mark.parametrize("seed", [0x5EED2000, 0x5EED2010])
def test_fp12_mul_proof_is_bit_identical_to_oracle(prover, seed):
    air = S.AIR_FP12_MUL
    x, y = random_fp12(seed), random_fp12(seed + 1)
    t, pis = S.trace_fp12_mul(x, y)
    cfg = S.StarkConfig.for_air(air)
    proof = prover.prove(air, cfg, t, pis)
    S.verify_stark_proof(air, cfg, proof)
    ref = O.prove(S.air_program(air), cfg, S.trace_rows_to_poly_values(t), pis)
    assert proof.size == ref.size and np.array_equal(proof, ref)


def test_fp12_mul_commitment_on_host_threads_and_on_the_device_give_the_same_proof():
    """FP12Mul's trace commitment has 32 leaves: by default host threads hash them with the challenger's permutation
    ("host_commit_leaves" = 64); with the option at 0 the device does (row form).  Same proof bytes, which are the oracle's."""
    air = S.AIR_FP12_MUL
    cfg = S.StarkConfig.for_air(air)
    t, pis = S.trace_fp12_mul(random_fp12(0x5EED2200), random_fp12(0x5EED2201))
    pv = S.Prover(0)
    try:
        on_host = pv.prove(air, cfg, t, pis)
        pv.set_option("host_commit_leaves", 0)
        on_device = pv.prove(air, cfg, t, pis)
        with pytest.raises(S.StarkhipError):
            pv.set_option("host_commit_leaves", 1 << 20)
    finally:
        pv.close()
    assert np.array_equal(on_host, on_device)
    assert np.array_equal(on_host, O.prove(S.air_program(air), cfg, S.trace_rows_to_poly_values(t), pis))


def test_page_locked_trace_buffer_is_reused_across_proofs(prover):
    """starkhip_host_alloc hand-over: two different traces generated into the same page-locked buffer, each proof
    bit-identical to the oracle's; the buffer outlives views of it and is released afterwards."""
    air = S.AIR_FP12_MUL
    cfg = S.StarkConfig.for_air(air)
    buf = prover.host_array((S.air_default_rows(air), S.air_columns(air)))
    assert buf.flags.c_contiguous and buf.dtype == np.uint64
    for seed in (0x5EED2100, 0x5EED2110):
        t, pis = S.trace_fp12_mul(random_fp12(seed), random_fp12(seed + 1), out=buf)
        assert t is buf
        proof = prover.prove(air, cfg, t, pis)
        ref = O.prove(S.air_program(air), cfg, S.trace_rows_to_poly_values(t.copy()), pis)
        assert np.array_equal(proof, ref)
    row = buf[3]
    del buf, t
    assert int(row[0]) == int(row[0])  # a view keeps the allocation alive
    with pytest.raises(S.StarkhipError) as e:
        prover.host_array((0,))
    assert e.value.code == S.ERR_BAD_SHAPE


def test_final_exp_proof_verifies_and_matches_golden_digest(prover):
    """Full-size FinalExponentiateStark (73527 x 8192): the product verifier accepts the GPU proof; the proof bytes
    hash to the digest of the CPU oracle's proof for the same input when that fixture exists
    (tests/golden/final_exp_aa_proof.sha256, made by tests/make_final_exp_golden.py on a 64+ GB host)."""
    import hashlib
    import os
    from bls_util import GOLDEN, fp_arr, native_vectors
    air = S.AIR_FINAL_EXP
    if not _available(air):
        pytest.skip("FinalExponentiateStark not restated yet")
    aa = fp_arr(*[int(s) for s in native_vectors()["final_exp_input_aa"]])
    t, pis = S.trace_final_exp(aa)
    cfg = S.StarkConfig.for_air(air)
    proof = prover.prove(air, cfg, t, pis)
    del t
    S.verify_stark_proof(air, cfg, proof)
    tm = prover.last_timings()
    assert tm["total"] > 0
    digest = hashlib.sha256(proof.tobytes()).hexdigest()
    path = os.path.join(GOLDEN, "final_exp_aa_proof.sha256")
    if os.path.exists(path):
        assert digest == open(path).read().split()[0]
    # tampering with an opening or the public output is rejected
    bad = proof.copy()
    bad[-1] = (int(bad[-1]) + 1) % S.P
    with pytest.raises(S.StarkhipError):
        S.verify_stark_proof(air, cfg, bad)
    # the same trace recorded as runs and expanded on the device: the same proof, bit for bit
    c, cpis = S.trace_final_exp(aa, compact=True)
    assert c.nbytes < 300e6 and np.array_equal(cpis, pis)
    assert np.array_equal(prover.prove(air, cfg, c, cpis), proof)


def _bls():
    from bls_util import native_vectors
    return {k: int(s) for k, s in native_vectors()["bls_signature"].items()}


def test_compact_traces_prove_to_the_same_bytes_as_dense_ones(prover):
    """On-device trace expansion: FP12Mul (16 rows), PairingPrecomp and MillerLoop (1024 rows) and ECCAgg (8192 rows) are
    proven from the recorded runs and from the dense rows; the proofs must be identical (FinalExp: in its own test above)."""
    from bls_util import fp_arr
    b = _bls()
    hm = (fp_arr(b["hm_x1"], b["hm_x2"]), fp_arr(b["hm_y1"], b["hm_y2"]), fp_arr(b["hm_z1"], b["hm_z2"]))
    cases = [
        (S.AIR_FP12_MUL, S.trace_fp12_mul, (random_fp12(0x5EED2200), random_fp12(0x5EED2201))),
        (S.AIR_PAIRING_PRECOMP, S.trace_pairing_precomp, hm),
        (S.AIR_MILLER_LOOP, S.trace_miller_loop, (fp_arr(b["gx"]), fp_arr(b["gy"])) + hm),
    ]
    from test_ecc_aggregate_cpu import pack, reference_vector
    pts, bits, _ = reference_vector()
    cases.append((S.AIR_ECC_AGGREGATE, S.trace_ecc_aggregate, pack(pts, bits)))
    for air, fn, args in cases:
        cfg = S.StarkConfig.for_air(air)
        dense, pis = fn(*args)
        compact, cpis = fn(*args, compact=True)
        want = prover.prove(air, cfg, dense, pis)
        got = prover.prove(air, cfg, compact, cpis)
        assert np.array_equal(got, want), S.AIR_NAMES[air]
        assert np.array_equal(prover.prove(air, cfg, compact, cpis), want)  # a log can be proven again
    # a log recorded for one AIR is refused for another
    with pytest.raises(S.StarkhipError) as e:
        prover.prove(S.AIR_MILLER_LOOP, S.StarkConfig.for_air(S.AIR_MILLER_LOOP), S.trace_fp12_mul(*cases[0][2], compact=True)[0], pis)
    assert e.value.code == S.ERR_BAD_SHAPE


def test_pairing_precomp_proof_is_bit_identical_to_oracle(prover):
    from bls_util import fp_arr
    air = S.AIR_PAIRING_PRECOMP
    if not _available(air):
        pytest.skip("PairingPrecompStark not restated yet")
    b = _bls()
    t, pis = S.trace_pairing_precomp(fp_arr(b["hm_x1"], b["hm_x2"]), fp_arr(b["hm_y1"], b["hm_y2"]), fp_arr(b["hm_z1"], b["hm_z2"]))
    cfg = S.StarkConfig.for_air(air)
    proof = prover.prove(air, cfg, t, pis)
    S.verify_stark_proof(air, cfg, proof)
    ref = O.prove(S.air_program(air), cfg, S.trace_rows_to_poly_values(t), pis)
    assert proof.size == ref.size and np.array_equal(proof, ref)


def test_miller_loop_proof_is_bit_identical_to_oracle(prover):
    from bls_util import fp_arr
    air = S.AIR_MILLER_LOOP
    if not _available(air):
        pytest.skip("MillerLoopStark not restated yet")
    b = _bls()
    t, pis = S.trace_miller_loop(fp_arr(b["gx"]), fp_arr(b["gy"]), fp_arr(b["s_x1"], b["s_x2"]), fp_arr(b["s_y1"], b["s_y2"]),
                                 fp_arr(b["s_z1"], b["s_z2"]))
    cfg = S.StarkConfig.for_air(air)
    proof = prover.prove(air, cfg, t, pis)
    S.verify_stark_proof(air, cfg, proof)
    ref = O.prove(S.air_program(air), cfg, S.trace_rows_to_poly_values(t), pis)
    assert proof.size == ref.size and np.array_equal(proof, ref)


def test_ecc_aggregate_proof_is_bit_identical_to_oracle(prover):
    """ECCAggStark (3339 x 8192, constraint degree 4 => three quotient chunks per challenge) on the reference's own
    aggregation vector (src/ecc_aggregate.rs:489-523, padded to the AIR's 512 operands)."""
    from test_ecc_aggregate_cpu import pack, reference_vector
    from bls_util import limbs
    air = S.AIR_ECC_AGGREGATE
    pts, bits, res = reference_vector()
    arr, b = pack(pts, bits)
    t, pis = S.trace_ecc_aggregate(arr, b)
    assert [int(x) for x in pis[-24:]] == limbs(res[0]) + limbs(res[1])
    cfg = S.StarkConfig.for_air(air)
    proof = prover.prove(air, cfg, t, pis)
    S.verify_stark_proof(air, cfg, proof)
    ref = O.prove(S.air_program(air), cfg, S.trace_rows_to_poly_values(t), pis)
    assert proof.size == ref.size and np.array_equal(proof, ref)
    bad = proof.copy()
    bad[-30] = (int(bad[-30]) + 1) % S.P  # a limb of the published aggregate
    with pytest.raises(S.StarkhipError):
        S.verify_stark_proof(air, cfg, bad)


def test_ecc_aggregate_invalid_witness_fails_like_the_reference(prover):
    """Degree 4 at blow-up 4 leaves one zero chunk in the quotient: a witness that breaks a constraint makes
    `quotient_poly.trim_to_len` fail -- starky's "Quotient has failed" error -- instead of yielding a proof."""
    from test_ecc_aggregate_cpu import pack, reference_vector
    air = S.AIR_ECC_AGGREGATE
    pts, bits, _ = reference_vector()
    t, pis = S.trace_ecc_aggregate(*pack(pts, bits))
    t[100, 526 + 50] = (int(t[100, 526 + 50]) + 1) % S.P
    with pytest.raises(S.StarkhipError) as e:
        prover.prove(air, S.StarkConfig.for_air(air), t, pis)
    assert e.value.code == S.ERR_QUOTIENT_NOT_DIVISIBLE


def test_two_contexts_prove_concurrently_and_agree_with_sequential(prover):
    """The bench keeps two proofs in flight on one GPU (two contexts, two host threads): results must not depend on it."""
    import threading
    from bls_util import fp_arr
    b = _bls()
    args = [(fp_arr(b["hm_x1"], b["hm_x2"]), fp_arr(b["hm_y1"], b["hm_y2"]), fp_arr(b["hm_z1"], b["hm_z2"])),
            (fp_arr(b["s_x1"], b["s_x2"]), fp_arr(b["s_y1"], b["s_y2"]), fp_arr(b["s_z1"], b["s_z2"]))]
    air = S.AIR_PAIRING_PRECOMP
    cfg = S.StarkConfig.for_air(air)
    traces = [S.trace_pairing_precomp(*a) for a in args]
    want = [prover.prove(air, cfg, t, pis) for t, pis in traces]
    got = [None, None]
    provers = [S.Prover(0), S.Prover(0)]

    def run(i):
        for _ in range(3):
            got[i] = provers[i].prove(air, cfg, traces[i][0], traces[i][1])
    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for pv in provers:
        pv.close()
    for i in range(2):
        assert np.array_equal(got[i], want[i])


def test_a_log_with_empty_records_and_late_zeros_expands_to_the_dense_trace(prover):
    """The recorder's set-then-clear corner cases on the device (kernels_trace.hip): a one-row run that was cleared again stays in
    the log as a record of ZERO rows -- the expansion's lane -> (limb, row) mapping divides by the run length and must skip it -- and a
    clear inside a longer run is a late zero.  The FP12Mul trace is replayed cell by cell through the recorder with both idioms put
    on columns that are zero in the trace; the device's expansion must equal the host replay and the matrix the writes stand for.
    The generators' own logs of all five AIRs go the same way (their proofs are compared elsewhere; here the cells)."""
    dense, _ = S.trace_fp12_mul(random_fp12(0x5EED2210), random_fp12(0x5EED2211))
    rows, cols = dense.shape
    zero_cols = np.flatnonzero(~dense.any(axis=0))
    c0, c1, c2 = (int(c) for c in zero_cols[:3])
    writes = [(5, c0, 1), (5, c0, 0)]                                   # empty record, first in the log
    writes += [(r, c1, 1) for r in range(2, 9)] + [(8, c1, 0), (7, c1, 0), (4, c1, 0)]  # shrinks twice, then a late zero
    r_idx, c_idx = np.nonzero(dense)
    order = np.lexsort((r_idx, c_idx))  # column by column, rows ascending: runs are extended as the generators extend them
    writes += [(int(r), int(c), int(dense[r, c])) for r, c in zip(r_idx[order], c_idx[order])]
    writes += [(0, c2, 9), (0, c2, 0)]                                  # another empty record, last in the log
    log = S.trace_from_writes(rows, cols, writes)
    assert log.overwrites() == (2, 1)
    expected = dense.copy()
    expected[2:7, c1] = 1
    expected[4, c1] = 0
    host, conflicts = log.expand()
    assert conflicts == 0 and np.array_equal(host, expected)
    assert np.array_equal(prover.expand_trace(log), expected)
    # a long run (>= 64 rows) taken back row by row to nothing, and records of many limbs beside it
    rows2, cols2 = 256, 40
    w2 = [(r, 3, 1) for r in range(100)] + [(r, 3, 0) for r in range(99, -1, -1)]
    w2 += [(r, c, 1 + c) for r in range(10, 200) for c in range(8, 30)]
    log2 = S.trace_from_writes(rows2, cols2, w2)
    want2 = np.zeros((rows2, cols2), dtype=np.uint64)
    for r, c, v in w2:
        want2[r, c] = v
    assert log2.overwrites()[0] == 1
    assert np.array_equal(prover.expand_trace(log2), want2)
    from bls_util import fp_arr
    b = _bls()
    hm = (fp_arr(b["hm_x1"], b["hm_x2"]), fp_arr(b["hm_y1"], b["hm_y2"]), fp_arr(b["hm_z1"], b["hm_z2"]))
    for fn, args in [(S.trace_fp12_mul, (random_fp12(0x5EED2212), random_fp12(0x5EED2213))), (S.trace_pairing_precomp, hm)]:
        d, _ = fn(*args)
        c, _ = fn(*args, compact=True)
        assert np.array_equal(prover.expand_trace(c), d)


def test_one_compact_trace_is_proven_by_several_contexts_at_once_and_recorded_on_threads(prover):
    """A finished trace log is immutable: four contexts on four host threads prove the same one concurrently (the bench's
    shape), and four threads record different traces at the same time (recording is armed per thread)."""
    import threading
    air = S.AIR_FP12_MUL
    cfg = S.StarkConfig.for_air(air)
    inputs = [(random_fp12(0x5EED2300 + 2 * i), random_fp12(0x5EED2301 + 2 * i)) for i in range(4)]
    want = []
    for x, y in inputs:
        t, pis = S.trace_fp12_mul(x, y)
        want.append(prover.prove(air, cfg, t, pis))
    # record on four threads at once
    logs = [None] * 4

    def record(i):
        logs[i] = S.trace_fp12_mul(*inputs[i], compact=True)
    th = [threading.Thread(target=record, args=(i,)) for i in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    provers = [S.Prover(0) for _ in range(4)]
    got = [[None] * 4 for _ in range(4)]

    def prove_all(k):  # context k proves every log, starting at a different one
        for j in range(4):
            i = (k + j) % 4
            got[k][i] = provers[k].prove(air, cfg, logs[i][0], logs[i][1])
    th = [threading.Thread(target=prove_all, args=(k,)) for k in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for pv in provers:
        pv.close()
    for k in range(4):
        for i in range(4):
            assert np.array_equal(got[k][i], want[i]), (k, i)


def test_both_constraint_evaluators_give_the_same_proof():
    """The tiled evaluator (default: quotient_plan.h, LDS-staged column tiles) and the op-stream interpreter are two
    independent device implementations of the same fold; proofs from either must be identical bytes, for a 1024-row AIR
    with several chunkings and for the small-domain path (FP12Mul: 32 points, less than a wave)."""
    from bls_util import fp_arr, random_fp12
    b = _bls()
    cases = [(S.AIR_PAIRING_PRECOMP, S.trace_pairing_precomp(fp_arr(b["hm_x1"], b["hm_x2"]), fp_arr(b["hm_y1"], b["hm_y2"]),
                                                              fp_arr(b["hm_z1"], b["hm_z2"]))),
             (S.AIR_FP12_MUL, S.trace_fp12_mul(random_fp12(7), random_fp12(8)))]
    pv = S.Prover(0)
    try:
        for air, (t, pis) in cases:
            cfg = S.StarkConfig.for_air(air)
            pv.set_option("quotient_impl", 1)
            ref = pv.prove(air, cfg, t, pis)
            pv.set_option("quotient_impl", 0)
            for chunks in (0, 1, 5):
                pv.set_option("quotient_chunks", chunks)
                assert np.array_equal(pv.prove(air, cfg, t, pis), ref), (S.AIR_NAMES[air], chunks)
        with pytest.raises(S.StarkhipError):
            pv.set_option("no_such_option", 1)
    finally:
        pv.close()


@pytest.mark.parametrize("air_name", ["fp12_mul", "pairing_precomp"])
def test_handoff_round_trip_of_gpu_proofs(prover, air_name):
    """Proof hand-off on what the GPU path produces: the blob of a real AIR's proof -> the nested
    StarkProofWithPublicInputs value (the shape src/aggregate_proof.rs:435-439 consumes) -> JSON -> blob is lossless, the value has
    the AIR's column / quotient / FRI-layer counts, and the round-tripped blob is still accepted by the verifier."""
    from starky_bls12_381_amd import handoff as H
    if air_name == "fp12_mul":
        air = S.AIR_FP12_MUL
        t, pis = S.trace_fp12_mul(random_fp12(0x5EED5000), random_fp12(0x5EED5001), compact=True)
    else:
        from test_gpu_pool import _precomp_args
        air = S.AIR_PAIRING_PRECOMP
        t, pis = S.trace_pairing_precomp(*_precomp_args(0x5EED5010), compact=True)
    cfg = S.StarkConfig.for_air(air)
    proof = prover.prove(air, cfg, t, pis)
    lay = S.proof_layout(proof)
    v = H.proof_to_value(proof)
    assert v["public_inputs"] == [int(x) for x in pis]
    r0 = v["proof"]["opening_proof"]["query_round_proofs"][0]
    leaf, path = r0["initial_trees_proof"]["evals_proofs"][0]
    assert len(leaf) == S.air_columns(air) == lay.n_columns and len(path["siblings"]) == lay.initial_sibling_count
    assert len(r0["initial_trees_proof"]["evals_proofs"][1][0]) == lay.n_quotient_polys
    assert len(v["proof"]["openings"]["local_values"]) == lay.n_columns and len(v["proof"]["openings"]["quotient_polys"]) == lay.n_quotient_polys
    assert len(v["proof"]["opening_proof"]["commit_phase_merkle_caps"]) == lay.n_fri_layers
    back = H.loads(H.dumps(proof), degree_bits=lay.degree_bits, rate_bits=cfg.rate_bits, arity_bits=cfg.arity_bits, num_challenges=cfg.num_challenges)
    assert np.array_equal(back, proof)
    S.verify_stark_proof(air, cfg, back)


@pytest.mark.parametrize("k", [0, 5, 255])
def test_zeta_on_the_coset_of_the_kept_values_is_proven_like_the_reference(k):
    """The product opens the trace polynomials from their values on the coset 7 H (it keeps no coefficients); the reference evaluates
    coefficients.  For zeta ON that coset the interpolation weights degenerate -- the opening is the value itself -- and rounds 4-5 returned
    ERR_ZETA_IN_SUBGROUP there, which starky reserves for zeta in H.  The case has probability 2^-115 in a transcript, so it is forced: the
    "zeta_on_coset" option of the context and the oracle's hook substitute zeta = 7 w_n^k in both, and the proof bytes must agree (the
    openings at zeta and at g zeta are then entries k and k + 1 of coset 0 of the LDE).  Such a proof is not a transcript, so the verifier is
    not asked."""
    cfg = S.StarkConfig.standard_fast_config()
    pv = S.Prover(0)
    try:
        for air, (t, pis) in ((S.AIR_TEST_FIBONACCI, S.trace_fibonacci(3, 5, 256)),
                              (S.AIR_FP12_MUL, S.trace_fp12_mul(random_fp12(0x5EED2300), random_fp12(0x5EED2301)))):
            c = cfg if air == S.AIR_TEST_FIBONACCI else S.StarkConfig.for_air(air)
            n = t.shape[0]
            usual = pv.prove(air, c, t, pis)
            pv.set_option("zeta_on_coset", (k % n) + 1)
            O.lib.oracle_set_zeta_on_coset((k % n) + 1)
            try:
                forced = pv.prove(air, c, t, pis)
                ref = O.prove(S.air_program(air), c, S.trace_rows_to_poly_values(t), pis)
            finally:
                pv.set_option("zeta_on_coset", 0)
                O.lib.oracle_set_zeta_on_coset(0)
            assert forced.size == ref.size and np.array_equal(forced, ref)
            assert not np.array_equal(forced, usual)
            # zeta is a base-field point: so are the openings (a zero second word), and the next-row openings of k are the local ones of k + 1
            lay = S.proof_layout(forced)
            cols = S.air_columns(air)
            local = forced[lay.off_local_values:lay.off_local_values + 2 * cols].reshape(cols, 2)
            nxt = forced[lay.off_next_values:lay.off_next_values + 2 * cols].reshape(cols, 2)
            assert not local[:, 1].any() and not nxt[:, 1].any()
            pv.set_option("zeta_on_coset", ((k + 1) % n) + 1)
            try:
                shifted = pv.prove(air, c, t, pis)
            finally:
                pv.set_option("zeta_on_coset", 0)
            assert np.array_equal(shifted[lay.off_local_values:lay.off_local_values + 2 * cols].reshape(cols, 2), nxt)
            assert np.array_equal(pv.prove(air, c, t, pis), usual)   # the option is off again
    finally:
        pv.close()
