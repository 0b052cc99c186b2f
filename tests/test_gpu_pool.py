"""The library's proof pool (starkhip_pool_*: submit / wait, merged trace commitments) on the GPU: proofs from the pool are
byte-identical to the ones from a plain context -- and so to the CPU oracle's -- whatever shares the chip with them."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import starky_bls12_381_amd as S
from bls_util import GOLDEN, random_fp, random_fp12, splitmix64

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fp2(gen):
    return np.concatenate([np.array([(random_fp(gen) >> (32 * i)) & 0xFFFFFFFF for i in range(12)], dtype=np.uint32) for _ in range(2)])


def _precomp_args(seed):
    g = splitmix64(seed)
    one = np.zeros(24, dtype=np.uint32)
    one[0] = 1
    return _fp2(g), _fp2(g), one


def test_merged_commitments_give_the_same_proofs_as_one_context(prover):
    """Seven small proofs of three AIRs submitted at once: the PairingPrecomp ones share one merged leaf-hash launch, the toy AIR's
    another; the FP12Mul commitments (32 leaves) are hashed by host threads; every proof equals the single-context proof byte for
    byte (FP12Mul also the oracle's)."""
    pool = S.ProofPool(0, big_contexts=1, small_contexts=7, generator_threads=4)
    try:
        jobs = []
        for k in range(3):
            x, y = random_fp12(0x5EED3000 + 2 * k), random_fp12(0x5EED3001 + 2 * k)
            jobs.append((S.AIR_FP12_MUL, S.trace_fp12_mul(x, y)))
        for k in range(2):
            jobs.append((S.AIR_PAIRING_PRECOMP, S.trace_pairing_precomp(*_precomp_args(0x5EED3100 + k))))
        for k in range(2):
            jobs.append((S.AIR_TEST_FIBONACCI, S.trace_fibonacci(3 + k, 5, 256)))
        tickets = [pool.submit(air, S.StarkConfig.for_air(air), t, pis) for air, (t, pis) in jobs]
        got = [pool.wait(t)[0] for t in tickets]
        stats = pool.stats()
    finally:
        pool.close()
    assert stats["small_commit_requests"] == 4 and stats["big_commit_launches"] == 0     # the three FP12Mul commitments never reach the scheduler
    assert stats["small_commit_launches"] < 4 and stats["max_merged_commitments"] >= 2   # some commitments shared a launch
    for (air, (t, pis)), proof in zip(jobs, got):
        cfg = S.StarkConfig.for_air(air)
        S.verify_stark_proof(air, cfg, proof)
        assert np.array_equal(proof, prover.prove(air, cfg, t, pis))
    air, (t, pis) = jobs[0]
    assert np.array_equal(got[0], O.prove(S.air_program(air), S.StarkConfig.for_air(air), S.trace_rows_to_poly_values(t), pis))


def test_witness_jobs_generate_inside_the_pool():
    """starkhip_pool_submit_witness: generate_trace + prove from the driver's operands, as src/aggregate_proof.rs:23-179 does in
    one function each; same bytes as recording the trace here and proving it on a context."""
    x, y = random_fp12(0x5EED3200), random_fp12(0x5EED3201)
    args = _precomp_args(0x5EED3210)
    pool = S.ProofPool(0, big_contexts=1, small_contexts=3, generator_threads=3)
    pv = S.Prover(0)
    try:
        t1 = pool.submit_witness(S.AIR_FP12_MUL, x, y)
        t2 = pool.submit_witness(S.AIR_PAIRING_PRECOMP, *args)
        t3 = pool.submit_witness(S.AIR_TEST_FIBONACCI, 3, 5)
        p1, info1 = pool.wait(t1)
        p2, _ = pool.wait(t2)
        p3, _ = pool.wait(t3)
        for air, proof, gen in ((S.AIR_FP12_MUL, p1, lambda: S.trace_fp12_mul(x, y, compact=True)),
                                (S.AIR_PAIRING_PRECOMP, p2, lambda: S.trace_pairing_precomp(*args, compact=True))):
            cfg = S.StarkConfig.for_air(air)
            trace, pis = gen()
            assert np.array_equal(proof[-pis.size:], pis)
            assert np.array_equal(proof, pv.prove(air, cfg, trace, pis))
        t, pis = S.trace_fibonacci(3, 5)
        assert np.array_equal(p3, pv.prove(S.AIR_TEST_FIBONACCI, S.StarkConfig.for_air(S.AIR_TEST_FIBONACCI), t, pis))
        tl = info1["timeline_s"]
        assert tl[0] <= tl[1] <= tl[2] <= tl[3] <= tl[4] and info1["phase_ms"]["total"] > 0
    finally:
        pv.close()
        pool.close()


def test_a_warmed_pool_hands_out_recycled_page_locked_blobs():
    """warm_up=1: every context reserves page-locked proof blobs (blob_arena.h); proofs come out of them bit-identical to the
    ones a lone context writes, starkhip_free hands them back, and they go away with the pool."""
    before = S.proof_blob_stats()
    x, y = random_fp12(0x5EED3300), random_fp12(0x5EED3301)
    args = _precomp_args(0x5EED3310)
    pool = S.ProofPool(0, big_contexts=1, small_contexts=2, generator_threads=2, warm_up=1)
    pv = S.Prover(0)
    try:
        held = S.proof_blob_stats()
        assert held["blobs"] - before["blobs"] == 2 + 2 * 2 and held["busy"] == before["busy"]   # FinalExp x 2; (MillerLoop, Precomp) per small context
        for rep in range(3):   # more proofs than blobs: they are recycled
            t1 = pool.submit_witness(S.AIR_FP12_MUL, x, y)
            t2 = pool.submit_witness(S.AIR_PAIRING_PRECOMP, *args)
            p1, _ = pool.wait(t1)
            p2, _ = pool.wait(t2)
        now = S.proof_blob_stats()
        assert now["taken"] - before["taken"] >= 6 and now["busy"] == before["busy"]
        for air, proof, gen in ((S.AIR_FP12_MUL, p1, lambda: S.trace_fp12_mul(x, y, compact=True)),
                                (S.AIR_PAIRING_PRECOMP, p2, lambda: S.trace_pairing_precomp(*args, compact=True))):
            trace, pis = gen()
            assert np.array_equal(proof, pv.prove(air, S.StarkConfig.for_air(air), trace, pis))
    finally:
        pv.close()
        pool.close()
    after = S.proof_blob_stats()
    assert after["blobs"] == before["blobs"] and after["bytes"] == before["bytes"]


def test_a_failing_job_reports_its_code_and_the_pool_goes_on():
    """A witness that does not satisfy the AIR (a trace cell changed) fails with the prover's own error code at wait();
    proofs submitted with it and after it are unaffected (a failed small proof must not hold the merged window open)."""
    air = S.AIR_PAIRING_PRECOMP   # degree 4 at blow-up 4: a broken constraint shows in the quotient's zero chunk ("Quotient has failed")
    cfg = S.StarkConfig.for_air(air)
    t, pis = S.trace_pairing_precomp(*_precomp_args(0x5EED3300))
    bad = t.copy()
    bad[200, 15000] = (int(bad[200, 15000]) + 1) % S.P
    pool = S.ProofPool(0, big_contexts=1, small_contexts=2, generator_threads=1)
    try:
        tb, tg = pool.submit(air, cfg, bad, pis), pool.submit(air, cfg, t, pis)
        with pytest.raises(S.StarkhipError) as e:
            pool.wait(tb)
        assert e.value.code == S.ERR_QUOTIENT_NOT_DIVISIBLE
        good, _ = pool.wait(tg)
        S.verify_stark_proof(air, cfg, good)
        with pytest.raises(S.StarkhipError) as e:
            pool.submit(air, cfg, t[:, :-1], pis)   # wrong shape: refused at submit
        assert e.value.code == S.ERR_BAD_SHAPE
        pis_bad = pis.copy()
        pis_bad[0] = np.uint64(S.P)                   # not canonical: refused by prove() before any GPU work, before the commitment
        t_early = pool.submit(air, cfg, t, pis_bad)
        again = pool.submit(air, cfg, t, pis)
        with pytest.raises(S.StarkhipError) as e:
            pool.wait(t_early)
        assert e.value.code == S.ERR_BAD_SHAPE
        assert np.array_equal(pool.wait(again)[0], good)
        with pytest.raises(S.StarkhipError):
            pool.wait(again)                          # a ticket is waited for once
    finally:
        pool.close()


def test_final_exp_four_in_flight_match_the_oracle_digest():
    """FOUR FinalExp proofs in flight on one GPU, each commitment a launch of its own.  All four prove the benchmark's first seeded
    input (0x5EED0001); every proof's SHA-256 equals the CPU oracle's (tests/golden/final_exp_seed_5eed0001_proof.sha256, made by
    tests/make_final_exp_golden.py --seed 0x5EED0001 on the GPU box's host) -- contention changes nothing in the bytes."""
    air = S.AIR_FINAL_EXP
    cfg = S.StarkConfig.for_air(air)
    want = open(os.path.join(GOLDEN, "final_exp_seed_5eed0001_proof.sha256")).read().split()[0]
    trace, pis = S.trace_final_exp(random_fp12(0x5EED0001))
    cols = S.trace_rows_to_poly_values(trace)   # what the reference hands to prove(): column vectors
    del trace
    pool = S.ProofPool(0, big_contexts=4, small_contexts=1, generator_threads=1)
    try:
        tickets = [pool.submit(air, cfg, cols, pis, layout=1) for _ in range(4)]
        got = [pool.wait(t) for t in tickets]
        digests = [hashlib.sha256(pr.tobytes()).hexdigest() for pr, _ in got]
        stats = pool.stats()
    finally:
        pool.close()
    assert digests == [want] * 4
    assert stats["big_commit_launches"] == 4
    # below five big contexts a big commitment goes out on its own, in the pair form (two lanes per leaf; kernels_hash.hip)
    assert {info["leaf_hash_form"] for _, info in got} == {"pair"}


def test_final_exp_in_lane_form_groups_match_the_oracle_digest(monkeypatch):
    """Pools with five or more FinalExp-class contexts send their big commitments out in GROUPS in the lane form (one lane per leaf,
    scheduled asm rounds; scheduler.cpp).  Forced here on four contexts: eight proofs of the benchmark's first seeded input, two groups
    of four commitments -- every proof's SHA-256 is the CPU oracle's."""
    air = S.AIR_FINAL_EXP
    cfg = S.StarkConfig.for_air(air)
    want = open(os.path.join(GOLDEN, "final_exp_seed_5eed0001_proof.sha256")).read().split()[0]
    trace, pis = S.trace_final_exp(random_fp12(0x5EED0001))
    cols = S.trace_rows_to_poly_values(trace)
    del trace
    monkeypatch.setenv("STARKHIP_POOL_BIG_LANE", "1")   # read by starkhip_pool_create
    pool = S.ProofPool(0, big_contexts=4, small_contexts=1, generator_threads=1)
    try:
        tickets = [pool.submit(air, cfg, cols, pis, layout=1) for _ in range(8)]
        digests = [hashlib.sha256(pool.wait(t)[0].tobytes()).hexdigest() for t in tickets]
        stats = pool.stats()
    finally:
        pool.close()
    assert digests == [want] * 8
    assert stats["big_commit_launches"] == 8


def test_failures_of_final_exp_class_jobs_inside_lane_groups():
    """scheduler.cpp's announce / withdraw / group logic for the 8192-row class with REAL kernels (the sanitizer harness runs it against
    a fake device): a pool of five big contexts sends FinalExp-class commitments out in lane-form groups that wait for proofs which
    have started.  (FinalExp itself -- degree 5 at blow-up 4, quotient_degree_factor == blow-up -- has no quotient chunk that must
    vanish, so like starky's prove() it cannot notice a broken trace; the 8192-row AIR that can is ECCAgg, degree 4: its commitment
    joins the same groups.)
      1. two proofs on five contexts are not held for a group that cannot fill (the scheduler's bound for jobs that have not started);
      2. three good FinalExp jobs + one whose public input is not canonical: that one fails BEFORE its commitment (ERR_BAD_SHAPE) and
         withdraws; the group of three goes out without waiting its bound for it;
      3. four good FinalExp jobs + an ECCAgg job with a changed trace cell: it fails AFTER its commitment (the quotient's top chunk:
         ERR_QUOTIENT_NOT_DIVISIBLE), having been a member of a group;
    every good proof hashes to the CPU oracle's digest of the benchmark's first seeded input, and the pool proves on afterwards."""
    from test_ecc_aggregate_cpu import pack, reference_vector
    air = S.AIR_FINAL_EXP
    cfg = S.StarkConfig.for_air(air)
    want = open(os.path.join(GOLDEN, "final_exp_seed_5eed0001_proof.sha256")).read().split()[0]
    x = random_fp12(0x5EED0001)
    compact, pis = S.trace_final_exp(x, compact=True)
    pis_bad = pis.copy()
    pis_bad[3] = np.uint64(S.P)   # not a canonical field element
    pts, bits, _ = reference_vector()
    ecc_cfg = S.StarkConfig.for_air(S.AIR_ECC_AGGREGATE)
    ecc_t, ecc_pis = S.trace_ecc_aggregate(*pack(pts, bits))
    ecc_bad = ecc_t.copy()
    ecc_bad[4000, 2000] = (int(ecc_bad[4000, 2000]) + 1) % S.P   # a cell the constraints of rows 3999 and 4000 read (not every cell is constrained on every row)
    sha = lambda pr: hashlib.sha256(pr.tobytes()).hexdigest()  # noqa: E731
    pool = S.ProofPool(0, big_contexts=5, small_contexts=1, generator_threads=3, warm_up=1)
    try:
        # 1.
        got = [pool.wait(t) for t in [pool.submit_witness(air, x) for _ in range(2)]]
        assert [sha(p) for p, _ in got] == [want] * 2
        assert all(i["timeline_s"][4] - i["timeline_s"][0] < 1.0 for _, i in got), [i["timeline_s"] for _, i in got]
        # 2.
        t_good = [pool.submit(air, cfg, compact, pis) for _ in range(3)]
        t_bad = pool.submit(air, cfg, compact, pis_bad)
        with pytest.raises(S.StarkhipError) as e:
            pool.wait(t_bad)
        assert e.value.code == S.ERR_BAD_SHAPE
        got = [pool.wait(t) for t in t_good]
        assert [sha(p) for p, _ in got] == [want] * 3
        # the three go out together -- unless the first had reached its commitment before this thread had submitted the others (seen once
        # in the full suite, never alone: the submissions are four Python calls): then it goes out alone in the pair form and the two others as a group
        forms = sorted((i["leaf_hash_form"], i["leaf_hash_group"]) for _, i in got)
        assert forms in ([("lane", 3)] * 3, [("lane", 2), ("lane", 2), ("pair", 1)]), forms
        assert all(i["timeline_s"][4] - i["timeline_s"][3] < 1.0 for _, i in got), [i["timeline_s"] for _, i in got]   # nobody waited 1 s for the one that withdrew
        # 3.
        t_ecc = pool.submit(S.AIR_ECC_AGGREGATE, ecc_cfg, ecc_bad, ecc_pis)
        t_good = [pool.submit(air, cfg, compact, pis) for _ in range(4)]
        with pytest.raises(S.StarkhipError) as e:
            pool.wait(t_ecc)
        assert e.value.code == S.ERR_QUOTIENT_NOT_DIVISIBLE
        got = [pool.wait(t) for t in t_good]
        assert [sha(p) for p, _ in got] == [want] * 4
        assert all(i["timeline_s"][4] - i["timeline_s"][3] < 1.5 for _, i in got)
        stats = pool.stats()
        assert stats["big_commit_launches"] == 2 + 3 + 5   # the ECCAgg job reached its commitment, the bad-input job did not
        # the pool goes on: the same ECCAgg statement with its real trace, and one more FinalExp proof
        ok = pool.wait(pool.submit(S.AIR_ECC_AGGREGATE, ecc_cfg, ecc_t, ecc_pis))[0]
        S.verify_stark_proof(S.AIR_ECC_AGGREGATE, ecc_cfg, ok)
        assert sha(pool.wait(pool.submit_witness(air, x))[0]) == want
    finally:
        pool.close()


def _signature_points(count, seed):
    """`count` different valid (pk, H(m), signature) triples on the curve (signature.synthetic_signatures over the reference's vector)."""
    from bls_util import native_vectors
    from starky_bls12_381_amd import signature as G
    return G.synthetic_signatures(count, native_vectors()["bls_signature"], seed=seed)


def test_batch_regime_miller_and_precomp_beside_a_lane_group_match_the_oracle(monkeypatch):
    """The regime the batch-of-signatures figure is measured in (tools/bench_signature.py, build/signature_demo --batch): MillerLoop and
    PairingPrecomp commitments merged in leaf_hash_multi_kernel windows WHILE a FinalExp lane-form group is on the chip.  Two
    MillerLoop + two PairingPrecomp witness jobs and four FinalExp jobs (the benchmark's first seeded input) go in together;
    every small proof equals the CPU oracle's proof of the same trace byte for byte, every FinalExp proof the oracle's digest."""
    monkeypatch.setenv("STARKHIP_POOL_BIG_LANE", "1")   # lane-form groups on four big contexts (pools of five or more do it by themselves)
    want_fe = open(os.path.join(GOLDEN, "final_exp_seed_5eed0001_proof.sha256")).read().split()[0]
    x_fe = random_fp12(0x5EED0001)
    sigs = _signature_points(2, 0x5EED3400)
    ml = [(pk[0], pk[1], hm[0], hm[1], hm[2]) for pk, hm, _ in sigs]     # miller_loop_main(pk, H(m)), src/aggregate_proof.rs:117-148
    pp = [(sig[0], sig[1], sig[2]) for _, _, sig in sigs]                  # calc_pairing_precomp(signature), :23-69
    pool = S.ProofPool(0, big_contexts=4, small_contexts=4, generator_threads=4, warm_up=1)
    try:
        t_fe = [pool.submit_witness(S.AIR_FINAL_EXP, x_fe) for _ in range(4)]
        t_ml = [pool.submit_witness(S.AIR_MILLER_LOOP, *a) for a in ml]
        t_pp = [pool.submit_witness(S.AIR_PAIRING_PRECOMP, *a) for a in pp]
        got_ml = [pool.wait(t) for t in t_ml]
        got_pp = [pool.wait(t) for t in t_pp]
        got_fe = [pool.wait(t) for t in t_fe]
        stats = pool.stats()
    finally:
        pool.close()
    assert [hashlib.sha256(p.tobytes()).hexdigest() for p, _ in got_fe] == [want_fe] * 4
    assert {info["leaf_hash_form"] for _, info in got_fe} == {"lane"} and max(info["leaf_hash_group"] for _, info in got_fe) >= 2
    assert stats["small_commit_requests"] == 4 and stats["big_commit_launches"] == 4
    # the small commitments went through the scheduler's merged windows (quad form, grid.y = proofs), beside the big ones
    assert {info["leaf_hash_form"] for _, info in got_ml + got_pp} == {"merged"}
    fe_span = (min(info["timeline_s"][3] for _, info in got_fe), max(info["timeline_s"][4] for _, info in got_fe))
    assert all(fe_span[0] < info["timeline_s"][4] and info["timeline_s"][3] < fe_span[1] for _, info in got_ml + got_pp)   # they overlapped in time
    for air, gen, args, got in ((S.AIR_MILLER_LOOP, S.trace_miller_loop, ml, got_ml), (S.AIR_PAIRING_PRECOMP, S.trace_pairing_precomp, pp, got_pp)):
        cfg = S.StarkConfig.for_air(air)
        for a, (proof, _) in zip(args, got):
            trace, pis = gen(*a)
            assert np.array_equal(proof, O.prove(S.air_program(air), cfg, S.trace_rows_to_poly_values(trace), pis))


def test_prove_from_separately_allocated_columns_matches_the_oracle(prover):
    """starky's literal `prove(stark, &config, trace_poly_values, ..)`: `Vec<PolynomialValues<F>>` is one heap allocation per column
    (src/aggregate_proof.rs:57-65 for FP12MulStark).  60 285 separately allocated numpy columns through starkhip_prove_columns and
    through starkhip_pool_submit_columns: the oracle's bytes both times; a NULL column or a wrong count is refused."""
    import ctypes as C
    air = S.AIR_FP12_MUL
    cfg = S.StarkConfig.for_air(air)
    trace, pis = S.trace_fp12_mul(random_fp12(0x5EED3500), random_fp12(0x5EED3501))
    columns = [trace[:, c].copy() for c in range(trace.shape[1])]   # 60 285 separate allocations
    assert len({c.ctypes.data for c in columns}) == len(columns)
    want = O.prove(S.air_program(air), cfg, S.trace_rows_to_poly_values(trace), pis)
    assert np.array_equal(prover.prove_columns(air, cfg, columns, pis), want)
    pool = S.ProofPool(0, big_contexts=1, small_contexts=2, generator_threads=1)
    try:
        t1 = pool.submit_columns(air, cfg, columns, pis)
        t2 = pool.submit_columns(air, cfg, columns, pis)
        assert np.array_equal(pool.wait(t1)[0], want) and np.array_equal(pool.wait(t2)[0], want)
        with pytest.raises(S.StarkhipError) as e:
            pool.submit_columns(air, cfg, columns[:-1], pis)
        assert e.value.code == S.ERR_BAD_SHAPE
    finally:
        pool.close()
    # a bigger shape, many halves of the staging: PairingPrecomp (29 376 columns x 1024 rows = 240 MB)
    air = S.AIR_PAIRING_PRECOMP
    cfg = S.StarkConfig.for_air(air)
    trace, pis = S.trace_pairing_precomp(*_precomp_args(0x5EED3510))
    columns = [trace[:, c].copy() for c in range(trace.shape[1])]
    assert np.array_equal(prover.prove_columns(air, cfg, columns, pis), prover.prove(air, cfg, trace, pis))
    table = (C.c_void_p * len(columns))(*[c.ctypes.data for c in columns])
    table[17] = None
    out, words = C.POINTER(C.c_uint64)(), C.c_size_t()
    rc = S.lib.starkhip_prove_columns(prover._ctx, air, C.byref(cfg), table, trace.shape[0], len(columns), pis.ctypes.data_as(C.POINTER(C.c_uint64)),
                                      pis.size, S.POW_SEARCH, C.byref(out), C.byref(words))
    assert rc == S.ERR_BAD_SHAPE


def test_pool_reports_its_reservation():
    """starkhip_pool_reservation: a warmed FinalExp-class context holds the LDE (the trace waits for it inside that buffer, uploads are
    staged in it, coefficients are not kept) and small buffers -- under 20 GB, where rounds 1-3 held 30 (values + coefficients + staging
    + LDE)."""
    pool = S.ProofPool(0, big_contexts=1, small_contexts=1, generator_threads=1, warm_up=1)
    try:
        r = pool.reservation()
    finally:
        pool.close()
    C_, n = S.air_columns(S.AIR_FINAL_EXP), 8192
    assert r["big_contexts"] == 1 and r["small_contexts"] == 1
    assert 8 * C_ * n * 4 <= r["big_context_device_bytes"] <= 20e9
    assert r["device_bytes"] >= r["big_context_device_bytes"] + r["small_context_device_bytes"] > r["big_context_device_bytes"]
    assert r["pinned_host_bytes"] >= 200 << 20


def test_cpp_demo_proves_a_batch_on_the_pool():
    """tools/signature_demo.cpp --batch 2: compiled host code above the C ABI only; 12 proofs in flight, all verified, linked and
    bound to their statements (exit code 0)."""
    exe = os.path.join(ROOT, "build", "signature_demo")
    ops = os.path.join(GOLDEN, "signature_operands_8.bin")
    if not os.path.exists(exe):
        pytest.skip("build/signature_demo not built (make demo)")
    r = subprocess.run([exe, "--batch", "2", "--operands", ops, "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert '"proofs_verified_after_timing": 12' in r.stdout and '"signatures_valid_linked_bound": 2' in r.stdout
