"""All six STARK proofs of one BLS signature check on one GPU (BASELINE.json configs[3], single-GPU form):
the reference's own test vector (src/native.rs:1480-1498), every proof accepted by the verifier, public inputs chained."""
import numpy as np
import pytest

import starky_bls12_381_amd as S
from starky_bls12_381_amd import aggregate as A
from test_aggregate_cpu import _bls_points

pytestmark = pytest.mark.gpu


def test_six_proofs_of_the_reference_signature(prover):
    _, pk, hm, sig = _bls_points()
    proofs, natives = A.prove_signature(prover, pk, hm, sig)
    assert sorted(proofs) == sorted(A.JOB_ORDER)
    assert A.signature_is_valid(natives)
    assert A.check_links(proofs)
    for name, (air, proof, cfg) in proofs.items():
        assert air == A.JOB_AIR[name]
        S.verify_stark_proof(air, cfg, proof)
        assert int(proof[0]) == 0x3130304652505353  # blob magic


def test_seven_proofs_of_the_mainnet_update(prover):
    """The light-client update the reference's main.rs proves (period 1053, keys from period 1052): ECCAggStark over the 512
    committee keys, then the six pairing proofs on its aggregate; every proof verifies and the public inputs chain."""
    import os
    from bls_util import GOLDEN
    from starky_bls12_381_amd import eth_input as E
    upd = E.load_update(os.path.join(GOLDEN, "light_client_update_period_1053.json"), os.path.join(GOLDEN, "light_client_update_period_1052.json"))
    ec = A.ec_aggregate_main(prover, upd["points"], upd["bits"])
    agg = np.asarray(ec[1][-24:], dtype=np.uint32)
    proofs, natives = A.prove_signature(prover, (agg[:12], agg[12:]), upd["hm"], upd["sig"])
    assert A.signature_is_valid(natives)
    proofs["ec"] = ec
    assert A.check_links(proofs)
    bad = dict(proofs)
    t = ec[1].copy()
    t[-1] ^= 1
    bad["ec"] = (ec[0], t, ec[2])
    assert not A.check_links(bad)


def test_cpp_driver_proves_the_reference_signature():
    """include/starkhip_driver.hpp (the C++ mirror of src/aggregate_proof.rs's drivers) through tools/signature_demo.cpp."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build", "signature_demo")
    if not os.path.exists(exe):
        pytest.skip("build/signature_demo not built (make demo)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "valid=1 linked=1" in r.stdout


def test_batch_of_eight_different_signatures_end_to_end():
    """BASELINE.json configs[4] in single-GPU form: EIGHT DIFFERENT valid signatures = 48 proofs through the end-to-end driver
    (operands -> trace generation on host threads, overlapped with proving on six contexts -> proofs).  Every proof is accepted
    by the verifier; links and statement hold for all eight; the statement check rejects another signature's points."""
    from bls_util import native_vectors
    from starky_bls12_381_amd import signature as G
    batch = 8
    sigs = G.synthetic_signatures(batch, native_vectors()["bls_signature"], seed=0x8516)
    mine = G.plan_batch(batch, 1)[0]
    assert len(mine) == 48
    # the product path, exactly what tools/bench_signature.py times: signature.one_step on the library's proof pool
    # (run_jobs_pool: submit_witness per job, natives on host threads, merged commitments, lane-form groups for the FinalExp proofs)
    pool = S.ProofPool(0, big_contexts=6, small_contexts=12, stream_priority=1, warm_up=1)
    try:
        elapsed, results, stats, sigs_seen, natives = G.one_step(None, batch, pool, mine, sigs)
        pstats = pool.stats()
    finally:
        pool.close()
    assert sorted(results) == sorted(mine) and elapsed > 0
    assert all(np.array_equal(a, b) for s0, s1 in zip(sigs, sigs_seen) for a, b in zip(s0, s1))
    assert pstats["big_commit_launches"] == 8 and pstats["small_commit_requests"] == 32   # (the eight FP12Mul commitments are hashed on the host)
    assert pstats["max_merged_commitments"] >= 2   # small commitments shared launches
    for (_, name), (air, proof, cfg) in results.items():
        assert air == A.JOB_AIR[name]
        S.verify_stark_proof(air, cfg, proof)
    digests = set()
    for i in range(batch):
        six = G.signature_proofs(results, i)
        assert A.check_links(six) and A.check_statement(six, sigs[i][0], sigs[i][1], sigs[i][2])
        assert A.signature_is_valid(natives[i], six)
        assert not A.check_statement(six, sigs[i][0], sigs[(i + 1) % batch][1], sigs[(i + 1) % batch][2])
        assert not A.check_statement(six, sigs[(i + 1) % batch][0], sigs[i][1], sigs[i][2])   # another key: not this statement
        digests.add(bytes(six["final_exp"][1][16:80]))
    assert len(digests) == batch   # eight different FinalExp proofs (different trace caps)
    assert stats["wall_s"] < stats["generate_s"] + stats["prove_s"]   # generation overlapped proving


def _demo(*args, timeout=900):
    import json
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "build", "signature_demo")
    if not os.path.exists(exe):
        pytest.skip("build/signature_demo not built (make demo)")
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout + r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_one_process_two_pools_gives_the_one_pool_proofs():
    """starkhip_multipool_*: the reference's single-process caller (src/aggregate_proof.rs:304-370) on several devices.  On the
    one-GPU box the two "devices" are two pools on the same card (`--devices 0,0`): a batch of eight different signatures = 48
    proofs, every one verified, linked and bound to its statement, and BYTE-EQUAL to the proofs one pool makes of the same
    operands (proof bytes do not depend on the device, pool or context); both pools proved FinalExp proofs -- four each."""
    import os
    ops = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "signature_operands_8.bin")
    common = ["--batch", 8, "--operands", ops, "--steps", 1, "--warmup", 0, "--digests"]
    one = _demo(*common, "--big", 6, "--small", 12)
    two = _demo(*common, "--devices", "0,0", "--big", 3, "--small", 6)
    assert one["pools"] == 1 and two["pools"] == 2
    for r in (one, two):
        assert r["proofs_verified_after_timing"] == 48 and r["signatures_valid_linked_bound"] == 8 and len(r["proof_digests"]) == 48
    assert one["proof_digests"] == two["proof_digests"]
    assert len(set(one["proof_digests"])) == 48
    assert [p["big"] for p in two["commit_launches_per_pool"]] == [4, 4]          # FinalExp-class proofs: longest first, spread evenly
    assert all(p["small_requests"] > 0 for p in two["commit_launches_per_pool"])
    assert two["hw_queues_late"] == 0   # a process of its own: the library's GPU_MAX_HW_QUEUES was in time


def test_multi_device_handle_from_python_places_jobs_and_matches_single_pool_bytes():
    """The same handle through ctypes: a batch of one signature's six jobs on THREE pools of the one card -- the plan is
    starkhip_plan_lpt's (FinalExp alone on slot 0, the two MillerLoops on slots 1 and 2, the rest behind the shorter queue) --,
    jobs named to a slot stay there, and the proofs equal those of a plain pool."""
    _, pk, hm, sig = _bls_points()
    jobs, natives = A.signature_jobs(pk, hm, sig)
    batch = [(A.JOB_AIR[name],) + tuple(jobs[name][1]) for name in A.JOB_ORDER]
    plan = S.api.plan_lpt([b[0] for b in batch], 3)
    ref = S.ProofPool(0, big_contexts=1, small_contexts=4, warm_up=1)
    try:
        want = [ref.wait(t)[0] for t in ref.submit_witness_batch(batch)]
    finally:
        ref.close()
    mp = S.ProofPool(big_contexts=1, small_contexts=2, warm_up=1, devices=[0, 0, 0])
    try:
        assert mp.n_pools == 3
        tickets = mp.submit_witness_batch(batch)
        assert [mp.slot_of(t) for t in tickets] == plan
        assert plan[A.JOB_ORDER.index("final_exp")] == 0 and sorted(plan[A.JOB_ORDER.index(n)] for n in ("ml1", "ml2")) == [1, 2]
        pinned = mp.submit_witness(batch[0][0], *batch[0][1:], slot=2)
        assert mp.slot_of(pinned) == 2
        got = [mp.wait(t)[0] for t in tickets]
        assert np.array_equal(mp.wait(pinned)[0], want[0])
        # the other hand-over forms through the handle: a recorded trace, dense rows, separately allocated columns
        air = A.JOB_AIR["pp1"]
        cfg = S.StarkConfig.for_air(air)
        args = jobs["pp1"][1]
        compact, cpis = S.trace_pairing_precomp(*args, compact=True)
        dense, pis = S.trace_pairing_precomp(*args)
        cols = [np.ascontiguousarray(dense[:, c]) for c in range(dense.shape[1])]
        t_forms = [mp.submit(air, cfg, compact, cpis), mp.submit(air, cfg, dense, pis, slot=1), mp.submit_columns(air, cfg, cols, pis)]
        assert mp.slot_of(t_forms[1]) == 1
        for t in t_forms:
            assert np.array_equal(mp.wait(t)[0], want[A.JOB_ORDER.index("pp1")])
        per_pool = mp.stats(per_pool=True)
    finally:
        mp.close()
    for name, a, b in zip(A.JOB_ORDER, got, want):
        assert np.array_equal(a, b), name
    assert per_pool[0]["big_commit_launches"] == 1 and per_pool[1]["big_commit_launches"] == 0
