"""Host driver around the hot path (starky_bls12_381_amd/aggregate.py, mirror of src/aggregate_proof.rs:23-179,304-370):
job plan, natives and public-input links of the six proofs of one signature check.  No GPU: proofs are stood in for by
blobs that end in the public inputs the trace generators produce."""
import numpy as np

import starky_bls12_381_amd as S
from starky_bls12_381_amd import aggregate as A
from bls_util import native_vectors


def _bls_points():
    b = {k: int(s) for k, s in native_vectors()["bls_signature"].items()}
    pk = (A.fp_limbs(b["pk_x"]), A.fp_limbs(b["pk_y"]))
    hm = (A.fp2_limbs(b["hm_x1"], b["hm_x2"]), A.fp2_limbs(b["hm_y1"], b["hm_y2"]), A.fp2_limbs(b["hm_z1"], b["hm_z2"]))
    sig = (A.fp2_limbs(b["s_x1"], b["s_x2"]), A.fp2_limbs(b["s_y1"], b["s_y2"]), A.fp2_limbs(b["s_z1"], b["s_z2"]))
    return b, pk, hm, sig


def test_neg_generator_constant_matches_the_reference_vector():
    from bls_util import BLS_P
    b, _, _, _ = _bls_points()
    # src/native.rs:1491-1492 holds G and negates the signature; src/aggregate_proof.rs:336-337 holds -G and keeps the signature
    assert (A.NEG_G1_X, A.NEG_G1_Y) == (b["gx"], BLS_P - b["gy"])


def test_signature_plan_covers_every_job_once_and_isolates_final_exp():
    for world in (1, 2, 3, 6, 8):
        plan = A.signature_plan(world)
        assert len(plan) == world
        flat = sorted(n for r in plan for n in r)
        assert flat == sorted(A.JOB_ORDER)
        if world >= 2:
            fe_rank = [r for r in plan if "final_exp" in r][0]
            assert fe_rank == ["final_exp"]  # FinalExp is ~73 % of the work: nothing else shares its GPU


def test_natives_say_the_reference_signature_is_valid_and_links_hold():
    _, pk, hm, sig = _bls_points()
    jobs, natives = A.signature_jobs(pk, hm, sig)
    assert list(jobs) == list(A.JOB_ORDER)
    assert A.signature_is_valid(natives)  # src/native.rs:1522-1526

    def blob(pis):
        return np.concatenate([np.zeros(3, dtype=np.uint64), np.asarray(pis, dtype=np.uint64)])
    proofs = {}
    for name in ("pp1", "pp2"):
        _, pis = S.trace_pairing_precomp(*jobs[name][1])
        proofs[name] = (S.AIR_PAIRING_PRECOMP, blob(pis), None)
    for name in ("ml1", "ml2"):
        _, pis = S.trace_miller_loop(*jobs[name][1])
        proofs[name] = (S.AIR_MILLER_LOOP, blob(pis), None)
        assert np.array_equal(pis[-144:], natives[name])
    _, pis = S.trace_fp12_mul(*jobs["fp12_mul"][1])
    proofs["fp12_mul"] = (S.AIR_FP12_MUL, blob(pis), None)
    assert np.array_equal(pis[288:], natives["product"])
    fe_pis = np.concatenate([natives["product"], natives["final"]]).astype(np.uint64)  # src/aggregate_proof.rs:160-165
    assert fe_pis.size == S.air_public_inputs(S.AIR_FINAL_EXP)
    proofs["final_exp"] = (S.AIR_FINAL_EXP, blob(fe_pis), None)
    assert A.check_links(proofs)
    # a different Miller-loop output breaks the chain
    bad = dict(proofs)
    t = proofs["ml2"][1].copy()
    t[-1] ^= 1
    bad["ml2"] = (S.AIR_MILLER_LOOP, t, None)
    assert not A.check_links(bad)
