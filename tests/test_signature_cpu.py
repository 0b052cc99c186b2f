"""Host side of the end-to-end signature driver (starky_bls12_381_amd/signature.py) without a GPU: synthetic valid
signatures, the batch plan, and the generate -> prove pipeline with a stand-in prover (real compact trace generators)."""
import numpy as np

import starky_bls12_381_amd as S
from starky_bls12_381_amd import aggregate as A
from starky_bls12_381_amd import signature as G
from bls_util import native_vectors


def _vector():
    return native_vectors()["bls_signature"]


def test_synthetic_signatures_are_valid_and_different():
    sigs = G.synthetic_signatures(4, _vector(), seed=5)
    seen = set()
    for pk, hm, sig in sigs:
        _, natives = A.signature_jobs(pk, hm, sig)
        assert A.signature_is_valid(natives)   # e(pk, H) * e(-G, sig) == 1 with the product's natives
        seen.add(bytes(pk[0]) + bytes(sig[0]))
    assert len(seen) == 4
    # a signature under another key does not check
    _, natives = A.signature_jobs(sigs[0][0], sigs[1][1], sigs[1][2])
    assert not A.signature_is_valid(natives)
    words = G.pack_operands(sigs)
    assert words.shape == (4, G.OPERAND_WORDS)
    back = G.unpack_operands(words)
    assert all(np.array_equal(a, b) for x, y in zip(sigs, back) for p, q in zip(x, y) for a, b in zip(p, q))


def test_batch_plan_for_eight_signatures_on_eight_ranks():
    plan = G.plan_batch(8, 8)
    flat = sorted(j for r in plan for j in r)
    assert flat == sorted((i, n) for i in range(8) for n in A.JOB_ORDER)
    assert all(sum(1 for _, n in r if n == "final_exp") == 1 for r in plan)   # one FinalExp per GPU (BASELINE configs[4])


def _fake_prove(pv, air, cfg, trace, pis):
    assert isinstance(trace, S.CompactTrace) and trace.shape[1] == S.air_columns(air)
    return np.concatenate([np.zeros(3, dtype=np.uint64), np.asarray(pis, dtype=np.uint64)])


def test_pipeline_generates_on_threads_and_links_hold_on_what_was_proven():
    sigs = G.synthetic_signatures(1, _vector(), seed=9)
    mine = G.plan_batch(1, 1)[0]
    args, natives = G.job_arguments(sigs, mine)
    results, stats = G.run_jobs([object(), object(), object()], mine, args, gen_threads=6, prove=_fake_prove)
    assert sorted(results) == sorted(mine)
    six = G.signature_proofs(results, 0)
    assert A.check_links(six)
    assert A.check_statement(six, sigs[0][1], sigs[0][2])
    assert A.signature_is_valid(natives[0], six)
    assert stats["generate_s"] > stats["wall_s"] * 0.5 or stats["wall_s"] < 2.0   # generation ran on several threads
    # the statement check is about the POINTS: proofs of another signature verify and link, but are not this statement
    other = G.synthetic_signatures(2, _vector(), seed=9)[1]
    assert not A.check_statement(six, other[1], other[2])
    # ... and a final_exp proof that attests to something else than 1 is not a valid signature
    bad = dict(six)
    air, blob, cfg = bad["final_exp"]
    blob = blob.copy()
    blob[-1] ^= np.uint64(1)
    bad["final_exp"] = (air, blob, cfg)
    assert not A.check_statement(bad, sigs[0][1], sigs[0][2])


def test_pipeline_reports_a_failing_job():
    import pytest

    def boom(pv, air, cfg, trace, pis):
        raise RuntimeError("device lost")
    sigs = G.synthetic_signatures(1, _vector(), seed=9)
    mine = [(0, "fp12_mul")]
    args, _ = G.job_arguments(sigs, mine)
    with pytest.raises(RuntimeError):
        G.run_jobs([object()], mine, args, gen_threads=1, prove=boom)
