"""Proof hand-off (starky_bls12_381_amd/handoff.py): blob <-> the nested StarkProofWithPublicInputs value, on an oracle
proof of the toy AIR (no GPU needed); the round-tripped blob is still accepted by the product verifier."""
import json

import numpy as np

import oracle_lib as O
import starky_bls12_381_amd as S
from starky_bls12_381_amd import handoff as H


def _proof(log_n=6):
    cfg = S.StarkConfig.standard_fast_config()
    t, pis = S.trace_fibonacci(3, 5, 1 << log_n)
    return cfg, O.prove(S.air_program(S.AIR_TEST_FIBONACCI), cfg, S.trace_rows_to_poly_values(t), pis), pis


def test_shape_follows_the_stark_proof_structs():
    cfg, proof, pis = _proof()
    v = H.proof_to_value(proof)
    assert list(v) == ["proof", "public_inputs"] and v["public_inputs"] == [int(x) for x in pis]
    p = v["proof"]
    assert list(p) == ["trace_cap", "permutation_zs_cap", "quotient_polys_cap", "openings", "opening_proof"]
    assert p["permutation_zs_cap"] is None and p["openings"]["permutation_zs"] is None
    assert len(p["trace_cap"]) == 1 << cfg.cap_height and all(len(h["elements"]) == 4 for h in p["trace_cap"])
    fri = p["opening_proof"]
    assert len(fri["query_round_proofs"]) == cfg.num_query_rounds
    r0 = fri["query_round_proofs"][0]
    assert len(r0["initial_trees_proof"]["evals_proofs"]) == 2  # trace oracle, quotient oracle
    leaf, mp = r0["initial_trees_proof"]["evals_proofs"][0]
    assert len(leaf) == S.air_columns(S.AIR_TEST_FIBONACCI) and len(mp["siblings"]) == 6 + cfg.rate_bits - cfg.cap_height
    assert len(r0["steps"]) == len(fri["commit_phase_merkle_caps"])
    assert all(len(st["evals"]) == 1 << cfg.arity_bits for st in r0["steps"])
    assert all(0 <= x < S.P for x in v["public_inputs"])


def test_json_round_trip_is_lossless_and_still_verifies():
    cfg, proof, _ = _proof(7)
    text = H.dumps(proof)
    assert json.loads(text)["proof"]["opening_proof"]["pow_witness"] == int(proof[-1 - len(H.proof_to_value(proof)["public_inputs"])])
    back = H.loads(text, degree_bits=7, rate_bits=cfg.rate_bits, arity_bits=cfg.arity_bits, num_challenges=cfg.num_challenges)
    assert back.dtype == np.uint64 and np.array_equal(back, proof)
    S.verify_stark_proof(S.AIR_TEST_FIBONACCI, cfg, back)
