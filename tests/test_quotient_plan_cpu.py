"""The tiled constraint plan the quotient kernel executes (csrc/quotient_plan.h), replayed on the CPU.

`starkhip_quotient_plan_check` builds the plan (supergroups, per-proof weights from the term contributions, column tiles,
pieces dealt to the eight waves of a workgroup, record streams), replays every stream on one random frame with the kernel's
own arithmetic (three 22-bit weight limbs, six unreduced 64-bit sums per alpha) and compares with the plain fold
acc = acc * alpha + mask * c_k over all constraints — the ConstraintConsumer semantics of the reference
(/root/reference/src/final_exponentiate.rs:907-1136 and the other eval_packed_generic bodies)."""
import pytest

import starky_bls12_381_amd as S

AIRS = [S.AIR_TEST_FIBONACCI, S.AIR_ECC_AGGREGATE, S.AIR_FP12_MUL, S.AIR_PAIRING_PRECOMP, S.AIR_MILLER_LOOP, S.AIR_FINAL_EXP]


@pytest.mark.parametrize("air", AIRS)
@pytest.mark.parametrize("chunks", [1, 7, 128])
def test_plan_replay_equals_plain_fold(air, chunks):
    st = S.quotient_plan_check(air, chunks, seed=0xC0FFEE + 31 * air + chunks)
    assert 1 <= st["chunks"] <= max(1, min(chunks, st["tiles"]))
    assert st["pieces"] >= st["supergroups"] - 1 or st["supergroups"] <= 1
    assert st["lds_cell_records"] + st["direct_loads"] > 0


def test_final_exp_plan_reads_each_cell_about_once():
    """What the design is for: per point the plan touches ~0.73 M staged cells and < 0.1 M direct loads where the
    interpreter issued 1.17 M loads (one per op)."""
    st = S.quotient_plan_check(S.AIR_FINAL_EXP, 8)
    # 15 982 distinct (kind, gates) in the program; 6 793 tiny ones (a selector times a linear form of six monomials, three gates times
    # "cell + constant") give one gate to their monomials and join 200 others (quotient_plan.h pass 1b): 9 389 with terms of their own
    assert st["supergroups"] == 9389
    assert st["contributions"] == 1101555          # every non-zero term of the 360 800 constraints is accounted for
    assert st["direct_loads"] < 100_000
    assert st["tiles"] == (S.air_columns(S.AIR_FINAL_EXP) + 63) // 64


def test_proof_layout_offsets():
    import numpy as np
    import oracle_lib as O
    air = S.AIR_TEST_FIBONACCI
    cfg = S.StarkConfig.standard_fast_config()
    t, pis = S.trace_fibonacci(3, 5, 64)
    proof = O.prove(S.air_program(air), cfg, S.trace_rows_to_poly_values(t), pis)
    L = S.proof_layout(proof)
    assert L.total_words == proof.size and L.n_columns == S.air_columns(air) and L.n_public_inputs == pis.size
    assert np.array_equal(proof[L.off_public_inputs:L.off_public_inputs + L.n_public_inputs], pis)
    assert L.off_query_rounds + L.n_query_rounds * L.query_round_words == L.off_final_poly
    assert L.off_pow_witness == L.off_final_poly + 2 * L.final_poly_len
    with pytest.raises(S.StarkhipError):
        S.proof_layout(proof[:-1])
