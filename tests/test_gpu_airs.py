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
