"""All six STARK proofs of one BLS signature check on one GPU (BASELINE.json configs[3], single-GPU form):
the reference's own test vector (src/native.rs:1480-1498), every proof accepted by the verifier, public inputs chained."""
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
