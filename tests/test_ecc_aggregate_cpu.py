"""ECCAggStark (src/ecc_aggregate.rs + src/g1.rs), CPU side: shape, the reference's own aggregation vector, every
constraint vanishing on the generated trace, oracle prove -> product verify on a reduced number of rows is not
possible (the AIR is fixed at 512 points), so the full-size proof parity lives in the GPU tests."""
import json
import os

import numpy as np

import oracle_lib as O
import starky_bls12_381_amd as S
from bls_util import BLS_P, GOLDEN, limbs

N = S.ECC_NUM_POINTS


def reference_vector():
    v = json.load(open(os.path.join(GOLDEN, "ecc_aggregate_vector.json")))
    pts = [(int(x), int(y)) for x, y in v["points"]]
    return pts, [bool(b) for b in v["bits"]], (int(v["res"][0]), int(v["res"][1]))


def pack(points, bits):
    """Pad to 512 operands: the tail repeats the last point with its bit cleared (skipped by the AIR, but each row block
    still carries a well-formed addition of it to the running sum)."""
    pts = list(points) + [points[-1]] * (N - len(points))
    b = list(bits) + [False] * (N - len(bits))
    arr = np.array([limbs(x) + limbs(y) for x, y in pts], dtype=np.uint32)
    return arr, np.array(b, dtype=bool)


def affine_add(a, b):
    lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, BLS_P) % BLS_P
    x3 = (lam * lam - a[0] - b[0]) % BLS_P
    return x3, (lam * (a[0] - x3) - a[1]) % BLS_P


def test_shape():
    assert (S.air_columns(S.AIR_ECC_AGGREGATE), S.air_public_inputs(S.AIR_ECC_AGGREGATE), S.air_constraint_degree(S.AIR_ECC_AGGREGATE),
            S.air_default_rows(S.AIR_ECC_AGGREGATE)) == (3339, 24 * N + N + 24, 4, 8192)  # README.md:40, src/ecc_aggregate.rs:17-21
    cfg = S.StarkConfig.for_air(S.AIR_ECC_AGGREGATE)
    assert cfg.rate_bits == 2  # src/aggregate_proof.rs:186-187


def test_native_aggregate_matches_the_reference_vector():
    pts, bits, res = reference_vector()
    arr, b = pack(pts, bits)
    out = S.native_g1_aggregate(arr, b)
    assert [int(x) for x in out] == limbs(res[0]) + limbs(res[1])
    # and plain affine arithmetic agrees with the reference's expected value
    acc = pts[0]
    for p in pts[1:4]:
        acc = affine_add(acc, p)
    assert acc == res


def test_trace_satisfies_every_constraint_and_publishes_the_aggregate():
    pts, bits, res = reference_vector()
    # a second case: first key absent, a few later ones absent
    rng = np.random.default_rng(7)
    more = pts[:]
    acc = pts[0]
    for _ in range(20):
        acc = affine_add(acc, pts[int(rng.integers(1, 5))] if acc[0] != pts[1][0] else pts[2])
        more.append(acc)
    bits2 = [False, True] + [bool(x) for x in rng.integers(0, 2, size=len(more) - 2)]
    for points, bb in ((pts, bits), (more, bits2)):
        arr, b = pack(points, bb)
        t, pis = S.trace_ecc_aggregate(arr, b)
        assert t.shape == (8192, 3339)
        want = S.native_g1_aggregate(arr, b)
        assert np.array_equal(pis[-24:], want)
        assert np.array_equal(pis[:24 * N], arr.reshape(-1)) and np.array_equal(pis[24 * N:25 * N], b.astype(np.uint64))
        blob = S.air_program(S.AIR_ECC_AGGREGATE)
        assert O.check_trace(blob, t, pis)[0] == 0
        bad = pis.copy()
        bad[-1] = (int(bad[-1]) + 1) % S.P
        assert O.check_trace(blob, t, bad)[0] > 0
        badt = t.copy()
        badt[30, 526 + 48] = (int(badt[30, 526 + 48]) + 1) % S.P  # x3 of the third addition
        assert O.check_trace(blob, badt, pis)[0] > 0
    if points is pts:
        assert [int(x) for x in pis[-24:]] == limbs(res[0]) + limbs(res[1])
