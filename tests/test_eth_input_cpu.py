"""Real-input driver (starky_bls12_381_amd/eth_input.py): the reference's own known answers for the isogeny and
hash-to-curve (src/hash_to_curve.rs:465-553), and the mainnet light-client update the reference's main.rs proves
(src/light_client_update_period_105{2,3}.json, kept as data fixtures): 512 keys -> aggregate -> pairing check."""
import os

import numpy as np

import starky_bls12_381_amd as S
from starky_bls12_381_amd import aggregate as A
from starky_bls12_381_amd import eth_input as E
from bls_util import GOLDEN


def test_isogeny_map_known_answer():  # src/hash_to_curve.rs:492-553
    a = ((3768960129599410557225162537737286003238400530051754572454824471200864202913026112975152396185116175737023068710834,
          2843653242501816279232983717246998149289638605923450990196321568072224346134709601553669097144892265594669670100681),
         (2136473314670056131183153764113091685196675640973971063848296586048702180604877062503412214120535118046733529576506,
          3717743359948639609414970569174500186381762539811697438986507840606082550875593852503699874848297189142874182531754))
    out = ((3219922746671482828210036408711997441423671614254909325234707044434520756052360285257107968950769890523504628275940,
            1689252599334450651431125834598273362703914442067213087777626885820814565104897473205802289043260096634945919754747),
           (3277365552217223927730141275188890184833071787772555827000840921808443941258778716588573376888715070179970391655322,
            583921403203359937897773959554466412643567032578544897698779952656397892876222999644067619700087458377600564507453))
    assert E.isogeny_map(a) == out


def test_hash_to_curve_known_answer():  # src/hash_to_curve.rs:465-490: empty message
    want = ((2484880953070652509895159898261749949971419256101265549903463729658081179969788208734336814677878439015289354663558,
             571286950361770968319560191831515067050084989489837870994029396792668285219017899793859671802388182901315402858724),
            (3945400848309661287520855376438021610375515007889273149322439985738679863089347725379973912108534346949384256127526,
             1067268791373784971379690868996146496995005458163356395218843329703930727067637736115073576974603814754170298346268))
    got = E.hash_to_curve_g2(b"")
    assert got == want
    x, y = got
    assert E.f2_sq(y) == E.f2_add(E.f2_mul(E.f2_sq(x), x), (4, 4))  # on E: y^2 = x^3 + 4(1 + i)


def test_mainnet_update_aggregates_and_verifies():
    upd = E.load_update(os.path.join(GOLDEN, "light_client_update_period_1053.json"), os.path.join(GOLDEN, "light_client_update_period_1052.json"))
    assert upd["points"].shape == (512, 24) and upd["bits"].size == 512 and upd["bits"].sum() > 340  # > 2/3 participation
    # the aggregate public key: product natives (the ECCAgg witness side) vs plain affine sums of the participating keys
    pk = S.native_g1_aggregate(upd["points"], upd["bits"])
    acc = None
    for (x, y), b in zip(upd["keys"], upd["bits"]):
        if not b:
            continue
        if acc is None:
            acc = (x, y)
        else:
            lam = (y - acc[1]) * pow(x - acc[0], -1, E.P) % E.P
            x3 = (lam * lam - acc[0] - x) % E.P
            acc = (x3, (lam * (acc[0] - x3) - acc[1]) % E.P)
    assert [int(v) for v in pk] == E._limbs(acc[0]) + E._limbs(acc[1])
    # e(apk, H(signing_root)) * e(-G1, signature) == 1 with the product's own Miller loop and final exponentiation
    _, natives = A.signature_jobs((pk[:12], pk[12:]), upd["hm"], upd["sig"])
    assert A.signature_is_valid(natives)
    # a different message does not verify
    other = E.hash_to_curve_g2(b"\x01" + upd["signing_root"][1:])
    fp2 = lambda a: np.array(E._limbs(a[0]) + E._limbs(a[1]), dtype=np.uint32)  # noqa: E731
    _, bad = A.signature_jobs((pk[:12], pk[12:]), (fp2(other[0]), fp2(other[1]), upd["hm"][2]), upd["sig"])
    assert not A.signature_is_valid(bad)
