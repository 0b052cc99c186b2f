#!/usr/bin/env python3
"""Extract the known-answer DATA the reference's own tests hold for the native BLS12-381 tower into
tests/golden/native_vectors.json.  Only decimal test vectors are taken (src/native.rs:1480-1498 the
verify_bls_signatures points, :1546-1557 the test_final_exponentiate input),
never source text.  Run in the build container (needs /root/reference)."""
import json
import re
import sys

src = open("/root/reference/src/native.rs").read()


def grab(name):
    m = re.search(r"let %s = BigUint::from_str\(\"(\d+)\"\)" % name, src)
    return m.group(1)


vec = {
    "source": "Electron-Labs/starky_bls12_381 src/native.rs:1477-1563 (test data only)",
    "bls_signature": {k: grab(k) for k in ["pk_x", "pk_y", "hm_x1", "hm_x2", "hm_y1", "hm_y2", "hm_z1", "hm_z2", "gx", "gy", "s_x1", "s_x2",
                                           "s_y1", "s_y2", "s_z1", "s_z2"]},
    "bls_signature_expect": "final_exponentiate(miller(-pk, Hm) * miller(g1, sig)) == Fp12::one()",
}
m = re.search(r"let aa = \[(.*?)\];", src, re.S)
vec["final_exp_input_aa"] = re.findall(r"\"(\d+)\"", m.group(1))
assert len(vec["final_exp_input_aa"]) == 12
vec["final_exp_expect"] = "Fp12::one()"
json.dump(vec, open("tests/golden/native_vectors.json", "w"), indent=1)
print("ok")
