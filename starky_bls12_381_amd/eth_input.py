"""Real-input driver: from an Ethereum light-client update to the operands of the five STARKs.

Mirrors the host-side preparation in /root/reference/src/main.rs:9-53 and src/aggregate_proof.rs:246-348, with the
milagro / eth_types calls replaced by plain integer arithmetic:
  * SSZ hash_tree_root of the attested BeaconBlockHeader and of SigningData{object_root, domain}   (main.rs:29-39)
  * participation bits, least-significant bit of each byte first                                  (aggregate_proof.rs:259-264)
  * G1 / G2 point decompression (ZCash serialisation of BLS12-381)                                (:248-256, :339-348)
  * hash_to_curve_g2 with DST "BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_" (RFC 9380 suite)     (:289-302)
The 3-isogeny coefficients are the table of src/hash_to_curve.rs:9-83; known answers: src/hash_to_curve.rs:465-553.

Host-only and not on the proving hot path: it produces (points, bits) for ECCAggStark and (pk, H(m), signature) for the
six pairing proofs (aggregate.py).  Field elements are Python ints; Fp2 = (c0, c1) = c0 + c1 * i with i^2 = -1.
"""
import hashlib
import json

import numpy as np

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
BLS_X = 0xD201000000010000  # |x| of the curve parameter x = -0xd201000000010000
DST = b"BLS_SIG_BLS12381G2_XMD:SHA-256_SSWU_RO_POP_"  # src/aggregate_proof.rs:234
MAINNET_DOMAIN = bytes.fromhex("070000006a95a1a967855d676d48be69883b712607f952d5198d0f5677564636")  # src/main.rs:26


# ------------------------------------------------------------------------------------------- Fp2
def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_neg(a):
    return (-a[0] % P, -a[1] % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_sq(a):
    return f2_mul(a, a)


def f2_scalar(a, k):
    return (a[0] * k % P, a[1] * k % P)


def f2_inv(a):
    n = pow((a[0] * a[0] + a[1] * a[1]) % P, -1, P)
    return (a[0] * n % P, -a[1] * n % P)


def f2_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_sq(a)
        e >>= 1
    return r


def f2_conj(a):  # Frobenius
    return (a[0], -a[1] % P)


def f2_is_square(a):
    n = (a[0] * a[0] + a[1] * a[1]) % P  # the norm is a square in Fp iff a is a square in Fp2
    return n == 0 or pow(n, (P - 1) // 2, P) == 1


def f2_sqrt(a):
    """A square root in Fp2 (p = 3 mod 4), or None."""
    if a == (0, 0):
        return (0, 0)
    a1 = f2_pow(a, (P - 3) // 4)
    alpha = f2_mul(f2_sq(a1), a)
    x0 = f2_mul(a1, a)
    if alpha == (P - 1, 0):
        x = f2_mul((0, 1), x0)
    else:
        b = f2_pow(f2_add((1, 0), alpha), (P - 1) // 2)
        x = f2_mul(b, x0)
    return x if f2_sq(x) == a else None


def f2_sgn0(a):  # RFC 9380 section 4.1, m = 2
    return (a[0] & 1) | ((a[0] == 0) & (a[1] & 1))


# ------------------------------------------------------------------------------------------- curves over Fp2 (affine, None = infinity)
def ec2_add(p1, p2, a_coef):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    if p1[0] == p2[0]:
        if f2_add(p1[1], p2[1]) == (0, 0):
            return None
        lam = f2_mul(f2_add(f2_scalar(f2_sq(p1[0]), 3), a_coef), f2_inv(f2_scalar(p1[1], 2)))
    else:
        lam = f2_mul(f2_sub(p2[1], p1[1]), f2_inv(f2_sub(p2[0], p1[0])))
    x3 = f2_sub(f2_sub(f2_sq(lam), p1[0]), p2[0])
    return (x3, f2_sub(f2_mul(lam, f2_sub(p1[0], x3)), p1[1]))


def ec2_neg(p):
    return None if p is None else (p[0], f2_neg(p[1]))


def ec2_mul(p, k, a_coef=(0, 0)):
    r = None
    while k:
        if k & 1:
            r = ec2_add(r, p, a_coef)
        p = ec2_add(p, p, a_coef)
        k >>= 1
    return r


# ------------------------------------------------------------------------------------------- hash to curve (G2)
def expand_message_xmd(msg, dst, n):
    ell = (n + 31) // 32
    dst_prime = dst + bytes([len(dst)])
    b0 = hashlib.sha256(b"\x00" * 64 + msg + n.to_bytes(2, "big") + b"\x00" + dst_prime).digest()
    b = [hashlib.sha256(b0 + b"\x01" + dst_prime).digest()]
    for i in range(2, ell + 1):
        b.append(hashlib.sha256(bytes(x ^ y for x, y in zip(b0, b[-1])) + bytes([i]) + dst_prime).digest())
    return b"".join(b)[:n]


def hash_to_field_fp2(msg, count=2, dst=DST):
    u = expand_message_xmd(msg, dst, count * 2 * 64)
    return [tuple(int.from_bytes(u[64 * (j + 2 * i):64 * (j + 2 * i) + 64], "big") % P for j in range(2)) for i in range(count)]


SSWU_A = (0, 240)              # E': y^2 = x^3 + 240 i x + 1012 (1 + i)
SSWU_B = (1012, 1012)
SSWU_Z = (P - 2, P - 1)        # -(2 + i)

# src/hash_to_curve.rs:9-83: [x_num, x_den, y_num, y_den], each [k3, k2, k1, k0] as (c0, c1)
ISO3 = [
    [(3557697382419259905260257622876359250272784728834673675850718343221361467102966990615722337003569479144794908942033, 0),
     (2668273036814444928945193217157269437704588546626005256888038757416021100327225242961791752752677109358596181706526,
      1334136518407222464472596608578634718852294273313002628444019378708010550163612621480895876376338554679298090853261),
     (0, 2668273036814444928945193217157269437704588546626005256888038757416021100327225242961791752752677109358596181706522),
     (889424345604814976315064405719089812568196182208668418962679585805340366775741747653930584250892369786198727235542,
      889424345604814976315064405719089812568196182208668418962679585805340366775741747653930584250892369786198727235542)],
    [(0, 0), (1, 0),
     (12, 4002409555221667393417789825735904156556882819939007885332058136124031650490837864442687629129015664037894272559775),
     (0, 4002409555221667393417789825735904156556882819939007885332058136124031650490837864442687629129015664037894272559715)],
    [(2816510427748580758331037284777117739799287910327449993381818688383577828123182200904113516794492504322962636245776, 0),
     (2668273036814444928945193217157269437704588546626005256888038757416021100327225242961791752752677109358596181706524,
      1334136518407222464472596608578634718852294273313002628444019378708010550163612621480895876376338554679298090853263),
     (0, 889424345604814976315064405719089812568196182208668418962679585805340366775741747653930584250892369786198727235518),
     (3261222600550988246488569487636662646083386001431784202863158481286248011511053074731078808919938689216061999863558,
      3261222600550988246488569487636662646083386001431784202863158481286248011511053074731078808919938689216061999863558)],
    [(1, 0),
     (18, 4002409555221667393417789825735904156556882819939007885332058136124031650490837864442687629129015664037894272559769),
     (0, 4002409555221667393417789825735904156556882819939007885332058136124031650490837864442687629129015664037894272559571),
     (4002409555221667393417789825735904156556882819939007885332058136124031650490837864442687629129015664037894272559355,
      4002409555221667393417789825735904156556882819939007885332058136124031650490837864442687629129015664037894272559355)],
]


def map_to_curve_sswu(u):
    """Simplified SWU onto E' (RFC 9380 section 6.6.2)."""
    zu2 = f2_mul(SSWU_Z, f2_sq(u))
    tv = f2_add(f2_sq(zu2), zu2)
    if tv == (0, 0):
        x1 = f2_mul(SSWU_B, f2_inv(f2_mul(SSWU_Z, SSWU_A)))
    else:
        x1 = f2_mul(f2_mul(f2_neg(SSWU_B), f2_inv(SSWU_A)), f2_add((1, 0), f2_inv(tv)))

    def g(x):
        return f2_add(f2_add(f2_mul(f2_sq(x), x), f2_mul(SSWU_A, x)), SSWU_B)
    gx1 = g(x1)
    if f2_is_square(gx1):
        x, y = x1, f2_sqrt(gx1)
    else:
        x = f2_mul(zu2, x1)
        y = f2_sqrt(g(x))
    if f2_sgn0(u) != f2_sgn0(y):
        y = f2_neg(y)
    return (x, y)


def isogeny_map(pt):
    """3-isogeny E' -> E (src/hash_to_curve.rs:203-248)."""
    x, y = pt
    x2 = f2_sq(x)
    x3 = f2_mul(x2, x)

    def poly(k, monic_cubic=False, monic_square=False):
        r = f2_add(k[3], f2_mul(x, k[2]))
        r = f2_add(r, x2 if monic_square else f2_mul(x2, k[1]))
        if monic_cubic:
            r = f2_add(r, x3)
        elif not monic_square:
            r = f2_add(r, f2_mul(x3, k[0]))
        return r
    x_num, x_den = poly(ISO3[0]), poly(ISO3[1], monic_square=True)
    y_num, y_den = poly(ISO3[2]), poly(ISO3[3], monic_cubic=True)
    return (f2_mul(x_num, f2_inv(x_den)), f2_mul(y, f2_mul(y_num, f2_inv(y_den))))


_PSI_X = f2_inv(f2_pow((1, 1), (P - 1) // 3))
_PSI_Y = f2_inv(f2_pow((1, 1), (P - 1) // 2))
_PSI2_X = pow(pow(2, (P - 1) // 3, P), -1, P)


def psi(p):
    return None if p is None else (f2_mul(f2_conj(p[0]), _PSI_X), f2_mul(f2_conj(p[1]), _PSI_Y))


def psi2(p):
    return None if p is None else (f2_scalar(p[0], _PSI2_X), f2_neg(p[1]))


def clear_cofactor_g2(p):
    """RFC 9380 appendix G.3 (the curve parameter is negative: c1 * Q = -(|c1| * Q))."""
    def c1(q):
        return ec2_neg(ec2_mul(q, BLS_X))
    add = lambda a, b: ec2_add(a, b, (0, 0))  # noqa: E731
    t1 = c1(p)
    t2 = psi(p)
    t3 = psi2(add(p, p))
    t3 = add(t3, ec2_neg(t2))
    t2 = c1(add(t1, t2))
    t3 = add(t3, t2)
    t3 = add(t3, ec2_neg(t1))
    return add(t3, ec2_neg(p))


def hash_to_curve_g2(msg, dst=DST):
    u0, u1 = hash_to_field_fp2(msg, 2, dst)
    r = ec2_add(map_to_curve_sswu(u0), map_to_curve_sswu(u1), SSWU_A)
    return clear_cofactor_g2(isogeny_map(r))


# ------------------------------------------------------------------------------------------- point decompression
def _flags(b):
    return b[0] >> 7, (b[0] >> 6) & 1, (b[0] >> 5) & 1, bytes([b[0] & 0x1F]) + b[1:]


def decompress_g1(b):
    """48-byte compressed G1 point -> (x, y)."""
    assert len(b) == 48
    c, inf, sort, rest = _flags(b)
    assert c == 1 and inf == 0, "uncompressed or infinity"
    x = int.from_bytes(rest, "big")
    y = pow((x * x * x + 4) % P, (P + 1) // 4, P)
    assert y * y % P == (x * x * x + 4) % P, "not on the curve"
    if (y > (P - 1) // 2) != bool(sort):
        y = P - y
    return x, y


def decompress_g2(b):
    """96-byte compressed G2 point (x.c1 first) -> ((x0, x1), (y0, y1))."""
    assert len(b) == 96
    c, inf, sort, rest = _flags(b[:48])
    assert c == 1 and inf == 0, "uncompressed or infinity"
    x = (int.from_bytes(b[48:], "big"), int.from_bytes(rest, "big"))
    y = f2_sqrt(f2_add(f2_mul(f2_sq(x), x), (4, 4)))
    assert y is not None, "not on the curve"
    larger = y[1] > (P - 1) // 2 or (y[1] == 0 and y[0] > (P - 1) // 2)
    if larger != bool(sort):
        y = f2_neg(y)
    return x, y


# ------------------------------------------------------------------------------------------- SSZ
def _merkle(chunks):
    n = 1
    while n < len(chunks):
        n *= 2
    layer = list(chunks) + [b"\x00" * 32] * (n - len(chunks))
    while len(layer) > 1:
        layer = [hashlib.sha256(layer[i] + layer[i + 1]).digest() for i in range(0, len(layer), 2)]
    return layer[0]


def beacon_header_root(h):
    u64 = lambda v: int(v).to_bytes(8, "little") + b"\x00" * 24  # noqa: E731
    hx = lambda s: bytes.fromhex(s[2:] if s.startswith("0x") else s)  # noqa: E731
    return _merkle([u64(h["slot"]), u64(h["proposer_index"]), hx(h["parent_root"]), hx(h["state_root"]), hx(h["body_root"])])


def signing_root(header, domain=MAINNET_DOMAIN):
    return _merkle([beacon_header_root(header), domain])


# ------------------------------------------------------------------------------------------- the update
def _limbs(v):
    return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(12)]


def load_update(update_path, prev_update_path, domain=MAINNET_DOMAIN):
    """Operands of the STARKs from two consecutive light-client updates (src/main.rs):
    the committee keys come from the previous period's `next_sync_committee`, bits / signature / header from this one."""
    upd = json.load(open(update_path))["data"]
    prev = json.load(open(prev_update_path))["data"]
    hx = lambda s: bytes.fromhex(s[2:] if s.startswith("0x") else s)  # noqa: E731
    keys = [decompress_g1(hx(k)) for k in prev["next_sync_committee"]["pubkeys"]]
    bit_bytes = hx(upd["sync_aggregate"]["sync_committee_bits"])
    bits = [bool((byte >> j) & 1) for byte in bit_bytes for j in range(8)]
    sig = decompress_g2(hx(upd["sync_aggregate"]["sync_committee_signature"]))
    root = signing_root(upd["attested_header"]["beacon"], domain)
    hm = hash_to_curve_g2(root)
    one = (1, 0)
    fp2 = lambda a: np.array(_limbs(a[0]) + _limbs(a[1]), dtype=np.uint32)  # noqa: E731
    return {
        "points": np.array([_limbs(x) + _limbs(y) for x, y in keys], dtype=np.uint32),
        "bits": np.array(bits, dtype=bool),
        "signing_root": root,
        "hm": (fp2(hm[0]), fp2(hm[1]), fp2(one)),
        "sig": (fp2(sig[0]), fp2(sig[1]), fp2(one)),
        "keys": keys,
    }
