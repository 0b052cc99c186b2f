/* TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU oracle for the starky prove() hot path: Goldilocks field, quadratic extension.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * PARITY UNPINNED at the prove() boundary: the reference keeps the prover in the
 * un-vendored crates starky 0.1.2 / plonky2 0.1.4 / plonky2_field 0.1.1
 * (Electron-Labs/plonky2 @ 666f31517353b29b3d847c6e18b26c9be8bf060b, see
 * /root/reference/Cargo.lock:1424-1427,1501-1504,2043-2046) and holds no golden proof
 * for any of the four AIRs.  This file restates the published algorithm
 * (SURVEY.md App. A.1).  What IS pinned: Poseidon permutation KATs and the native
 * BLS12-381 vectors of /root/reference/src/native.rs:1477-1563 (tests/golden/).
 *
 * Deliberately written independently of starky_bls12_381_amd/csrc/gl.h: the
 * product reduces with the 2^96 = -1 split; this file folds twice with
 * 2^64 = 2^32 - 1 and finishes with compares (and keeps a `%` version, f_mul_slow,
 * that tests cross-check), so a bug in either reduction shows up as a mismatch.
 */
#ifndef ORACLE_FIELD_H
#define ORACLE_FIELD_H
#include <stdint.h>

typedef uint64_t fe;
typedef unsigned __int128 u128;
#define OR_P 0xFFFFFFFF00000001ULL

static inline fe f_add(fe a, fe b) { u128 s = (u128)a + b; if (s >= OR_P) s -= OR_P; return (fe)s; }
static inline fe f_sub(fe a, fe b) { return a >= b ? a - b : (fe)((u128)a + OR_P - b); }
static inline fe f_mul_slow(fe a, fe b) { return (fe)(((u128)a * b) % OR_P); }
/* reduce a value < 2^128: fold the high word twice using 2^64 = 2^32 - 1 (mod p) */
static inline fe f_red128(u128 x) {
    u128 y = (u128)(uint64_t)(x >> 64) * 0xFFFFFFFFULL + (uint64_t)x; /* < 2^97 */
    u128 z = (u128)(uint64_t)(y >> 64) * 0xFFFFFFFFULL + (uint64_t)y; /* < 2^65 */
    if (z >= ((u128)OR_P << 1)) z -= ((u128)OR_P << 1);
    if (z >= OR_P) z -= OR_P;
    return (fe)z;
}
static inline fe f_mul(fe a, fe b) { return f_red128((u128)a * b); }
static inline fe f_neg(fe a) { return a ? OR_P - a : 0; }
static inline fe f_pow(fe b, uint64_t e) {
    fe r = 1;
    while (e) {
        if (e & 1) r = f_mul(r, b);
        b = f_mul(b, b);
        e >>= 1;
    }
    return r;
}
static inline fe f_inv(fe a) { return f_pow(a, OR_P - 2); }
/* primitive 2^k-th root of unity: POWER_OF_TWO_GENERATOR^(2^(32-k)) */
static inline fe f_root(unsigned k) {
    fe r = 1753635133440165772ULL;
    for (unsigned i = k; i < 32; i++) r = f_mul(r, r);
    return r;
}

/* F[X]/(X^2-7) */
typedef struct { fe a0, a1; } fe2;
static inline fe2 e_make(fe a0, fe a1) { fe2 r = {a0, a1}; return r; }
static inline fe2 e_base(fe a) { return e_make(a, 0); }
static inline fe2 e_add(fe2 a, fe2 b) { return e_make(f_add(a.a0, b.a0), f_add(a.a1, b.a1)); }
static inline fe2 e_sub(fe2 a, fe2 b) { return e_make(f_sub(a.a0, b.a0), f_sub(a.a1, b.a1)); }
static inline fe2 e_mul(fe2 a, fe2 b) {
    return e_make(f_add(f_mul(a.a0, b.a0), f_mul(7, f_mul(a.a1, b.a1))),
                  f_add(f_mul(a.a0, b.a1), f_mul(a.a1, b.a0)));
}
static inline fe2 e_mulb(fe2 a, fe b) { return e_make(f_mul(a.a0, b), f_mul(a.a1, b)); }
static inline fe2 e_inv(fe2 a) {
    fe n = f_sub(f_mul(a.a0, a.a0), f_mul(7, f_mul(a.a1, a.a1)));
    fe ni = f_inv(n);
    return e_make(f_mul(a.a0, ni), f_mul(f_neg(a.a1), ni));
}
static inline fe2 e_pow(fe2 b, uint64_t e) {
    fe2 r = e_make(1, 0);
    while (e) {
        if (e & 1) r = e_mul(r, b);
        b = e_mul(b, b);
        e >>= 1;
    }
    return r;
}
static inline int e_eq(fe2 a, fe2 b) { return a.a0 == b.a0 && a.a1 == b.a1; }

static inline uint32_t bitrev(uint32_t x, unsigned bits) {
    uint32_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}
#endif
