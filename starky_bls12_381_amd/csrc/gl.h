// Goldilocks field p = 2^64 - 2^32 + 1 and its quadratic extension F[X]/(X^2 - 7),
// shared by host C++ and gfx950 device code.
//
// Semantics follow plonky2_field 0.1.1 GoldilocksField / QuadraticExtension as used
// by the reference at /root/reference/src/aggregate_proof.rs:235-237 (F, D = 2);
// SURVEY.md App. A.1.  All values stored by this library are canonical (< p).
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GL_HD __host__ __device__ __forceinline__
#else
#define GL_HD inline
#endif

typedef uint64_t gl_t;

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL  // 2^64 mod p = 2^32 - 1

// multiplicative generator 7; 2^32-th root of unity 7^((p-1)/2^32)
#define GL_GENERATOR 7ULL
#define GL_POWER_OF_TWO_GENERATOR 1753635133440165772ULL
#define GL_TWO_ADICITY 32

GL_HD gl_t gl_add(gl_t a, gl_t b) {
    // a, b < p  =>  a + b < 2p < 2^65
    gl_t s = a + b;
    // if carry: true sum = s + 2^64 = s + eps (mod p), and s < p - 1 so no second wrap.
    // Written with masks: on the host these are data-dependent and a branch would mispredict half the time.
    s += (gl_t)(0 - (gl_t)(s < a)) & GL_EPS;
    s -= (gl_t)(0 - (gl_t)(s >= GL_P)) & GL_P;
    return s;
}

GL_HD gl_t gl_sub(gl_t a, gl_t b) {
    gl_t d = a - b;
    d += (gl_t)(0 - (gl_t)(a < b)) & GL_P;  // wraps mod 2^64 to the right residue
    return d;
}

GL_HD gl_t gl_neg(gl_t a) { return a ? GL_P - a : 0; }

GL_HD gl_t gl_double(gl_t a) { return gl_add(a, a); }

// reduce a 128-bit value hi:lo to canonical form
GL_HD gl_t gl_reduce128(uint64_t hi, uint64_t lo) {
    uint64_t hi_hi = hi >> 32;
    uint64_t hi_lo = hi & GL_EPS;
    // 2^96 = -1, 2^64 = eps (mod p)
    uint64_t t0 = lo - hi_hi;
    t0 -= (uint64_t)(0 - (uint64_t)(lo < hi_hi)) & GL_EPS;  // borrow: add p == subtract eps mod 2^64
    uint64_t t1 = (hi_lo << 32) - hi_lo;                    // hi_lo * eps, < 2^64
    uint64_t r = t0 + t1;
    r += (uint64_t)(0 - (uint64_t)(r < t1)) & GL_EPS;       // carry: 2^64 = eps
    r -= (uint64_t)(0 - (uint64_t)(r >= GL_P)) & GL_P;
    return r;
}

GL_HD gl_t gl_mul(gl_t a, gl_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t lo = a * b;
    uint64_t hi = __umul64hi(a, b);
#else
    unsigned __int128 m = (unsigned __int128)a * b;
    uint64_t lo = (uint64_t)m, hi = (uint64_t)(m >> 64);
#endif
    return gl_reduce128(hi, lo);
}

GL_HD gl_t gl_sqr(gl_t a) { return gl_mul(a, a); }

// a*b + c
GL_HD gl_t gl_mad(gl_t a, gl_t b, gl_t c) { return gl_add(gl_mul(a, b), c); }

GL_HD gl_t gl_pow(gl_t b, uint64_t e) {
    gl_t r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, b);
        b = gl_sqr(b);
        e >>= 1;
    }
    return r;
}

GL_HD gl_t gl_inv(gl_t a) { return gl_pow(a, GL_P - 2); }

// primitive 2^k-th root of unity
GL_HD gl_t gl_root_of_unity(unsigned k) {
    gl_t r = GL_POWER_OF_TWO_GENERATOR;
    for (unsigned i = k; i < GL_TWO_ADICITY; i++) r = gl_sqr(r);
    return r;
}

GL_HD gl_t gl_from_u64(uint64_t x) { return x >= GL_P ? x - GL_P : x; }

// ---------------------------------------------------------------- extension
struct gl2_t {
    gl_t a0, a1;
};

#define GL2_W 7ULL  // X^2 = 7

GL_HD gl2_t gl2_make(gl_t a0, gl_t a1) {
    gl2_t r;
    r.a0 = a0;
    r.a1 = a1;
    return r;
}
GL_HD gl2_t gl2_from_base(gl_t a) { return gl2_make(a, 0); }
GL_HD gl2_t gl2_zero() { return gl2_make(0, 0); }
GL_HD gl2_t gl2_one() { return gl2_make(1, 0); }
GL_HD bool gl2_eq(gl2_t a, gl2_t b) { return a.a0 == b.a0 && a.a1 == b.a1; }
GL_HD gl2_t gl2_add(gl2_t a, gl2_t b) { return gl2_make(gl_add(a.a0, b.a0), gl_add(a.a1, b.a1)); }
GL_HD gl2_t gl2_sub(gl2_t a, gl2_t b) { return gl2_make(gl_sub(a.a0, b.a0), gl_sub(a.a1, b.a1)); }
GL_HD gl2_t gl2_neg(gl2_t a) { return gl2_make(gl_neg(a.a0), gl_neg(a.a1)); }
GL_HD gl2_t gl2_mul(gl2_t a, gl2_t b) {
    gl_t c0 = gl_add(gl_mul(a.a0, b.a0), gl_mul(GL2_W, gl_mul(a.a1, b.a1)));
    gl_t c1 = gl_add(gl_mul(a.a0, b.a1), gl_mul(a.a1, b.a0));
    return gl2_make(c0, c1);
}
GL_HD gl2_t gl2_mul_base(gl2_t a, gl_t b) { return gl2_make(gl_mul(a.a0, b), gl_mul(a.a1, b)); }
GL_HD gl2_t gl2_sqr(gl2_t a) { return gl2_mul(a, a); }
GL_HD gl2_t gl2_inv(gl2_t a) {
    // (a0 + a1 X)^-1 = (a0 - a1 X) / (a0^2 - 7 a1^2)
    gl_t norm = gl_sub(gl_sqr(a.a0), gl_mul(GL2_W, gl_sqr(a.a1)));
    gl_t ni = gl_inv(norm);
    return gl2_make(gl_mul(a.a0, ni), gl_mul(gl_neg(a.a1), ni));
}
GL_HD gl2_t gl2_pow(gl2_t b, uint64_t e) {
    gl2_t r = gl2_one();
    while (e) {
        if (e & 1) r = gl2_mul(r, b);
        b = gl2_sqr(b);
        e >>= 1;
    }
    return r;
}

GL_HD uint32_t gl_bitrev(uint32_t x, unsigned bits) {
    if (bits == 0) return 0;
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(x) >> (32 - bits);
#else
    uint32_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
#endif
}
