// gfx950 device-only Goldilocks arithmetic with lazy ("non-canonical") reduction.
//
// A value is any 64-bit representative of its class mod p = 2^64 - 2^32 + 1 (so p..2^64-1 are allowed
// aliases of 0..2^32-2).  Each routine states which operands may be arbitrary and which must be canonical;
// every step is exact mod p, so a kernel canonicalises once (gl_canon) where results leave it.
// The 64 x 64 multiply is written as exactly four v_mad_u64_u32 (quarter-rate on CDNA4, the dominant cost)
// and the reduction uses 2^64 = 2^32 - 1 and 2^96 = -1 (mod p) with two conditional corrections.
#pragma once
#include <hip/hip_runtime.h>

#include "gl.h"

namespace starkhip {

__device__ __forceinline__ gl_t gl_canon(gl_t x) { return x >= GL_P ? x - GL_P : x; }

// (hi * 2^64 + lo) mod p, any representative; hi, lo arbitrary
__device__ __forceinline__ gl_t gl_reduce128_nc(uint64_t hi, uint64_t lo) {
    const uint32_t hi_lo = (uint32_t)hi, hi_hi = (uint32_t)(hi >> 32);
    // carries are taken from the add / subtract themselves (__builtin_*_overflow), which costs fewer instructions than
    // comparing afterwards; the corrections are adds of a selected constant, not selects between two 64-bit candidates
    uint64_t t0, r;
    const bool borrow = __builtin_sub_overflow(lo, (uint64_t)hi_hi, &t0);
    t0 -= borrow ? GL_EPS : 0;
    const bool carry = __builtin_add_overflow(t0, (uint64_t)hi_lo * 0xFFFFFFFFu, &r);  // hi_lo * eps < p
    r += carry ? GL_EPS : 0;
    return r;
}

// a * b mod p, any representative in [0, 2^64); a, b arbitrary 64-bit
__device__ __forceinline__ gl_t gl_mul_nc(gl_t a, gl_t b) {
    // 64 x 64 -> 128 as exactly four v_mad_u64_u32 (32 x 32 + 64): each partial sum below fits 64 bits
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t p00 = (uint64_t)a0 * b0;
    const uint64_t p01 = (uint64_t)a0 * b1 + (p00 >> 32);
    const uint64_t p10 = (uint64_t)a1 * b0 + (uint32_t)p01;
    const uint64_t hi = (uint64_t)a1 * b1 + (p01 >> 32) + (p10 >> 32);
    const uint64_t lo = (p10 << 32) | (uint32_t)p00;
    return gl_reduce128_nc(hi, lo);
}

// a * b + c mod p, any representative; a, b, c arbitrary 64-bit (a*b + c < 2^128)
__device__ __forceinline__ gl_t gl_mad_nc(gl_t a, gl_t b, gl_t c) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t p00 = (uint64_t)a0 * b0;
    const uint64_t p01 = (uint64_t)a0 * b1 + (p00 >> 32);
    const uint64_t p10 = (uint64_t)a1 * b0 + (uint32_t)p01;
    uint64_t hi = (uint64_t)a1 * b1 + (p01 >> 32) + (p10 >> 32);  // <= 2^64 - 2
    const uint64_t lo0 = (p10 << 32) | (uint32_t)p00;
    uint64_t lo;
    hi += __builtin_add_overflow(lo0, c, &lo) ? 1 : 0;
    return gl_reduce128_nc(hi, lo);
}

// a arbitrary, b canonical (< p)
__device__ __forceinline__ gl_t gl_add_nc(gl_t a, gl_t b) {
    uint64_t s;
    const bool c = __builtin_add_overflow(a, b, &s);
    return s + (c ? GL_EPS : 0);  // wrapped: s <= p - 2, so + eps cannot wrap again
}

// a - b; a arbitrary, b canonical (< p)
__device__ __forceinline__ gl_t gl_sub_nc(gl_t a, gl_t b) {
    uint64_t d;
    const bool c = __builtin_sub_overflow(a, b, &d);
    return d - (c ? GL_EPS : 0);  // wrapped: d >= 2^64 - p + 1 > eps, so - eps cannot wrap again
}

// ---- both operands arbitrary representatives (two corrections: the second wrap needs both inputs >= p - 1)
__device__ __forceinline__ gl_t gl_add_nn(gl_t a, gl_t b) {
    uint64_t s, s2;
    const bool c1 = __builtin_add_overflow(a, b, &s);
    const bool c2 = __builtin_add_overflow(s, c1 ? GL_EPS : 0, &s2);  // second wrap only when both were >= p - 1
    return s2 + (c2 ? GL_EPS : 0);
}
__device__ __forceinline__ gl_t gl_sub_nn(gl_t a, gl_t b) {
    uint64_t d, d2;
    const bool b1 = __builtin_sub_overflow(a, b, &d);
    const bool b2 = __builtin_sub_overflow(d, b1 ? GL_EPS : 0, &d2);
    return d2 - (b2 ? GL_EPS : 0);
}

// x * 2^e mod p for a compile-time-foldable 0 <= e < 96; x arbitrary, result any representative.
// 2^64 = 2^32 - 1, 2^96 = -1, 2^128 = -2^32 (mod p).
__device__ __forceinline__ gl_t gl_mul_pow2_nn(gl_t x, int e) {
    if (e == 0) return x;
    const int a = e >> 5, b = e & 31;
    const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32);
    uint32_t y0, y1, y2;  // x << b as three words, y2 < 2^31
    if (b == 0) {
        y0 = x0; y1 = x1; y2 = 0;
    } else {
        y0 = x0 << b;
        y1 = (x1 << b) | (x0 >> (32 - b));
        y2 = x1 >> (32 - b);
    }
    if (a == 0) {  // (y1:y0) + y2 * eps
        const uint64_t lo = ((uint64_t)y1 << 32) | y0;
        const uint64_t t = ((uint64_t)y2 << 32) - y2;  // < p
        return gl_add_nc(lo, t);
    }
    if (a == 1) {  // y0 * 2^32 + y1 * eps - y2
        const uint64_t A = (uint64_t)y0 << 32;             // < p
        const uint64_t B = ((uint64_t)y1 << 32) - y1;      // < p
        return gl_sub_nc(gl_add_nc(A, B), (gl_t)y2);
    }
    // a == 2: y0 * eps - y1 - y2 * 2^32
    const uint64_t A = ((uint64_t)y0 << 32) - y0;          // < p
    const uint64_t B = ((uint64_t)y2 << 32) + y1;          // < 2^63 + 2^32 < p
    return gl_sub_nc(A, B);
}

}  // namespace starkhip
