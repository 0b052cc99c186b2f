// gfx950 device-only Goldilocks arithmetic with lazy ("non-canonical") reduction.
//
// A value is any 64-bit representative of its class mod p = 2^64 - 2^32 + 1 (so p..2^64-1 are allowed
// aliases of 0..2^32-2).  Each routine states which operands may be arbitrary and which must be canonical;
// every step is exact mod p, so a kernel canonicalises once (gl_canon) where results leave it.
// Every integer VALU instruction issues at the same quarter rate on CDNA4, so the routines below are written for
// instruction COUNT: a 64 x 64 multiply-reduce is 13 VALU instructions (four v_mad_u64_u32 + two moves for the product,
// seven for the reduction with 2^64 = 2^32 - 1 and 2^96 = -1 mod p), with carries kept in scalar flag registers.
#pragma once
#include <hip/hip_runtime.h>

#include "gl.h"

namespace starkhip {

__device__ __forceinline__ gl_t gl_canon(gl_t x) { return x >= GL_P ? x - GL_P : x; }

// ---- reduction of a 128-bit value given as words: (h1 : h0 : l1 : l0) + cin * 2^96, any representative out.
// cin_mask is a per-lane flag in a scalar register pair (the carry-out of a v_mad_u64_u32), or absent (HAS_CIN = false).
//
//   x = (l1:l0) - h1 - cin + h0 * eps          (2^64 = eps = 2^32 - 1, 2^96 = -1 mod p)
//
// The subtract's borrow and the multiply-add's carry are taken from the instructions' own scalar flag outputs (the
// compiler would spend a 64-bit compare on each) and settled with ONE correction:
//   D = (l1:l0) - h1 - cin mod 2^64, borrow b: true value D - b * 2^64 = D - b * eps
//   r = D + h0 * eps mod 2^64, carry c:        true value r + c * 2^64 = r + c * eps
//   b == c: r;  c only: r + eps (r < h0 * eps <= 2^64 - 2^33 + 1: no wrap);  b only: r - eps (r >= D >= 2^64 - 2^32: no wrap)
// 7 VALU instructions, none on the scalar ALU.  A VALU write of a scalar register needs two wait states before a VALU
// reads it: the s_nop's.
template <bool HAS_CIN>
__device__ __forceinline__ gl_t gl_reduce_words(uint32_t l0, uint32_t l1, uint32_t h0, uint32_t h1, uint64_t cin_mask) {
    uint32_t d0, d1;
    uint64_t borrow_mask, carry_mask, r;
    if (HAS_CIN)
        asm("v_subb_co_u32_e64 %0, %2, %3, %4, %6\n\ts_nop 1\n\tv_subb_co_u32_e64 %1, %2, %5, 0, %2"
            : "=&v"(d0), "=&v"(d1), "=&s"(borrow_mask)
            : "v"(l0), "v"(h1), "v"(l1), "s"(cin_mask));
    else
        asm("v_sub_co_u32_e64 %0, %2, %3, %4\n\ts_nop 1\n\tv_subb_co_u32_e64 %1, %2, %5, 0, %2"
            : "=&v"(d0), "=&v"(d1), "=&s"(borrow_mask)
            : "v"(l0), "v"(h1), "v"(l1));
    const uint64_t D = ((uint64_t)d1 << 32) | d0;
    // r + (c - b) * eps without touching the scalar ALU (a scalar XOR/AND of the two flags and selects on the result
    // would put a VALU -> SALU -> VALU round trip into every multiply):  t = c - b in {-1, 0, 1} from the flags as
    // carry/borrow inputs, then r - t (signed multiply-add by -1) and t added to the high word (+ t * 2^32).
    uint32_t t;
    uint64_t scratch_mask;
    asm("v_mad_u64_u32 %0, %2, %4, -1, %5\n\t"
        "v_subb_co_u32_e64 %1, %3, 0, 0, %6\n\t"  // t = -b; with the s_nop the two wait states between the carry flag's write and read
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %1, %3, %1, 0, %2\n\t"  // t = c - b
        "v_mad_i64_i32 %0, %3, %1, -1, %0"
        : "=&v"(r), "=&v"(t), "=&s"(carry_mask), "=&s"(scratch_mask)
        : "v"(h0), "v"(D), "s"(borrow_mask));  // h0 * eps < p
    return r + ((uint64_t)t << 32);
}

// (hi * 2^64 + lo) mod p, any representative; hi, lo arbitrary
__device__ __forceinline__ gl_t gl_reduce128_nc(uint64_t hi, uint64_t lo) {
    return gl_reduce_words<false>((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32), 0);
}

// a * b mod p, any representative in [0, 2^64); a, b arbitrary 64-bit.
// 64 x 64 -> 128 as four v_mad_u64_u32 (32 x 32 + 64) and two moves:
//   p0 = a0 b0;  m1 = a0 b1 + p0_hi  (fits);  m2 + cm 2^64 = a1 b0 + m1  (the carry-out cm stays in a scalar register
//   pair and enters the reduction as -cm, since 2^96 = -1);  p3 = a1 b1 + m2_hi  (fits)
//   a b = p3 2^64 + cm 2^96 + (m2_lo : p0_lo)
// cm is read by the reduction's first subtract, after the move and the multiply-add that form p3: two wait states.
__device__ __forceinline__ gl_t gl_mul_nc(gl_t a, gl_t b) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t p0 = (uint64_t)a0 * b0;
    const uint64_t m1 = (uint64_t)a0 * b1 + (p0 >> 32);
    uint64_t m2, cm;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(m2), "=s"(cm) : "v"(a1), "v"(b0), "v"(m1));
    const uint64_t p3 = (uint64_t)a1 * b1 + (m2 >> 32);
    return gl_reduce_words<true>((uint32_t)p0, (uint32_t)m2, (uint32_t)p3, (uint32_t)(p3 >> 32), cm);
}

// a * b + c mod p, any representative; a, b, c arbitrary 64-bit.  As gl_mul_nc with c as the addend of the first
// multiply-add; its carry-out c0 (weight 2^64) joins m2_hi in the addend of the last one (m2_hi + c0 <= 2^32, and
// a1 b1 + 2^32 still fits 64 bits).
__device__ __forceinline__ gl_t gl_mad_nc(gl_t a, gl_t b, gl_t c) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    uint64_t p0, c0, m2, cm, scratch_mask;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(p0), "=s"(c0) : "v"(a0), "v"(b0), "v"(c));
    const uint64_t m1 = (uint64_t)a0 * b1 + (p0 >> 32);
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(m2), "=s"(cm) : "v"(a1), "v"(b0), "v"(m1));
    uint32_t t0, t1;  // (t1 : t0) = m2_hi + c0; c0 is three VALU instructions old here
    asm("v_addc_co_u32_e64 %0, %2, %3, 0, %4\n\ts_nop 1\n\tv_addc_co_u32_e64 %1, %2, 0, 0, %2"
        : "=&v"(t0), "=&v"(t1), "=&s"(scratch_mask)
        : "v"((uint32_t)(m2 >> 32)), "s"(c0));
    const uint64_t p3 = (uint64_t)a1 * b1 + (((uint64_t)t1 << 32) | t0);
    return gl_reduce_words<true>((uint32_t)p0, (uint32_t)m2, (uint32_t)p3, (uint32_t)(p3 >> 32), cm);
}

// gl_mad_nc with a WAVE-UNIFORM b (a kernel argument, an op-stream constant): its halves are scalar-register operands of
// the two hand-written multiply-adds instead of being copied to vector registers first (two moves less).  b must be
// uniform across the wave -- a divergent b would silently be read from the first lane.
__device__ __forceinline__ gl_t gl_mad_nc_ub(gl_t a, gl_t b, gl_t c) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    uint64_t p0, c0, m2, cm, scratch_mask;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(p0), "=s"(c0) : "v"(a0), "s"(b0), "v"(c));
    const uint64_t m1 = (uint64_t)a0 * b1 + (p0 >> 32);
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(m2), "=s"(cm) : "v"(a1), "s"(b0), "v"(m1));
    uint32_t t0, t1;
    asm("v_addc_co_u32_e64 %0, %2, %3, 0, %4\n\ts_nop 1\n\tv_addc_co_u32_e64 %1, %2, 0, 0, %2"
        : "=&v"(t0), "=&v"(t1), "=&s"(scratch_mask)
        : "v"((uint32_t)(m2 >> 32)), "s"(c0));
    const uint64_t p3 = (uint64_t)a1 * b1 + (((uint64_t)t1 << 32) | t0);
    return gl_reduce_words<true>((uint32_t)p0, (uint32_t)m2, (uint32_t)p3, (uint32_t)(p3 >> 32), cm);
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
