// gfx950 device-side Poseidon-Goldilocks, written for the integer VALU of CDNA4 (every instruction but a plain
// 32-bit add issues in 4 cycles, v_mad_u64_u32 included -- so: as few instructions as possible):
//  * the MDS layer (30 per permutation, 144 small-constant MACs each) multiplies the 32-bit halves of the state
//    words into two 64-bit accumulators per output with v_mad_u64_u32 and folds them once (2^64 = 2^32 - 1);
//  * "quad" form for the big Merkle-leaf kernel: one permutation spread over the 4 lanes of a DPP quad (lane l
//    owns state elements 3l .. 3l+2), so the kernel fields 4x as many lanes as there are leaves -- a trace
//    commitment has only N = 32768 leaves of 9191 sequential permutations each, far too few lanes for 256 CUs
//    with one lane per leaf.  The other lanes' words arrive through v_mov_b32 quad_perm rotations;
//  * values inside a permutation are lazily reduced 64-bit representatives (gl_dev.h).
//  * the 22 partial rounds of the quad form are taken three at a time (only element 0 is non-linear in them): one dense
//    12 x 12 layer with per-lane integer coefficients, two dot products and three S-boxes per triple (bottom of this file).
// Semantics are exactly those of poseidon.h (same permutation); tests compare both against the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>

#include "gl_dev.h"
#include "poseidon.h"

namespace starkhip {

// ---------------------------------------------------------------- shared pieces (lazy reduction, gl_dev.h)
// The MDS matrix is circulant with 6-bit entries: out[r] = sum_i CIRC[i] * s[(i + r) % 12] (+ 8 * s[0] for r = 0).  Every
// product is taken on the 32-bit halves of the state words into two 64-bit accumulators, one v_mad_u64_u32 each
// (that instruction issues at the same 4 cycles as a 24-bit multiply on this part, tools/valu_rate_bench.hip).
__device__ __forceinline__ uint64_t mad32(uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * b + c; }  // one v_mad_u64_u32

__device__ __forceinline__ gl_t sbox_nc(gl_t x) {
    const gl_t x2 = gl_mul_nc(x, x), x4 = gl_mul_nc(x2, x2), x3 = gl_mul_nc(x2, x);
    return gl_mul_nc(x3, x4);
}

// A + B * 2^32 mod p for A, B < 2^44, any representative
__device__ __forceinline__ gl_t combine_lohi_nc(uint64_t A, uint64_t B) {
    // B * 2^32 = b1 * 2^64 + b0 * 2^32 = b1 * eps + b0 * 2^32 (mod p): one multiply-add, then only the high words add
    const uint32_t b0 = (uint32_t)B, b1 = (uint32_t)(B >> 32);
    const uint64_t T = mad32(b1, 0xFFFFFFFFu, A);         // A + b1 * eps < 2^45
    uint32_t hi;
    const bool c = __builtin_add_overflow((uint32_t)(T >> 32), b0, &hi);
    uint64_t r = ((uint64_t)hi << 32) | (uint32_t)T;
    asm("" : "+v"(r));  // keep (T_lo, hi) one register pair: otherwise the two adds are re-associated through three extra moves
    return r + (c ? GL_EPS : 0);                          // after a wrap hi < 2^13, r < 2^45: cannot wrap again
}

// ---------------------------------------------------------------- one permutation per lane
// (Merkle levels, proof-of-work grinding, FRI leaves: few, short launches.)  Any representative in and out.
__device__ __forceinline__ void poseidon_mds_dev(gl_t* s) {
    constexpr uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    uint32_t lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        lo[i] = (uint32_t)s[i];
        hi[i] = (uint32_t)(s[i] >> 32);
    }
#pragma unroll
    for (int r = 0; r < 12; r++) {
        uint64_t A = 0, B = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const int j = (i + r) % 12;
            const uint32_t k = CIRC[i] + ((r == 0 && i == 0) ? 8u : 0u);
            A = mad32(lo[j], k, A);
            B = mad32(hi[j], k, B);
        }
        s[r] = combine_lohi_nc(A, B);  // A, B <= (276 + 8) * 2^32
    }
}

// In: canonical or not; out: canonical.
__device__ __forceinline__ void poseidon_permute_dev(gl_t* s) {
    const uint64_t* RC = POSEIDON_RC_DEV;
    int rc = 0;
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = sbox_nc(gl_add_nc(s[i], RC[rc + i]));
        rc += 12;
        poseidon_mds_dev(s);
    }
#pragma unroll 1
    for (int r = 0; r < 22; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl_add_nc(s[i], RC[rc + i]);
        rc += 12;
        s[0] = sbox_nc(s[0]);
        poseidon_mds_dev(s);
    }
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = sbox_nc(gl_add_nc(s[i], RC[rc + i]));
        rc += 12;
        poseidon_mds_dev(s);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
}

__device__ __forceinline__ void poseidon_two_to_one_dev(const gl_t* a, const gl_t* b, gl_t* out) {
    gl_t s[12];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        s[i] = a[i];
        s[4 + i] = b[i];
        s[8 + i] = 0;
    }
    poseidon_permute_dev(s);
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = s[i];
}

__device__ __forceinline__ void poseidon_hash_or_noop_dev(const gl_t* in, size_t len, size_t stride, gl_t* out) {
    if (len <= 4) {
        for (size_t i = 0; i < 4; i++) out[i] = i < len ? in[i * stride] : 0;
        return;
    }
    gl_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    for (size_t off = 0; off < len; off += 8) {
        const size_t k = len - off < 8 ? len - off : 8;
        for (size_t i = 0; i < k; i++) s[i] = in[(off + i) * stride];
        poseidon_permute_dev(s);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = s[i];
}

// ---------------------------------------------------------------- one permutation per DPP quad
// Lane l (= lane id & 3) owns the three state elements l, l + 4, l + 8 in (s0, s1, s2) ("slots" 0, 1, 2).  The sponge's rate
// (elements 0 .. 7) is then slots 0 and 1 of every lane -- each lane absorbs two columns per block -- and the capacity
// (8 .. 11) is slot 2 of every lane, so the last linear layer of a permutation that is followed by a full absorb needs ONE
// output per lane instead of three (the rate outputs would be overwritten): 59 instructions less per permutation.
// A quad rotation by r lanes brings element ((l + r) & 3) + 4 m' to slot m' of lane l; the circulant coefficient of that
// operand for output l + 4 m is CIRC[(((l + r) & 3) - l + 4 (m' - m)) mod 12] -- it depends on the lane through the wrap of
// (l + r), so every lane holds its twelve coefficients cf[3 r + (m' - m) mod 3] in registers (the first layout, lane l owning
// 3l .. 3l+2, had lane-uniform inline constants but no way to skip the rate outputs).
//
// Cost model (measured on MI355X, tools/valu_rate_bench.hip): every integer VALU instruction other than a plain
// 32-bit add/logic op issues in 4 cycles per wave -- v_mad_u64_u32 (32 x 32 + 64) included -- so the MDS layer is
// written to minimise instruction COUNT: the 64-bit state words are multiplied as two 32-bit halves into two
// 64-bit accumulators (2 mads per coefficient, 74 per lane and round) after 18 quad-rotation moves, instead of
// 22-bit limbs with 24-bit multiplies (108 multiplies + 54 three-operand adds + limb split and merge).
// Values are kept as ANY 64-bit representative of their residue; the caller canonicalises what leaves the
// permutation (gl_canon).
template <int ROT>
__device__ __forceinline__ uint32_t quad_rot(uint32_t v) {
    // lane i of the quad reads lane (i + ROT) & 3
    constexpr int sel = ((0 + ROT) & 3) | (((1 + ROT) & 3) << 2) | (((2 + ROT) & 3) << 4) | (((3 + ROT) & 3) << 6);
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, sel, 0xF, 0xF, true);
}

// The same rotation on the LDS pipe (ds_swizzle_b32 touches no LDS memory): it takes no VALU issue slot, and the MDS layer
// is VALU-bound.  Its latency is that of an LDS access, so it is used where other work is at hand (the MDS layer starts
// with the lane's own elements) and not inside the S-box chain.
template <int ROT>
__device__ __forceinline__ uint32_t quad_rot_lds(uint32_t v) {
    constexpr int sel = ((0 + ROT) & 3) | (((1 + ROT) & 3) << 2) | (((2 + ROT) & 3) << 4) | (((3 + ROT) & 3) << 6);
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x8000 | sel);
}

#ifndef STARKHIP_MDS_SWIZZLE_MASK
#define STARKHIP_MDS_SWIZZLE_MASK 0
#endif
template <int ROT>
__device__ __forceinline__ uint32_t mds_rot(uint32_t v) {
    return ((STARKHIP_MDS_SWIZZLE_MASK >> ROT) & 1) ? quad_rot_lds<ROT>(v) : quad_rot<ROT>(v);
}

template <int ROT>
__device__ __forceinline__ gl_t quad_rot64(gl_t v) {
    return (gl_t)quad_rot<ROT>((uint32_t)v) | ((gl_t)quad_rot<ROT>((uint32_t)(v >> 32)) << 32);
}
__device__ __forceinline__ gl_t quad_lane0_64(gl_t v) {  // every lane of the quad reads lane 0's value
    const uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)v, 0x00, 0xF, 0xF, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(v >> 32), 0x00, 0xF, 0xF, true);
    return (gl_t)lo | ((gl_t)hi << 32);
}

// Partial rounds: x^7 of lane 0's element only.  The quad's other lanes would repeat lane 0's four multiplies for
// nothing, so lane 1 is put to work instead: after x^2, lane 0 forms x^3 = x^2 * x while lane 1 forms x^4 = x^2 * x^2 in
// the SAME multiply, then lane 0 fetches x^4 from lane 1 for x^7 = x^3 * x^4: three multiplies deep instead of four.
__device__ __forceinline__ gl_t sbox_lane0_nc(gl_t s0, bool lane0) {
    const gl_t x2 = quad_lane0_64(gl_mul_nc(s0, s0));          // lane 0's x^2 in every lane
    const gl_t y = gl_mul_nc(x2, lane0 ? s0 : x2);              // lane 0: x^3, lanes 1..3: x^4
    const gl_t x7 = gl_mul_nc(y, quad_rot64<1>(y));             // lane 0: x^3 * (lane 1's x^4)
    return lane0 ? x7 : s0;
}

// Round constants as the kernels stage them in LDS: per constant two 64-bit words (low half, high half), so each
// is directly the 64-bit addend of the first multiply-add of its accumulator.
struct RcPair {
    uint64_t lo, hi;
};

// MDS layer; the accumulators start from this lane's three round constants of the NEXT round.
// cf: this lane's twelve circulant coefficients (above); diag0 = 8 on lane 0, 0 elsewhere (MDS_MATRIX_DIAG has a single non-zero
// entry, at element 0 = lane 0's slot 0).  CAP_ONLY: only slot 2 (the capacity element) is computed, s0 and s1 are left as
// they are -- for a permutation whose rate outputs the next absorb overwrites.
template <bool CAP_ONLY>
__device__ __forceinline__ void poseidon_mds_quad(gl_t& s0, gl_t& s1, gl_t& s2, const uint32_t (&cf)[12], uint32_t diag0, const RcPair& c0,
                                                  const RcPair& c1, const RcPair& c2) {
    uint32_t lo[4][3], hi[4][3];  // [r][m']: halves of element ((l + r) & 3) + 4 m'
    lo[0][0] = (uint32_t)s0; hi[0][0] = (uint32_t)(s0 >> 32);
    lo[0][1] = (uint32_t)s1; hi[0][1] = (uint32_t)(s1 >> 32);
    lo[0][2] = (uint32_t)s2; hi[0][2] = (uint32_t)(s2 >> 32);
#pragma unroll
    for (int m = 0; m < 3; m++) {
        lo[1][m] = mds_rot<1>(lo[0][m]); hi[1][m] = mds_rot<1>(hi[0][m]);
        lo[2][m] = mds_rot<2>(lo[0][m]); hi[2][m] = mds_rot<2>(hi[0][m]);
        lo[3][m] = mds_rot<3>(lo[0][m]); hi[3][m] = mds_rot<3>(hi[0][m]);
    }
    gl_t out[3] = {s0, s1, s2};
#pragma unroll
    for (int mo = CAP_ONLY ? 2 : 0; mo < 3; mo++) {
        const RcPair& c = mo == 0 ? c0 : mo == 1 ? c1 : c2;
        uint64_t A = c.lo, B = c.hi;
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int m = 0; m < 3; m++) {
                const uint32_t k = cf[3 * r + (m - mo + 3) % 3];
                A = mad32(lo[r][m], k, A);
                B = mad32(hi[r][m], k, B);
                // keep the chain a chain: without these (empty) barriers the sum is re-associated so that the round
                // constant is added by an instruction of its own instead of being the addend of the first multiply-add
                asm("" : "+v"(A));
                asm("" : "+v"(B));
            }
        if (mo == 0) {
            A = mad32(lo[0][0], diag0, A);
            B = mad32(hi[0][0], diag0, B);
        }
        out[mo] = combine_lohi_nc(A, B);  // A, B <= 2^32 + (276 + 8) * 2^32 < 2^41
    }
    s0 = out[0];
    s1 = out[1];
    s2 = out[2];
}

// ---------------------------------------------------------------- merged partial rounds
// Only element 0 passes the S-box in a partial round, so three consecutive partial rounds are linear in the eleven other
// elements.  With M the MDS matrix, Mz = M with row 0 zeroed, and x_k the k-th S-box output:
//     y1 = (M u')[0] + c1[0]                              u' = (x_1, u_1, .., u_11),  x_1 = u_0^7
//     y2 = (M Mz u')[0] + M00 x_2 + (M c1z)[0] + c2[0]     x_2 = y1^7
//     out = M Mz Mz u' + (M Mz e0) x_3' ...               (see build_quad_merged_tables for every term)
// i.e. ONE dense 12 x 12 layer (72 multiply-adds per lane, entries of M Mz Mz < 2^21: still 32-bit multiplicands with
// 64-bit accumulators), two 12-term dot products for the intermediate element 0, and the three S-boxes -- instead of
// three dense layers.  The matrices are not circulant (the diagonal term and the zeroed rows break that), so each lane
// holds its coefficients in registers: n3[mo][3r+m] = (M Mz Mz)[l + 4 mo][col], col = ((l + r) & 3) + 4 m, the lane's own
// columns of row 0 of M and of M Mz, and the columns that multiply x_2 and x_3.
struct QuadMergedCoef {
    uint32_t n3[3][12];
    uint32_t n1[3], n2[3];  // row 0 of M and of M Mz at this lane's OWN three columns: the two intermediate dot products
                            // are per-lane partial sums added up across the quad (6 multiply-adds + a butterfly, not 24)
    uint32_t m00;           // M[0][0] on lane 0, 0 elsewhere (the x_2 term of y2 is added once)
    uint32_t b2[3], b3[3];  // (M Mz)[l + 4 mo][0], M[l + 4 mo][0]
    uint32_t cf[12];        // the circulant coefficients of the plain layers (poseidon_mds_quad)
};
static const int QUAD_MERGED_TRIPLES = 7;  // = POSEIDON_MERGED_TRIPLES (poseidon_merged.h): partial rounds 0..20; the 22nd stays a plain round

// y is the same value in the four lanes of a quad; returns y^7 in all of them.  Even lanes form x^3, odd lanes x^4 in one
// multiply, and every lane finds the other factor in its right-hand neighbour.
__device__ __forceinline__ gl_t sbox_quad_uniform(gl_t y, bool even_lane) {
    const gl_t x2 = gl_mul_nc(y, y);
    const gl_t v = gl_mul_nc(x2, even_lane ? y : x2);
    return gl_mul_nc(v, quad_rot64<1>(v));
}

__device__ __forceinline__ void quad_dot12(const uint32_t (&lo)[4][3], const uint32_t (&hi)[4][3], const uint32_t* coef, uint64_t& A,
                                           uint64_t& B) {
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int m = 0; m < 3; m++) {
            A = mad32(lo[r][m], coef[3 * r + m], A);
            B = mad32(hi[r][m], coef[3 * r + m], B);
            asm("" : "+v"(A));  // keep the chain a chain (see poseidon_mds_quad)
            asm("" : "+v"(B));
        }
}

// a += the same accumulator of the lane's neighbour at distance 1, then at distance 2: the quad's total in all four lanes
__device__ __forceinline__ uint64_t quad_sum64(uint64_t a) {
    const uint32_t l1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)a, 0xB1, 0xF, 0xF, true);          // quad_perm [1,0,3,2]
    const uint32_t h1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(a >> 32), 0xB1, 0xF, 0xF, true);
    uint64_t n1 = (uint64_t)l1 | ((uint64_t)h1 << 32);
    asm("" : "+v"(n1));  // one 64-bit add of a register pair (else: one add per word, each with a zeroed partner register)
    a += n1;
    const uint32_t l2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)a, 0x4E, 0xF, 0xF, true);          // quad_perm [2,3,0,1]
    const uint32_t h2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(a >> 32), 0x4E, 0xF, 0xF, true);
    uint64_t n2 = (uint64_t)l2 | ((uint64_t)h2 << 32);
    asm("" : "+v"(n2));
    return a + n2;
}

// Three partial rounds.  In: (s0, s1, s2) with this round's constants already added (as poseidon_mds_quad leaves them);
// out: the state three rounds later with the following round's constants added.  k1, k2: the constants of the two
// intermediate element-0 values, each divided by four (every lane seeds its partial sum with it); k3: this lane's three
// output constants.
__device__ __forceinline__ void poseidon_partial3_quad(gl_t& s0, gl_t& s1, gl_t& s2, const QuadMergedCoef& c, const RcPair& k1,
                                                       const RcPair& k2, const RcPair* k3, bool lane0, bool even_lane) {
    s0 = sbox_lane0_nc(s0, lane0);  // lane 0: x_1; the other lanes keep their element
    uint32_t lo[4][3], hi[4][3];
    lo[0][0] = (uint32_t)s0; hi[0][0] = (uint32_t)(s0 >> 32);
    lo[0][1] = (uint32_t)s1; hi[0][1] = (uint32_t)(s1 >> 32);
    lo[0][2] = (uint32_t)s2; hi[0][2] = (uint32_t)(s2 >> 32);
#pragma unroll
    for (int m = 0; m < 3; m++) {
        lo[1][m] = quad_rot<1>(lo[0][m]); hi[1][m] = quad_rot<1>(hi[0][m]);
        lo[2][m] = quad_rot<2>(lo[0][m]); hi[2][m] = quad_rot<2>(hi[0][m]);
        lo[3][m] = quad_rot<3>(lo[0][m]); hi[3][m] = quad_rot<3>(hi[0][m]);
    }
    uint64_t A = k1.lo, B = k1.hi;
#pragma unroll
    for (int m = 0; m < 3; m++) {
        A = mad32(lo[0][m], c.n1[m], A);
        B = mad32(hi[0][m], c.n1[m], B);
    }
    const gl_t x2 = sbox_quad_uniform(combine_lohi_nc(quad_sum64(A), quad_sum64(B)), even_lane);  // sums < 2^40
    const uint32_t x2l = (uint32_t)x2, x2h = (uint32_t)(x2 >> 32);
    A = k2.lo; B = k2.hi;
#pragma unroll
    for (int m = 0; m < 3; m++) {
        A = mad32(lo[0][m], c.n2[m], A);
        B = mad32(hi[0][m], c.n2[m], B);
    }
    A = mad32(x2l, c.m00, A);
    B = mad32(x2h, c.m00, B);
    const gl_t x3 = sbox_quad_uniform(combine_lohi_nc(quad_sum64(A), quad_sum64(B)), even_lane);  // sums < 2^48
    const uint32_t x3l = (uint32_t)x3, x3h = (uint32_t)(x3 >> 32);
    gl_t out[3];
#pragma unroll
    for (int mo = 0; mo < 3; mo++) {
        A = k3[mo].lo; B = k3[mo].hi;
        quad_dot12(lo, hi, c.n3[mo], A, B);
        A = mad32(x2l, c.b2[mo], A);
        B = mad32(x2h, c.b2[mo], B);
        A = mad32(x3l, c.b3[mo], A);
        B = mad32(x3h, c.b3[mo], B);
        out[mo] = combine_lohi_nc(A, B);  // A, B < 2^57: 12 terms of (< 2^21) x (< 2^32) and the two single terms
    }
    s0 = out[0];
    s1 = out[1];
    s2 = out[2];
}

// One permutation of the quad form, the partial rounds taken three at a time.  rc: this lane's view of the round constants,
// rc[3 r + m] = split(RC[12 r + l + 4 m]), r < 30, followed by three zeros; tk[t] = {k1, k2} of triple t, tk3 = this lane's k3
// constants, three per triple.  In: canonical or not; out: any representative (canonicalise with gl_canon before it leaves the
// kernel).  CAP_ONLY: the last layer leaves the rate elements (s0, s1) unspecified -- the caller overwrites them.
template <bool CAP_ONLY>
__device__ __forceinline__ void poseidon_permute_quad_merged(gl_t& s0, gl_t& s1, gl_t& s2, uint32_t diag0, const RcPair* __restrict__ rc,
                                                             const QuadMergedCoef& c, const RcPair* __restrict__ tk,
                                                             const RcPair* __restrict__ tk3, bool lane0, bool even_lane) {
    s0 = gl_add_nc(s0, rc[0].lo | (rc[0].hi << 32));
    s1 = gl_add_nc(s1, rc[1].lo | (rc[1].hi << 32));
    s2 = gl_add_nc(s2, rc[2].lo | (rc[2].hi << 32));
    int r = 0;
#pragma unroll 1
    for (; r < 4; r++) {
        s0 = sbox_nc(s0);
        s1 = sbox_nc(s1);
        s2 = sbox_nc(s2);
        poseidon_mds_quad<false>(s0, s1, s2, c.cf, diag0, rc[3 * r + 3], rc[3 * r + 4], rc[3 * r + 5]);
    }
#pragma unroll 1
    for (int t = 0; t < QUAD_MERGED_TRIPLES; t++) poseidon_partial3_quad(s0, s1, s2, c, tk[2 * t], tk[2 * t + 1], tk3 + 3 * t, lane0, even_lane);
    r = 4 + 3 * QUAD_MERGED_TRIPLES;  // 25: the last partial round
    s0 = sbox_lane0_nc(s0, lane0);
    poseidon_mds_quad<false>(s0, s1, s2, c.cf, diag0, rc[3 * r + 3], rc[3 * r + 4], rc[3 * r + 5]);
    r++;
#pragma unroll 1
    for (; r < 29; r++) {
        s0 = sbox_nc(s0);
        s1 = sbox_nc(s1);
        s2 = sbox_nc(s2);
        poseidon_mds_quad<false>(s0, s1, s2, c.cf, diag0, rc[3 * r + 3], rc[3 * r + 4], rc[3 * r + 5]);
    }
    s0 = sbox_nc(s0);
    s1 = sbox_nc(s1);
    s2 = sbox_nc(s2);
    poseidon_mds_quad<CAP_ONLY>(s0, s1, s2, c.cf, diag0, rc[3 * r + 3], rc[3 * r + 4], rc[3 * r + 5]);
}

// ---------------------------------------------------------------- one permutation per DPP row (16 lanes)
// For commitments with few leaves (a 1024-row AIR has 2048 .. 4096, FP12Mul 32) the quad form is a latency chain: 128 .. 256
// waves of up to 12 167 sequential permutations on 1024 SIMDs, and a lone wave issues one instruction per ~5 cycles
// whether it depends on the previous one or not.  What shortens that chain is fewer instructions PER WAVE and permutation,
// at any cost in lanes: here lane e < 12 of a row owns state element e (lanes 12 .. 15 mirror lanes 0 .. 3), so an S-box layer
// is ONE x^7 (52 instructions instead of 3 x 52) and a linear layer is 24 multiply-adds + 28 row moves (instead of 72 + 18).
// A wave then carries 4 leaves instead of 16: 4x the lane-instructions of the quad form, which is why the big commitments and
// the pool's merged launches stay with the quad form (kernels_hash.hip picks).
//
// Rotation by k on 12 lanes with 16-lane row shifts: with x16 = x on lanes 0 .. 11 and x[0 .. 3] again on lanes 12 .. 15,
// (row_shl:k x16)[i] = x[(i + k) mod 12] for every i < 12 as long as k <= 4; larger k go through z = rot 4 and w = rot 8,
// mirrored again.
template <int K>
__device__ __forceinline__ uint32_t row_shl(uint32_t v) {  // lane i of the row reads lane i + K (lanes past the row's end: 0)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + K, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t row_mirror(uint32_t v) {  // lanes 12 .. 15 <- lanes 0 .. 3 (row_shr:12, bank 3 only)
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x11C, 0xF, 0x8, false);
}
template <int K>
__device__ __forceinline__ uint32_t row_ror(uint32_t v) {  // rotation over all 16 lanes of the row
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + K, 0xF, 0xF, false);
}

// out_e = seed + sum_k coef[k] * x[(e + k) mod 12]: the circulant layer with coef = CIRC (+ 8 at k = 0 on lane 0), or any
// 12 x 12 layer with per-lane coefficients coef[k] = Mat[e][(e + k) mod 12] < 2^21.  (lo, hi): the halves of this lane's x.
__device__ __forceinline__ gl_t row_layer(uint32_t lo, uint32_t hi, const uint32_t (&coef)[12], uint64_t A, uint64_t B) {
#define STARKHIP_ROW_TERM(L, H, K)            \
    A = mad32((L), coef[K], A);               \
    B = mad32((H), coef[K], B);               \
    asm("" : "+v"(A));                        \
    asm("" : "+v"(B));
    STARKHIP_ROW_TERM(lo, hi, 0)
    const uint32_t xl = row_mirror(lo), xh = row_mirror(hi);
    STARKHIP_ROW_TERM(row_shl<1>(xl), row_shl<1>(xh), 1)
    STARKHIP_ROW_TERM(row_shl<2>(xl), row_shl<2>(xh), 2)
    STARKHIP_ROW_TERM(row_shl<3>(xl), row_shl<3>(xh), 3)
    const uint32_t z0l = row_shl<4>(xl), z0h = row_shl<4>(xh);
    STARKHIP_ROW_TERM(z0l, z0h, 4)
    const uint32_t zl = row_mirror(z0l), zh = row_mirror(z0h);
    STARKHIP_ROW_TERM(row_shl<1>(zl), row_shl<1>(zh), 5)
    STARKHIP_ROW_TERM(row_shl<2>(zl), row_shl<2>(zh), 6)
    STARKHIP_ROW_TERM(row_shl<3>(zl), row_shl<3>(zh), 7)
    const uint32_t w0l = row_shl<4>(zl), w0h = row_shl<4>(zh);
    STARKHIP_ROW_TERM(w0l, w0h, 8)
    const uint32_t wl = row_mirror(w0l), wh = row_mirror(w0h);
    STARKHIP_ROW_TERM(row_shl<1>(wl), row_shl<1>(wh), 9)
    STARKHIP_ROW_TERM(row_shl<2>(wl), row_shl<2>(wh), 10)
    STARKHIP_ROW_TERM(row_shl<3>(wl), row_shl<3>(wh), 11)
#undef STARKHIP_ROW_TERM
    return combine_lohi_nc(A, B);  // A, B < 2^33 + 12 * 2^21 * 2^32 < 2^57
}

// The circulant layer again, as ONE scheduled asm block (tools/gen_row_layer_asm.py -> row_layer_asm.inc): a lone wave pays a
// whole issue slot for every wait state hipcc fills with s_nop, and from the C++ above it makes 31 of them per round (one
// temporary for every rotated operand: a write-after-read wait before each move; moves that read a register written two
// instructions earlier).  The block rotates six temporaries and issues each group's moves under the previous group's
// multiply-adds: 52 instructions + one s_nop.  The coefficients are inline constants; c0 = 17 (+ 8 on lane 0).
#include "row_layer_asm.inc"
__device__ __forceinline__ gl_t row_layer_circ(uint32_t lo, uint32_t hi, uint32_t c0, uint64_t A, uint64_t B) {
    uint32_t t0, t1, t2, t3, t4, t5, zl, zh;
    uint64_t carry_sink;
    asm(STARKHIP_ROW_LAYER_CIRC_ASM
        : "+v"(A), "+v"(B), "+v"(lo), "+v"(hi), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(zl), "=&v"(zh),
          "=&s"(carry_sink)
        : "v"(c0));
    return combine_lohi_nc(A, B);  // A, B < 2^33 + 284 * 2^32
}

// x^7 of lane 0's element only (partial rounds): lane 0 forms x^3 while lane 1 forms x^4 in the same multiply.
__device__ __forceinline__ gl_t sbox_row_lane0_nc(gl_t s, bool lane0) {
    const gl_t sq = gl_mul_nc(s, s);
    // lanes 0 and 1 of each quad read lane 0 (quad_perm [0,0,2,3]); only the row's first quad matters
    const uint32_t x2l = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)sq, 0xE0, 0xF, 0xF, true);
    const uint32_t x2h = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(sq >> 32), 0xE0, 0xF, 0xF, true);
    const gl_t x2 = (gl_t)x2l | ((gl_t)x2h << 32);
    const gl_t y = gl_mul_nc(x2, lane0 ? s : x2);  // lane 0: x^3, lane 1: x^4
    // lane 0 reads lane 1 (quad_perm [1,1,2,3])
    const uint32_t nl = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)y, 0xE5, 0xF, 0xF, true);
    const uint32_t nh = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(y >> 32), 0xE5, 0xF, 0xF, true);
    const gl_t x7 = gl_mul_nc(y, (gl_t)nl | ((gl_t)nh << 32));
    return lane0 ? x7 : s;
}

// ---- whole rounds as scheduled asm blocks on fixed registers (tools/gen_row_round_asm.py -> row_round_asm.inc): the S-box's flag
// hand-offs are filled with the other multiply of x^3 / x^4 (full rounds) or with the whole layer over the eleven elements that do
// not pass the S-box (partial rounds: M s' = M (s with element 0 zeroed) + column 0 of M times x0^7).  120 and 113 issue slots per
// round against 134 and 127 from the C++ above.
#include "row_round_asm.inc"
typedef uint32_t row_u32x4 __attribute__((ext_vector_type(4)));
struct RowConsts {
    uint32_t c0, col0;   // CIRC[0] (+ 8 on lane 0); column 0 of the MDS matrix at this lane
    uint32_t za, zb;     // zero, opaque to the compiler: the upper halves of the multiplies' addend pairs stay in their registers
    uint64_t mask0;      // lane 0 of every row
    uint64_t maske;      // even lanes
    // merged triples (poseidon_merged.h), this lane's views: n3k[k] = N3[e][(e + k) mod 12]; misc0 = (M[0][e], N2[0][e], N2[e][0],
    // M[e][0]); misc1 = (N3[e][0], M[0][0] on lane 0, N2[0][0] on lane 0, -).  All zero on lanes 12 .. 15.
    row_u32x4 n3k0, n3k1, n3k2, misc0, misc1;
};
// Per-lane coefficient rows of the merged triples, built on the host once per device (kernels_hash.hip)
struct RowMergedTables {
    uint32_t coef[16][20];                       // [lane]: n3k[12], then misc0[4], misc1[4]
    RcPair k1[7], k2[7], k3[7][12];              // POSEIDON_MERGED_TRIPLES = 7
};
__device__ __forceinline__ void row_consts_init(RowConsts& K, unsigned e, const uint32_t* __restrict__ coef /* RowMergedTables::coef[e] */) {
    constexpr uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    K.c0 = e == 0 ? 25u : 17u;
    K.col0 = e == 0 ? 25u : e < 12 ? CIRC[(12 - e) % 12] : 0u;
    K.za = K.zb = 0;
    K.mask0 = 0x0001000100010001ull;
    K.maske = 0x5555555555555555ull;
    asm volatile("" : "+v"(K.za), "+v"(K.zb), "+s"(K.mask0), "+s"(K.maske));
    K.n3k0 = row_u32x4{coef[0], coef[1], coef[2], coef[3]};
    K.n3k1 = row_u32x4{coef[4], coef[5], coef[6], coef[7]};
    K.n3k2 = row_u32x4{coef[8], coef[9], coef[10], coef[11]};
    K.misc0 = row_u32x4{coef[12], coef[13], coef[14], coef[15]};
    K.misc1 = row_u32x4{coef[16], coef[17], coef[18], coef[19]};
}
// One round; OFF = byte offset of the constant AFTER the next round's in this lane's row of the LDS table (the block reloads the seed
// registers from there as soon as it has consumed them).  sa / sb: the next round's constant as two 64-bit addends, in and out.
template <int OFF>
__device__ __forceinline__ gl_t row_round_full_asm(gl_t s, uint64_t& sa, uint64_t& sb, uint32_t rc_lds, const RowConsts& K) {
    uint64_t out;
    asm(STARKHIP_ROW_FULL_ROUND_ASM
        : STARKHIP_ROW_STATE_OUT(out), STARKHIP_ROW_SEED_A(sa), STARKHIP_ROW_SEED_B(sb)
        : STARKHIP_ROW_STATE_LO((uint32_t)s), STARKHIP_ROW_STATE_HI((uint32_t)(s >> 32)), STARKHIP_ROW_ADDR(rc_lds), STARKHIP_ROW_C0(K.c0), STARKHIP_ROW_ZA(K.za),
          STARKHIP_ROW_ZB(K.zb), [off] "n"(OFF)
        : STARKHIP_ROW_CLOBBERS);
    return out;
}
template <int OFF>
__device__ __forceinline__ gl_t row_round_partial_asm(gl_t s, uint64_t& sa, uint64_t& sb, uint32_t rc_lds, const RowConsts& K) {
    uint64_t out;
    asm(STARKHIP_ROW_PARTIAL_ROUND_ASM
        : STARKHIP_ROW_STATE_OUT(out), STARKHIP_ROW_SEED_A(sa), STARKHIP_ROW_SEED_B(sb)
        : STARKHIP_ROW_STATE_LO((uint32_t)s), STARKHIP_ROW_STATE_HI((uint32_t)(s >> 32)), STARKHIP_ROW_ADDR(rc_lds), STARKHIP_ROW_C0(K.c0),
          STARKHIP_ROW_COL0(K.col0), STARKHIP_ROW_ZA(K.za), STARKHIP_ROW_ZB(K.zb), STARKHIP_ROW_MASK0(K.mask0), [off] "n"(OFF)
        : STARKHIP_ROW_CLOBBERS);
    return out;
}
// Three partial rounds at once.  k1 / k2 / k3: this triple's constants, each (low half, high half) as two 64-bit addends (k1 and k2
// on lane 0 only), in; the next triple's, fetched from OFF1 / OFF2 / OFF3 of the lane's LDS row, out.
template <int OFF1, int OFF2, int OFF3>
__device__ __forceinline__ gl_t row_triple_asm(gl_t s, row_u32x4& k1, row_u32x4& k2, row_u32x4& k3, uint32_t rc_lds, const RowConsts& K) {
    uint64_t out;
    asm(STARKHIP_ROW_TRIPLE_ASM
        : STARKHIP_ROW_STATE_OUT(out), STARKHIP_ROW_K1(k1), STARKHIP_ROW_K2(k2), STARKHIP_ROW_K3(k3)
        : STARKHIP_ROW_STATE_LO((uint32_t)s), STARKHIP_ROW_STATE_HI((uint32_t)(s >> 32)), STARKHIP_ROW_ADDR(rc_lds), STARKHIP_ROW_N3K0(K.n3k0),
          STARKHIP_ROW_N3K1(K.n3k1), STARKHIP_ROW_N3K2(K.n3k2), STARKHIP_ROW_MISC0(K.misc0), STARKHIP_ROW_MISC1(K.misc1), STARKHIP_ROW_ZA(K.za),
          STARKHIP_ROW_ZB(K.zb), STARKHIP_ROW_MASK0(K.mask0), STARKHIP_ROW_MASKE(K.maske), [off1] "n"(OFF1), [off2] "n"(OFF2), [off3] "n"(OFF3)
        : STARKHIP_ROW_CLOBBERS, STARKHIP_ROW_TRIPLE_CLOBBERS);
    return out;
}
static const int ROW_TRIPLE_BASE = 32;  // LDS row: RcPair[0 .. 31] round constants (30, 31 zero), then k1, k2, k3 of triple t at 32 + 3 t
template <int T>
__device__ __forceinline__ gl_t row_triples_from(gl_t s, row_u32x4& k1, row_u32x4& k2, row_u32x4& k3, uint32_t rc_lds, const RowConsts& K) {
    if constexpr (T < 7) {
        constexpr int O = (ROW_TRIPLE_BASE + 3 * (T + 1)) * (int)sizeof(RcPair);  // the next triple's (beyond the last: unused entries of the row)
        s = row_triple_asm<O, O + 16, O + 32>(s, k1, k2, k3, rc_lds, K);
        return row_triples_from<T + 1>(s, k1, k2, k3, rc_lds, K);
    } else {
        return s;
    }
}
template <int R, int END>
__device__ __forceinline__ gl_t row_rounds_range(gl_t s, uint64_t& sa, uint64_t& sb, uint32_t rc_lds, const RowConsts& K) {
    if constexpr (R < END) {
        constexpr int OFF = (R + 2) * (int)sizeof(RcPair);
        if constexpr (R < 4 || R >= 26) s = row_round_full_asm<OFF>(s, sa, sb, rc_lds, K);
        else s = row_round_partial_asm<OFF>(s, sa, sb, rc_lds, K);
        return row_rounds_range<R + 1, END>(s, sa, sb, rc_lds, K);
    } else {
        return s;
    }
}
// rc: this lane's row of the LDS table, RcPair rc[64]: round constants, then the merged triples' constants
__device__ __forceinline__ gl_t poseidon_permute_row_merged_asm(gl_t s, const RcPair* __restrict__ rc, const RowConsts& K) {
    s = gl_add_nc(s, rc[0].lo | (rc[0].hi << 32));
    uint64_t sa = rc[1].lo, sb = rc[1].hi;
    const uint32_t rc_lds = (uint32_t)(uintptr_t)rc;
    s = row_rounds_range<0, 4>(s, sa, sb, rc_lds, K);                      // leaves rc[4] added: the first partial round's constants
    auto pair4 = [](const RcPair& c) { return row_u32x4{(uint32_t)c.lo, (uint32_t)(c.lo >> 32), (uint32_t)c.hi, (uint32_t)(c.hi >> 32)}; };
    row_u32x4 k1 = pair4(rc[ROW_TRIPLE_BASE]), k2 = pair4(rc[ROW_TRIPLE_BASE + 1]), k3 = pair4(rc[ROW_TRIPLE_BASE + 2]);
    s = row_triples_from<0>(s, k1, k2, k3, rc_lds, K);                     // rounds 4 .. 24; leaves rc[25] added
    sa = rc[26].lo;
    sb = rc[26].hi;
    return row_rounds_range<25, 30>(s, sa, sb, rc_lds, K);                 // the 22nd partial round, then four full rounds
}

template <int R>
__device__ __forceinline__ gl_t row_rounds_from(gl_t s, uint64_t& sa, uint64_t& sb, uint32_t rc_lds, const RowConsts& K) {
    if constexpr (R < 30) {
        constexpr int OFF = (R + 2) * (int)sizeof(RcPair);  // rc[R + 1] is in the seed registers; rc[R + 2] is fetched (rc[30], rc[31] are zero)
        if constexpr (R < 4 || R >= 26) s = row_round_full_asm<OFF>(s, sa, sb, rc_lds, K);
        else s = row_round_partial_asm<OFF>(s, sa, sb, rc_lds, K);
        return row_rounds_from<R + 1>(s, sa, sb, rc_lds, K);
    } else {
        return s;
    }
}
// rc: this lane's row of the LDS table, RcPair rc[32] (rounds 0 .. 29, then zeros)
__device__ __forceinline__ gl_t poseidon_permute_row_asm(gl_t s, const RcPair* __restrict__ rc, const RowConsts& K) {
    s = gl_add_nc(s, rc[0].lo | (rc[0].hi << 32));
    uint64_t sa = rc[1].lo, sb = rc[1].hi;
    const uint32_t rc_lds = (uint32_t)(uintptr_t)rc;  // the low half of a generic pointer into LDS is the LDS address
    return row_rounds_from<0>(s, sa, sb, rc_lds, K);
}

// One permutation of the row form, plain rounds.  rc: this lane's round constants split in halves, rc[r] for round r < 30 and a
// zero at rc[30]; c0 = 17 (+ 8 on lane 0): the k = 0 coefficient.  Lanes 12 .. 15 compute on mirrored copies and are never read.
// In: canonical or not; out: any representative.
__device__ __forceinline__ gl_t poseidon_permute_row(gl_t s, const RcPair* __restrict__ rc, uint32_t c0, bool lane0) {
    s = gl_add_nc(s, rc[0].lo | (rc[0].hi << 32));
    int r = 0;
#pragma unroll 1
    for (; r < 4; r++) {
        s = sbox_nc(s);
        s = row_layer_circ((uint32_t)s, (uint32_t)(s >> 32), c0, rc[r + 1].lo, rc[r + 1].hi);
    }
#pragma unroll 1
    for (; r < 26; r++) {
        s = sbox_row_lane0_nc(s, lane0);
        s = row_layer_circ((uint32_t)s, (uint32_t)(s >> 32), c0, rc[r + 1].lo, rc[r + 1].hi);
    }
#pragma unroll 1
    for (; r < 30; r++) {
        s = sbox_nc(s);
        s = row_layer_circ((uint32_t)s, (uint32_t)(s >> 32), c0, rc[r + 1].lo, rc[r + 1].hi);
    }
    return s;
}

// ---------------------------------------------------------------- one permutation per LANE, partial rounds merged
// The quad form spends 4 lanes on a permutation because one big commitment alone has too few leaves for the chip (32 768 for
// FinalExp: 512 waves).  With SEVERAL big commitments in flight that reason is gone, and the quad form's price shows: in the 22
// partial rounds all four lanes execute the single S-box (three multiplies per round and quad), so a permutation costs
// 4346 / 16 = 272 wave-instructions against 176 with the whole state in one lane (11.3 K issue slots per 64 permutations: full
// rounds with their circulant layer on the matrix pipe, partial rounds four at a time; the block sizes are in lane_round_asm.inc).
// Everything is uniform over the wave here -- round constants, the merged layers' coefficients -- so it comes from one LDS image by
// broadcast reads (a scalar-register formulation would need some 230 coefficients per merge in 100 SGPRs).
typedef const __attribute__((address_space(3))) gl_t* lds_gl_ptr;
struct LaneTables {
    RcPair rc[31][12];         // round constants in halves; rc[30] = 0 (the "next round" of the last one)
    RcPair kf[5][3];           // k1, k2, k3 of the merged fours (poseidon_merged.h)
    RcPair k4[5][12];
    uint32_t row[12][16];      // per output row of the dense layer: N4[r][0 .. 11], N3[r][0], N2[r][0], M[r][0], -
    uint32_t m0[12], n20[12];  // row 0 of M and of N2 (the first two intermediate dot products) ...
    uint32_t n30[16];          // ... and of N3, then N2[0][0] (the third)
    // The rounds whose circulant layer runs on the matrix pipe (full rounds 0 .. 3 and 26 .. 28, the plain partial rounds 24 and 25;
    // lane_round_asm.inc, tools/gen_lane_round_asm.py): per round, byte plane and LANE the fourth dword of the weight tile -- the constant
    // bytes that ride in the spare K-values (kernels_hash.hip: build_lane_tables)
    uint32_t rcb[9][8][64];
    gl_t rc0[12];              // the first round's constants as whole words (added to the state at the start of every permutation)
};

// ---- the lane form's rounds as scheduled asm blocks (tools/gen_lane_round_asm.py -> lane_round_asm.inc; the block sizes are in its
// header), LDS loads a row ahead with counted waits -- hipcc made 1337 slots of a full round and 1464 of three merged partial rounds, a
// third of them wait states (history section 5).  The state lives in three 8-register tuples bound to v[80:103]: the rate is the first
// two, the capacity the third.  Partial rounds 4 .. 23 run FOUR at a time (poseidon_merged.h), 24 and 25 as plain rounds.
#include "lane_round_asm.inc"
typedef uint32_t lane_u32x8 __attribute__((ext_vector_type(8)));
struct LaneState {
    lane_u32x8 t0, t1, t2;
};
struct LaneZeros {
    uint32_t za, zb;
};
__device__ __forceinline__ void lane_zeros_init(LaneZeros& Z) {
    Z.za = Z.zb = 0;
    asm volatile("" : "+v"(Z.za), "+v"(Z.zb));
}
#define STARKHIP_LANE_ROUND_BLOCK(NAME, TEXT)                                                                                              \
    __device__ __forceinline__ void NAME(LaneState& st, uint32_t seed_lds, const LaneZeros& Z) {                                           \
        asm(TEXT                                                                                                                           \
            : STARKHIP_LANE_STATE0(st.t0), STARKHIP_LANE_STATE1(st.t1), STARKHIP_LANE_STATE2(st.t2)                                        \
            : STARKHIP_LANE_A_SEED(seed_lds), STARKHIP_LANE_ZA(Z.za), STARKHIP_LANE_ZB(Z.zb)                                               \
            : STARKHIP_LANE_CLOBBERS);                                                                                                     \
    }
STARKHIP_LANE_ROUND_BLOCK(lane_full_round_asm, STARKHIP_LANE_FULL_ROUND_ASM)
STARKHIP_LANE_ROUND_BLOCK(lane_last_round_asm, STARKHIP_LANE_LAST_ROUND_ASM)
STARKHIP_LANE_ROUND_BLOCK(lane_partial_round_asm, STARKHIP_LANE_PARTIAL_ROUND_ASM)
#undef STARKHIP_LANE_ROUND_BLOCK
__device__ __forceinline__ void lane_four_asm(LaneState& st, uint32_t k4_lds, uint32_t kf_lds, uint32_t coef_lds, const LaneZeros& Z) {
    asm(STARKHIP_LANE_FOUR_ASM
        : STARKHIP_LANE_STATE0(st.t0), STARKHIP_LANE_STATE1(st.t1), STARKHIP_LANE_STATE2(st.t2)
        : STARKHIP_LANE_A_K3(k4_lds), STARKHIP_LANE_A_K12(kf_lds), STARKHIP_LANE_A_COEF(coef_lds), STARKHIP_LANE_ZA(Z.za), STARKHIP_LANE_ZB(Z.zb)
        : STARKHIP_LANE_CLOBBERS);
}
// The operands of the matrix-pipe rounds that live across the whole kernel: the weight tile of v_mfma_i32_32x32x32_i8 for THIS lane
// (two copies of its dwords 0 .. 2; dword 3 is loaded per plane inside the blocks) and the constant fourth dword of the state-side tuples.
struct LaneMfma {
    uint32_t aw[2][4];
    uint32_t bc[4];
};
// Weight tile: lane (row = lane & 31, half = lane >> 5) holds the 16 K-values of that row that meet the K-values of the lane half `half`
// of the other operand.  Output g = (row & 3) + 4 (row >> 3) of a permutation lands in result register g of its own lane when the row's
// bit 2 equals the lane half: those rows carry M[g][0 .. 11] (circulant, + 8 at [0][0]); every other (row, half) is zero -- a block
// diagonal tile that multiplies the two lane halves' states separately (tools/experiments/mfma_mds_probe.hip).
__device__ __forceinline__ void lane_mfma_init(LaneMfma& M, unsigned lane) {
    constexpr uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    const unsigned row = lane & 31u, half = lane >> 5, g = (row & 3u) + 4u * (row >> 3);
    const bool live = ((row >> 2) & 1u) == half && g < 12u;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        uint32_t w = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const unsigned j = 4 * d + i;
            uint32_t c = 0;
#pragma unroll
            for (unsigned gg = 0; gg < 12; gg++)  // (a table walk the compiler folds; g is per lane)
                if (gg == g) c = CIRC[(j + 12u - gg) % 12u] + ((gg == 0 && j == 0) ? 8u : 0u);
            w |= c << (8 * i);
        }
        M.aw[0][d] = M.aw[1][d] = live ? w : 0u;
    }
    M.aw[0][3] = M.aw[1][3] = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) M.bc[k] = STARKHIP_LANE_B_CONST;
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
        for (int d = 0; d < 4; d++) asm volatile("" : "+v"(M.aw[k][d]));
#pragma unroll
    for (int k = 0; k < 4; k++) asm volatile("" : "+v"(M.bc[k]));
}
#define STARKHIP_LANE_MFMA_BLOCK(NAME, TEXT)                                                                                               \
    __device__ __forceinline__ void NAME(LaneState& st, uint32_t rcb_lds, const LaneZeros& Z, LaneMfma& M) {                               \
        asm(TEXT                                                                                                                           \
            : STARKHIP_LANE_STATE0(st.t0), STARKHIP_LANE_STATE1(st.t1), STARKHIP_LANE_STATE2(st.t2), STARKHIP_LANE_AW03(M.aw[0][3]),       \
              STARKHIP_LANE_AW13(M.aw[1][3])                                                                                               \
            : STARKHIP_LANE_A_RCB(rcb_lds), STARKHIP_LANE_ZA(Z.za), STARKHIP_LANE_ZB(Z.zb), STARKHIP_LANE_AW00(M.aw[0][0]),                \
              STARKHIP_LANE_AW01(M.aw[0][1]), STARKHIP_LANE_AW02(M.aw[0][2]), STARKHIP_LANE_AW10(M.aw[1][0]), STARKHIP_LANE_AW11(M.aw[1][1]), \
              STARKHIP_LANE_AW12(M.aw[1][2]), STARKHIP_LANE_BC0(M.bc[0]), STARKHIP_LANE_BC1(M.bc[1]), STARKHIP_LANE_BC2(M.bc[2]),          \
              STARKHIP_LANE_BC3(M.bc[3]), STARKHIP_LANE_S_SEL_A(STARKHIP_LANE_SEL_A_VALUE), STARKHIP_LANE_S_SEL_B(STARKHIP_LANE_SEL_B_VALUE), \
              STARKHIP_LANE_S_SEL_C(STARKHIP_LANE_SEL_C_VALUE), STARKHIP_LANE_S_SEL_D(STARKHIP_LANE_SEL_D_VALUE),                          \
              STARKHIP_LANE_S_X80(0x80808080u), STARKHIP_LANE_S_K64K(65536u)                                                               \
            : STARKHIP_LANE_CLOBBERS, STARKHIP_LANE_MFMA_CLOBBERS);                                                                        \
    }
STARKHIP_LANE_MFMA_BLOCK(lane_full_round_mfma_asm, STARKHIP_LANE_FULL_ROUND_MFMA_ASM)
STARKHIP_LANE_MFMA_BLOCK(lane_partial_round_mfma_asm, STARKHIP_LANE_PARTIAL_ROUND_MFMA_ASM)
#undef STARKHIP_LANE_MFMA_BLOCK
__device__ __forceinline__ gl_t lane_get(const lane_u32x8& t, int i) { return (gl_t)t[2 * i] | ((gl_t)t[2 * i + 1] << 32); }
__device__ __forceinline__ void lane_set(lane_u32x8& t, int i, gl_t x) {
    t[2 * i] = (uint32_t)x;
    t[2 * i + 1] = (uint32_t)(x >> 32);
}
// One permutation.  CAP_ONLY: only the capacity (st.t2) of the result is computed -- the caller overwrites the rate.
// Rounds 0 .. 3, 24, 25 (the plain partial rounds) and 26 .. 28 run their circulant layer on the matrix pipe (844 and 268 slots against 980
// and 418); the merged fours (29-bit coefficients: four weight planes, no gain) and the capacity-only last round stay multiply-add chains.
// -DSTARKHIP_LANE_NO_MFMA: the round-3 form throughout (A/B measurements).
template <bool CAP_ONLY>
__device__ __forceinline__ void poseidon_permute_lane_asm(LaneState& st, const LaneTables* __restrict__ T, const LaneZeros& Z, LaneMfma& M, unsigned lane) {
    {
        // the first round's constants are read HERE, every permutation: hoisted out of the caller's loop they are 48 registers hipcc
        // spills to scratch and reloads one after the other (the pointer goes through an empty asm so that it cannot)
        // (... as an LDS pointer: through a generic one they are flat loads, which count on vmcnt too -- the wait for them would also wait
        // for the NEXT permutation's columns, requested from HBM a moment ago)
        uint32_t rc0_lds = (uint32_t)(uintptr_t)&T->rc0[0];
        asm volatile("" : "+v"(rc0_lds));
        const lds_gl_ptr rc0 = (lds_gl_ptr)(uintptr_t)rc0_lds;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            lane_set(st.t0, i, gl_add_nc(lane_get(st.t0, i), rc0[i]));
            lane_set(st.t1, i, gl_add_nc(lane_get(st.t1, i), rc0[4 + i]));
            lane_set(st.t2, i, gl_add_nc(lane_get(st.t2, i), rc0[8 + i]));
        }
    }
    const uint32_t rc_lds = (uint32_t)(uintptr_t)&T->rc[0][0], k4_lds = (uint32_t)(uintptr_t)&T->k4[0][0], kf_lds = (uint32_t)(uintptr_t)&T->kf[0][0],
                   coef_lds = (uint32_t)(uintptr_t)&T->row[0][0];
    constexpr uint32_t RC_ROW = 12 * sizeof(RcPair), KF_ROW = 3 * sizeof(RcPair);
#ifdef STARKHIP_LANE_NO_MFMA
    (void)M;
    (void)lane;
#pragma unroll 1
    for (uint32_t r = 0; r < 4; r++) lane_full_round_asm(st, rc_lds + (r + 1) * RC_ROW, Z);
#pragma unroll 1
    for (uint32_t t = 0; t < 5; t++) lane_four_asm(st, k4_lds + t * RC_ROW, kf_lds + t * KF_ROW, coef_lds, Z);
    lane_partial_round_asm(st, rc_lds + 25 * RC_ROW, Z);
    lane_partial_round_asm(st, rc_lds + 26 * RC_ROW, Z);
#pragma unroll 1
    for (uint32_t r = 26; r < 29; r++) lane_full_round_asm(st, rc_lds + (r + 1) * RC_ROW, Z);
#else
    const uint32_t rcb_lds = (uint32_t)(uintptr_t)&T->rcb[0][0][0] + lane * 4u;
    constexpr uint32_t RCB_ROUND = 8 * 64 * 4;
#pragma unroll 1
    for (uint32_t m = 0; m < 4; m++) lane_full_round_mfma_asm(st, rcb_lds + m * RCB_ROUND, Z, M);
#pragma unroll 1
    for (uint32_t t = 0; t < 5; t++) lane_four_asm(st, k4_lds + t * RC_ROW, kf_lds + t * KF_ROW, coef_lds, Z);
#pragma unroll 1
    for (uint32_t m = 4; m < 6; m++) lane_partial_round_mfma_asm(st, rcb_lds + m * RCB_ROUND, Z, M);
#pragma unroll 1
    for (uint32_t m = 6; m < 9; m++) lane_full_round_mfma_asm(st, rcb_lds + m * RCB_ROUND, Z, M);
#endif
    if (CAP_ONLY) lane_last_round_asm(st, rc_lds + 30 * RC_ROW, Z);
    else lane_full_round_asm(st, rc_lds + 30 * RC_ROW, Z);
}

// ---- the PAIR form (tools/gen_pair_round_asm.py -> pair_round_asm.inc): lanes l and l + 32 of a wave share one permutation -- the lower
// lane holds state elements 0 .. 5, the upper lane 6 .. 11 -- so a commitment of 32 768 leaves is 1 024 waves of 256 registers: one per SIMD,
// the form a LONE FinalExp-class commitment takes (the lane form's 512 waves would leave half the chip idle, the quad form costs 272 issue
// slots per permutation against 212 here).  Full rounds and the lone partial round: six S-boxes per lane, then the circulant layer on the
// matrix pipe with a DENSE weight tile and two byte planes per instruction (four v_mfma_i32_32x32x32_i8 per round; the operand maps are
// in the generator).  Partial rounds 4 .. 23 four at a time (poseidon_merged.h): the three dot products are partial sums over a lane's own
// elements added across the pair with v_permlane32_swap_b32; the dense layer reads the partner's six elements (one exchange per merge) and
// computes the lane's six outputs.
#ifdef STARKHIP_PAIR_INC   // experiment builds: another schedule of the same blocks (tools/experiments/pair_variants.sh)
#include STARKHIP_PAIR_INC
#else
#include "pair_round_asm.inc"
#endif
constexpr int PAIR_MFMA_ROUNDS = 10;   // full rounds 0 .. 3, the plain partial rounds 24 and 25, full rounds 26 .. 29
struct PairTables {
    gl_t rc0[2][6];                 // [half]: the first round's constants of the half's elements
    RcPair kf[5][2][3];             // [merged four][half]: k1, k2, k3 -- in the lower half only (the sums are added across the pair), zero in the upper
    RcPair k4[5][2][6];             // [merged four][half][local output]
    // per half, 480 bytes: rows 0 of M, N2 and N3 against the half's own six elements (8 dwords each; N2[0][0] in dword 6 of the third),
    // then per local output r (g = 6 half + r) sixteen dwords: N4[g][own six], N4[g][the partner's six, neighbours crossed], N3[g][0],
    // N2[g][0], M[g][0], 0
    uint32_t coef[2][120];
    uint32_t rcb[PAIR_MFMA_ROUNDS][4][64];   // per matrix-pipe round, instruction and LANE: dword 3 of the weight tile (the constants' bytes)
};
typedef uint32_t pair_u32x4 __attribute__((ext_vector_type(4)));
struct PairState {
    pair_u32x4 t0, t1, t2;          // the lane's six elements: (0, 1), (2, 3), (4, 5), low dword first
};
struct PairMfma {
    uint32_t aw[2][4];
    uint32_t bc[4];
};
// Weight tile: lane (row = lane & 31, khalf = lane >> 5) holds the 16 weights of `row` that meet the 16 K-values of lane half `khalf` of
// the state operand: K-value (dword d < 3, byte i) = byte plane p + (i >> 1) of the half's element 2 d + (i & 1).  Row -> result register
// i = (row & 3) + 4 (row >> 3) of the lane half (row >> 2) & 1: output g = 6 half + i % 6 against plane p + i / 6 (i < 12).
__device__ __forceinline__ void pair_mfma_init(PairMfma& M, unsigned lane) {
    constexpr uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    const unsigned row = lane & 31u, khalf = lane >> 5, i = (row & 3u) + 4u * (row >> 3), out_half = (row >> 2) & 1u;
    const bool live = i < 12u;
    const unsigned g = 6u * out_half + i % 6u, pp = i / 6u;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        uint32_t w = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const unsigned j = 6u * khalf + 2u * d + (b & 1u);
            uint32_t c = 0;
#pragma unroll
            for (unsigned gg = 0; gg < 12; gg++)
#pragma unroll
                for (unsigned jj = 0; jj < 12; jj++)
                    if (gg == g && jj == j) c = CIRC[(jj + 12u - gg) % 12u] + ((gg == 0 && jj == 0) ? 8u : 0u);
            if ((unsigned)(b >> 1) == pp) w |= c << (8 * b);
        }
        M.aw[0][d] = M.aw[1][d] = live ? w : 0u;
    }
    M.aw[0][3] = M.aw[1][3] = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) M.bc[k] = STARKHIP_LANE_B_CONST;
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
        for (int d = 0; d < 4; d++) asm volatile("" : "+v"(M.aw[k][d]));
#pragma unroll
    for (int k = 0; k < 4; k++) asm volatile("" : "+v"(M.bc[k]));
}
#define STARKHIP_PAIR_MFMA_BLOCK(NAME, TEXT)                                                                                               \
    __device__ __forceinline__ void NAME(PairState& st, uint32_t rcb_lds, const LaneZeros& Z, PairMfma& M, uint64_t mask_lo) {             \
        asm(TEXT                                                                                                                           \
            : STARKHIP_PAIR_STATE0(st.t0), STARKHIP_PAIR_STATE1(st.t1), STARKHIP_PAIR_STATE2(st.t2), STARKHIP_PAIR_AW03(M.aw[0][3]),       \
              STARKHIP_PAIR_AW13(M.aw[1][3])                                                                                               \
            : STARKHIP_PAIR_A_RCB(rcb_lds), STARKHIP_PAIR_ZA(Z.za), STARKHIP_PAIR_ZB(Z.zb), STARKHIP_PAIR_AW00(M.aw[0][0]),                \
              STARKHIP_PAIR_AW01(M.aw[0][1]), STARKHIP_PAIR_AW02(M.aw[0][2]), STARKHIP_PAIR_AW10(M.aw[1][0]), STARKHIP_PAIR_AW11(M.aw[1][1]), \
              STARKHIP_PAIR_AW12(M.aw[1][2]), STARKHIP_PAIR_BC0(M.bc[0]), STARKHIP_PAIR_BC1(M.bc[1]), STARKHIP_PAIR_BC2(M.bc[2]),          \
              STARKHIP_PAIR_BC3(M.bc[3]), STARKHIP_PAIR_S_SEL_A(STARKHIP_LANE_SEL_A_VALUE), STARKHIP_PAIR_S_SEL_B(STARKHIP_LANE_SEL_B_VALUE), \
              STARKHIP_PAIR_S_X80(0x80808080u), STARKHIP_PAIR_S_K64K(65536u), STARKHIP_PAIR_MASK_LO(mask_lo)                               \
            : STARKHIP_PAIR_CLOBBERS, STARKHIP_PAIR_MFMA_CLOBBERS);                                                                        \
    }
STARKHIP_PAIR_MFMA_BLOCK(pair_full_round_asm, STARKHIP_PAIR_FULL_ROUND_ASM)
STARKHIP_PAIR_MFMA_BLOCK(pair_last_round_asm, STARKHIP_PAIR_LAST_ROUND_ASM)
STARKHIP_PAIR_MFMA_BLOCK(pair_partial_round_asm, STARKHIP_PAIR_PARTIAL_ROUND_ASM)
#undef STARKHIP_PAIR_MFMA_BLOCK
__device__ __forceinline__ void pair_four_asm(PairState& st, uint32_t k4_lds, uint32_t kf_lds, uint32_t coef_lds, const LaneZeros& Z, uint64_t mask_lo) {
    asm(STARKHIP_PAIR_FOUR_ASM
        : STARKHIP_PAIR_STATE0(st.t0), STARKHIP_PAIR_STATE1(st.t1), STARKHIP_PAIR_STATE2(st.t2)
        : STARKHIP_PAIR_A_K3(k4_lds), STARKHIP_PAIR_A_K12(kf_lds), STARKHIP_PAIR_A_COEF(coef_lds), STARKHIP_PAIR_ZA(Z.za), STARKHIP_PAIR_ZB(Z.zb),
          STARKHIP_PAIR_MASK_LO(mask_lo)
        : STARKHIP_PAIR_CLOBBERS);
}
__device__ __forceinline__ gl_t pair_get(const pair_u32x4& t, int i) { return (gl_t)t[2 * i] | ((gl_t)t[2 * i + 1] << 32); }
__device__ __forceinline__ void pair_set(pair_u32x4& t, int i, gl_t x) {
    t[2 * i] = (uint32_t)x;
    t[2 * i + 1] = (uint32_t)(x >> 32);
}
// One permutation of the pair's state.  CAP_ONLY: of the result only each lane's elements 2 .. 5 are computed (the capacity is the upper
// lane's 2 .. 5; the caller overwrites the rest).  `half` = lane >> 5.
template <bool CAP_ONLY>
__device__ __forceinline__ void poseidon_permute_pair_asm(PairState& st, const PairTables* __restrict__ T, const LaneZeros& Z, PairMfma& M, unsigned lane,
                                                          uint64_t mask_lo) {
    const unsigned half = lane >> 5;
    {
        uint32_t rc0_lds = (uint32_t)(uintptr_t)&T->rc0[half][0];   // (an LDS pointer: see poseidon_permute_lane_asm)
        asm volatile("" : "+v"(rc0_lds));
        const lds_gl_ptr rc0 = (lds_gl_ptr)(uintptr_t)rc0_lds;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            pair_set(st.t0, i, gl_add_nc(pair_get(st.t0, i), rc0[i]));
            pair_set(st.t1, i, gl_add_nc(pair_get(st.t1, i), rc0[2 + i]));
            pair_set(st.t2, i, gl_add_nc(pair_get(st.t2, i), rc0[4 + i]));
        }
    }
    const uint32_t k4_lds = (uint32_t)(uintptr_t)&T->k4[0][half][0], kf_lds = (uint32_t)(uintptr_t)&T->kf[0][half][0],
                   coef_lds = (uint32_t)(uintptr_t)&T->coef[half][0], rcb_lds = (uint32_t)(uintptr_t)&T->rcb[0][0][0] + lane * 4u;
    constexpr uint32_t K4_FOUR = 2 * 6 * sizeof(RcPair), KF_FOUR = 2 * 3 * sizeof(RcPair), RCB_ROUND = 4 * 64 * 4;
#pragma unroll 1
    for (uint32_t m = 0; m < 4; m++) pair_full_round_asm(st, rcb_lds + m * RCB_ROUND, Z, M, mask_lo);
#pragma unroll 1
    for (uint32_t t = 0; t < 5; t++) pair_four_asm(st, k4_lds + t * K4_FOUR, kf_lds + t * KF_FOUR, coef_lds, Z, mask_lo);
#pragma unroll 1
    for (uint32_t m = 4; m < 6; m++) pair_partial_round_asm(st, rcb_lds + m * RCB_ROUND, Z, M, mask_lo);
#pragma unroll 1
    for (uint32_t m = 6; m < 9; m++) pair_full_round_asm(st, rcb_lds + m * RCB_ROUND, Z, M, mask_lo);
    if (CAP_ONLY) pair_last_round_asm(st, rcb_lds + 9 * RCB_ROUND, Z, M, mask_lo);
    else pair_full_round_asm(st, rcb_lds + 9 * RCB_ROUND, Z, M, mask_lo);
}

}  // namespace starkhip
