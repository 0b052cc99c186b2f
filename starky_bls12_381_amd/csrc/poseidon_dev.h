// gfx950 device-side Poseidon-Goldilocks, tuned for the VALU mix of CDNA4:
//  * the MDS layer (30 per permutation, 144 small-constant MACs each) is done on 22-bit limbs with
//    v_mad_u32_u24 (full rate) instead of 64-bit multiplies (quarter rate): three limb sums per output stay
//    below 2^31 (12 * 41 * 2^22), one cheap reduction per output;
//  * "quad" form: one permutation spread over the 4 lanes of a DPP quad (lane l owns state elements
//    l, l+4, l+8), so a Merkle-leaf kernel can field 4x as many lanes as there are leaves -- the trace
//    commitment has only N = 32768 leaves of 9191 sequential permutations each, far too few lanes for
//    256 CUs with one lane per leaf.  The other lanes' limbs arrive through v_mov_b32 quad_perm broadcasts.
// Semantics are exactly those of poseidon.h (same permutation); tests compare both against the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>

#include "gl_dev.h"
#include "poseidon.h"

namespace starkhip {

__device__ __forceinline__ void limbs22(gl_t x, uint32_t& a0, uint32_t& a1, uint32_t& a2) {
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    a0 = lo & 0x3FFFFFu;
    a1 = __builtin_amdgcn_alignbit(hi, lo, 22) & 0x3FFFFFu;
    a2 = hi >> 12;
}

// S0 + S1 * 2^22 + S2 * 2^44  (each S < 2^31)  mod p, canonical
__device__ __forceinline__ gl_t combine22(uint32_t S0, uint32_t S1, uint32_t S2) {
    uint64_t t = (uint64_t)S0 + ((uint64_t)S1 << 22);           // < 2^54
    uint64_t u = (uint64_t)(S2 & 0xFFFFFu) << 44;               // low 64 bits of S2 << 44
    uint64_t hi = S2 >> 20;                                       // < 2^11
    uint64_t lo = t + u;
    hi += lo < t;
    // value = hi * 2^64 + lo,  2^64 = eps (mod p), hi * eps < 2^44
    uint64_t r = lo + ((hi << 32) - hi);
    if (r < lo) r += GL_EPS;
    if (r >= GL_P) r -= GL_P;
    return r;
}

__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) { return __umul24(a, b) + c; }

// ---------------------------------------------------------------- one permutation per lane
__device__ __forceinline__ void poseidon_mds_dev(gl_t* s) {
    const uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    uint32_t a0[12], a1[12], a2[12];
#pragma unroll
    for (int i = 0; i < 12; i++) limbs22(s[i], a0[i], a1[i], a2[i]);
#pragma unroll
    for (int r = 0; r < 12; r++) {
        uint32_t S0 = 0, S1 = 0, S2 = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const int j = (i + r) % 12;
            const uint32_t k = CIRC[i] + ((r == 0 && i == 0) ? 8u : 0u);
            S0 = mad24(a0[j], k, S0);
            S1 = mad24(a1[j], k, S1);
            S2 = mad24(a2[j], k, S2);
        }
        s[r] = combine22(S0, S1, S2);
    }
}

__device__ __forceinline__ void poseidon_permute_dev(gl_t* s) {
    const uint64_t* RC = POSEIDON_RC_DEV;
    int rc = 0;
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = poseidon_sbox(gl_add(s[i], RC[rc + i]));
        rc += 12;
        poseidon_mds_dev(s);
    }
#pragma unroll 1
    for (int r = 0; r < 22; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], RC[rc + i]);
        rc += 12;
        s[0] = poseidon_sbox(s[0]);
        poseidon_mds_dev(s);
    }
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = poseidon_sbox(gl_add(s[i], RC[rc + i]));
        rc += 12;
        poseidon_mds_dev(s);
    }
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
// Lane l (= lane id & 3) owns state elements l, l + 4, l + 8 in (s0, s1, s2).
// Inside the quad permutation values are kept as ANY 64-bit representative of their residue (no canonical
// "subtract p" after each operation): limbs22 / the 128-bit product treat the register as a plain integer,
// so every step stays exact mod p; the caller canonicalises what leaves the permutation (gl_canon).
struct QuadConsts {
    uint32_t k[3][12];  // k[m][e] = MDS coefficient of state element e in output row l + 4 m
};

__device__ __forceinline__ gl_t sbox_nc(gl_t x) {
    const gl_t x2 = gl_mul_nc(x, x), x4 = gl_mul_nc(x2, x2), x3 = gl_mul_nc(x2, x);
    return gl_mul_nc(x3, x4);
}
// S0 + S1 * 2^22 + S2 * 2^44 + c  (S < 2^31, c < 2^64) mod p, any representative
__device__ __forceinline__ gl_t combine22_add_nc(uint32_t S0, uint32_t S1, uint32_t S2, gl_t c) {
    const uint64_t t = (uint64_t)S0 + ((uint64_t)S1 << 22);
    const uint64_t u = (uint64_t)(S2 & 0xFFFFFu) << 44;
    uint64_t hi = S2 >> 20;
    uint64_t lo = t + u;
    hi += lo < t;
    const uint64_t lo2 = lo + c;
    hi += lo2 < lo;
    uint64_t r = lo2 + ((hi << 32) - hi);  // hi <= 2^11 + 1
    if (r < lo2) r += GL_EPS;
    return r;
}

template <int J>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    // every lane of the quad reads lane J's value
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, J * 0x55, 0xF, 0xF, true);
}

__device__ __forceinline__ void quad_consts_init(QuadConsts& q, unsigned l) {
    const uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
        for (int e = 0; e < 12; e++) {
            // out[r] = sum_i s[(i + r) % 12] * CIRC[i]  =>  coefficient of s[e] in row r is CIRC[(e - r) mod 12]
            uint32_t c = 0;
#pragma unroll
            for (int ll = 0; ll < 4; ll++) {
                const int r = ll + 4 * m;
                const uint32_t v = CIRC[(e - r + 24) % 12] + ((r == 0 && e == 0) ? 8u : 0u);
                c = (l == (unsigned)ll) ? v : c;
            }
            // keep the 36 coefficients resident in VGPRs (do not rematerialise them from the lane id every round)
            asm volatile("" : "+v"(c));
            q.k[m][e] = c;
        }
}

// MDS layer; also adds this lane's three round constants of the NEXT round (c0, c1, c2) before reducing
__device__ __forceinline__ void poseidon_mds_quad(gl_t& s0, gl_t& s1, gl_t& s2, const QuadConsts& q, gl_t c0, gl_t c1, gl_t c2) {
    uint32_t own[3][3];
    limbs22(s0, own[0][0], own[0][1], own[0][2]);
    limbs22(s1, own[1][0], own[1][1], own[1][2]);
    limbs22(s2, own[2][0], own[2][1], own[2][2]);
    uint32_t S[3][3];
#pragma unroll
    for (int m = 0; m < 3; m++) S[m][0] = S[m][1] = S[m][2] = 0;
#pragma unroll
    for (int slot = 0; slot < 3; slot++) {
#pragma unroll
        for (int limb = 0; limb < 3; limb++) {
            const uint32_t v = own[slot][limb];
            const uint32_t e0 = quad_bcast<0>(v), e1 = quad_bcast<1>(v), e2 = quad_bcast<2>(v), e3 = quad_bcast<3>(v);
#pragma unroll
            for (int m = 0; m < 3; m++) {
                uint32_t acc = S[m][limb];
                acc = mad24(e0, q.k[m][4 * slot + 0], acc);
                acc = mad24(e1, q.k[m][4 * slot + 1], acc);
                acc = mad24(e2, q.k[m][4 * slot + 2], acc);
                acc = mad24(e3, q.k[m][4 * slot + 3], acc);
                S[m][limb] = acc;
            }
        }
    }
    s0 = combine22_add_nc(S[0][0], S[0][1], S[0][2], c0);
    s1 = combine22_add_nc(S[1][0], S[1][1], S[1][2], c1);
    s2 = combine22_add_nc(S[2][0], S[2][1], S[2][2], c2);
}

// rc: this lane's view of the round constants, rc[r * 3 + m] = RC[12 r + l + 4 m], r < 30, followed by three zeros.
// In: canonical or not; out: any representative (canonicalise with gl_canon before it leaves the kernel).
__device__ __forceinline__ void poseidon_permute_quad(gl_t& s0, gl_t& s1, gl_t& s2, const QuadConsts& q, const gl_t* __restrict__ rc, bool lane0) {
    s0 = gl_add_nc(s0, rc[0]);
    s1 = gl_add_nc(s1, rc[1]);
    s2 = gl_add_nc(s2, rc[2]);
    int r = 0;
#pragma unroll 1
    for (; r < 4; r++) {
        s0 = sbox_nc(s0);
        s1 = sbox_nc(s1);
        s2 = sbox_nc(s2);
        poseidon_mds_quad(s0, s1, s2, q, rc[3 * r + 3], rc[3 * r + 4], rc[3 * r + 5]);
    }
#pragma unroll 1
    for (; r < 26; r++) {
        const gl_t t = sbox_nc(s0);
        s0 = lane0 ? t : s0;
        poseidon_mds_quad(s0, s1, s2, q, rc[3 * r + 3], rc[3 * r + 4], rc[3 * r + 5]);
    }
#pragma unroll 1
    for (; r < 30; r++) {
        s0 = sbox_nc(s0);
        s1 = sbox_nc(s1);
        s2 = sbox_nc(s2);
        poseidon_mds_quad(s0, s1, s2, q, rc[3 * r + 3], rc[3 * r + 4], rc[3 * r + 5]);
    }
}

}  // namespace starkhip
