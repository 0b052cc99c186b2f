// Poseidon-Goldilocks permutation (width 12, rate 8, x^7, 4 + 22 + 4 rounds) for host and
// gfx950 device code; plus the sponge conventions plonky2 0.1.4 uses for Merkle trees and
// the Fiat-Shamir challenger.  The reference selects this hash via
// `type C = PoseidonGoldilocksConfig` at /root/reference/src/aggregate_proof.rs:236.
#pragma once
#include "gl.h"
#include "poseidon_consts.h"

namespace starkhip {

#define POSEIDON_WIDTH 12
#define POSEIDON_RATE 8

#if defined(__HIPCC__)
__constant__ const uint64_t POSEIDON_RC_DEV[POSEIDON_RC_COUNT] = POSEIDON_RC_TABLE;
#endif
static const uint64_t POSEIDON_RC_HOST[POSEIDON_RC_COUNT] = POSEIDON_RC_TABLE;

GL_HD gl_t poseidon_sbox(gl_t x) {
    gl_t x2 = gl_sqr(x);
    gl_t x4 = gl_sqr(x2);
    gl_t x3 = gl_mul(x2, x);
    return gl_mul(x3, x4);
}

// 128-bit accumulate of s * small, then one reduction.
struct acc128 {
    uint64_t lo, hi;
};
GL_HD void acc_mad(acc128& a, uint64_t s, uint32_t k) {
    // s * k < 2^70: split s into 32-bit halves
    uint64_t p0 = (s & 0xFFFFFFFFULL) * k;  // < 2^38
    uint64_t p1 = (s >> 32) * k;            // < 2^38, weight 2^32
    uint64_t add_lo = p0 + (p1 << 32);
    uint64_t carry = add_lo < p0;
    uint64_t nlo = a.lo + add_lo;
    carry += nlo < a.lo;
    a.lo = nlo;
    a.hi += (p1 >> 32) + carry;
}

// MDS layer: out[r] = sum_i s[(i + r) % 12] * CIRC[i] + (r == 0) * 8 * s[0]
GL_HD void poseidon_mds(gl_t* s) {
    const uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    // split into 32-bit halves: sums stay below 2^32 * 12 * 41 < 2^42, no carries needed
    // duplicated (lo[i + 12] == lo[i]) so the circulant index i + r needs no modulo (host compilers do not unroll this)
    uint64_t lo[24], hi[24];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        lo[i] = lo[i + 12] = s[i] & 0xFFFFFFFFULL;
        hi[i] = hi[i + 12] = s[i] >> 32;
    }
    gl_t out[12];
#pragma unroll
    for (int r = 0; r < 12; r++) {
        uint64_t al = 0, ah = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            al += lo[i + r] * CIRC[i];
            ah += hi[i + r] * CIRC[i];
        }
        if (r == 0) {
            al += lo[0] * 8;
            ah += hi[0] * 8;
        }
        // value = al + ah * 2^32, al, ah < 2^42  => 128-bit hi:lo
        uint64_t l = al + (ah << 32);
        uint64_t h = (ah >> 32) + (l < al ? 1 : 0);
        out[r] = gl_reduce128(h, l);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = out[i];
}

GL_HD void poseidon_permute(gl_t* s) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t* RC = POSEIDON_RC_DEV;
#else
    const uint64_t* RC = POSEIDON_RC_HOST;
#endif
    int rc = 0;
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = poseidon_sbox(gl_add(s[i], RC[rc + i]));
        rc += 12;
        poseidon_mds(s);
    }
    for (int r = 0; r < 22; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], RC[rc + i]);
        rc += 12;
        s[0] = poseidon_sbox(s[0]);
        poseidon_mds(s);
    }
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = poseidon_sbox(gl_add(s[i], RC[rc + i]));
        rc += 12;
        poseidon_mds(s);
    }
}

// two_to_one(a, b): state = [a, b, 0,0,0,0]; permute; first four lanes.
GL_HD void poseidon_two_to_one(const gl_t* a, const gl_t* b, gl_t* out) {
    gl_t s[12];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        s[i] = a[i];
        s[4 + i] = b[i];
        s[8 + i] = 0;
    }
    poseidon_permute(s);
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = s[i];
}

// hash_no_pad over a strided sequence (element i at in[i * stride]): overwrite-mode sponge.
GL_HD void poseidon_hash_no_pad(const gl_t* in, size_t len, size_t stride, gl_t* out) {
    gl_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    for (size_t off = 0; off < len; off += 8) {
        size_t k = len - off < 8 ? len - off : 8;
        for (size_t i = 0; i < k; i++) s[i] = in[(off + i) * stride];
        poseidon_permute(s);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) out[i] = s[i];
}

// hash_or_noop: leaves with <= 4 elements are copied (zero padded), longer ones hashed.
GL_HD void poseidon_hash_or_noop(const gl_t* in, size_t len, size_t stride, gl_t* out) {
    if (len <= 4) {
        for (size_t i = 0; i < 4; i++) out[i] = i < len ? in[i * stride] : 0;
    } else {
        poseidon_hash_no_pad(in, len, stride, out);
    }
}

// host permutation tuned for the challenger's long sequential absorbs (poseidon_host.cpp)
void poseidon_permute_host(gl_t* s);

// ---------------------------------------------------------------- Challenger (host only)
struct Challenger {
    gl_t state[12];
    gl_t in[8];
    int n_in;
    gl_t out[8];
    int n_out;
    Challenger() : n_in(0), n_out(0) {
        for (int i = 0; i < 12; i++) state[i] = 0;
    }
    void duplex() {
        for (int i = 0; i < n_in; i++) state[i] = in[i];
        n_in = 0;
#if defined(__HIP_DEVICE_COMPILE__)
        poseidon_permute(state);
#else
        poseidon_permute_host(state);
#endif
        for (int i = 0; i < 8; i++) out[i] = state[i];
        n_out = 8;
    }
    void observe(gl_t x) {
        n_out = 0;
        in[n_in++] = x;
        if (n_in == 8) duplex();
    }
    void observe_ext(gl2_t x) {
        observe(x.a0);
        observe(x.a1);
    }
    void observe_many(const gl_t* x, size_t n) {
        for (size_t i = 0; i < n; i++) observe(x[i]);
    }
    gl_t get() {
        if (n_in > 0 || n_out == 0) duplex();
        return out[--n_out];  // pops from the END of the squeeze buffer
    }
    gl2_t get_ext() {
        gl_t a = get();
        gl_t b = get();
        return gl2_make(a, b);
    }
};

}  // namespace starkhip
