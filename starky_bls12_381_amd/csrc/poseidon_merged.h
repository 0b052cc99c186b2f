// Host-side tables for taking Poseidon's partial rounds three at a time (used by the leaf-hash kernel's quad and row forms,
// poseidon_dev.h, and by the AVX-512 host permutation, poseidon_host.cpp) and four at a time (the lane and pair forms; at the end).
//
// Only element 0 passes the S-box in a partial round.  With M the MDS matrix, Mz = M with row 0 zeroed, u the state at the
// start of partial round r (its constants added), x1 = u0^7, ut = (x1, u1 .. u11), and c1, c2, c3 the constants of rounds
// r+1, r+2, r+3 (c?z = with element 0 zeroed):
//     y1  = (M ut)[0] + k1                       k1 = c1[0]                  x2 = y1^7
//     y2  = (N2 ut)[0] + M[0][0] x2 + k2          k2 = (M c1z)[0] + c2[0]     x3 = y2^7      N2 = M Mz
//     out = N3 ut + N2[:,0] x2 + M[:,0] x3 + k3   k3 = N2 c1z + M c2z + c3                   N3 = M Mz Mz
// `out` is the state at the start of round r+3 (constants added).  The matrices are exact integers (entries < 2^21).
#pragma once
#include <stdint.h>

#include "poseidon.h"

namespace starkhip {

static const int POSEIDON_MERGED_TRIPLES = 7;  // partial rounds 0..20; the 22nd stays a plain round

struct PoseidonMergedTables {
    uint64_t M[12][12], N2[12][12], N3[12][12];
    gl_t k1[POSEIDON_MERGED_TRIPLES], k2[POSEIDON_MERGED_TRIPLES], k3[POSEIDON_MERGED_TRIPLES][12];
};

inline void build_poseidon_merged_tables(PoseidonMergedTables& T) {
    static const uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    uint64_t Mz[12][12];
    for (int i = 0; i < 12; i++)
        for (int j = 0; j < 12; j++) {
            T.M[i][j] = CIRC[(j - i + 12) % 12] + ((i == 0 && j == 0) ? 8 : 0);
            Mz[i][j] = i == 0 ? 0 : T.M[i][j];
        }
    auto mul = [](const uint64_t (&a)[12][12], const uint64_t (&b)[12][12], uint64_t (&o)[12][12]) {
        for (int i = 0; i < 12; i++)
            for (int j = 0; j < 12; j++) {
                uint64_t acc = 0;
                for (int k = 0; k < 12; k++) acc += a[i][k] * b[k][j];
                o[i][j] = acc;
            }
    };
    mul(T.M, Mz, T.N2);
    mul(T.N2, Mz, T.N3);
    auto matvec_mod = [](const uint64_t (&a)[12][12], const gl_t* v, gl_t* o) {
        for (int i = 0; i < 12; i++) {
            unsigned __int128 acc = 0;
            for (int j = 0; j < 12; j++) acc += (unsigned __int128)a[i][j] * v[j];
            o[i] = (gl_t)(acc % GL_P);
        }
    };
    const uint64_t* RC = POSEIDON_RC_HOST;
    for (int t = 0; t < POSEIDON_MERGED_TRIPLES; t++) {
        const int r = 4 + 3 * t;
        gl_t c1z[12], c2z[12], a[12], b[12];
        for (int i = 0; i < 12; i++) {
            c1z[i] = i ? RC[12 * (r + 1) + i] : 0;
            c2z[i] = i ? RC[12 * (r + 2) + i] : 0;
        }
        matvec_mod(T.M, c1z, a);
        T.k1[t] = RC[12 * (r + 1)];
        T.k2[t] = gl_add(a[0], RC[12 * (r + 2)]);
        matvec_mod(T.N2, c1z, a);
        matvec_mod(T.M, c2z, b);
        for (int i = 0; i < 12; i++) T.k3[t][i] = gl_add(gl_add(a[i], b[i]), RC[12 * (r + 3) + i]);
    }
}

// ---- FOUR partial rounds at once (the lane and pair forms of the leaf hash: tools/gen_lane_round_asm.py, gen_pair_round_asm.py).
// With N_k = M Mz^(k-1) and c1 .. c4 the constants of rounds r + 1 .. r + 4:
//     y1  = (M ut)[0] + k1                                          x2 = y1^7      k1 = c1[0]
//     y2  = (N2 ut)[0] + M[0][0] x2 + k2                             x3 = y2^7      k2 = (M c1z)[0] + c2[0]
//     y3  = (N3 ut)[0] + N2[0][0] x2 + M[0][0] x3 + k3               x4 = y3^7      k3 = (N2 c1z)[0] + (M c2z)[0] + c3[0]
//     out = N4 ut + N3[:,0] x2 + N2[:,0] x3 + M[:,0] x4 + k4                        k4 = N3 c1z + N2 c2z + M c3z + c4
// N4's entries are below 2^29; a row of it plus its three x-coefficients sums to less than 0.83 * 2^32, so sums of products with 32-bit
// halves stay below 2^64 (checked below).  Partial rounds 4 .. 23 are five such merges; rounds 24 and 25 stay plain rounds.
static const int POSEIDON_MERGED_FOURS = 5;

struct PoseidonMergedFours {
    uint64_t M[12][12], N2[12][12], N3[12][12], N4[12][12];
    gl_t k1[POSEIDON_MERGED_FOURS], k2[POSEIDON_MERGED_FOURS], k3[POSEIDON_MERGED_FOURS], k4[POSEIDON_MERGED_FOURS][12];
    bool sums_fit;   // every row of N4 with its x-coefficients, times 2^32 - 1, plus a 32-bit seed, is below 2^64
};

inline void build_poseidon_merged_fours(PoseidonMergedFours& T) {
    static const uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    uint64_t Mz[12][12];
    for (int i = 0; i < 12; i++)
        for (int j = 0; j < 12; j++) {
            T.M[i][j] = CIRC[(j - i + 12) % 12] + ((i == 0 && j == 0) ? 8 : 0);
            Mz[i][j] = i == 0 ? 0 : T.M[i][j];
        }
    auto mul = [](const uint64_t (&a)[12][12], const uint64_t (&b)[12][12], uint64_t (&o)[12][12]) {
        for (int i = 0; i < 12; i++)
            for (int j = 0; j < 12; j++) {
                uint64_t acc = 0;
                for (int k = 0; k < 12; k++) acc += a[i][k] * b[k][j];
                o[i][j] = acc;
            }
    };
    mul(T.M, Mz, T.N2);
    mul(T.N2, Mz, T.N3);
    mul(T.N3, Mz, T.N4);
    T.sums_fit = true;
    for (int g = 0; g < 12; g++) {
        unsigned __int128 tot = T.N3[g][0] + T.N2[g][0] + T.M[g][0];
        for (int j = 0; j < 12; j++) tot += T.N4[g][j];
        // the bound the FOLD needs, not only "the accumulators fit 64 bits": fold_big (tools/gen_lane_round_asm.py, gen_pair_round_asm.py)
        // adds B_hi + carry with the carry-out dropped, which is right for B < 2^64 - 2^32 (today's constants give 0.83 * 2^64)
        const unsigned __int128 limit = ((unsigned __int128)1 << 64) - ((unsigned __int128)1 << 32);
        if (tot * 0xFFFFFFFFull + 0xFFFFFFFFull >= limit) T.sums_fit = false;
    }
    auto matvec_mod = [](const uint64_t (&a)[12][12], const gl_t* v, gl_t* o) {
        for (int i = 0; i < 12; i++) {
            unsigned __int128 acc = 0;
            for (int j = 0; j < 12; j++) acc += (unsigned __int128)a[i][j] * v[j];
            o[i] = (gl_t)(acc % GL_P);
        }
    };
    const uint64_t* RC = POSEIDON_RC_HOST;
    for (int t = 0; t < POSEIDON_MERGED_FOURS; t++) {
        const int r = 4 + 4 * t;
        gl_t c1z[12], c2z[12], c3z[12], a[12], b[12], c[12];
        for (int i = 0; i < 12; i++) {
            c1z[i] = i ? RC[12 * (r + 1) + i] : 0;
            c2z[i] = i ? RC[12 * (r + 2) + i] : 0;
            c3z[i] = i ? RC[12 * (r + 3) + i] : 0;
        }
        matvec_mod(T.M, c1z, a);
        T.k1[t] = RC[12 * (r + 1)];
        T.k2[t] = gl_add(a[0], RC[12 * (r + 2)]);
        matvec_mod(T.N2, c1z, a);
        matvec_mod(T.M, c2z, b);
        T.k3[t] = gl_add(gl_add(a[0], b[0]), RC[12 * (r + 3)]);
        matvec_mod(T.N3, c1z, a);
        matvec_mod(T.N2, c2z, b);
        matvec_mod(T.M, c3z, c);
        for (int i = 0; i < 12; i++) T.k4[t][i] = gl_add(gl_add(gl_add(a[i], b[i]), c[i]), RC[12 * (r + 4) + i]);
    }
}

}  // namespace starkhip
