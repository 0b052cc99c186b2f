// Host-side Poseidon permutation used by the Fiat-Shamir challenger and the verifier.  The challenger absorbs
// 2 * (2C + Q) field elements per proof (36.8 K sequential permutations for FinalExp, 48.7 K for MillerLoop), so this
// sits on a proof's critical path between the openings and the FRI combination (19 % of FinalExp's latency, 33 % of
// MillerLoop's with the plain loop at 2.08 us per permutation on the GPU box's EPYC 9575F, tools/host_perm_rate.py).
//
// AVX-512 form (used when the CPU has avx512f/dq/vl, otherwise the portable loop of poseidon.h): the 12 state words
// live in one 8-lane and one 4-lane vector.
//  * S-box: 64 x 64 -> 128 from four 32 x 32 vpmuludq per lane, Goldilocks reduction with mask compares; x^7 for all
//    lanes in the full rounds, scalar for element 0 in the partial rounds.
//  * MDS: the 32-bit halves are stored twice in a row (lo[24], hi[24]), so the circulant's input rotated by i is an
//    UNALIGNED LOAD at offset i -- no shuffles; 12 x (vpmuludq by a broadcast coefficient + add) per half.
//  * partial rounds four at a time (poseidon_merged.h): per merge three dot products with a small-integer row and ONE dense
//    12 x 12 layer instead of three circulant layers -- a shorter dependency chain per round, which is what a sequential
//    sponge is bound by (2.05 -> 1.86 us per permutation on the GPU box).
// An earlier auto-vectorised AVX2 variant of the plain loop was slower than scalar (2.74 us) and was dropped.
#include <immintrin.h>

#include "poseidon.h"
#include "poseidon_merged.h"

namespace starkhip {

namespace {

#define AVX512_TARGET __attribute__((target("avx512f,avx512dq,avx512vl")))

struct V12 {  // 12 x u64: elements 0..7 and 8..11
    __m512i a;
    __m256i b;
};

const uint64_t EPS = 0xFFFFFFFFull;

// ---- scalar, lazy: any 64-bit representative in and out (the partial rounds' S-boxes are one dependent chain)
inline gl_t reduce128_nc(uint64_t hi, uint64_t lo) {
    const uint64_t hi_hi = hi >> 32, hi_lo = hi & EPS;
    uint64_t t0 = lo - hi_hi;
    t0 -= (uint64_t)(0 - (uint64_t)(lo < hi_hi)) & EPS;
    const uint64_t t1 = (hi_lo << 32) - hi_lo;
    uint64_t r = t0 + t1;
    r += (uint64_t)(0 - (uint64_t)(r < t1)) & EPS;
    return r;
}
inline gl_t mul_nc(gl_t a, gl_t b) {
    const unsigned __int128 m = (unsigned __int128)a * b;
    return reduce128_nc((uint64_t)(m >> 64), (uint64_t)m);
}
inline gl_t sbox_nc(gl_t x) {
    const gl_t x2 = mul_nc(x, x), x4 = mul_nc(x2, x2), x3 = mul_nc(x2, x);
    return mul_nc(x3, x4);
}

// ---- 8 lanes
// Values between the layers of one permutation are ANY 64-bit representative of their class (lazy reduction): a product's
// operands may be arbitrary, and so may the addends of the sums below; the state is canonicalised once, when it leaves
// permute_avx512.  That removes a compare + masked subtract (4 cycles of latency) from every reduction of the sequential chain.
AVX512_TARGET inline __m512i reduce128_8(__m512i hi, __m512i lo) {  // any representative
    const __m512i eps = _mm512_set1_epi64((long long)EPS);
    const __m512i hh = _mm512_srli_epi64(hi, 32), hl = _mm512_and_si512(hi, eps);
    __m512i t0 = _mm512_sub_epi64(lo, hh);
    t0 = _mm512_mask_sub_epi64(t0, _mm512_cmplt_epu64_mask(lo, hh), t0, eps);  // borrowed: - eps
    const __m512i t1 = _mm512_sub_epi64(_mm512_slli_epi64(hl, 32), hl);          // hl * eps < p
    __m512i r = _mm512_add_epi64(t0, t1);
    return _mm512_mask_add_epi64(r, _mm512_cmplt_epu64_mask(r, t1), r, eps);      // wrapped: + eps (t1 < p, so no second wrap)
}
AVX512_TARGET inline __m512i canon_8(__m512i r) {
    const __m512i p = _mm512_set1_epi64((long long)GL_P);
    return _mm512_mask_sub_epi64(r, _mm512_cmpge_epu64_mask(r, p), r, p);
}
AVX512_TARGET inline __m512i mul_8(__m512i x, __m512i y) {
    const __m512i m32 = _mm512_set1_epi64((long long)EPS);
    const __m512i x1 = _mm512_srli_epi64(x, 32), y1 = _mm512_srli_epi64(y, 32);
    const __m512i p00 = _mm512_mul_epu32(x, y), p01 = _mm512_mul_epu32(x, y1), p10 = _mm512_mul_epu32(x1, y), p11 = _mm512_mul_epu32(x1, y1);
    const __m512i mid = _mm512_add_epi64(p01, _mm512_srli_epi64(p00, 32));                       // < 2^64
    const __m512i mid2 = _mm512_add_epi64(p10, _mm512_and_si512(mid, m32));                      // < 2^64
    const __m512i lo = _mm512_or_si512(_mm512_slli_epi64(mid2, 32), _mm512_and_si512(p00, m32));
    const __m512i hi = _mm512_add_epi64(p11, _mm512_add_epi64(_mm512_srli_epi64(mid, 32), _mm512_srli_epi64(mid2, 32)));
    return reduce128_8(hi, lo);
}
AVX512_TARGET inline __m512i add_8(__m512i x, __m512i y) {  // x any representative, y canonical (a constant); any representative out
    const __m512i eps = _mm512_set1_epi64((long long)EPS);
    const __m512i s = _mm512_add_epi64(x, y);
    return _mm512_mask_add_epi64(s, _mm512_cmplt_epu64_mask(s, x), s, eps);  // wrapped: s <= p - 2, + eps cannot wrap again
}
AVX512_TARGET inline __m512i sbox_8(__m512i x) {
    const __m512i x2 = mul_8(x, x), x4 = mul_8(x2, x2), x3 = mul_8(x2, x);
    return mul_8(x3, x4);
}

// ---- 4 lanes (same code on 256-bit vectors)
AVX512_TARGET inline __m256i reduce128_4(__m256i hi, __m256i lo) {
    const __m256i eps = _mm256_set1_epi64x((long long)EPS);
    const __m256i hh = _mm256_srli_epi64(hi, 32), hl = _mm256_and_si256(hi, eps);
    __m256i t0 = _mm256_sub_epi64(lo, hh);
    t0 = _mm256_mask_sub_epi64(t0, _mm256_cmplt_epu64_mask(lo, hh), t0, eps);
    const __m256i t1 = _mm256_sub_epi64(_mm256_slli_epi64(hl, 32), hl);
    __m256i r = _mm256_add_epi64(t0, t1);
    return _mm256_mask_add_epi64(r, _mm256_cmplt_epu64_mask(r, t1), r, eps);
}
AVX512_TARGET inline __m256i canon_4(__m256i r) {
    const __m256i p = _mm256_set1_epi64x((long long)GL_P);
    return _mm256_mask_sub_epi64(r, _mm256_cmpge_epu64_mask(r, p), r, p);
}
AVX512_TARGET inline __m256i mul_4(__m256i x, __m256i y) {
    const __m256i m32 = _mm256_set1_epi64x((long long)EPS);
    const __m256i x1 = _mm256_srli_epi64(x, 32), y1 = _mm256_srli_epi64(y, 32);
    const __m256i p00 = _mm256_mul_epu32(x, y), p01 = _mm256_mul_epu32(x, y1), p10 = _mm256_mul_epu32(x1, y), p11 = _mm256_mul_epu32(x1, y1);
    const __m256i mid = _mm256_add_epi64(p01, _mm256_srli_epi64(p00, 32));
    const __m256i mid2 = _mm256_add_epi64(p10, _mm256_and_si256(mid, m32));
    const __m256i lo = _mm256_or_si256(_mm256_slli_epi64(mid2, 32), _mm256_and_si256(p00, m32));
    const __m256i hi = _mm256_add_epi64(p11, _mm256_add_epi64(_mm256_srli_epi64(mid, 32), _mm256_srli_epi64(mid2, 32)));
    return reduce128_4(hi, lo);
}
AVX512_TARGET inline __m256i add_4(__m256i x, __m256i y) {
    const __m256i eps = _mm256_set1_epi64x((long long)EPS);
    const __m256i s = _mm256_add_epi64(x, y);
    return _mm256_mask_add_epi64(s, _mm256_cmplt_epu64_mask(s, x), s, eps);
}
AVX512_TARGET inline __m256i sbox_4(__m256i x) {
    const __m256i x2 = mul_4(x, x), x4 = mul_4(x2, x2), x3 = mul_4(x2, x);
    return mul_4(x3, x4);
}

#define MV_A(x) _mm512_load_si512((const void*)(x))
#define MV_B(x) _mm256_load_si256((const __m256i*)((x) + 8))

// MDS layer: out[r] = sum_i CIRC[i] * s[(i + r) % 12] + (r == 0) * 8 * s[0]
// With `partial` the vector's element 0 is zero and x0 is the value it stands for: its column of the matrix is added at
// the end, so the scalar S-box that produces x0 runs beside the vector part instead of in front of it.
template <bool partial>
AVX512_TARGET inline void mds_t(V12& s, gl_t x0) {
    static const uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    alignas(64) uint64_t lo[24], hi[24];
    const __m512i m32 = _mm512_set1_epi64((long long)EPS);
    const __m256i m32h = _mm256_set1_epi64x((long long)EPS);
    const __m512i alo = _mm512_and_si512(s.a, m32), ahi = _mm512_srli_epi64(s.a, 32);
    const __m256i blo = _mm256_and_si256(s.b, m32h), bhi = _mm256_srli_epi64(s.b, 32);
    _mm512_store_si512((void*)lo, alo);
    _mm256_store_si256((__m256i*)(lo + 8), blo);
    _mm512_storeu_si512((void*)(lo + 12), alo);
    _mm256_storeu_si256((__m256i*)(lo + 20), blo);
    _mm512_store_si512((void*)hi, ahi);
    _mm256_store_si256((__m256i*)(hi + 8), bhi);
    _mm512_storeu_si512((void*)(hi + 12), ahi);
    _mm256_storeu_si256((__m256i*)(hi + 20), bhi);
    __m512i La = _mm512_setzero_si512(), Ha = _mm512_setzero_si512();
    __m256i Lb = _mm256_setzero_si256(), Hb = _mm256_setzero_si256();
    for (int i = 0; i < 12; i++) {  // fully unrolled by the compiler
        const __m512i c = _mm512_set1_epi64((long long)CIRC[i]);
        const __m256i ch = _mm256_set1_epi64x((long long)CIRC[i]);
        La = _mm512_add_epi64(La, _mm512_mul_epu32(_mm512_loadu_si512((const void*)(lo + i)), c));      // outputs r = 0..7 see s[r + i]
        Ha = _mm512_add_epi64(Ha, _mm512_mul_epu32(_mm512_loadu_si512((const void*)(hi + i)), c));
        Lb = _mm256_add_epi64(Lb, _mm256_mul_epu32(_mm256_loadu_si256((const __m256i*)(lo + 8 + i)), ch));  // r = 8..11
        Hb = _mm256_add_epi64(Hb, _mm256_mul_epu32(_mm256_loadu_si256((const __m256i*)(hi + 8 + i)), ch));
    }
    if (partial) {
        // column 0: CIRC[(12 - r) % 12] for output r, + 8 on output 0
        static const long long C0A[8] = {17 + 8, 20, 34, 18, 39, 13, 13, 28};
        static const long long C0B[4] = {2, 16, 41, 15};
        const __m512i ca = _mm512_loadu_si512((const void*)C0A);
        const __m256i cb = _mm256_loadu_si256((const __m256i*)C0B);
        const long long xl = (long long)(x0 & EPS), xh = (long long)(x0 >> 32);
        La = _mm512_add_epi64(La, _mm512_mul_epu32(_mm512_set1_epi64(xl), ca));
        Ha = _mm512_add_epi64(Ha, _mm512_mul_epu32(_mm512_set1_epi64(xh), ca));
        Lb = _mm256_add_epi64(Lb, _mm256_mul_epu32(_mm256_set1_epi64x(xl), cb));
        Hb = _mm256_add_epi64(Hb, _mm256_mul_epu32(_mm256_set1_epi64x(xh), cb));
    } else {
        // + 8 * s[0] on output 0 only
        La = _mm512_mask_add_epi64(La, 1, La, _mm512_slli_epi64(alo, 3));
        Ha = _mm512_mask_add_epi64(Ha, 1, Ha, _mm512_slli_epi64(ahi, 3));
    }
    // value = L + H * 2^32 with L, H < 2^42
    {
        const __m512i l = _mm512_add_epi64(La, _mm512_slli_epi64(Ha, 32));
        const __m512i h = _mm512_mask_add_epi64(_mm512_srli_epi64(Ha, 32), _mm512_cmplt_epu64_mask(l, La), _mm512_srli_epi64(Ha, 32),
                                                _mm512_set1_epi64(1));
        s.a = reduce128_8(h, l);
    }
    {
        const __m256i l = _mm256_add_epi64(Lb, _mm256_slli_epi64(Hb, 32));
        const __m256i h = _mm256_mask_add_epi64(_mm256_srli_epi64(Hb, 32), _mm256_cmplt_epu64_mask(l, Lb), _mm256_srli_epi64(Hb, 32),
                                                _mm256_set1_epi64x(1));
        s.b = reduce128_4(h, l);
    }
}

// The same layer with the INPUTS broadcast from registers and the matrix COLUMNS as constants (out = sum_j column_j * s_j), as the
// merged triples' dense layer does: no store + unaligned reload of the state (a 64-byte load that straddles two stores is
// not forwarded), and vpmuludq takes the low halves by itself.
struct alignas(64) MdsColumns {
    uint64_t col[12][16];  // column j: rows 0..7 (zmm), rows 8..11 (ymm), 4 unused
};
const MdsColumns& mds_columns() {
    static const MdsColumns C = [] {
        static const uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
        MdsColumns c = {};
        for (int j = 0; j < 12; j++)
            for (int r = 0; r < 12; r++) c.col[j][r] = CIRC[(j - r + 12) % 12] + ((r == 0 && j == 0) ? 8 : 0);
        return c;
    }();
    return C;
}
AVX512_TARGET inline void mds_bcast(V12& s) {
    const MdsColumns& C = mds_columns();
    __m512i La = _mm512_setzero_si512(), Ha = _mm512_setzero_si512();
    __m256i Lb = _mm256_setzero_si256(), Hb = _mm256_setzero_si256();
    const __m512i b512 = _mm512_zextsi256_si512(s.b);
#define MDS_TERM(SRC, LANE, J)                                                                            \
    {                                                                                                     \
        const __m512i x = _mm512_permutexvar_epi64(_mm512_set1_epi64(LANE), SRC);                         \
        const __m512i xh = _mm512_srli_epi64(x, 32);                                                      \
        const __m512i ca = MV_A(C.col[J]);                                                                \
        const __m256i cb = MV_B(C.col[J]);                                                                \
        La = _mm512_add_epi64(La, _mm512_mul_epu32(x, ca));                                               \
        Ha = _mm512_add_epi64(Ha, _mm512_mul_epu32(xh, ca));                                              \
        Lb = _mm256_add_epi64(Lb, _mm256_mul_epu32(_mm512_castsi512_si256(x), cb));                       \
        Hb = _mm256_add_epi64(Hb, _mm256_mul_epu32(_mm512_castsi512_si256(xh), cb));                      \
    }
    MDS_TERM(s.a, 0, 0) MDS_TERM(s.a, 1, 1) MDS_TERM(s.a, 2, 2) MDS_TERM(s.a, 3, 3) MDS_TERM(s.a, 4, 4) MDS_TERM(s.a, 5, 5)
    MDS_TERM(s.a, 6, 6) MDS_TERM(s.a, 7, 7) MDS_TERM(b512, 0, 8) MDS_TERM(b512, 1, 9) MDS_TERM(b512, 2, 10) MDS_TERM(b512, 3, 11)
#undef MDS_TERM
    {
        const __m512i l = _mm512_add_epi64(La, _mm512_slli_epi64(Ha, 32));
        const __m512i h = _mm512_mask_add_epi64(_mm512_srli_epi64(Ha, 32), _mm512_cmplt_epu64_mask(l, La), _mm512_srli_epi64(Ha, 32),
                                                _mm512_set1_epi64(1));
        s.a = reduce128_8(h, l);
    }
    {
        const __m256i l = _mm256_add_epi64(Lb, _mm256_slli_epi64(Hb, 32));
        const __m256i h = _mm256_mask_add_epi64(_mm256_srli_epi64(Hb, 32), _mm256_cmplt_epu64_mask(l, Lb), _mm256_srli_epi64(Hb, 32),
                                                _mm256_set1_epi64x(1));
        s.b = reduce128_4(h, l);
    }
}

#ifdef STARKHIP_HOST_MDS_RELOAD
AVX512_TARGET inline void mds(V12& s) { mds_t<false>(s, 0); }
#else
AVX512_TARGET inline void mds(V12& s) { mds_bcast(s); }
#endif

// ---- partial rounds FOUR at a time (poseidon_merged.h; round 6 -- three at a time before).  The three intermediate element-0 values are
// dot products of the state with one small-integer row (a multiply per half-vector and a horizontal add), the state after the fourth
// round is ONE dense 12 x 12 layer: out = sum_j (column j of N4) * ut[j], twelve broadcast-multiply-adds per half like the circulant
// layer.  Shorter dependency chain per round (the challenger's 36.8 K permutations are one sequential chain) and three dense layers
// less per merge.  N4's entries are below 2^29 and a row of it with its x-coefficients sums to less than 0.83 * 2^32
// (PoseidonMergedFours::sums_fit), so every 64-bit lane sum of (32-bit half) x (entry) below stays under 2^64.
struct alignas(64) MergedVectors {   // each vector as 16 words: rows 0..7 (zmm), rows 8..11 (ymm), 4 unused
    uint64_t n4[12][16];  // column j of N4
    uint64_t r1[16], r2[16], r3[16], r4[16], b2[16], b3[16], b4[16], k4[POSEIDON_MERGED_FOURS][16];
    gl_t k1[POSEIDON_MERGED_FOURS], k2[POSEIDON_MERGED_FOURS], k3[POSEIDON_MERGED_FOURS];
    uint64_t m00;
};

const MergedVectors& merged_vectors() {
    static const MergedVectors V = [] {
        static PoseidonMergedFours P;
        build_poseidon_merged_fours(P);
        if (!P.sums_fit) abort();  // the constants are fixed: cannot happen
        MergedVectors v = {};
        for (int j = 0; j < 12; j++)
            for (int i = 0; i < 12; i++) v.n4[j][i] = P.N4[i][j];
        for (int i = 0; i < 12; i++) {
            v.r1[i] = P.M[0][i];
            v.r2[i] = P.N2[0][i];
            v.r3[i] = P.N3[0][i];
            v.r4[i] = P.N4[0][i];
            v.b2[i] = P.N3[i][0];
            v.b3[i] = P.N2[i][0];
            v.b4[i] = P.M[i][0];
        }
        for (int t = 0; t < POSEIDON_MERGED_FOURS; t++) {
            for (int i = 0; i < 12; i++) v.k4[t][i] = P.k4[t][i];
            v.k1[t] = P.k1[t];
            v.k2[t] = P.k2[t];
            v.k3[t] = P.k3[t];
        }
        v.m00 = P.M[0][0];
        return v;
    }();
    return V;
}

// (sum over the 12 lanes of lo * row) + (sum of hi * row) * 2^32 as an integer (each sum below 0.83 * 2^64, the value below 2^97)
AVX512_TARGET inline unsigned __int128 dot_row(__m512i alo, __m512i ahi, __m256i blo, __m256i bhi, __m512i ra, __m256i rb) {
    const uint64_t L = (uint64_t)_mm512_reduce_add_epi64(_mm512_mul_epu32(alo, ra)) +
                       (uint64_t)_mm512_reduce_add_epi64(_mm512_zextsi256_si512(_mm256_mul_epu32(blo, rb)));
    const uint64_t H = (uint64_t)_mm512_reduce_add_epi64(_mm512_mul_epu32(ahi, ra)) +
                       (uint64_t)_mm512_reduce_add_epi64(_mm512_zextsi256_si512(_mm256_mul_epu32(bhi, rb)));
    return (unsigned __int128)L + ((unsigned __int128)H << 32);
}

// s: the state at the start of a partial round (constants added); on return the state four rounds later (constants added).
// The four S-boxes are ONE dependent chain (x1 -> y1 -> x2 -> y2 -> x3 -> y3 -> x4), and a sequential sponge is bound by it: everything
// that does not depend on them -- the dot products and the dense layer over elements 1 .. 11 -- is computed from the state
// with element 0 zeroed, beside the chain; each link then costs a few scalar multiply-adds and a reduction
// (y1 = P1 + M00 x1 + k1, y2 = P2 + N2_00 x1 + M00 x2 + k2, y3 = P3 + N3_00 x1 + N2_00 x2 + M00 x3 + k3), and x1 .. x4 enter the dense layer
// as four last terms.
// e0: element 0 of the state as a scalar, in and out -- the next merge's S-box input is row 0 of the dense layer, formed here as
// one more scalar link (P4 + N4_00 x1 + N3_00 x2 + N2_00 x3 + M00 x4) instead of waiting for the vector layer, its fold and reduction and a
// move back to a general register: the chain from merge to merge never leaves the scalar unit.  Lane 0 of s.a is not read.
AVX512_TARGET inline void partial4(V12& s, const MergedVectors& V, int t, gl_t& e0) {
    const __m512i m32 = _mm512_set1_epi64((long long)EPS);
    const __m256i m32h = _mm256_set1_epi64x((long long)EPS);
    const gl_t x1 = sbox_nc(e0);  // the chain starts at once
    const __m512i uz = _mm512_maskz_mov_epi64(0xFE, s.a);  // element 0 zeroed
    const __m512i alo = _mm512_and_si512(uz, m32), ahi = _mm512_srli_epi64(uz, 32);
    const __m256i blo = _mm256_and_si256(s.b, m32h), bhi = _mm256_srli_epi64(s.b, 32);
    const unsigned __int128 P1 = dot_row(alo, ahi, blo, bhi, MV_A(V.r1), MV_B(V.r1)) + V.k1[t];
    const unsigned __int128 P2 = dot_row(alo, ahi, blo, bhi, MV_A(V.r2), MV_B(V.r2)) + V.k2[t];
    const unsigned __int128 P3 = dot_row(alo, ahi, blo, bhi, MV_A(V.r3), MV_B(V.r3)) + V.k3[t];
    const unsigned __int128 P4 = dot_row(alo, ahi, blo, bhi, MV_A(V.r4), MV_B(V.r4)) + V.k4[t][0];
    // the dense layer over elements 1 .. 11: out = N4 ut + N3[:,0] x2 + N2[:,0] x3 + M[:,0] x4 + k4, column 0 of N4 (times x1) added below
    alignas(64) uint64_t lo[12], hi[12];
    _mm512_store_si512((void*)lo, alo);
    _mm256_store_si256((__m256i*)(lo + 8), blo);
    _mm512_store_si512((void*)hi, ahi);
    _mm256_store_si256((__m256i*)(hi + 8), bhi);
    __m512i La = _mm512_setzero_si512(), Ha = _mm512_setzero_si512();
    __m256i Lb = _mm256_setzero_si256(), Hb = _mm256_setzero_si256();
    for (int j = 1; j < 12; j++) {
        const __m512i l8 = _mm512_set1_epi64((long long)lo[j]), h8 = _mm512_set1_epi64((long long)hi[j]);
        const __m256i l4 = _mm256_set1_epi64x((long long)lo[j]), h4 = _mm256_set1_epi64x((long long)hi[j]);
        const __m512i ca = MV_A(V.n4[j]);
        const __m256i cb = MV_B(V.n4[j]);
        La = _mm512_add_epi64(La, _mm512_mul_epu32(l8, ca));
        Ha = _mm512_add_epi64(Ha, _mm512_mul_epu32(h8, ca));
        Lb = _mm256_add_epi64(Lb, _mm256_mul_epu32(l4, cb));
        Hb = _mm256_add_epi64(Hb, _mm256_mul_epu32(h4, cb));
    }
#define ADD_TERM(x, col)                                                                          /* += (column) * x, x any 64-bit value */ \
    {                                                                                                                                    \
        const long long xl = (long long)((x) & EPS), xh = (long long)((x) >> 32);                                                        \
        La = _mm512_add_epi64(La, _mm512_mul_epu32(_mm512_set1_epi64(xl), MV_A(col)));                                                   \
        Ha = _mm512_add_epi64(Ha, _mm512_mul_epu32(_mm512_set1_epi64(xh), MV_A(col)));                                                   \
        Lb = _mm256_add_epi64(Lb, _mm256_mul_epu32(_mm256_set1_epi64x(xl), MV_B(col)));                                                  \
        Hb = _mm256_add_epi64(Hb, _mm256_mul_epu32(_mm256_set1_epi64x(xh), MV_B(col)));                                                  \
    }
    ADD_TERM(x1, V.n4[0]);
    // the chain: entries of M .. N4 are < 2^29, so every sum below is < 2^97 + 4 * 2^93: the high word stays far below 2^64
    const unsigned __int128 v1 = P1 + (unsigned __int128)V.r1[0] * x1;
    const gl_t x2 = sbox_nc(reduce128_nc((uint64_t)(v1 >> 64), (uint64_t)v1));
    ADD_TERM(x2, V.b2);
    const unsigned __int128 v2 = P2 + (unsigned __int128)V.r2[0] * x1 + (unsigned __int128)V.m00 * x2;
    const gl_t x3 = sbox_nc(reduce128_nc((uint64_t)(v2 >> 64), (uint64_t)v2));
    ADD_TERM(x3, V.b3);
    const unsigned __int128 v3 = P3 + (unsigned __int128)V.r3[0] * x1 + (unsigned __int128)V.r2[0] * x2 + (unsigned __int128)V.m00 * x3;
    const gl_t x4 = sbox_nc(reduce128_nc((uint64_t)(v3 >> 64), (uint64_t)v3));
    const unsigned __int128 v4 = P4 + (unsigned __int128)V.r4[0] * x1 + (unsigned __int128)V.r3[0] * x2 + (unsigned __int128)V.r2[0] * x3 +
                                 (unsigned __int128)V.m00 * x4;
    e0 = reduce128_nc((uint64_t)(v4 >> 64), (uint64_t)v4);
    ADD_TERM(x4, V.b4);
#undef ADD_TERM
    {  // value = L + H * 2^32 (L, H < 0.83 * 2^64), then + k4 (canonical)
        const __m512i l = _mm512_add_epi64(La, _mm512_slli_epi64(Ha, 32));
        const __m512i h = _mm512_mask_add_epi64(_mm512_srli_epi64(Ha, 32), _mm512_cmplt_epu64_mask(l, La), _mm512_srli_epi64(Ha, 32),
                                                _mm512_set1_epi64(1));
        s.a = add_8(reduce128_8(h, l), MV_A(V.k4[t]));
    }
    {
        const __m256i l = _mm256_add_epi64(Lb, _mm256_slli_epi64(Hb, 32));
        const __m256i h = _mm256_mask_add_epi64(_mm256_srli_epi64(Hb, 32), _mm256_cmplt_epu64_mask(l, Lb), _mm256_srli_epi64(Hb, 32),
                                                _mm256_set1_epi64x(1));
        s.b = add_4(reduce128_4(h, l), MV_B(V.k4[t]));
    }
}

AVX512_TARGET void permute_avx512(gl_t* st) {
    const uint64_t* RC = POSEIDON_RC_HOST;
    V12 s;
    s.a = _mm512_loadu_si512((const void*)st);
    s.b = _mm256_loadu_si256((const __m256i*)(st + 8));
    int rc = 0;
    for (int r = 0; r < 4; r++, rc += 12) {
        s.a = sbox_8(add_8(s.a, _mm512_loadu_si512((const void*)(RC + rc))));
        s.b = sbox_4(add_4(s.b, _mm256_loadu_si256((const __m256i*)(RC + rc + 8))));
        mds(s);
    }
    // partial rounds: five merges of four, then the 21st and 22nd alone
    const MergedVectors& V = merged_vectors();
    s.a = add_8(s.a, _mm512_loadu_si512((const void*)(RC + rc)));
    s.b = add_4(s.b, _mm256_loadu_si256((const __m256i*)(RC + rc + 8)));
    gl_t e0 = (gl_t)_mm_cvtsi128_si64(_mm512_castsi512_si128(s.a));
    for (int t = 0; t < POSEIDON_MERGED_FOURS; t++) partial4(s, V, t, e0);
    rc += 12 * 4 * POSEIDON_MERGED_FOURS;  // the constants of this round are already in s
    for (int single = 0; single < 2; single++) {
        if (single) {  // the first of the two left this round's constants to be added
            s.a = add_8(s.a, _mm512_loadu_si512((const void*)(RC + rc)));
            s.b = add_4(s.b, _mm256_loadu_si256((const __m256i*)(RC + rc + 8)));
            e0 = (gl_t)_mm_cvtsi128_si64(_mm512_castsi512_si128(s.a));
        }
        const gl_t x0 = sbox_nc(e0);  // element 0, scalar
        s.a = _mm512_maskz_mov_epi64(0xFE, s.a);  // the vector part does not wait for x0
        mds_t<true>(s, x0);
        rc += 12;
    }
    for (int r = 0; r < 4; r++, rc += 12) {
        s.a = sbox_8(add_8(s.a, _mm512_loadu_si512((const void*)(RC + rc))));
        s.b = sbox_4(add_4(s.b, _mm256_loadu_si256((const __m256i*)(RC + rc + 8))));
        mds(s);
    }
    _mm512_storeu_si512((void*)st, canon_8(s.a));
    _mm256_storeu_si256((__m256i*)(st + 8), canon_4(s.b));
}

bool have_avx512() {
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl");
    return ok;
}

}  // namespace

// Inputs must be canonical (they are: the challenger absorbs proof data and its own canonical state).
void poseidon_permute_host(gl_t* s) {
    if (have_avx512()) permute_avx512(s);
    else poseidon_permute(s);
}

// Host replay of the permutation with its partial rounds 4 .. 23 taken FOUR at a time from build_poseidon_merged_fours' tables, in the
// arithmetic the lane and pair forms of the leaf hash use (sums of products with the 32-bit halves of the state in 64-bit accumulators,
// folded mod p), against the plain permutation on `n` pseudo-random states and the all-(p - 1) state: a CPU-side check of the tables and
// of the claim that the accumulators fit (tests/test_field_hash_cpu.py).  Returns the number of mismatching states, -1 if a sum overflowed.
int merged_fours_selfcheck(unsigned n) {
    static PoseidonMergedFours T;
    build_poseidon_merged_fours(T);
    if (!T.sums_fit) return -1;
    const uint64_t* RC = POSEIDON_RC_HOST;
    bool overflow = false;
    auto acc2 = [&](const uint64_t* coef, unsigned cnt, const gl_t* v, gl_t seed) {  // seed + sum coef[j] v[j] mod p, through the two accumulators
        unsigned __int128 A = seed & EPS, B = seed >> 32;
        for (unsigned j = 0; j < cnt; j++) {
            A += (unsigned __int128)(v[j] & EPS) * coef[j];
            B += (unsigned __int128)(v[j] >> 32) * coef[j];
        }
        if ((A >> 64) || (B >> 64)) overflow = true;
        return (gl_t)((A + (B << 32)) % GL_P);
    };
    auto full = [&](gl_t* s, int next_round) {  // S-box layer, MDS layer, the next round's constants
        gl_t t[12];
        for (int i = 0; i < 12; i++) t[i] = sbox_nc(s[i]);
        for (int g = 0; g < 12; g++) s[g] = acc2(T.M[g], 12, t, next_round < 30 ? RC[12 * next_round + g] : 0);
    };
    auto partial = [&](gl_t* s, int next_round) {
        s[0] = sbox_nc(s[0]);
        gl_t t[12];
        for (int i = 0; i < 12; i++) t[i] = s[i];
        for (int g = 0; g < 12; g++) s[g] = acc2(T.M[g], 12, t, RC[12 * next_round + g]);
    };
    int bad = 0;
    uint64_t seed = 0x9E3779B97F4A7C15ull;
    for (unsigned it = 0; it <= n; it++) {
        gl_t s[12], want[12];
        for (int i = 0; i < 12; i++) {
            seed = seed * 6364136223846793005ull + 1442695040888963407ull;
            s[i] = it == n ? GL_P - 1 : (seed ^ (seed >> 29)) % GL_P;
            want[i] = s[i];
        }
        poseidon_permute(want);
        for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], RC[i]);
        for (int r = 0; r < 4; r++) full(s, r + 1);
        for (int t = 0; t < POSEIDON_MERGED_FOURS; t++) {
            gl_t u[15];  // ut, x2, x3, x4
            u[0] = sbox_nc(s[0]);
            for (int i = 1; i < 12; i++) u[i] = s[i];
            uint64_t row[15];
            const gl_t y1 = acc2(T.M[0], 12, u, T.k1[t]);
            u[12] = sbox_nc(y1);
            for (int j = 0; j < 12; j++) row[j] = T.N2[0][j];
            row[12] = T.M[0][0];
            const gl_t y2 = acc2(row, 13, u, T.k2[t]);
            u[13] = sbox_nc(y2);
            for (int j = 0; j < 12; j++) row[j] = T.N3[0][j];
            row[12] = T.N2[0][0];
            row[13] = T.M[0][0];
            const gl_t y3 = acc2(row, 14, u, T.k3[t]);
            u[14] = sbox_nc(y3);
            for (int g = 0; g < 12; g++) {
                for (int j = 0; j < 12; j++) row[j] = T.N4[g][j];
                row[12] = T.N3[g][0];
                row[13] = T.N2[g][0];
                row[14] = T.M[g][0];
                s[g] = acc2(row, 15, u, T.k4[t][g]);
            }
        }
        partial(s, 25);
        partial(s, 26);
        for (int r = 26; r < 30; r++) full(s, r + 1);
        for (int i = 0; i < 12; i++)
            if (s[i] % GL_P != want[i]) {
                bad++;
                break;
            }
    }
    return overflow ? -1 : bad;
}

}  // namespace starkhip
