// Poseidon-Goldilocks hashing kernels for gfx950: Merkle leaf digests over the coset-major LDE,
// 2-to-1 compression levels, proof-of-work grinding and a raw permutation batch for tests.
// Restates plonky2 MerkleTree::new / hash_or_noop / two_to_one, which the
// reference reaches through prove() at /root/reference/src/aggregate_proof.rs:59.
#include <hip/hip_runtime.h>

#include <string.h>

#include <mutex>

#include "kernels.h"
#include "poseidon_merged.h"
#include "poseidon_dev.h"

namespace starkhip {

// ---- per-lane tables of the quad permutation (poseidon_dev.h), built on the host once per device
struct QuadMergedTables {
    uint32_t coef[4][64];  // per lane: n3[3][12], n1[3], n2[3], m00 (lane 0 only), b2[3], b3[3], pad to 50, cf[12] at 50, pad
    RcPair tk[2 * QUAD_MERGED_TRIPLES];       // k1, k2 per triple
    RcPair tk3[4][3 * QUAD_MERGED_TRIPLES];   // per lane: k3[mo] per triple
};
__constant__ QuadMergedTables QUAD_MERGED;

// Lane l owns state elements l, l + 4, l + 8 (slots 0, 1, 2); its rotated operand (r, m) is element ((l + r) & 3) + 4 m.
static inline int quad_elem(int l, int m) { return l + 4 * m; }
static inline int quad_col(int l, int r, int m) { return ((l + r) & 3) + 4 * m; }

// The per-lane views of poseidon_merged.h's tables and of the circulant MDS matrix.
static void build_quad_merged_tables(QuadMergedTables& T) {
    static PoseidonMergedTables P;
    build_poseidon_merged_tables(P);
    static const uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    auto split = [](gl_t v) { return RcPair{v & 0xFFFFFFFFull, v >> 32}; };
    for (int l = 0; l < 4; l++) {
        uint32_t* c = T.coef[l];
        for (int r = 0; r < 4; r++)
            for (int m = 0; m < 3; m++) {
                const int col = quad_col(l, r, m);
                for (int mo = 0; mo < 3; mo++) c[12 * mo + 3 * r + m] = (uint32_t)P.N3[quad_elem(l, mo)][col];
            }
        for (int m = 0; m < 3; m++) {
            c[36 + m] = (uint32_t)P.M[0][quad_elem(l, m)];   // the lane's own columns of row 0
            c[39 + m] = (uint32_t)P.N2[0][quad_elem(l, m)];
        }
        c[42] = l == 0 ? (uint32_t)P.M[0][0] : 0;
        for (int mo = 0; mo < 3; mo++) {
            c[43 + mo] = (uint32_t)P.N2[quad_elem(l, mo)][0];
            c[46 + mo] = (uint32_t)P.M[quad_elem(l, mo)][0];
        }
        c[49] = 0;
        // cf[3 r + d]: coefficient of the operand (r, m') for the output slot m with (m' - m) mod 3 = d:
        // CIRC[(col - out) mod 12] with col - out = ((l + r) & 3) - l + 4 d
        for (int r = 0; r < 4; r++)
            for (int d = 0; d < 3; d++) c[50 + 3 * r + d] = CIRC[((((l + r) & 3) - l + 4 * d) % 12 + 12) % 12];
        c[62] = c[63] = 0;
    }
    // every lane seeds its partial sum of y1 / y2 with a quarter of the constant (4^-1 = (3p + 1) / 4 mod p)
    const gl_t quarter = (gl_t)((((unsigned __int128)3 * GL_P) + 1) / 4);
    for (int t = 0; t < QUAD_MERGED_TRIPLES; t++) {
        T.tk[2 * t] = split(gl_mul(P.k1[t], quarter));
        T.tk[2 * t + 1] = split(gl_mul(P.k2[t], quarter));
        for (int l = 0; l < 4; l++)
            for (int mo = 0; mo < 3; mo++) T.tk3[l][3 * t + mo] = split(P.k3[t][quad_elem(l, mo)]);
    }
}

// Host replay of the quad formulation with exactly the tables the kernel gets (per-lane coefficient views included), against
// the plain host permutation: a CPU-side check of build_quad_merged_tables (tests/test_field_hash_cpu.py).  Returns the
// number of mismatching states out of `n`.
int quad_merged_tables_selfcheck(unsigned n) {
    static QuadMergedTables T;
    build_quad_merged_tables(T);
    auto join = [](const RcPair& c) { return (gl_t)(c.lo | (c.hi << 32)); };
    // the plain layer as the kernel computes it: per lane, twelve coefficients by rotation and slot difference
    auto mds_lanes = [&](gl_t* s) {
        gl_t out[12];
        for (int l = 0; l < 4; l++)
            for (int mo = 0; mo < 3; mo++) {
                gl_t acc = 0;
                for (int r = 0; r < 4; r++)
                    for (int m = 0; m < 3; m++) acc = gl_add(acc, gl_mul(s[quad_col(l, r, m)], T.coef[l][50 + 3 * r + (m - mo + 3) % 3]));
                if (l == 0 && mo == 0) acc = gl_add(acc, gl_mul(s[0], 8));
                out[quad_elem(l, mo)] = acc;
            }
        for (int i = 0; i < 12; i++) s[i] = out[i];
    };
    int bad = 0;
    uint64_t seed = 0x9E3779B97F4A7C15ull;
    for (unsigned it = 0; it < n; it++) {
        gl_t s[12], want[12];
        for (int i = 0; i < 12; i++) {
            seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17;
            s[i] = it == 0 ? 0 : it == 1 ? GL_P - 1 : seed % GL_P;
            want[i] = s[i];
        }
        poseidon_permute(want);
        const uint64_t* RC = POSEIDON_RC_HOST;
        int r = 0;
        for (; r < 4; r++) {
            for (int i = 0; i < 12; i++) s[i] = poseidon_sbox(gl_add(s[i], RC[12 * r + i]));
            mds_lanes(s);
        }
        for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], RC[12 * r + i]);
        for (int t = 0; t < QUAD_MERGED_TRIPLES; t++, r += 3) {
            gl_t u[12];
            for (int i = 0; i < 12; i++) u[i] = s[i];
            u[0] = poseidon_sbox(u[0]);
            // y1, y2: per-lane partial sums with the quartered constants, exactly as the kernel adds them up
            gl_t y1 = 0, y2p = 0;
            for (int l = 0; l < 4; l++) {
                const uint32_t* c = T.coef[l];
                gl_t a = join(T.tk[2 * t]), b = join(T.tk[2 * t + 1]);
                for (int m = 0; m < 3; m++) {
                    a = gl_add(a, gl_mul(u[quad_elem(l, m)], c[36 + m]));
                    b = gl_add(b, gl_mul(u[quad_elem(l, m)], c[39 + m]));
                }
                y1 = gl_add(y1, a);
                y2p = gl_add(y2p, b);
            }
            const gl_t x2 = poseidon_sbox(y1);
            gl_t y2 = y2p;
            for (int l = 0; l < 4; l++) y2 = gl_add(y2, gl_mul(x2, T.coef[l][42]));
            const gl_t x3 = poseidon_sbox(y2);
            for (int l = 0; l < 4; l++) {
                const uint32_t* c = T.coef[l];
                for (int mo = 0; mo < 3; mo++) {
                    gl_t acc = join(T.tk3[l][3 * t + mo]);
                    for (int rr = 0; rr < 4; rr++)
                        for (int m = 0; m < 3; m++) acc = gl_add(acc, gl_mul(u[quad_col(l, rr, m)], c[12 * mo + 3 * rr + m]));
                    acc = gl_add(acc, gl_mul(x2, c[43 + mo]));
                    acc = gl_add(acc, gl_mul(x3, c[46 + mo]));
                    s[quad_elem(l, mo)] = acc;
                }
            }
        }
        s[0] = poseidon_sbox(s[0]);  // round 25, plain
        mds_lanes(s);
        r++;
        for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], RC[12 * r + i]);
        for (; r < 30; r++) {
            for (int i = 0; i < 12; i++) s[i] = poseidon_sbox(s[i]);
            mds_lanes(s);
            if (r + 1 < 30)
                for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], RC[12 * (r + 1) + i]);
        }
        for (int i = 0; i < 12; i++)
            if (s[i] != want[i]) {
                bad++;
                break;
            }
    }
    return bad;
}

// ---- the row form's merged triples: per-lane coefficient rows and constants (poseidon_dev.h: RowMergedTables)
__constant__ RowMergedTables ROW_MERGED;
static void build_row_merged_tables(RowMergedTables& T) {
    static PoseidonMergedTables P;
    build_poseidon_merged_tables(P);
    auto split = [](gl_t v) { return RcPair{v & 0xFFFFFFFFull, v >> 32}; };
    for (int e = 0; e < 16; e++) {
        uint32_t* c = T.coef[e];
        for (int k = 0; k < 20; k++) c[k] = 0;
        if (e >= 12) continue;
        for (int k = 0; k < 12; k++) c[k] = (uint32_t)P.N3[e][(e + k) % 12];
        c[12] = (uint32_t)P.M[0][e];
        c[13] = (uint32_t)P.N2[0][e];
        c[14] = (uint32_t)P.N2[e][0];
        c[15] = (uint32_t)P.M[e][0];
        c[16] = (uint32_t)P.N3[e][0];
        c[17] = e == 0 ? (uint32_t)P.M[0][0] : 0;
        c[18] = e == 0 ? (uint32_t)P.N2[0][0] : 0;
    }
    for (int t = 0; t < POSEIDON_MERGED_TRIPLES; t++) {
        T.k1[t] = split(P.k1[t]);
        T.k2[t] = split(P.k2[t]);
        for (int e = 0; e < 12; e++) T.k3[t][e] = split(P.k3[t][e]);
    }
}
static hipError_t ensure_row_merged_tables() {
    static std::mutex mu;
    static bool done[64] = {false};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> g(mu);
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (done[dev]) return hipSuccess;
    static RowMergedTables T;
    static bool built = false;
    if (!built) {
        build_row_merged_tables(T);
        built = true;
    }
    e = hipMemcpyToSymbol(HIP_SYMBOL(ROW_MERGED), &T, sizeof T);
    if (e == hipSuccess) done[dev] = true;
    return e;
}

static hipError_t ensure_quad_merged_tables() {
    static std::mutex mu;
    static bool done[64] = {false};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> g(mu);
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (done[dev]) return hipSuccess;
    static QuadMergedTables T;  // zero-initialised; filled once
    static bool built = false;
    if (!built) {
        build_quad_merged_tables(T);
        built = true;
    }
    e = hipMemcpyToSymbol(HIP_SYMBOL(QUAD_MERGED), &T, sizeof T);
    if (e == hipSuccess) done[dev] = true;
    return e;
}

// Leaf digests of a column-major matrix laid out coset-major (see kernels_ntt.hip):
//   element (column c, physical point q) at mat[c * N + q], q = s * n + k  <->  natural index i = k * R + s.
// Leaf position j in the tree holds natural row bitrev_logN(j) (plonky2 reverse_index_bits_in_place),
// so the thread that owns physical point q writes digest slot j = bitrev(i).
// Four lanes (one DPP quad) walk one row; adjacent quads read adjacent k => each load touches whole 128-byte runs.
__device__ __forceinline__ void leaf_hash_body(const gl_t* __restrict__ mat, size_t n_cols, unsigned log_n, unsigned rate_bits,
                                               gl_t* __restrict__ digests) {
    // lane l of the quad owns sponge state elements l, l + 4, l + 8 (poseidon_dev.h)
    __shared__ RcPair rcs[4][96];  // per-lane view of the round constants, split in halves, + 3 zeros ("next round" of the last round)
    for (unsigned idx = threadIdx.x; idx < 4 * 96; idx += blockDim.x) {
        const unsigned ll = idx / 96, w = idx % 96;
        const gl_t c = w < 90 ? POSEIDON_RC_DEV[12 * (w / 3) + ll + 4 * (w % 3)] : 0;
        rcs[ll][w].lo = c & 0xFFFFFFFFull;
        rcs[ll][w].hi = c >> 32;
    }
    __shared__ RcPair tks[2 * QUAD_MERGED_TRIPLES];
    __shared__ RcPair tk3s[4][3 * QUAD_MERGED_TRIPLES];
    for (unsigned idx = threadIdx.x; idx < 2 * QUAD_MERGED_TRIPLES; idx += blockDim.x) tks[idx] = QUAD_MERGED.tk[idx];
    for (unsigned idx = threadIdx.x; idx < 4 * 3 * QUAD_MERGED_TRIPLES; idx += blockDim.x)
        tk3s[idx / (3 * QUAD_MERGED_TRIPLES)][idx % (3 * QUAD_MERGED_TRIPLES)] = QUAD_MERGED.tk3[idx / (3 * QUAD_MERGED_TRIPLES)][idx % (3 * QUAD_MERGED_TRIPLES)];
    __syncthreads();
    const unsigned log_N = log_n + rate_bits;
    const size_t N = (size_t)1 << log_N;
    const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const unsigned l = (unsigned)tid & 3u;
    const size_t q = tid >> 2;
    if (q >= N) return;  // whole quads drop out together
    const size_t s = q >> log_n, k = q & (((size_t)1 << log_n) - 1);
    const size_t i = (k << rate_bits) + s;
    const size_t j = gl_bitrev((uint32_t)i, log_N);
    const gl_t* col = mat + q;
    if (n_cols <= 4) {  // hash_or_noop: short leaves are copied, zero padded
        digests[4 * j + l] = l < n_cols ? col[(size_t)l * N] : 0;
        return;
    }
    const uint32_t diag0 = l == 0 ? 8u : 0u;
    const RcPair* rc = rcs[l];
    QuadMergedCoef mc;
    {
        const uint32_t* c = QUAD_MERGED.coef[l];
#pragma unroll
        for (int e = 0; e < 36; e++) mc.n3[e / 12][e % 12] = c[e];
#pragma unroll
        for (int e = 0; e < 3; e++) {
            mc.n1[e] = c[36 + e];
            mc.n2[e] = c[39 + e];
            mc.b2[e] = c[43 + e];
            mc.b3[e] = c[46 + e];
        }
        mc.m00 = c[42];
#pragma unroll
        for (int e = 0; e < 12; e++) mc.cf[e] = c[50 + e];
    }
    const RcPair* tk3 = tk3s[l];
    const bool even_lane = (l & 1u) == 0;
    gl_t s0 = 0, s1 = 0, s2 = 0;
    // Overwrite-mode sponge, rate 8: block b overwrites state elements 0 .. 7 = slots 0 and 1 of the four lanes with columns
    // 8 b + l and 8 b + l + 4.  The next block's two cells are requested before the current permutation (about 10 us of
    // arithmetic) so that their latency -- column stride N * 8 bytes, a new page per load -- is never waited for with only two
    // waves per SIMD.  A permutation that is followed by another full block computes only the capacity in its last layer.
    const gl_t* mine = col + (size_t)l * N;
    const size_t n_full = n_cols / 8, rem = n_cols % 8;
    gl_t n0 = 0, n1 = 0;
    if (n_full) {
        n0 = mine[0];
        n1 = mine[4 * N];
    }
    for (size_t b = 0; b < n_full; b++) {
        s0 = n0;
        s1 = n1;
        if (b + 1 < n_full) {
            n0 = mine[(8 * (b + 1)) * N];
            n1 = mine[(8 * (b + 1) + 4) * N];
            poseidon_permute_quad_merged<true>(s0, s1, s2, diag0, rc, mc, tks, tk3, l == 0, even_lane);
        } else {
            poseidon_permute_quad_merged<false>(s0, s1, s2, diag0, rc, mc, tks, tk3, l == 0, even_lane);
        }
    }
    if (rem) {  // the last, partial block overwrites elements 0 .. rem - 1 only
        const size_t off = 8 * n_full;
        if (l < rem) s0 = mine[off * N];
        if (l + 4 < rem) s1 = mine[(off + 4) * N];
        poseidon_permute_quad_merged<false>(s0, s1, s2, diag0, rc, mc, tks, tk3, l == 0, even_lane);
    }
    digests[4 * j + l] = gl_canon(s0);  // digest = state elements 0 .. 3: slot 0 of the four lanes
}

__global__ __launch_bounds__(256) void leaf_hash_kernel(const gl_t* __restrict__ mat, size_t n_cols, unsigned log_n, unsigned rate_bits,
                                                         gl_t* __restrict__ digests) { STARKHIP_PRIO_ENTRY
    leaf_hash_body(mat, n_cols, log_n, rate_bits, digests);
}

// The same over K matrices of ONE shape in one launch (grid.y = matrix): the commitments of K proofs of the same AIR.  A
// 1024-row AIR has 2048 .. 4096 leaves, i.e. 128 .. 256 waves of up to 12 167 sequential permutations each -- a latency chain
// that leaves 7/8 of the chip idle; K of them side by side fill it (scheduler.hip gathers them).
__global__ __launch_bounds__(256) void leaf_hash_multi_kernel(LeafHashBatch B, size_t n_cols, unsigned log_n, unsigned rate_bits) { STARKHIP_PRIO_ENTRY
    leaf_hash_body(B.mat[blockIdx.y], n_cols, log_n, rate_bits, B.digests[blockIdx.y]);
}

// ---- the row form (poseidon_dev.h): 16 lanes per leaf, for commitments whose quad launch would leave most of the chip idle.
// Same digests as leaf_hash_kernel.  Lane e < 8 of a row absorbs column 8 b + e of block b (the rate), lanes 8 .. 11 carry the
// capacity, lanes 12 .. 15 idle as mirrors.  One wave = 4 leaves; adjacent rows read adjacent points of a column.
__global__ __launch_bounds__(256) void leaf_hash_row_kernel(const gl_t* __restrict__ mat, size_t n_cols, unsigned log_n, unsigned rate_bits,
                                                             gl_t* __restrict__ digests) { STARKHIP_PRIO_ENTRY
    // [lane of the row][entry]: entries 0 .. 31 the round constants (zero beyond round 29), 32 + 3 t + {0, 1, 2} the merged triples' k1, k2
    // (lane 0 only) and k3; everything zero on lanes 12 .. 15
    __shared__ RcPair rcs[16][64];
    for (unsigned idx = threadIdx.x; idx < 16 * 64; idx += blockDim.x) {
        const unsigned e = idx / 64, r = idx % 64;
        RcPair c = {0, 0};
        if (e < 12 && r < 30) {
            const gl_t v = POSEIDON_RC_DEV[12 * r + e];
            c.lo = v & 0xFFFFFFFFull;
            c.hi = v >> 32;
        } else if (e < 12 && r >= 32 && r < 32 + 3 * 7) {
            const unsigned t = (r - 32) / 3, w = (r - 32) % 3;
            if (w == 2) c = ROW_MERGED.k3[t][e];
            else if (e == 0) c = w == 0 ? ROW_MERGED.k1[t] : ROW_MERGED.k2[t];
        }
        rcs[e][r] = c;
    }
    __syncthreads();
    const unsigned log_N = log_n + rate_bits;
    const size_t N = (size_t)1 << log_N;
    const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const unsigned e = (unsigned)tid & 15u;
    const size_t q = tid >> 4;
    if (q >= N) return;  // whole rows drop out together
    const size_t sidx = q >> log_n, k = q & (((size_t)1 << log_n) - 1);
    const size_t i = (k << rate_bits) + sidx;
    const size_t j = gl_bitrev((uint32_t)i, log_N);
    const gl_t* col = mat + q;
    if (n_cols <= 4) {  // hash_or_noop: short leaves are copied, zero padded
        if (e < 4) digests[4 * j + e] = e < n_cols ? col[(size_t)e * N] : 0;
        return;
    }
#ifdef STARKHIP_ROW_CPP_ROUNDS  // the C++ rounds (poseidon_permute_row): what hipcc schedules by itself, kept for comparison
    const uint32_t c0 = e == 0 ? 17u + 8u : 17u;  // CIRC[0] + MDS_MATRIX_DIAG[0] on lane 0
#define STARKHIP_ROW_PERMUTE(s) poseidon_permute_row(s, rc, c0, e == 0)
#else
    RowConsts K;
    row_consts_init(K, e, ROW_MERGED.coef[e]);
#ifdef STARKHIP_ROW_PLAIN_ROUNDS  // asm rounds without the merged triples
#define STARKHIP_ROW_PERMUTE(s) poseidon_permute_row_asm(s, rc, K)
#else
#define STARKHIP_ROW_PERMUTE(s) poseidon_permute_row_merged_asm(s, rc, K)
#endif
#endif
    const RcPair* rc = rcs[e];
    const bool absorbs = e < 8;
    gl_t s = 0;
    const gl_t* mine = col + (size_t)(absorbs ? e : 0) * N;  // lanes 8 .. 15 never load
    const size_t n_full = n_cols / 8, rem = n_cols % 8;
    gl_t nx = 0;
    if (n_full && absorbs) nx = mine[0];
    for (size_t b = 0; b < n_full; b++) {
        if (absorbs) s = nx;
        if (b + 1 < n_full && absorbs) nx = mine[(8 * (b + 1)) * N];  // requested one permutation ahead
        s = STARKHIP_ROW_PERMUTE(s);
    }
    if (rem) {  // the last, partial block overwrites elements 0 .. rem - 1 only
        if (e < rem) s = mine[(8 * n_full) * N];
        s = STARKHIP_ROW_PERMUTE(s);
    }
    if (e < 4) digests[4 * j + e] = gl_canon(s);
#undef STARKHIP_ROW_PERMUTE
}

// ---- the lane form (poseidon_dev.h): one lane per leaf, for big commitments when several are in flight (the pool picks).
// Same digests as leaf_hash_kernel.  64 adjacent points of a column per wave: every load is one 512-byte run.
__constant__ LaneTables LANE_TABLES;
__global__ __launch_bounds__(256, 2) void leaf_hash_lane_kernel(const gl_t* __restrict__ mat, size_t n_cols, unsigned log_n, unsigned rate_bits,
                                                              gl_t* __restrict__ digests) {
    __shared__ LaneTables T;
    {
        const uint32_t* src = (const uint32_t*)&LANE_TABLES;
        uint32_t* dst = (uint32_t*)&T;
        for (unsigned idx = threadIdx.x; idx < sizeof(LaneTables) / 4; idx += blockDim.x) dst[idx] = src[idx];
    }
    __syncthreads();
    const unsigned log_N = log_n + rate_bits;
    const size_t N = (size_t)1 << log_N;
    // Every lane of a wave stays active to the end: the matrix-pipe rounds read the operand registers of all 64 lanes whatever EXEC
    // says (a lane that left early would feed garbage weights into its partner half's sums).  Lanes beyond the last leaf shadow it.
    const size_t q_raw = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const bool live = q_raw < N;
    const size_t q = live ? q_raw : N - 1;
    const size_t sidx = q >> log_n, k = q & (((size_t)1 << log_n) - 1);
    const size_t i = (k << rate_bits) + sidx;
    const size_t j = gl_bitrev((uint32_t)i, log_N);
    const gl_t* col = mat + q;
    if (n_cols <= 4) {
        if (live)
            for (unsigned e = 0; e < 4; e++) digests[4 * j + e] = e < n_cols ? col[(size_t)e * N] : 0;
        return;
    }
    const size_t n_full = n_cols / 8, rem = n_cols % 8;
    gl_t nx[8];
    if (n_full) {
#pragma unroll
        for (int e = 0; e < 8; e++) nx[e] = col[(size_t)e * N];
    }
    LaneZeros Z;
    lane_zeros_init(Z);
    LaneMfma M;
    const unsigned lane = threadIdx.x & 63u;
    lane_mfma_init(M, lane);
    LaneState st;
#pragma unroll
    for (int w = 0; w < 8; w++) st.t0[w] = st.t1[w] = st.t2[w] = 0;
    for (size_t b = 0; b < n_full; b++) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            lane_set(st.t0, e, nx[e]);
            lane_set(st.t1, e, nx[4 + e]);
        }
        if (b + 1 < n_full) {
#pragma unroll
            for (int e = 0; e < 8; e++) nx[e] = col[(8 * (b + 1) + e) * N];  // requested one permutation ahead
            poseidon_permute_lane_asm<true>(st, &T, Z, M, lane);
        } else {
            poseidon_permute_lane_asm<false>(st, &T, Z, M, lane);
        }
    }
    if (rem) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if ((size_t)e < rem) lane_set(st.t0, e, col[(8 * n_full + e) * N]);
            if ((size_t)(4 + e) < rem) lane_set(st.t1, e, col[(8 * n_full + 4 + e) * N]);
        }
        poseidon_permute_lane_asm<false>(st, &T, Z, M, lane);
    }
    if (live) {
#pragma unroll
        for (int e = 0; e < 4; e++) digests[4 * j + e] = gl_canon(lane_get(st.t0, e));
    }
}
static bool build_lane_tables(LaneTables& T) {
    static PoseidonMergedFours P;
    build_poseidon_merged_fours(P);
    if (!P.sums_fit) return false;   // (a property of the MDS matrix, checked where the tables are made: the accumulators' 64 bits)
    auto split = [](gl_t v) { return RcPair{v & 0xFFFFFFFFull, v >> 32}; };
    memset(&T, 0, sizeof T);
    for (int r = 0; r < 30; r++)
        for (int e = 0; e < 12; e++) T.rc[r][e] = split(POSEIDON_RC_HOST[12 * r + e]);
    for (int e = 0; e < 12; e++) T.rc0[e] = POSEIDON_RC_HOST[e];
    for (int t = 0; t < POSEIDON_MERGED_FOURS; t++) {
        T.kf[t][0] = split(P.k1[t]);
        T.kf[t][1] = split(P.k2[t]);
        T.kf[t][2] = split(P.k3[t]);
        for (int e = 0; e < 12; e++) T.k4[t][e] = split(P.k4[t][e]);
    }
    for (int r = 0; r < 12; r++) {
        for (int c = 0; c < 12; c++) T.row[r][c] = (uint32_t)P.N4[r][c];
        T.row[r][12] = (uint32_t)P.N3[r][0];
        T.row[r][13] = (uint32_t)P.N2[r][0];
        T.row[r][14] = (uint32_t)P.M[r][0];
        T.m0[r] = (uint32_t)P.M[0][r];
        T.n20[r] = (uint32_t)P.N2[0][r];
        T.n30[r] = (uint32_t)P.N3[0][r];
    }
    T.n30[12] = (uint32_t)P.N2[0][0];
    // The matrix-pipe rounds (poseidon_dev.h: poseidon_permute_lane_asm): table m serves the round whose layer is seeded with rc[next[m]].
    // The products see signed bytes (byte - 128) and the spare K-values add 34 818 = STARKHIP_LANE_K_OFFSET to every plane, so the 64-bit
    // constant whose bytes ride in the weight tile is  RC[g] = rc[g] - (34 818 - 128 rowsum[g]) * 0x0101010101010101  mod p.
    static const int CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    static const int NEXT_ROUND[9] = {1, 2, 3, 4, 25, 26, 27, 28, 29};
    for (int m = 0; m < 9; m++)
        for (unsigned lane = 0; lane < 64; lane++) {
            const unsigned row = lane & 31u, half = lane >> 5, g = (row & 3u) + 4u * (row >> 3);
            const bool live = ((row >> 2) & 1u) == half && g < 12u;
            gl_t RC = 0;
            if (live) {
                uint64_t rowsum = 0;
                for (int j = 0; j < 12; j++) rowsum += (uint64_t)CIRC[(j + 12 - (int)g) % 12] + ((g == 0 && j == 0) ? 8u : 0u);
                const gl_t off = gl_mul((gl_t)(STARKHIP_LANE_K_OFFSET - 128 * rowsum), 0x0101010101010101ull % GL_P);
                RC = gl_sub(POSEIDON_RC_HOST[12 * NEXT_ROUND[m] + g], off);
            }
            for (int b = 0; b < 8; b++) {
                const uint32_t byte = (uint32_t)(RC >> (8 * b)) & 0xFFu;
                T.rcb[m][b][lane] = live ? ((byte & 0x7Fu) | ((2u * (byte >> 7) + 40u) << 8) | (127u << 16) | (127u << 24)) : 0u;
            }
        }
    return true;
}
static hipError_t ensure_lane_tables() {
    static std::mutex mu;
    static bool done[64] = {false};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> g(mu);
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (done[dev]) return hipSuccess;
    static LaneTables T;
    static bool built = false;
    if (!built) {
        if (!build_lane_tables(T)) return hipErrorInvalidValue;
        built = true;
    }
    e = hipMemcpyToSymbol(HIP_SYMBOL(LANE_TABLES), &T, sizeof T);
    if (e == hipSuccess) done[dev] = true;
    return e;
}

// ---- the pair form (poseidon_dev.h): two lanes per leaf, for a LONE commitment of >= 32 768 leaves (1 024 waves: one per SIMD).
// Same digests as leaf_hash_kernel.  Lane l < 32 of a wave absorbs columns 8 b .. 8 b + 5 of block b at point 32 w + l, lane l + 32
// columns 8 b + 6, 8 b + 7 of the same point and carries the capacity: every load is a 256-byte run.
__constant__ PairTables PAIR_TABLES;
__global__ __launch_bounds__(256, 2) void leaf_hash_pair_kernel(const gl_t* __restrict__ mat, size_t n_cols, unsigned log_n, unsigned rate_bits,
                                                              gl_t* __restrict__ digests) {
    __shared__ PairTables T;
    {
        const uint32_t* src = (const uint32_t*)&PAIR_TABLES;
        uint32_t* dst = (uint32_t*)&T;
        for (unsigned idx = threadIdx.x; idx < sizeof(PairTables) / 4; idx += blockDim.x) dst[idx] = src[idx];
    }
    __syncthreads();
    const unsigned log_N = log_n + rate_bits;
    const size_t N = (size_t)1 << log_N;
    const unsigned lane = threadIdx.x & 63u, half = lane >> 5;
    // every lane stays active to the end (the matrix-pipe rounds read all 64 lanes' operand registers); pairs beyond the last leaf shadow it
    const size_t q_raw = (blockIdx.x * (size_t)(blockDim.x >> 6) + (threadIdx.x >> 6)) * 32u + (lane & 31u);
    const bool live = q_raw < N;
    const size_t q = live ? q_raw : N - 1;
    const size_t sidx = q >> log_n, k = q & (((size_t)1 << log_n) - 1);
    const size_t i = (k << rate_bits) + sidx;
    const size_t j = gl_bitrev((uint32_t)i, log_N);
    if (n_cols <= 4) {
        if (live && half == 0)
            for (unsigned e = 0; e < 4; e++) digests[4 * j + e] = e < n_cols ? mat[q + (size_t)e * N] : 0;
        return;
    }
    const size_t n_full = n_cols / 8, rem = n_cols % 8;
    const unsigned first = half ? 6u : 0u;            // this lane's columns inside a block of eight: first .. first + cnt - 1
    const gl_t* col = mat + q + (size_t)first * N;
    gl_t nx[6];
    if (n_full) {
        nx[0] = col[0];
        nx[1] = col[N];
        if (half == 0) {
#pragma unroll
            for (int e = 2; e < 6; e++) nx[e] = col[(size_t)e * N];
        }
    }
    LaneZeros Z;
    lane_zeros_init(Z);
    PairMfma M;
    pair_mfma_init(M, lane);
    uint64_t mask_lo = 0xFFFFFFFFull;
    asm volatile("" : "+s"(mask_lo));
    PairState st;
#pragma unroll
    for (int w = 0; w < 4; w++) st.t0[w] = st.t1[w] = st.t2[w] = 0;
    for (size_t b = 0; b < n_full; b++) {
        pair_set(st.t0, 0, nx[0]);
        pair_set(st.t0, 1, nx[1]);
        if (half == 0) {   // the upper lane's elements 2 .. 5 are the capacity
            pair_set(st.t1, 0, nx[2]);
            pair_set(st.t1, 1, nx[3]);
            pair_set(st.t2, 0, nx[4]);
            pair_set(st.t2, 1, nx[5]);
        }
        if (b + 1 < n_full) {
            const gl_t* nc = col + 8 * (b + 1) * N;   // requested one permutation ahead
            nx[0] = nc[0];
            nx[1] = nc[N];
            if (half == 0) {
#pragma unroll
                for (int e = 2; e < 6; e++) nx[e] = nc[(size_t)e * N];
            }
            poseidon_permute_pair_asm<true>(st, &T, Z, M, lane, mask_lo);
        } else {
            poseidon_permute_pair_asm<false>(st, &T, Z, M, lane, mask_lo);
        }
    }
    if (rem) {
        const gl_t* nc = col + 8 * n_full * N;
        const unsigned cnt = half ? 2u : 6u;
#pragma unroll
        for (unsigned e = 0; e < 6; e++) {
            if (e < cnt && first + e < rem) {
                const gl_t x = nc[(size_t)e * N];
                if (e < 2) pair_set(st.t0, e, x);
                else if (e < 4) pair_set(st.t1, e - 2, x);
                else pair_set(st.t2, e - 4, x);
            }
        }
        poseidon_permute_pair_asm<false>(st, &T, Z, M, lane, mask_lo);
    }
    if (live && half == 0) {
        digests[4 * j + 0] = gl_canon(pair_get(st.t0, 0));
        digests[4 * j + 1] = gl_canon(pair_get(st.t0, 1));
        digests[4 * j + 2] = gl_canon(pair_get(st.t1, 0));
        digests[4 * j + 3] = gl_canon(pair_get(st.t1, 1));
    }
}
static bool build_pair_tables(PairTables& T) {
    static PoseidonMergedFours P;
    build_poseidon_merged_fours(P);
    if (!P.sums_fit) return false;
    auto split = [](gl_t v) { return RcPair{v & 0xFFFFFFFFull, v >> 32}; };
    memset(&T, 0, sizeof T);
    for (unsigned h = 0; h < 2; h++) {
        for (unsigned e = 0; e < 6; e++) T.rc0[h][e] = POSEIDON_RC_HOST[6 * h + e];
        for (int t = 0; t < POSEIDON_MERGED_FOURS; t++)
            for (unsigned r = 0; r < 6; r++) T.k4[t][h][r] = split(P.k4[t][6 * h + r]);
        uint32_t* c = T.coef[h];
        for (unsigned e = 0; e < 6; e++) {
            c[e] = (uint32_t)P.M[0][6 * h + e];
            c[8 + e] = (uint32_t)P.N2[0][6 * h + e];
            c[16 + e] = (uint32_t)P.N3[0][6 * h + e];
        }
        c[16 + 6] = (uint32_t)P.N2[0][0];
        for (unsigned r = 0; r < 6; r++) {
            const unsigned g = 6 * h + r;
            uint32_t* row = c + 24 + 16 * r;
            for (unsigned jj = 0; jj < 12; jj++) row[jj] = (uint32_t)P.N4[g][(6 * h + jj) % 12];   // own six, then the partner's
            row[12] = (uint32_t)P.N3[g][0];
            row[13] = (uint32_t)P.N2[g][0];
            row[14] = (uint32_t)P.M[g][0];
        }
    }
    for (int t = 0; t < POSEIDON_MERGED_FOURS; t++) {   // the constants of the three dot products enter once: through the lower half
        T.kf[t][0][0] = split(P.k1[t]);
        T.kf[t][0][1] = split(P.k2[t]);
        T.kf[t][0][2] = split(P.k3[t]);
    }
    // the matrix-pipe rounds' constants, as in build_lane_tables (the same offsets: a row still sums twelve signed bytes)
    static const int CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    static const int NEXT_ROUND[PAIR_MFMA_ROUNDS] = {1, 2, 3, 4, 25, 26, 27, 28, 29, 30};
    for (int m = 0; m < PAIR_MFMA_ROUNDS; m++)
        for (unsigned lane = 0; lane < 64; lane++) {
            const unsigned row = lane & 31u, khalf = lane >> 5, i = (row & 3u) + 4u * (row >> 3), out_half = (row >> 2) & 1u;
            if (khalf != 0 || i >= 12u) continue;
            const unsigned g = 6u * out_half + i % 6u, pp = i / 6u;
            uint64_t rowsum = 0;
            for (int jj = 0; jj < 12; jj++) rowsum += (uint64_t)CIRC[(jj + 12 - (int)g) % 12] + ((g == 0 && jj == 0) ? 8u : 0u);
            const gl_t off = gl_mul((gl_t)(STARKHIP_LANE_K_OFFSET - 128 * rowsum), 0x0101010101010101ull % GL_P);
            const gl_t rc = NEXT_ROUND[m] < 30 ? POSEIDON_RC_HOST[12 * NEXT_ROUND[m] + g] : 0;
            const gl_t RC = gl_sub(rc, off);
            for (unsigned qq = 0; qq < 4; qq++) {
                const uint32_t byte = (uint32_t)(RC >> (8 * (2 * qq + pp))) & 0xFFu;
                T.rcb[m][qq][lane] = (byte & 0x7Fu) | ((2u * (byte >> 7) + 40u) << 8) | (127u << 16) | (127u << 24);
            }
        }
    return true;
}
static hipError_t ensure_pair_tables() {
    static std::mutex mu;
    static bool done[64] = {false};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> g(mu);
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (done[dev]) return hipSuccess;
    static PairTables T;
    static bool built = false;
    if (!built) {
        if (!build_pair_tables(T)) return hipErrorInvalidValue;
        built = true;
    }
    e = hipMemcpyToSymbol(HIP_SYMBOL(PAIR_TABLES), &T, sizeof T);
    if (e == hipSuccess) done[dev] = true;
    return e;
}

// Leaves stored row-major and already in tree order: leaf j = rows[j][0..width)
__global__ __launch_bounds__(64) void leaf_hash_rows_kernel(const gl_t* __restrict__ rows, size_t width, size_t n_leaves,
                                                             gl_t* __restrict__ digests) { STARKHIP_PRIO_ENTRY
    size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (j >= n_leaves) return;
    gl_t out[4];
    poseidon_hash_or_noop_dev(rows + j * width, width, 1, out);
#pragma unroll
    for (int e = 0; e < 4; e++) digests[4 * j + e] = out[e];
}

__global__ __launch_bounds__(64) void merkle_level_kernel(const gl_t* __restrict__ child, gl_t* __restrict__ parent, size_t n_parent) { STARKHIP_PRIO_ENTRY
    size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (j >= n_parent) return;
    gl_t out[4];
    poseidon_two_to_one_dev(child + 8 * j, child + 8 * j + 4, out);
#pragma unroll
    for (int e = 0; e < 4; e++) parent[4 * j + e] = out[e];
}

__global__ void permute_batch_kernel(gl_t* states, size_t n) { STARKHIP_PRIO_ENTRY
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    gl_t s[12];
#pragma unroll
    for (int e = 0; e < 12; e++) s[e] = states[12 * i + e];
    poseidon_permute_dev(s);
#pragma unroll
    for (int e = 0; e < 12; e++) states[12 * i + e] = s[e];
}

// Proof-of-work grinding (plonky2 fri_proof_of_work, App. A.8): the challenger's sponge state with its
// pending inputs already written in; candidate nonce goes to lane `pos`; the response is state[7]
// after one permutation.  Keeps the MINIMUM valid nonce in *best (initialised to UINT64_MAX).
__global__ void pow_grind_kernel(const gl_t* __restrict__ base_state, int pos, unsigned pow_bits, uint64_t start, uint64_t count,
                                 unsigned long long* best) { STARKHIP_PRIO_ENTRY
    uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= count) return;
    uint64_t w = start + t;
    gl_t s[12];
#pragma unroll
    for (int e = 0; e < 12; e++) s[e] = base_state[e];
    s[pos] = w;
    poseidon_permute_dev(s);
    if ((s[7] >> (64 - pow_bits)) == 0) atomicMin(best, (unsigned long long)w);
}

static inline unsigned nblocks(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

hipError_t launch_leaf_hash(const gl_t* mat, size_t n_cols, unsigned log_n, unsigned rate_bits, gl_t* digests, hipStream_t st) {
    size_t N = (size_t)1 << (log_n + rate_bits);
    if (hipError_t e = ensure_quad_merged_tables(); e != hipSuccess) return e;
    hipLaunchKernelGGL(leaf_hash_kernel, dim3(nblocks(4 * N, 256)), dim3(256), 0, st, mat, n_cols, log_n, rate_bits, digests);
    return hipGetLastError();
}
hipError_t launch_leaf_hash_row(const gl_t* mat, size_t n_cols, unsigned log_n, unsigned rate_bits, gl_t* digests, hipStream_t st) {
    size_t N = (size_t)1 << (log_n + rate_bits);
    if (hipError_t e = ensure_row_merged_tables(); e != hipSuccess) return e;
    hipLaunchKernelGGL(leaf_hash_row_kernel, dim3(nblocks(16 * N, 256)), dim3(256), 0, st, mat, n_cols, log_n, rate_bits, digests);
    return hipGetLastError();
}
hipError_t launch_leaf_hash_lane(const gl_t* mat, size_t n_cols, unsigned log_n, unsigned rate_bits, gl_t* digests, hipStream_t st) {
    size_t N = (size_t)1 << (log_n + rate_bits);
    if (hipError_t e = ensure_lane_tables(); e != hipSuccess) return e;
    hipLaunchKernelGGL(leaf_hash_lane_kernel, dim3(nblocks(N, 256)), dim3(256), 0, st, mat, n_cols, log_n, rate_bits, digests);
    return hipGetLastError();
}
hipError_t launch_leaf_hash_pair(const gl_t* mat, size_t n_cols, unsigned log_n, unsigned rate_bits, gl_t* digests, hipStream_t st) {
    size_t N = (size_t)1 << (log_n + rate_bits);
    if (hipError_t e = ensure_pair_tables(); e != hipSuccess) return e;
    hipLaunchKernelGGL(leaf_hash_pair_kernel, dim3(nblocks(N, 128)), dim3(256), 0, st, mat, n_cols, log_n, rate_bits, digests);
    return hipGetLastError();
}
hipError_t launch_leaf_hash_multi(const LeafHashBatch& B, unsigned count, size_t n_cols, unsigned log_n, unsigned rate_bits, hipStream_t st) {
    if (count == 0 || count > LEAF_HASH_MAX_BATCH) return hipErrorInvalidValue;
    size_t N = (size_t)1 << (log_n + rate_bits);
    if (hipError_t e = ensure_quad_merged_tables(); e != hipSuccess) return e;
    if (count == 1) hipLaunchKernelGGL(leaf_hash_kernel, dim3(nblocks(4 * N, 256)), dim3(256), 0, st, B.mat[0], n_cols, log_n, rate_bits, B.digests[0]);
    else hipLaunchKernelGGL(leaf_hash_multi_kernel, dim3(nblocks(4 * N, 256), count), dim3(256), 0, st, B, n_cols, log_n, rate_bits);
    return hipGetLastError();
}
hipError_t launch_leaf_hash_rows(const gl_t* rows, size_t width, size_t n_leaves, gl_t* digests, hipStream_t st) {
    hipLaunchKernelGGL(leaf_hash_rows_kernel, dim3(nblocks(n_leaves, 64)), dim3(64), 0, st, rows, width, n_leaves, digests);
    return hipGetLastError();
}
// levels: digests buffer holds level 0 (n_leaves * 4) followed by level 1 (n_leaves/2 * 4) ... down to the cap level.
hipError_t launch_merkle_levels(gl_t* digests, unsigned log_leaves, unsigned cap_h, hipStream_t st) {
    gl_t* child = digests;
    for (unsigned lv = log_leaves; lv > cap_h; lv--) {
        size_t n_parent = (size_t)1 << (lv - 1);
        gl_t* parent = child + ((size_t)4 << lv);
        hipLaunchKernelGGL(merkle_level_kernel, dim3(nblocks(n_parent, 64)), dim3(64), 0, st, child, parent, n_parent);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        child = parent;
    }
    return hipSuccess;
}
hipError_t launch_permute_batch(gl_t* states, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(permute_batch_kernel, dim3(nblocks(n, 64)), dim3(64), 0, st, states, n);
    return hipGetLastError();
}
hipError_t launch_pow_grind(const gl_t* base_state, int pos, unsigned pow_bits, uint64_t start, uint64_t count, unsigned long long* best,
                            hipStream_t st) {
    hipLaunchKernelGGL(pow_grind_kernel, dim3(nblocks(count, 256)), dim3(256), 0, st, base_state, pos, pow_bits, start, count, best);
    return hipGetLastError();
}

}  // namespace starkhip
