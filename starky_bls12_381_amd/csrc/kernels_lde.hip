// Trace-column IFFT + coset LDE for 2^8 .. 2^13 rows: the device twin of plonky2's
// PolynomialBatch::from_values (values -> ifft -> lde onto the coset 7 * <w_N>), which the
// reference reaches through starky::prover::prove (/root/reference/src/aggregate_proof.rs:59-65).
//
// One workgroup owns whole columns (n / 16 threads per column, 16 elements per thread) and runs every
// transform of that column back to back: 1 inverse NTT, then 2^rate_bits forward NTTs whose inputs are the
// coefficients, still in registers, times the coset powers.  HBM sees each value once in, each coefficient and
// LDE point once out: 8 * (n + n + N) bytes per column.
//
// Each transform is a Stockham auto-sort NTT in radix-16 passes (plus one radix-2/4/8 pass when log n is not a
// multiple of 4).  Inside a pass the 16-point sub-transform runs in registers and all of its twiddles are
// powers of two -- in Goldilocks 2 has order 192 and plonky2's 64th root of unity is 2^39 -- so they cost
// shifts and adds, not 64x64 multiplies; only the n * 15/16 inter-pass twiddles per pass are generic
// multiplies, read from tables laid out [i][j mod Ns] so that a wave reads them as whole lines.  Passes
// exchange data through one padded LDS image of the column (row of 16 elements + 1 pad => the stride-16
// scatter of the first pass and the unit-stride gathers of the next are both conflict-free).
// Arithmetic is the lazy-reduction form of gl_dev.h; values are canonicalised when stored to HBM.
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <vector>

#include "gl_dev.h"
#include "kernels.h"

namespace starkhip {

// ---------------------------------------------------------------- register sub-transforms
constexpr int bitrev_c(int k, int bits) {
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((k >> i) & 1) << (bits - 1 - i);
    return r;
}
constexpr int ilog2_c(int x) { return x <= 1 ? 0 : 1 + ilog2_c(x >> 1); }

// Decimation-in-frequency radix-2 network of size R on v[BASE + STRIDE * i], root w_R = 2^(39 * 64 / R) (or its
// inverse).  Leaves X[k] at i = bitrev(k).
template <int R, bool INV, int BASE, int STRIDE>
struct Dif {
    static __device__ __forceinline__ void run(gl_t (&v)[16]) {
        constexpr int H = R / 2;
        constexpr int E_FWD = (39 * (64 / R)) % 192;
        constexpr int E = INV ? (192 - E_FWD) % 192 : E_FWD;
#pragma unroll
        for (int i = 0; i < H; i++) {
            const int e = (E * i) % 192;
            const gl_t a = v[BASE + STRIDE * i], b = v[BASE + STRIDE * (i + H)];
#ifdef STARKHIP_LDE_NN_BUTTERFLY  // the round-2 form: both operands arbitrary representatives, two wrap corrections per sum and difference
            v[BASE + STRIDE * i] = gl_add_nn(a, b);
            v[BASE + STRIDE * (i + H)] = e < 96 ? gl_mul_pow2_nn(gl_sub_nn(a, b), e) : gl_mul_pow2_nn(gl_sub_nn(b, a), e - 96);
#else
            // One operand canonical (3 instructions) makes both the sum and the difference single-correction forms (4 + 5
            // instead of 7 + 8): the second wrap of a + b or a - b needs BOTH operands >= p - 1 (gl_dev.h).  The subtrahend is
            // the canonical one: b for (a - b) 2^e, a for the negated form (b - a) 2^(e - 96).
            if (e < 96) {
                const gl_t bc = gl_canon(b);
                v[BASE + STRIDE * i] = gl_add_nc(a, bc);
                v[BASE + STRIDE * (i + H)] = gl_mul_pow2_nn(gl_sub_nc(a, bc), e);
            } else {
                const gl_t ac = gl_canon(a);
                v[BASE + STRIDE * i] = gl_add_nc(b, ac);
                v[BASE + STRIDE * (i + H)] = gl_mul_pow2_nn(gl_sub_nc(b, ac), e - 96);
            }
#endif
        }
        Dif<H, INV, BASE, STRIDE>::run(v);
        Dif<H, INV, BASE + STRIDE * H, STRIDE>::run(v);
    }
};
template <bool INV, int BASE, int STRIDE>
struct Dif<1, INV, BASE, STRIDE> {
    static __device__ __forceinline__ void run(gl_t (&)[16]) {}
};

// S = 16 / R independent size-R transforms: transform m lives in v[m + S * i]; natural order in and out.
template <int R, bool INV, int M>
struct SubNtts {
    static __device__ __forceinline__ void run(gl_t (&v)[16]) {
        constexpr int S = 16 / R;
        Dif<R, INV, M, S>::run(v);
        if constexpr (M + 1 < S) SubNtts<R, INV, M + 1>::run(v);
    }
};
template <int R>
__device__ __forceinline__ void unscramble(gl_t (&v)[16]) {
    constexpr int S = 16 / R, LOGR = ilog2_c(R);
    gl_t w[16];
#pragma unroll
    for (int m = 0; m < S; m++)
#pragma unroll
        for (int k = 0; k < R; k++) w[m + S * k] = v[m + S * bitrev_c(k, LOGR)];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = w[i];
}

// ---------------------------------------------------------------- pass structure
template <int LOGN>
struct LdePlan {
    static constexpr int N = 1 << LOGN;
    static constexpr int T = N / 16;                       // threads per column
    static constexpr int FULL = LOGN / 4;                  // radix-16 passes
    static constexpr int TAIL = LOGN % 4;                  // log2 of the last pass's radix (0: none)
    static constexpr int NP = FULL + (TAIL ? 1 : 0);
    static constexpr int radix(int p) { return p < FULL ? 16 : (1 << TAIL); }
    static constexpr int ns(int p) { return p == 0 ? 1 : ns(p - 1) * radix(p - 1); }
    // twiddle table: passes 1 .. NP-1, each radix(p) rows of ns(p) entries
    static constexpr int tw_off(int p) { return p <= 1 ? 0 : tw_off(p - 1) + radix(p - 1) * ns(p - 1); }
    static constexpr int tw_words() { return tw_off(NP); }
    static constexpr int THREADS = T < 256 ? 256 : T;
    static constexpr int CPB = THREADS / T;                // columns per workgroup
    static constexpr int LDS_COL = N + N / 16;             // padded elements per column
};

__device__ __forceinline__ int lds_pad(int a) { return a + (a >> 4); }
#ifdef STARKHIP_LDE_FULL_BARRIER
__device__ __forceinline__ void lde_lds_barrier() { __syncthreads(); }
#else
__device__ __forceinline__ void lde_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

// The inter-pass twiddles of pass P for this thread: w_{NS*R}^{i * (j mod NS)}, j = t + m * T; the inverse transform's last pass also
// carries n^-1 (row 0).  Uniform row base + 32-bit lane offset => scalar-base loads, no per-load address registers.  They depend on the
// thread only, not on the data: the loads of pass P + 1 are issued BEFORE pass P's exchange barriers, so their L2 latency passes under the
// LDS round trip instead of after it (17.10 -> 16.82 ms for FinalExp; -DSTARKHIP_LDE_NO_TW_PREFETCH loads them where they are used).
template <int LOGN, int P, bool INV>
__device__ __forceinline__ void lde_load_twiddles(gl_t (&w)[16], const gl_t* __restrict__ tw, int t) {
    using PL = LdePlan<LOGN>;
    if constexpr (P > 0 && P < PL::NP) {
        constexpr int R = PL::radix(P), S = 16 / R, NS = PL::ns(P), T = PL::T;
        constexpr bool LAST = P == PL::NP - 1;
        constexpr int TW_OFF = PL::tw_off(P);  // constexpr variable: otherwise the recursive helper survives as a CALL in the 2^13 kernel
        const gl_t* twp = tw + TW_OFF;
#pragma unroll
        for (int m = 0; m < S; m++) {
            const uint32_t jj8 = (uint32_t)((t + m * T) % NS) * 8u;
#pragma unroll
            for (int i = (INV && LAST) ? 0 : 1; i < R; i++) w[m + S * i] = *(const gl_t*)((const char*)(twp + i * NS) + jj8);
        }
    }
}

// One Stockham pass on the 16 values a thread holds (element i' = column index t + i' * T).
// P > 0: inputs come from the LDS image written by pass P-1.  Last pass: results stay in registers (natural index
// t + i' * T); otherwise they are scattered to the LDS image for pass P+1.
template <int LOGN, int P, bool INV>
__device__ __forceinline__ void lde_pass(gl_t (&v)[16], gl_t (&w)[16], gl_t* __restrict__ lds, const gl_t* __restrict__ tw, int t) {
    using PL = LdePlan<LOGN>;
    constexpr int R = PL::radix(P), S = 16 / R, NS = PL::ns(P), T = PL::T;
    constexpr bool LAST = P == PL::NP - 1;
    if constexpr (P > 0) {
        // T is a multiple of 16, so pad(t + i T) = pad(t) + i * (T + T/16): one address register, immediate offsets
        const int tpad = lds_pad(t);
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = lds[tpad + i * (T + T / 16)];
#ifdef STARKHIP_LDE_NO_TW_PREFETCH
        lde_load_twiddles<LOGN, P, INV>(w, tw, t);
#endif
#pragma unroll
        for (int m = 0; m < S; m++) {
#pragma unroll
            for (int i = (INV && LAST) ? 0 : 1; i < R; i++) v[m + S * i] = gl_mul_nc(v[m + S * i], w[m + S * i]);
        }
    }
    SubNtts<R, INV, 0>::run(v);
    unscramble<R>(v);
    if constexpr (!LAST) {
#ifndef STARKHIP_LDE_NO_TW_PREFETCH
        lde_load_twiddles<LOGN, P + 1, INV>(w, tw, t);
#endif
        // The two barriers of an exchange order LDS accesses only: they wait for this wave's LDS operations (lgkmcnt), NOT for its
        // global ones -- __syncthreads() also waits vmcnt(0), i.e. for the sixteen result stores of the previous coset transform to reach
        // memory, at the first exchange of every transform.  (Global data is never handed from thread to thread in this kernel: a
        // thread re-reads only the coefficients it stored itself.)
        lde_lds_barrier();  // every thread of the column has read its inputs
#pragma unroll
        for (int m = 0; m < S; m++) {
            const int j = t + m * T;
            if constexpr (NS == 1) {  // first pass (radix 16): element 16 j + k -> padded 17 j + k
#pragma unroll
                for (int k = 0; k < R; k++) lds[17 * j + k] = v[m + S * k];
            } else {  // NS is a multiple of 16: pad(base + k NS) = pad(base) + k * (NS + NS/16)
                const int pbase = lds_pad((j / NS) * NS * R + (j % NS));
#pragma unroll
                for (int k = 0; k < R; k++) lds[pbase + k * (NS + NS / 16)] = v[m + S * k];
            }
        }
        lde_lds_barrier();
    }
}

template <int LOGN, int P, bool INV>
struct LdePasses {
    static __device__ __forceinline__ void run(gl_t (&v)[16], gl_t* lds, const gl_t* tw, int t) {
        gl_t w[16];
        run_with(v, w, lds, tw, t);
    }
    static __device__ __forceinline__ void run_with(gl_t (&v)[16], gl_t (&w)[16], gl_t* lds, const gl_t* tw, int t) {
        lde_pass<LOGN, P, INV>(v, w, lds, tw, t);
        if constexpr (P + 1 < LdePlan<LOGN>::NP) LdePasses<LOGN, P + 1, INV>::run_with(v, w, lds, tw, t);
    }
};

// values [C][n] (or coefficients when from_coeffs) -> lde [C][2^rate][n], coset-major, and the coefficients at coeffs[c * cf_stride + k].
// The coefficients are the kernel's own scratch between the inverse transform and the coset transforms.  The prover does not keep them
// (openings and the FRI combination read coset 0 of the LDE: kernels_fri.hip): it passes the LAST coset slot of the column's own LDE
// block (cf_stride = 2^rate n, cf_keep = 0) -- a thread reads its 16 coefficient words for the last coset before it writes exactly those
// 16 positions with that coset's values.  starkhip_lde_batch asks for them (cf_stride = n, cf_keep = 1), possibly in place of `values`.
// `values` itself MAY lie inside `lde` (the prover parks a trace in the last quarter of the buffer the LDE is written to and launches
// ranges of columns whose blocks cover only columns already transformed: prover.hip run_lde_trace); within a workgroup every input word
// is in registers, behind a barrier in the closed-form branch, before any word is written -- so no pointer here is `__restrict__`.
template <int LOGN>
__global__ __launch_bounds__(LdePlan<LOGN>::THREADS, 4) void lde_columns_v2_kernel(const gl_t* values, gl_t* coeffs, unsigned cf_stride, int cf_keep,
                                                                                    gl_t* lde, unsigned n_cols, unsigned rate_bits,
                                                                                    const gl_t* __restrict__ tw_fwd,
                                                                                    const gl_t* __restrict__ tw_inv, const gl_t* __restrict__ cs,
                                                                                    const gl_t* __restrict__ oh, int from_coeffs) { STARKHIP_PRIO_ENTRY
    using PL = LdePlan<LOGN>;
    constexpr int T = PL::T, n = PL::N;
    extern __shared__ gl_t lds_all[];
    __shared__ unsigned cls[3];  // closed-form classes (below): [0] flags, [1] number of ones, [2] row of a one
    if constexpr (PL::CPB == 1) {
        if (threadIdx.x == 0) cls[0] = cls[1] = 0;
    }
    const int cib_raw = threadIdx.x / T, t = threadIdx.x % T;
    const unsigned col0 = blockIdx.x * PL::CPB;           // first column of this workgroup (always < n_cols)
    const bool live = col0 + cib_raw < n_cols;            // idle columns of the last workgroup shadow col0 and store nothing
    const int cib = live ? cib_raw : 0;
    gl_t* lds = lds_all + (size_t)cib_raw * PL::LDS_COL;
    // every HBM access below is (uniform workgroup base) + (32-bit lane byte offset) + (compile-time i * T * 8)
    const char* in_base = (const char*)(values + (size_t)col0 * n);
    const uint32_t in_off = (uint32_t)(cib * n + t) * 8u;
    gl_t v[16];
    // The coefficients are not held in registers across the coset transforms (16 x 64-bit more per lane would halve
    // the occupancy): each thread re-reads exactly the 16 words it stored itself, which are still in L2 / MALL.
    const char* cf_base = in_base;
    uint32_t cf_off = in_off;
    if (!from_coeffs) {
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = *(const gl_t*)(in_base + in_off + (uint32_t)(i * T * 8));
        if constexpr (PL::CPB == 1) {
            // Columns whose transforms have a closed form skip all five of them.  A FinalExp trace repeats every intermediate
            // Fp12 of its 32 operations on all 8192 rows (/root/reference/src/final_exponentiate.rs:137-228: constant columns) and
            // carries a one-hot row selector per row (:242-245, :931-956): 17.6 % of its 73 527 columns.
            //   constant c:   coefficients (c, 0, ..., 0), LDE = c everywhere;
            //   unit vector e_r (one 1 in row r, zeros elsewhere): coefficients n^-1 w_n^(-r j) = oh[(r j) mod n], and its LDE is
            //   the LDE of e_0 rotated by r inside every coset: lde_r[s][k] = lde_0[s][(k - r) mod n] = oh[n + s n + ((k - r) mod n)].
            // Both are the exact field values the transforms would produce (canonical), so the proof bytes do not change.  The
            // class is found from the values themselves (one barrier per column); one workgroup = one column here.
            if (oh != nullptr) {
                const gl_t first = *(const gl_t*)in_base;  // uniform
                unsigned flags = 0, ones = 0, row = 0;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    flags |= v[i] != first ? 1u : 0u;
                    flags |= v[i] > 1 ? 2u : 0u;
                    if (v[i] == 1) {
                        ones++;
                        row = (unsigned)(t + i * T);
                    }
                }
                __syncthreads();  // cls[] cleared
                if (flags) atomicOr(&cls[0], flags);
                if (ones) {
                    atomicAdd(&cls[1], ones);
                    cls[2] = row;
                }
                __syncthreads();
                const unsigned all_flags = cls[0], all_ones = cls[1];
                const bool is_const = (all_flags & 1u) == 0, is_unit = (all_flags & 2u) == 0 && all_ones == 1;
                if (is_const || is_unit) {  // uniform over the workgroup
                    const unsigned n_cosets = 1u << rate_bits;
                    char* cf_out = (char*)(coeffs + (size_t)col0 * cf_stride);  // one column per workgroup here
                    char* out_base = (char*)(lde + (size_t)col0 * n_cosets * n);
                    // 16-byte stores (two adjacent words per lane: 8-byte stores reach 0.5 - 0.7 of their rate): thread t owns words
                    // 2 t, 2 t + 1 of every block of 2 T
                    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                    if (is_const) {
                        const gl_t c = gl_canon(first);
                        if (cf_keep) {
#pragma unroll
                            for (int i = 0; i < 8; i++) {
                                u64x2 w = {0, 0};
                                if (t == 0 && i == 0) w.x = c;
                                *(u64x2*)(cf_out + (uint32_t)((2 * t + i * 2 * T) * 8)) = w;
                            }
                        }
                        const u64x2 cc = {c, c};
                        for (unsigned s = 0; s < n_cosets; s++) {
#pragma unroll
                            for (int i = 0; i < 8; i++) *(u64x2*)(out_base + (uint32_t)((s * n + 2 * t + i * 2 * T) * 8)) = cc;
                        }
                    } else {
                        const unsigned r = cls[2];
                        if (cf_keep) {
#pragma unroll
                            for (int i = 0; i < 8; i++) {
                                const unsigned j = (unsigned)(2 * t + i * 2 * T);
                                const u64x2 w = {oh[(r * j) & (unsigned)(n - 1)], oh[(r * (j + 1)) & (unsigned)(n - 1)]};
                                *(u64x2*)(cf_out + (uint32_t)(j * 8)) = w;
                            }
                        }
                        for (unsigned s = 0; s < n_cosets; s++) {
                            const gl_t* e0 = oh + n + (size_t)s * n;
#pragma unroll
                            for (int i = 0; i < 8; i++) {
                                const unsigned j = (unsigned)(2 * t + i * 2 * T);
                                const u64x2 w = {e0[(j - r) & (unsigned)(n - 1)], e0[(j + 1 - r) & (unsigned)(n - 1)]};
                                *(u64x2*)(out_base + (uint32_t)((s * n + j) * 8)) = w;
                            }
                        }
                    }
                    return;
                }
            }
        }
        LdePasses<LOGN, 0, true>::run(v, lds, tw_inv, t);
        char* cf_out = (char*)(coeffs + (size_t)col0 * cf_stride);
        cf_off = (uint32_t)(cib * cf_stride + t) * 8u;
        if (live) {
#pragma unroll
            for (int i = 0; i < 16; i++) *(gl_t*)(cf_out + cf_off + (uint32_t)(i * T * 8)) = gl_canon(v[i]);
        }
        cf_base = cf_out;
    }
    const unsigned n_cosets = 1u << rate_bits;
    char* out_base = (char*)(lde + (size_t)col0 * n_cosets * n);
    for (unsigned s = 0; s < n_cosets; s++) {
        // (no barrier needed here: pass 0 synchronises before it overwrites the LDS image the previous transform read)
        const char* cs_base = (const char*)(cs + (size_t)s * n);
        const uint32_t t8 = (uint32_t)t * 8u;
        const char* cfb = cf_base;
        asm volatile("" : "+s"(cfb));  // opaque: keeps the re-read a load (no forwarding from the stores above, no hoisting out of the loop)
#pragma unroll
        for (int i = 0; i < 16; i++)
            v[i] = gl_mul_nc(*(const gl_t*)(cfb + cf_off + (uint32_t)(i * T * 8)), *(const gl_t*)(cs_base + t8 + (uint32_t)(i * T * 8)));
        LdePasses<LOGN, 0, false>::run(v, lds, tw_fwd, t);
        if (live) {
            const uint32_t out_off = (uint32_t)((cib * n_cosets + s) * n + t) * 8u;  // < CPB * 2^rate * n * 8 <= 2^22
#pragma unroll
            for (int i = 0; i < 16; i++) *(gl_t*)(out_base + out_off + (uint32_t)(i * T * 8)) = gl_canon(v[i]);
        }
    }
}

// ---------------------------------------------------------------- wave-resident transforms for 2^13 rows
// The kernel the FinalExp-class traces take (tools/lde_wave_model.py is its executable specification: every index map, LDS address
// function and table below is checked there against a plain NTT and against the LDS banking rules).
//
// lde_columns_v2_kernel<13> moves a column through the WHOLE workgroup's LDS image three times per transform, each time between two
// s_barriers of eight waves: thirty barriers per column, waves of a workgroup in lock step, and the 17/16 padding conflict-free for the
// first exchange only.  Here a column's 13 index bits are split 4 (registers) + 6 (lanes) + 3 (waves) so that
//   * only ONE exchange per transform crosses waves (it has to: the three wave bits must reach the registers once), and its image is
//     laid out by READER -- after it every wave reads its own eighth of the LDS only;
//   * the other exchange moves bits between registers and LANES: it stays inside the wave's own eighth, needs no barrier, and waves run
//     at their own pace between the two barriers of the crossing exchange;
//   * the last index bit is a lane bit (lane ^ 32): v_permlane32_swap_b32 pairs the registers of the two half-waves, no LDS at all;
//   * the inverse transform ENDS in the layout the forward transforms START from, so a coefficient never leaves the thread that made
//     it: the first coset is transformed from registers, the others re-read the thread's own sixteen words (thread-major scratch in
//     the column's last coset slot, raw representatives, no canonicalisation);
//   * every LDS access pattern is conflict-free by construction (unit strides, or a 5-bit XOR swizzle inside 32-word rows);
//   * n^-1 is folded into the coset table.
// Two LDS exchanges and two barriers per transform instead of three and six; the forward transform's stores are two runs of 32
// consecutive points per wave and register (256 B each).
struct LdeWaveTables {
    const gl_t* tw1_inv;  // [16][512]  w_n^(-j2 k1), j2 = thread
    const gl_t* tw1_fwd;  // [16][512]  w_n^(+j2 k1), j2 = the low nine bits of the coefficient index a thread holds
    const gl_t* tw2_inv;  // [16][32]   w_512^(-b k2)
    const gl_t* tw2_fwd;  // [16][32]
    const gl_t* cs;       // [2^rate][16][512]  n^-1 (7 w_N^s)^c, c = the coefficient register i of thread t holds
};
static constexpr int LDE_WAVE_LOGN = 13;

// Development builds only (make variant DEFS=-DSTARKHIP_LDE_ABLATE=mask): leave out one kind of work to see what the rest costs -- wrong
// results.  1: table loads become constants, 2: no coefficient re-read, 4: no result stores, 8: no LDS exchanges, 16: no barriers.
#ifndef STARKHIP_LDE_ABLATE
#define STARKHIP_LDE_ABLATE 0
#endif
#define LDE_TABLE_LOAD(expr) ((STARKHIP_LDE_ABLATE & 1) ? (gl_t)(0x9E3779B97F4A7C15ULL + threadIdx.x) : (expr))
template <bool INV>
__device__ __forceinline__ void lde_dft16(gl_t (&v)[16]) {
    SubNtts<16, INV, 0>::run(v);
    unscramble<16>(v);
}
// LDS operations of ONE wave execute in order, so a wave's reads see its own earlier writes; what has to be kept is the compiler's order
__device__ __forceinline__ void lde_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lde_x_barrier() {
    if (!(STARKHIP_LDE_ABLATE & 16)) lde_lds_barrier();
}
__device__ __forceinline__ void lde_lds_put(char* p, gl_t x) {
    if (!(STARKHIP_LDE_ABLATE & 8)) *(gl_t*)p = x;
}
__device__ __forceinline__ gl_t lde_lds_get(const char* p, gl_t keep) { return (STARKHIP_LDE_ABLATE & 8) ? keep : *(const gl_t*)p; }

// exponent E of the radix-2 step's twiddle 2^E = w_32^(+-k4): w_64 = 2^39, so w_32 = 2^78 and its inverse 2^114
template <bool INV>
constexpr int lde_w32_exp(int k4) { return ((INV ? 114 : 78) * k4) % 192; }

// (A, B) <- (A + B', A - B') where B' = B * 2^E was formed WITHOUT its sign (2^96 = -1): E >= 96 swaps the outputs.  The element in B
// has k4 = K_LO in the lower half-wave and K_HI in the upper one.
template <bool INV, int K_LO, int K_HI>
__device__ __forceinline__ void lde_last_bfly(gl_t& A, gl_t& B, bool hi) {
    constexpr bool NEG_LO = lde_w32_exp<INV>(K_LO) >= 96, NEG_HI = lde_w32_exp<INV>(K_HI) >= 96;
    const gl_t bc = gl_canon(B);
    const gl_t s = gl_add_nc(A, bc), d = gl_sub_nc(A, bc);
    if constexpr (NEG_LO == NEG_HI) {
        A = NEG_LO ? d : s;
        B = NEG_LO ? s : d;
    } else {
        const bool neg = hi ? NEG_HI : NEG_LO;
        A = neg ? d : s;
        B = neg ? s : d;
    }
}
template <bool INV, int K>
__device__ __forceinline__ void lde_w32_twiddles(gl_t (&v)[16]) {  // v[k4] *= |w_32^(+-k4)| (sign left to lde_last_bfly)
    v[K] = gl_mul_pow2_nn(v[K], lde_w32_exp<INV>(K) % 96);
    if constexpr (K + 1 < 16) lde_w32_twiddles<INV, K + 1>(v);
}
// rows 2, 3 of `a` <-> rows 0, 1 of `b` (one row = 16 lanes): afterwards the lower half-wave holds (own a, the upper half's a) in
// (a, b), the upper half-wave (the lower half's b, own b)
__device__ __forceinline__ void lde_swap_halves(gl_t& a, gl_t& b) {
    // (the builtin, not inline assembly: v_permlane32_swap_b32 needs two wait states after a VALU write of its operands, which the
    // compiler's hazard recogniser inserts for the builtin and does not look for inside an asm block)
    const auto lo = __builtin_amdgcn_permlane32_swap((uint32_t)a, (uint32_t)b, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((uint32_t)(a >> 32), (uint32_t)(b >> 32), false, false);
    a = ((uint64_t)hi[0] << 32) | lo[0];
    b = ((uint64_t)hi[1] << 32) | lo[1];
}

// The grid is PERSISTENT: a workgroup takes its first column from its index and every further one from a counter (`next`, zero at
// launch), so a launch is 2 workgroups per CU however many columns it has: no workgroup turnover (LDS allocation, wave launch, the
// first loads' HBM latency with nothing to hide behind) between columns, and short columns (closed forms) even out by themselves.
__global__ __launch_bounds__(512, 4) void lde_columns_wave_kernel(const gl_t* values, gl_t* lde, unsigned n_cols, unsigned rate_bits, LdeWaveTables tb,
                                                                   const gl_t* __restrict__ oh, unsigned* next) { STARKHIP_PRIO_ENTRY
    constexpr int n = 1 << LDE_WAVE_LOGN, T = n / 16;
    extern __shared__ gl_t lds_all[];  // 8192 words, no padding
    __shared__ unsigned cls[3], next_col;
    const unsigned n_cosets = 1u << rate_bits;
    gl_t v[16], tw[16];
    unsigned col = blockIdx.x;
  while (col < n_cols) {  // uniform over the workgroup
    // (the thread index is made opaque per column: otherwise every per-lane table and image address below is an invariant of this loop,
    // hoisted out of it and kept in registers across it -- 140 spilled registers)
    unsigned t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const unsigned w = t >> 6, l = t & 63;
    const bool hi = l >= 32;
    const uint32_t t8 = t * 8u;
    if (t == 0) {
        cls[0] = cls[1] = 0;
        next_col = gridDim.x + atomicAdd(next, 1u);  // read by everybody behind a barrier further down
    }
    const char* in_base = (const char*)(values + (size_t)col * n);
    char* out_base = (char*)(lde + (size_t)col * n_cosets * n);
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = *(const gl_t*)(in_base + t8 + (uint32_t)(i * T * 8));
#pragma unroll
    for (int k = 1; k < 16; k++) tw[k] = LDE_TABLE_LOAD(*(const gl_t*)((const char*)tb.tw1_inv + t8 + (uint32_t)(k * T * 8)));

    // ---- closed forms (constant and unit-vector columns), as in lde_columns_v2_kernel: exact values, no transforms
    if (oh != nullptr) {
        const gl_t first = *(const gl_t*)in_base;  // uniform
        unsigned flags = 0, ones = 0, row = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            flags |= v[i] != first ? 1u : 0u;
            flags |= v[i] > 1 ? 2u : 0u;
            if (v[i] == 1) {
                ones++;
                row = t + (unsigned)i * T;
            }
        }
        // per WAVE first (ballots), then one LDS atomic per wave: 512 lanes' atomics on one address serialise -- in a real trace nearly
        // every thread holds a cell that differs from the first and a cell that is 1 (2 x 512 conflicting LDS operations per column)
        const bool w_differs = __ballot(flags & 1u) != 0, w_big = __ballot(flags & 2u) != 0;
        const unsigned w_ones = (unsigned)__popcll(__ballot(ones == 1)) + 2u * (unsigned)(__ballot(ones > 1) != 0);  // exact while it matters (<= 1)
        __syncthreads();  // cls[] cleared
        if (l == 0) {
            if (w_differs || w_big) atomicOr(&cls[0], (w_differs ? 1u : 0u) | (w_big ? 2u : 0u));
            if (w_ones) atomicAdd(&cls[1], w_ones);
        }
        if (w_ones == 1 && ones == 1) cls[2] = row;  // one lane of the wave
        __syncthreads();
        const unsigned all_flags = cls[0], all_ones = cls[1];
        const bool is_const = (all_flags & 1u) == 0, is_unit = (all_flags & 2u) == 0 && all_ones == 1;
        if (is_const || is_unit) {  // uniform over the workgroup
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            if (is_const) {
                const gl_t c = gl_canon(first);
                const u64x2 cc = {c, c};
                for (unsigned s = 0; s < n_cosets; s++) {
#pragma unroll
                    for (int i = 0; i < 8; i++) *(u64x2*)(out_base + (uint32_t)((s * n + 2 * t + i * 2 * T) * 8)) = cc;
                }
            } else {
                const unsigned r = cls[2];
                for (unsigned s = 0; s < n_cosets; s++) {
                    const gl_t* e0 = oh + n + (size_t)s * n;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const unsigned j = 2 * t + (unsigned)i * 2 * T;
                        const u64x2 wv = {e0[(j - r) & (unsigned)(n - 1)], e0[(j + 1 - r) & (unsigned)(n - 1)]};
                        *(u64x2*)(out_base + (uint32_t)((s * n + j) * 8)) = wv;
                    }
                }
            }
            col = next_col;  // (written before the two barriers above)
            __syncthreads();  // nobody is still reading cls[] / next_col when thread 0 writes them again
            continue;
        }
    }

    char* lds = (char*)lds_all;
    // ================================================================ inverse transform
    lde_dft16<true>(v);  // j12..9 -> k1
#pragma unroll
    for (int k = 1; k < 16; k++) v[k] = gl_mul_nc(v[k], tw[k]);
    {  // next twiddles: w_512^(-b k2), b = l & 31
        const uint32_t b8 = (l & 31u) * 8u;
#pragma unroll
        for (int k = 1; k < 16; k++) tw[k] = LDE_TABLE_LOAD(*(const gl_t*)((const char*)tb.tw2_inv + b8 + (uint32_t)(k * 32 * 8)));
    }
    lde_x_barrier();  // (nothing of this workgroup is in the image yet; kept: one code path for the crossing exchange)
#pragma unroll
    for (int k = 0; k < 16; k++) lde_lds_put(lds + t8 + (uint32_t)(k * 512 * 8), v[k]);  // image[k1][j2]
    lde_x_barrier();
    {   // this thread's k1 = 2 w + (l >> 5), b = l & 31; register a = j8..5
        const uint32_t base = ((2u * w + (l >> 5)) * 512u + (l & 31u)) * 8u;
#pragma unroll
        for (int a = 0; a < 16; a++) v[a] = lde_lds_get(lds + base + (uint32_t)(a * 32 * 8), v[a]);
    }
    lde_dft16<true>(v);  // j8..5 -> k2
#pragma unroll
    for (int k = 1; k < 16; k++) v[k] = gl_mul_nc(v[k], tw[k]);
    {   // exchange inside the wave's slice: word b * 32 + ((2 k2 + k1bit) ^ b)
        const uint32_t b = l & 31u, k1bit = l >> 5;
        const uint32_t wbase = ((w * 1024u + b * 32u) | (k1bit ^ b)) * 8u;
#pragma unroll
        for (int k = 0; k < 16; k++) lde_lds_put(lds + (wbase ^ (uint32_t)(2 * k * 8)), v[k]);
        lde_wave_sync();
        // now: k1bit = l & 1, k2 = (l >> 1) & 15, e = l >> 5; register d = j4..1, b = 2 d + e
        const uint32_t e = l >> 5, m5 = l & 31u;
        const uint32_t rbase = ((w * 1024u + e * 32u) | (m5 ^ e)) * 8u;
#pragma unroll
        for (int d = 0; d < 16; d++) v[d] = lde_lds_get(lds + ((rbase ^ (uint32_t)(2 * d * 8)) + (uint32_t)(d * 64 * 8)), v[d]);
    }
    lde_dft16<true>(v);  // j4..1 -> k4
    if (hi) lde_w32_twiddles<true, 0>(v);  // the e = 1 elements
    // registers (2 m, 2 m + 1) of the two half-waves -> (e = 0, e = 1) of k4 = 2 m + hi; radix 2 -> k5
#define LDE_INV_LAST(M) lde_swap_halves(v[2 * M], v[2 * M + 1]); lde_last_bfly<true, 2 * M, 2 * M + 1>(v[2 * M], v[2 * M + 1], hi);
    LDE_INV_LAST(0) LDE_INV_LAST(1) LDE_INV_LAST(2) LDE_INV_LAST(3) LDE_INV_LAST(4) LDE_INV_LAST(5) LDE_INV_LAST(6) LDE_INV_LAST(7)
#undef LDE_INV_LAST
    // coefficient register i = k5 * 8 + m is v[2 m + k5]; thread-major scratch in the column's last coset slot
    char* cf = out_base + (size_t)(n_cosets - 1) * n * 8;
    if (n_cosets > 1) {
#pragma unroll
        for (int i = 0; i < 16; i++) *(gl_t*)(cf + t8 + (uint32_t)(i * T * 8)) = v[2 * (i & 7) + (i >> 3)];
    }
    // ================================================================ forward transforms
    // coefficient index bits of this thread: c0 = l0, c7..4 = l4..1, c8 = l5, c3..1 = w
    const uint32_t c0 = l & 1u, a_in = ((l >> 5) << 3) | ((l >> 2) & 7u), c4 = (l >> 1) & 1u;
    const uint32_t fw_wbase = ((w * 1024u + ((a_in << 1) | c0) * 32u) | (c4 ^ ((c0 << 1) | ((a_in & 3u) << 2)))) * 8u;
    // after the exchange: k1 = l & 15, c4 = (l >> 4) & 1, c0 = l >> 5
    const uint32_t r_c0 = l >> 5, r_c4 = (l >> 4) & 1u, r_k1 = l & 15u;
    const uint32_t fw_rbase = ((w * 1024u + r_c0 * 32u) | (((r_k1 << 1) | r_c4) ^ (r_c0 << 1))) * 8u;
    const uint32_t fw_b = (r_c4 << 4) | (w << 1) | r_c0;                     // low five bits of the 512-point sub-transform's index
    const uint32_t fx_wbase = (fw_b * 32u + r_k1) * 8u;                      // + (k2 >> 1) * 1024 + (k2 & 1) * 16
    const uint32_t fx_rbase = (w * 1024u + (l >> 5) * 32u + (l & 31u)) * 8u; // + d * 64
    const uint32_t st_off = ((l & 31u) | (w << 5) | ((l >> 5) << 11)) * 8u;  // + (reg & 7) * 256 + (reg >> 3) * 4096
    {   // the first coset straight from the registers: v[i] <- coefficient i * cs[0][i]
#pragma unroll
        for (int i = 0; i < 16; i++) tw[i] = LDE_TABLE_LOAD(*(const gl_t*)((const char*)tb.cs + t8 + (uint32_t)(i * T * 8)));
        gl_t c[16];
#pragma unroll
        for (int i = 0; i < 16; i++) c[i] = gl_mul_nc(v[2 * (i & 7) + (i >> 3)], tw[i]);
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = c[i];
    }
    for (unsigned s = 0; s < n_cosets; s++) {
#pragma unroll
        for (int k = 1; k < 16; k++) tw[k] = LDE_TABLE_LOAD(*(const gl_t*)((const char*)tb.tw1_fwd + t8 + (uint32_t)(k * T * 8)));
        lde_dft16<false>(v);  // c12..9 -> k1
#pragma unroll
        for (int k = 1; k < 16; k++) v[k] = gl_mul_nc(v[k], tw[k]);
        {
            const uint32_t b8 = fw_b * 8u;
#pragma unroll
            for (int k = 1; k < 16; k++) tw[k] = LDE_TABLE_LOAD(*(const gl_t*)((const char*)tb.tw2_fwd + b8 + (uint32_t)(k * 32 * 8)));
        }
        // exchange inside the wave's slice: row (2 a + c0), word ((2 k1 + c4) ^ g), g = 2 c0 + 4 (a & 3)
#pragma unroll
        for (int k = 0; k < 16; k++) lde_lds_put(lds + (fw_wbase ^ (uint32_t)(2 * k * 8)), v[k]);
        lde_wave_sync();
#pragma unroll
        for (int a = 0; a < 16; a++) v[a] = lde_lds_get(lds + ((fw_rbase ^ (uint32_t)(((a & 3) << 2) * 8)) + (uint32_t)(a * 64 * 8)), v[a]);
        lde_dft16<false>(v);  // c8..5 -> k2
#pragma unroll
        for (int k = 1; k < 16; k++) v[k] = gl_mul_nc(v[k], tw[k]);
        lde_x_barrier();  // every wave is done with its slice
        const unsigned nxt = next_col;  // between the two barriers: thread 0 cannot have gone on to the next column's counter yet
#pragma unroll
        for (int k = 0; k < 16; k++) lde_lds_put(lds + fx_wbase + (uint32_t)(((k >> 1) * 1024 + (k & 1) * 16) * 8), v[k]);
        lde_x_barrier();

#pragma unroll
        for (int d = 0; d < 16; d++) v[d] = lde_lds_get(lds + fx_rbase + (uint32_t)(d * 64 * 8), v[d]);
        lde_dft16<false>(v);  // c4..1 -> k4
        if (hi) lde_w32_twiddles<false, 0>(v);
#define LDE_FWD_LAST(R) lde_swap_halves(v[R], v[R + 8]); lde_last_bfly<false, R, R + 8>(v[R], v[R + 8], hi);
        LDE_FWD_LAST(0) LDE_FWD_LAST(1) LDE_FWD_LAST(2) LDE_FWD_LAST(3) LDE_FWD_LAST(4) LDE_FWD_LAST(5) LDE_FWD_LAST(6) LDE_FWD_LAST(7)
#undef LDE_FWD_LAST
        char* ob = out_base + (size_t)s * n * 8;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if ((STARKHIP_LDE_ABLATE & 4) && v[r] != 12345) continue;  // (data-dependent, so the arithmetic stays)
            *(gl_t*)(ob + st_off + (uint32_t)(((r & 7) * 256 + (r >> 3) * 4096) * 8)) = gl_canon(v[r]);
        }
        if (s + 1 < n_cosets) {  // the next coset: the thread's own sixteen words back from the scratch, times that coset's powers
            const char* cs_base = (const char*)(tb.cs + (size_t)(s + 1) * n);
            const char* cfb = cf;
            asm volatile("" : "+s"(cfb));  // opaque: keeps the re-read a load (no forwarding from the stores above)
#pragma unroll
            for (int i = 0; i < 16; i++)
                v[i] = gl_mul_nc((STARKHIP_LDE_ABLATE & 2) ? v[i] : *(const gl_t*)(cfb + t8 + (uint32_t)(i * T * 8)),
                                 LDE_TABLE_LOAD(*(const gl_t*)(cs_base + t8 + (uint32_t)(i * T * 8))));
        } else {
            col = nxt;
        }
    }
  }
}

// ---------------------------------------------------------------- host side
template <int LOGN>
static void fill_tw(std::vector<gl_t>& out, bool inv) {
    using PL = LdePlan<LOGN>;
    out.assign(PL::tw_words() ? PL::tw_words() : 1, 1);
    const gl_t ninv = gl_inv((gl_t)PL::N);
    if (PL::NP == 1) out[0] = inv ? ninv : 1;
    for (int p = 1; p < PL::NP; p++) {
        const int R = PL::radix(p), NS = PL::ns(p);
        gl_t w = gl_root_of_unity(ilog2_c(NS * R));
        if (inv) w = gl_inv(w);
        const gl_t scale = (inv && p == PL::NP - 1) ? ninv : 1;
        for (int i = 0; i < R; i++) {
            const gl_t wi = gl_pow(w, i);
            gl_t acc = scale;
            for (int jj = 0; jj < NS; jj++) {
                out[PL::tw_off(p) + i * NS + jj] = acc;
                acc = gl_mul(acc, wi);
            }
        }
    }
}

template <int LOGN>
static hipError_t launch_v2(const gl_t* values, gl_t* coeffs, gl_t* lde, size_t n_cols, unsigned rate_bits, const gl_t* tw_fwd,
                            const gl_t* tw_inv, const gl_t* cs, const gl_t* oh, int from_coeffs, hipStream_t st) {
    using PL = LdePlan<LOGN>;
    const size_t lds_bytes = (size_t)PL::CPB * PL::LDS_COL * sizeof(gl_t);
    if (lds_bytes > 64 * 1024) {  // per device, so not cached in a static
        hipError_t e = hipFuncSetAttribute((const void*)lde_columns_v2_kernel<LOGN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    const unsigned blocks = (unsigned)((n_cols + PL::CPB - 1) / PL::CPB);
    // no coefficient output wanted: they pass through the last coset slot of the column's LDE block (see the kernel)
    const bool keep = coeffs != nullptr;
    gl_t* cf = keep ? coeffs : lde + (((size_t)1 << rate_bits) - 1) * PL::N;
    const unsigned cf_stride = keep ? (unsigned)PL::N : (unsigned)PL::N << rate_bits;
    hipLaunchKernelGGL(lde_columns_v2_kernel<LOGN>, dim3(blocks), dim3(PL::THREADS), lds_bytes, st, values, cf, cf_stride, keep ? 1 : 0, lde,
                       (unsigned)n_cols, rate_bits, tw_fwd, tw_inv, cs, oh, from_coeffs);
    return hipGetLastError();
}

bool lde_v2_supported(unsigned log_n) { return log_n >= 8 && log_n <= 13; }

size_t lde_v2_tw_words(unsigned log_n) {
    switch (log_n) {
        case 8: return LdePlan<8>::tw_words();
        case 9: return LdePlan<9>::tw_words();
        case 10: return LdePlan<10>::tw_words();
        case 11: return LdePlan<11>::tw_words();
        case 12: return LdePlan<12>::tw_words();
        case 13: return LdePlan<13>::tw_words();
        default: return 0;
    }
}

// Host-built tables: forward / inverse inter-pass twiddles, and cs[s][k] = (7 * w_N^s)^k (the coset shift of
// coset s; N = n * 2^rate_bits).  The three device buffers must hold lde_v2_tw_words(log_n) (x2) and N words.
// d_oh (n + N words, used when one workgroup owns one column: log_n >= 12): the closed forms of a unit-vector column --
// oh[i] = n^-1 w_n^(-i) for i < n, then the LDE of e_0, coset-major: oh[n + s n + k] = n^-1 (x^n - 1) / (x - 1) at x = 7 w_N^(4 k + s).
size_t lde_v2_oh_words(unsigned log_n, unsigned rate_bits) { return log_n >= 12 ? ((size_t)1 << log_n) + ((size_t)1 << (log_n + rate_bits)) : 0; }

hipError_t lde_v2_upload_tables(unsigned log_n, unsigned rate_bits, gl_t* d_tw_fwd, gl_t* d_tw_inv, gl_t* d_cs, gl_t* d_oh, hipStream_t st) {
    std::vector<gl_t> fwd, inv;
    switch (log_n) {
        case 8: fill_tw<8>(fwd, false); fill_tw<8>(inv, true); break;
        case 9: fill_tw<9>(fwd, false); fill_tw<9>(inv, true); break;
        case 10: fill_tw<10>(fwd, false); fill_tw<10>(inv, true); break;
        case 11: fill_tw<11>(fwd, false); fill_tw<11>(inv, true); break;
        case 12: fill_tw<12>(fwd, false); fill_tw<12>(inv, true); break;
        case 13: fill_tw<13>(fwd, false); fill_tw<13>(inv, true); break;
        default: return hipErrorInvalidValue;
    }
    const size_t n = (size_t)1 << log_n, n_cosets = (size_t)1 << rate_bits;
    std::vector<gl_t> cs(n * n_cosets);
    const gl_t wN = gl_root_of_unity(log_n + rate_bits);
    for (size_t s = 0; s < n_cosets; s++) {
        const gl_t shift = gl_mul(GL_GENERATOR, gl_pow(wN, s));
        gl_t acc = 1;
        for (size_t k = 0; k < n; k++) {
            cs[s * n + k] = acc;
            acc = gl_mul(acc, shift);
        }
    }
    hipError_t e;
    std::vector<gl_t> oh;
    if (d_oh && lde_v2_oh_words(log_n, rate_bits)) {
        oh.resize(lde_v2_oh_words(log_n, rate_bits));
        const gl_t ninv = gl_inv((gl_t)n), wn_inv = gl_inv(gl_root_of_unity(log_n));
        gl_t acc = ninv;
        for (size_t i = 0; i < n; i++) {
            oh[i] = acc;
            acc = gl_mul(acc, wn_inv);
        }
        // x^n = 7^n w_N^(s n) (w_N^(4 k n) = 1); geometric sum n^-1 (x^n - 1) / (x - 1); x - 1 != 0 on the coset
        const gl_t g_n = gl_pow(GL_GENERATOR, n);
        for (size_t s = 0; s < n_cosets; s++) {
            const gl_t num = gl_mul(ninv, gl_sub(gl_mul(g_n, gl_pow(wN, s * n)), 1));
            const gl_t step = gl_pow(wN, n_cosets);  // w_n
            gl_t x = gl_mul(GL_GENERATOR, gl_pow(wN, s));
            for (size_t k = 0; k < n; k++) {
                oh[n + s * n + k] = gl_mul(num, gl_inv(gl_sub(x, 1)));
                x = gl_mul(x, step);
            }
        }
        if ((e = hipMemcpyAsync(d_oh, oh.data(), oh.size() * 8, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
    }
    if ((e = hipMemcpyAsync(d_tw_fwd, fwd.data(), fwd.size() * 8, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(d_tw_inv, inv.data(), inv.size() * 8, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(d_cs, cs.data(), cs.size() * 8, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
    return hipStreamSynchronize(st);  // the host vectors go out of scope
}

hipError_t launch_lde_columns_v2(const gl_t* values, gl_t* coeffs, gl_t* lde, size_t n_cols, unsigned log_n, unsigned rate_bits,
                                 const gl_t* tw_fwd, const gl_t* tw_inv, const gl_t* cs, const gl_t* oh, int from_coeffs, hipStream_t st) {
    if (n_cols == 0) return hipSuccess;
    switch (log_n) {
        case 8: return launch_v2<8>(values, coeffs, lde, n_cols, rate_bits, tw_fwd, tw_inv, cs, oh, from_coeffs, st);
        case 9: return launch_v2<9>(values, coeffs, lde, n_cols, rate_bits, tw_fwd, tw_inv, cs, oh, from_coeffs, st);
        case 10: return launch_v2<10>(values, coeffs, lde, n_cols, rate_bits, tw_fwd, tw_inv, cs, oh, from_coeffs, st);
        case 11: return launch_v2<11>(values, coeffs, lde, n_cols, rate_bits, tw_fwd, tw_inv, cs, oh, from_coeffs, st);
        case 12: return launch_v2<12>(values, coeffs, lde, n_cols, rate_bits, tw_fwd, tw_inv, cs, oh, from_coeffs, st);
        case 13: return launch_v2<13>(values, coeffs, lde, n_cols, rate_bits, tw_fwd, tw_inv, cs, oh, from_coeffs, st);
        default: return hipErrorInvalidValue;
    }
}

// ---- wave-resident kernel (2^13 rows): tables as tools/lde_wave_model.py builds them
bool lde_wave_supported(unsigned log_n) { return log_n == (unsigned)LDE_WAVE_LOGN; }
size_t lde_wave_table_words(unsigned rate_bits) { return 2 * 16 * 512 + 2 * 16 * 32 + ((size_t)16 * 512 << rate_bits); }

static unsigned lde_wave_coef_index(unsigned t, unsigned i) {
    const unsigned w = t >> 6, l = t & 63;
    return (i << 9) | ((l >> 5) << 8) | (((l >> 1) & 15) << 4) | (w << 1) | (l & 1);
}

hipError_t lde_wave_upload_tables(unsigned rate_bits, gl_t* d_tab, hipStream_t st) {
    constexpr unsigned n = 1u << LDE_WAVE_LOGN, T = n / 16;
    std::vector<gl_t> tab(lde_wave_table_words(rate_bits));
    gl_t* tw1[2] = {tab.data(), tab.data() + 16 * T};            // inverse, forward
    gl_t* tw2[2] = {tab.data() + 2 * 16 * T, tab.data() + 2 * 16 * T + 16 * 32};
    gl_t* cs = tab.data() + 2 * 16 * T + 2 * 16 * 32;
    const gl_t wn = gl_root_of_unity(LDE_WAVE_LOGN);
    for (int dir = 0; dir < 2; dir++) {
        const gl_t w = dir == 0 ? gl_inv(wn) : wn;
        std::vector<gl_t> pw(n);  // w^e
        pw[0] = 1;
        for (unsigned e = 1; e < n; e++) pw[e] = gl_mul(pw[e - 1], w);
        for (unsigned t = 0; t < T; t++) {
            const unsigned j2 = dir == 0 ? t : (lde_wave_coef_index(t, 0) & 511u);
            for (unsigned k1 = 0; k1 < 16; k1++) tw1[dir][k1 * T + t] = pw[(j2 * k1) & (n - 1)];
        }
        for (unsigned k2 = 0; k2 < 16; k2++)
            for (unsigned b = 0; b < 32; b++) tw2[dir][k2 * 32 + b] = pw[(16 * b * k2) & (n - 1)];
    }
    const gl_t wN = gl_root_of_unity(LDE_WAVE_LOGN + rate_bits), ninv = gl_inv((gl_t)n);
    for (unsigned s = 0; s < (1u << rate_bits); s++) {
        const gl_t shift = gl_mul(GL_GENERATOR, gl_pow(wN, s));
        std::vector<gl_t> sp(n);  // n^-1 shift^c
        sp[0] = ninv;
        for (unsigned c = 1; c < n; c++) sp[c] = gl_mul(sp[c - 1], shift);
        for (unsigned i = 0; i < 16; i++)
            for (unsigned t = 0; t < T; t++) cs[(size_t)s * n + i * T + t] = sp[lde_wave_coef_index(t, i)];
    }
    hipError_t e = hipMemcpyAsync(d_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(st);  // the host vector goes out of scope
}

// values [C][8192] -> lde [C][2^rate][8192]; `next`: a device word holding 0 (the launch's column counter); `oh`: the closed-form tables of lde_v2_upload_tables (or null: every column is transformed)
hipError_t launch_lde_columns_wave(const gl_t* values, gl_t* lde, size_t n_cols, unsigned rate_bits, const gl_t* d_tab, const gl_t* oh, unsigned* next,
                                   hipStream_t st) {
    if (n_cols == 0) return hipSuccess;
    constexpr size_t T = (1u << LDE_WAVE_LOGN) / 16, lds_bytes = (size_t)8 << LDE_WAVE_LOGN;
    hipError_t e = hipFuncSetAttribute((const void*)lde_columns_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);  // per device: not cached
    if (e != hipSuccess) return e;
    LdeWaveTables tb;
    tb.tw1_inv = d_tab;
    tb.tw1_fwd = d_tab + 16 * T;
    tb.tw2_inv = d_tab + 2 * 16 * T;
    tb.tw2_fwd = tb.tw2_inv + 16 * 32;
    tb.cs = tb.tw2_fwd + 16 * 32;
    // two workgroups fit a CU (64 KB of LDS each); `next` must be zero (the caller clears it on the same stream)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned grid = (unsigned)std::min<size_t>(n_cols, (size_t)2 * (size_t)std::max(cus, 1));
    hipLaunchKernelGGL(lde_columns_wave_kernel, dim3(grid), dim3(512), lds_bytes, st, values, lde, (unsigned)n_cols, rate_bits, tb, oh, next);
    return hipGetLastError();
}

}  // namespace starkhip
