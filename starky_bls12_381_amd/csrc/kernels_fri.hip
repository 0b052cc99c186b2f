// Openings, FRI batch combination, FRI commit-phase helpers and query gathers for gfx950.
// Restates StarkOpeningSet::new, PolynomialBatch::prove_openings and fri_committed_trees of
// starky 0.1.2 / plonky2 0.1.4, reached from the reference through prove()
// at /root/reference/src/aggregate_proof.rs:59.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace starkhip {

// Sums of products w_k * v_k (field elements) taken WITHOUT reducing each product: the 128-bit products are added into a 192-bit
// accumulator (fewer than 2^64 terms fit) and the sum is reduced once -- ten to fifteen instructions per term where a reduced
// multiply-add is fifty (gl_mul + gl_add: the openings and the FRI combination, 1.2 * 10^9 terms each for a FinalExp trace, were bound
// by exactly those instructions).  2^128 = -2^32 (mod p).
struct Acc192 {
    uint64_t lo, mid, hi;
};
__device__ __forceinline__ void acc192_mad(Acc192& A, uint64_t w, uint64_t v) {
    const uint64_t pl = w * v, ph = __umul64hi(w, v);
    const uint64_t lo = A.lo + pl;
    const uint64_t c0 = lo < pl ? 1u : 0u;
    uint64_t mid = A.mid + ph;
    uint64_t c1 = mid < ph ? 1u : 0u;
    mid += c0;
    c1 += mid < c0 ? 1u : 0u;
    A.lo = lo;
    A.mid = mid;
    A.hi += c1;
}
__device__ __forceinline__ gl_t acc192_reduce(const Acc192& A) {  // hi < 2^32 (fewer than 2^32 terms)
    return gl_sub(gl_reduce128(A.mid, A.lo), gl_reduce128(0, A.hi << 32));
}
struct Acc192x2 {  // an element of the extension: both words against the same base-field factor
    Acc192 a0, a1;
};
__device__ __forceinline__ void acc192x2_mad(Acc192x2& A, gl2_t w, gl_t v) {
    acc192_mad(A.a0, w.a0, v);
    acc192_mad(A.a1, w.a1, v);
}
__device__ __forceinline__ gl2_t acc192x2_reduce(const Acc192x2& A) { return gl2_make(acc192_reduce(A.a0), acc192_reduce(A.a1)); }

// out[i] = base^i in the extension, i < count (SoA-free: array of gl2_t)
__global__ void ext_powers_kernel(gl2_t* out, gl2_t base, size_t count) { STARKHIP_PRIO_ENTRY
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < count) out[i] = gl2_pow(base, i);
}

// ---------------------------------------------------------------- openings (App. A.7)
// The prover keeps no coefficients of the trace polynomials (a FinalExp context would hold 4.8 GB of them for these two sums): a
// polynomial of degree < n is evaluated from its n values on coset 0 of the LDE, x_k = 7 w_n^k, by the interpolation formula
//     p(z) = sum_k p(x_k) L_k(z),   L_k(z) = (z^n - 7^n) x_k / (n 7^n (z - x_k))
// (Z(x) = x^n - 7^n vanishes on the coset and Z'(x_k) = n 7^n / x_k).  Exact field arithmetic: the same element of the extension as
// sum_k c_k z^k, so the proof bytes are those of the reference's coefficient form (StarkOpeningSet::new -> eval).  L_k(w_n z) =
// L_(k-1)(z): the weights of the next-row opening are the same vector rotated by one.  `scale` = (z^n - 7^n) / (n 7^n) from the host.
__global__ void coset_weights_kernel(gl2_t* __restrict__ wz, gl2_t* __restrict__ wgz, gl2_t z, gl2_t scale, unsigned log_n) { STARKHIP_PRIO_ENTRY
    const size_t n = (size_t)1 << log_n;
    const size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (k >= n) return;
    const gl_t x = gl_mul(GL_GENERATOR, gl_pow(gl_root_of_unity(log_n), k));
    const gl2_t d = gl2_sub(z, gl2_make(x, 0));
    // z ON the coset (z = x_k for one k; then scale = 0): L_j(x_k) is 1 for j = k and 0 elsewhere -- the opening is the value itself.  The
    // reference evaluates coefficients and never notices (only z in the subgroup H fails there: src of "Opening point is in the subgroup")
    const gl2_t w = (d.a0 == 0 && d.a1 == 0) ? gl2_one() : gl2_mul(gl2_mul_base(scale, x), gl2_inv(d));
    wz[k] = w;
    wgz[(k + 1) & (n - 1)] = w;
}

// A workgroup takes OPEN_COLS polynomials: out_z[c] = sum_k vec[c * stride + k] * zpow[k], out_gz[c] likewise with gzpow.  vec =
// coefficients and zpow = powers of z (the quotient polynomials), or vec = values on coset 0 and zpow = the weights above (the trace).
// vec is read exactly once.  zpow / gzpow (32 bytes per k) come from L2 and are used for all OPEN_COLS columns: with one column per
// workgroup they were four times the bytes of the column itself -- 19 GB through L2 for a FinalExp trace, 8 TB/s, the kernel's bound
// (2.3 ms; 4.8 GB of HBM reads would take 1).
static const unsigned OPEN_COLS = 4;
__global__ __launch_bounds__(256) void openings_kernel(const gl_t* __restrict__ coeffs, size_t stride, size_t n_polys, size_t n,
                                                       const gl2_t* __restrict__ zpow, const gl2_t* __restrict__ gzpow,
                                                       gl2_t* __restrict__ out_z, gl2_t* __restrict__ out_gz) { STARKHIP_PRIO_ENTRY
    const size_t c0 = (size_t)blockIdx.x * OPEN_COLS;
    const unsigned nc = (unsigned)(n_polys - c0 < OPEN_COLS ? n_polys - c0 : OPEN_COLS);
    const gl_t* col[OPEN_COLS];
#pragma unroll
    for (unsigned q = 0; q < OPEN_COLS; q++) col[q] = coeffs + (c0 + (q < nc ? q : 0)) * stride;  // (idle slots shadow the first column)
    Acc192x2 sum_a[OPEN_COLS], sum_b[OPEN_COLS];
#pragma unroll
    for (unsigned q = 0; q < OPEN_COLS; q++) sum_a[q] = sum_b[q] = Acc192x2{{0, 0, 0}, {0, 0, 0}};
    // the loads of step it + 1 are issued before the arithmetic of step it (hipcc keeps them inside their own iteration otherwise, and
    // four waves per SIMD do not cover an HBM round trip per step)
    const size_t n_steps = (n + blockDim.x - 1) / blockDim.x;
    gl_t v_next[OPEN_COLS];
    gl2_t wz_next = gl2_zero(), wg_next = gl2_zero();
    auto fetch = [&](size_t it) {
        const size_t k = it * blockDim.x + threadIdx.x;
        const bool in = it < n_steps && k < n;
        const size_t kk = in ? k : 0;
        wz_next = zpow[kk];
        if (out_gz) wg_next = gzpow[kk];
#pragma unroll
        for (unsigned q = 0; q < OPEN_COLS; q++) v_next[q] = in ? col[q][kk] : 0;  // a zero value adds nothing
    };
    fetch(0);
    for (size_t it = 0; it < n_steps; it++) {
        const gl2_t wz = wz_next, wg = wg_next;
        gl_t v[OPEN_COLS];
#pragma unroll
        for (unsigned q = 0; q < OPEN_COLS; q++) v[q] = v_next[q];
        fetch(it + 1);
#pragma unroll
        for (unsigned q = 0; q < OPEN_COLS; q++) {
            acc192x2_mad(sum_a[q], wz, v[q]);
            if (out_gz) acc192x2_mad(sum_b[q], wg, v[q]);
        }
    }
    gl2_t a[OPEN_COLS], b[OPEN_COLS];
#pragma unroll
    for (unsigned q = 0; q < OPEN_COLS; q++) {
        a[q] = acc192x2_reduce(sum_a[q]);
        b[q] = acc192x2_reduce(sum_b[q]);
    }
    __shared__ gl2_t sa[256], sb[256];
#pragma unroll
    for (unsigned q = 0; q < OPEN_COLS; q++) {  // (all of them, so that a[] and b[] stay in registers: the idle slots' sums are not stored)
        sa[threadIdx.x] = a[q];
        sb[threadIdx.x] = b[q];
        __syncthreads();
        for (int h = 128; h > 0; h >>= 1) {
            if ((int)threadIdx.x < h) {
                sa[threadIdx.x] = gl2_add(sa[threadIdx.x], sa[threadIdx.x + h]);
                sb[threadIdx.x] = gl2_add(sb[threadIdx.x], sb[threadIdx.x + h]);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0 && q < nc) {
            out_z[c0 + q] = sa[0];
            if (out_gz) out_gz[c0 + q] = sb[0];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- FRI batch combine (App. A.8)
// partial[jc][k] = sum_{j in chunk jc} apow[j0 + j] * coeffs[j * stride + k]: linear, so it is taken on the trace's coset-0 values as
// well as on coefficients (the quotient polynomials); the prover turns the sum of the former into coefficients with one inverse
// coset transform of two vectors (prover.hip)
__global__ __launch_bounds__(256) void fri_combine_kernel(const gl_t* __restrict__ coeffs, size_t stride, size_t n_polys, size_t n,
                                                          const gl2_t* __restrict__ apow, size_t polys_per_chunk,
                                                          gl2_t* __restrict__ partial) { STARKHIP_PRIO_ENTRY
    size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (k >= n) return;
    size_t j0 = (size_t)blockIdx.y * polys_per_chunk;
    size_t j1 = j0 + polys_per_chunk < n_polys ? j0 + polys_per_chunk : n_polys;
    Acc192x2 acc = {{0, 0, 0}, {0, 0, 0}};
    for (size_t j = j0; j < j1; j++) acc192x2_mad(acc, apow[j], coeffs[j * stride + k]);
    partial[(size_t)blockIdx.y * n + k] = acc192x2_reduce(acc);
}
// out[k] = sum_jc partial[jc][k], as two vectors of base-field words: out[k], out[n + k] (what launch_ntt_global transforms)
__global__ void ext_reduce_kernel(const gl2_t* __restrict__ partial, size_t n_chunks, size_t n, gl_t* __restrict__ out) { STARKHIP_PRIO_ENTRY
    size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (k >= n) return;
    gl2_t acc = gl2_zero();
    for (size_t c = 0; c < n_chunks; c++) acc = gl2_add(acc, partial[c * n + k]);
    out[k] = acc.a0;
    out[n + k] = acc.a1;
}

// ---------------------------------------------------------------- FRI commit phase
// values SoA [2][len] natural order  ->  leaf rows [len / arity][2 * arity]: row r, slot e = value at bitrev(r * arity + e)
__global__ void fri_leaves_kernel(const gl_t* __restrict__ vals, unsigned log_len, unsigned arity_bits, gl_t* __restrict__ rows) { STARKHIP_PRIO_ENTRY
    size_t len = (size_t)1 << log_len;
    size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (j >= len) return;
    size_t src = gl_bitrev((uint32_t)j, log_len);
    rows[2 * j] = vals[src];
    rows[2 * j + 1] = vals[len + src];
}
// coefficient fold: out[k] = sum_{i < arity} beta^i * in[k * arity + i]; in SoA [2][len], out SoA [2][len / arity]
__global__ void fri_fold_kernel(const gl_t* __restrict__ in, size_t len, unsigned arity_bits, gl2_t beta, gl_t* __restrict__ out) { STARKHIP_PRIO_ENTRY
    size_t arity = (size_t)1 << arity_bits, olen = len >> arity_bits;
    size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (k >= olen) return;
    gl2_t acc = gl2_zero();
    for (size_t i = arity; i-- > 0;) acc = gl2_add(gl2_mul(acc, beta), gl2_make(in[k * arity + i], in[len + k * arity + i]));
    out[k] = acc.a0;
    out[olen + k] = acc.a1;
}

static inline unsigned nb(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

hipError_t launch_ext_powers(gl2_t* out, gl2_t base, size_t count, hipStream_t st) {
    hipLaunchKernelGGL(ext_powers_kernel, dim3(nb(count, 256)), dim3(256), 0, st, out, base, count);
    return hipGetLastError();
}
hipError_t launch_coset_weights(gl2_t* wz, gl2_t* wgz, gl2_t z, gl2_t scale, unsigned log_n, hipStream_t st) {
    hipLaunchKernelGGL(coset_weights_kernel, dim3(nb((size_t)1 << log_n, 256)), dim3(256), 0, st, wz, wgz, z, scale, log_n);
    return hipGetLastError();
}
hipError_t launch_openings(const gl_t* coeffs, size_t stride, size_t n_polys, size_t n, const gl2_t* zpow, const gl2_t* gzpow, gl2_t* out_z,
                           gl2_t* out_gz, hipStream_t st) {
    if (!n_polys) return hipSuccess;
    hipLaunchKernelGGL(openings_kernel, dim3((unsigned)((n_polys + OPEN_COLS - 1) / OPEN_COLS)), dim3(256), 0, st, coeffs, stride, n_polys, n, zpow,
                       gzpow, out_z, out_gz);
    return hipGetLastError();
}
hipError_t launch_fri_combine(const gl_t* coeffs, size_t stride, size_t n_polys, size_t n, const gl2_t* apow, size_t polys_per_chunk,
                              size_t n_chunks, gl2_t* partial, hipStream_t st) {
    hipLaunchKernelGGL(fri_combine_kernel, dim3(nb(n, 256), (unsigned)n_chunks), dim3(256), 0, st, coeffs, stride, n_polys, n, apow,
                       polys_per_chunk, partial);
    return hipGetLastError();
}
hipError_t launch_ext_reduce(const gl2_t* partial, size_t n_chunks, size_t n, gl_t* out, hipStream_t st) {
    hipLaunchKernelGGL(ext_reduce_kernel, dim3(nb(n, 256)), dim3(256), 0, st, partial, n_chunks, n, out);
    return hipGetLastError();
}
hipError_t launch_fri_leaves(const gl_t* vals, unsigned log_len, unsigned arity_bits, gl_t* rows, hipStream_t st) {
    size_t len = (size_t)1 << log_len;
    hipLaunchKernelGGL(fri_leaves_kernel, dim3(nb(len, 256)), dim3(256), 0, st, vals, log_len, arity_bits, rows);
    return hipGetLastError();
}
hipError_t launch_fri_fold(const gl_t* in, size_t len, unsigned arity_bits, gl2_t beta, gl_t* out, hipStream_t st) {
    hipLaunchKernelGGL(fri_fold_kernel, dim3(nb(len >> arity_bits, 256)), dim3(256), 0, st, in, len, arity_bits, beta, out);
    return hipGetLastError();
}

}  // namespace starkhip
