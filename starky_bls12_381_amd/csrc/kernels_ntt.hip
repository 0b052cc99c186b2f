// Goldilocks NTT kernels for gfx950.
//
//  * lde_columns_kernel: the trace commitment's hot loops 1+2 (plonky2
//    PolynomialBatch::from_values, App. A.3): per column  values -> ifft -> coeffs, then for each
//    of the R = 2^rate_bits cosets  coeffs * (7 w_N^s)^k -> fft.  One workgroup owns one column
//    (a batch of columns for short traces), the whole transform lives in LDS, HBM is touched
//    exactly once per element in and (1 + R) times out: 8*C*(n + n + N) bytes.
//  * ntt_global_kernel: in-place transform of a few long vectors (quotient polys, FRI layers) in
//    global memory; not on the bandwidth-critical path.
//
// Device LDE layout ("coset-major"): lde[c][s][k], s < R, k < n holds the evaluation at the
// NATURAL point index i = k*R + s, i.e. x = 7 * w_N^i.  Every consumer walks k with adjacent
// lanes, so all HBM traffic is coalesced; the reference's bit-reversed leaf order only shows up
// as the address at which a leaf digest is stored.
#include <hip/hip_runtime.h>

#include "gl.h"
#include "kernels.h"

namespace starkhip {

// ---------------------------------------------------------------- twiddle tables
// tw[j] = root^j for j < n/2
__global__ void fill_powers_kernel(gl_t* out, gl_t base, gl_t mult, size_t count) { STARKHIP_PRIO_ENTRY
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < count) out[i] = gl_mul(base, gl_pow(mult, i));
}

// coset scale table: sc[s][k] = n^-1 * (7 * w_N^s)^k
__global__ void fill_coset_scale_kernel(gl_t* out, unsigned log_n, unsigned rate_bits) { STARKHIP_PRIO_ENTRY
    size_t n = (size_t)1 << log_n;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= (n << rate_bits)) return;
    size_t s = i >> log_n, k = i & (n - 1);
    gl_t shift = gl_mul(GL_GENERATOR, gl_pow(gl_root_of_unity(log_n + rate_bits), s));
    out[i] = gl_mul(gl_inv((gl_t)n), gl_pow(shift, k));
}

// ---------------------------------------------------------------- row-major -> column-major
// in [rows][cols] -> out [cols][rows]; 32x32 tiles through LDS
__global__ void transpose_kernel(const gl_t* __restrict__ in, gl_t* __restrict__ out, size_t rows, size_t cols) { STARKHIP_PRIO_ENTRY
    __shared__ gl_t tile[32][33];
    size_t c0 = (size_t)blockIdx.x * 32, r0 = (size_t)blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        size_t r = r0 + j, c = c0 + threadIdx.x;
        if (r < rows && c < cols) tile[j][threadIdx.x] = in[r * cols + c];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        size_t c = c0 + j, r = r0 + threadIdx.x;
        if (r < rows && c < cols) out[c * rows + r] = tile[threadIdx.x][j];
    }
}

// ---------------------------------------------------------------- in-LDS radix-2 DIT
// data in LDS in bit-reversed order on entry, natural order on exit.  tw = powers of the size-n root
// (forward) or of its inverse.  All threads of the block cooperate; n/2 butterflies per stage.
template <int LOGN>
__device__ __forceinline__ void lds_ntt_dit(gl_t* a, const gl_t* __restrict__ tw, unsigned tw_log, int tid, int nthreads) {
    constexpr int n = 1 << LOGN;
#pragma unroll 1
    for (int st = 0; st < LOGN; st++) {
        const int half = 1 << st;
        for (int b = tid; b < n / 2; b += nthreads) {
            int j = b & (half - 1);
            int i0 = ((b >> st) << (st + 1)) + j;
            int i1 = i0 + half;
            gl_t w = tw[(size_t)j << (tw_log - 1 - st)];
            gl_t u = a[i0];
            gl_t v = gl_mul(a[i1], w);
            a[i0] = gl_add(u, v);
            a[i1] = gl_sub(u, v);
        }
        __syncthreads();
    }
}

// One workgroup transforms COLS_PER_BLOCK columns (each n = 2^LOGN long).
//   values  [C][n]      column-major input
//   coeffs  [C][n]      output (may be null)
//   lde     [C][R][n]   output, coset-major
// LDS: COLS_PER_BLOCK * n * 8 bytes (64 KiB at n = 8192).
template <int LOGN, int COLS_PER_BLOCK, int THREADS>
__global__ __launch_bounds__(THREADS) void lde_columns_kernel(const gl_t* values, gl_t* coeffs,  // coeffs may be values (in place): every word of a column is in LDS, behind a barrier, before any is written
                                                               gl_t* lde,  // not restrict: the trace columns wait inside the LDE buffer (prover.hip: trace_in_lde), so `values` may alias it
                                                               size_t n_cols, unsigned rate_bits,
                                                               const gl_t* __restrict__ tw_fwd, const gl_t* __restrict__ tw_inv,
                                                               unsigned tw_log, const gl_t* __restrict__ coset_scale, int from_coeffs) { STARKHIP_PRIO_ENTRY
    constexpr int n = 1 << LOGN;
    constexpr int TPC = THREADS / COLS_PER_BLOCK;  // threads cooperating on one column
    constexpr int EPT = (n + TPC - 1) / TPC;       // coefficients kept in registers per thread
    extern __shared__ gl_t smem[];
    const int sub = threadIdx.x / TPC, tid = threadIdx.x % TPC;
    const size_t col = (size_t)blockIdx.x * COLS_PER_BLOCK + sub;
    const bool active = col < n_cols;
    gl_t* a = smem + (size_t)sub * n;
    const unsigned R = 1u << rate_bits;

    gl_t c[EPT];
    if (from_coeffs) {
        // input already holds coefficients; pre-multiply by n because coset_scale carries n^-1
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            int k = tid + e * TPC;
            c[e] = (k < n && active) ? gl_mul(values[col * n + k], (gl_t)n) : 0;
        }
    } else {
        if (active)
            for (int k = tid; k < n; k += TPC) a[gl_bitrev(k, LOGN)] = values[col * n + k];
        __syncthreads();
        lds_ntt_dit<LOGN>(a, tw_inv, tw_log, tid, TPC);  // unscaled inverse transform (n^-1 folded into coset_scale)
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            int k = tid + e * TPC;
            c[e] = (k < n) ? a[k] : 0;
        }
    }
    if (coeffs && active && !from_coeffs) {
        const gl_t ninv = coset_scale[0];  // (7 w^0)^0 * n^-1
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            int k = tid + e * TPC;
            if (k < n) coeffs[col * n + k] = gl_mul(c[e], ninv);
        }
    }
    for (unsigned s = 0; s < R; s++) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            int k = tid + e * TPC;
            if (k < n) a[gl_bitrev(k, LOGN)] = gl_mul(c[e], coset_scale[(size_t)s * n + k]);
        }
        __syncthreads();
        lds_ntt_dit<LOGN>(a, tw_fwd, tw_log, tid, TPC);
        if (active)
            for (int k = tid; k < n; k += TPC) lde[(col * R + s) * n + k] = a[k];
    }
}

template <int LOGN, int CPB, int THREADS>
static hipError_t launch_lde_t(const gl_t* values, gl_t* coeffs, gl_t* lde, size_t n_cols, unsigned rate_bits, const gl_t* tw_fwd,
                               const gl_t* tw_inv, unsigned tw_log, const gl_t* coset_scale, int from_coeffs, hipStream_t st) {
    size_t lds = (size_t)CPB * sizeof(gl_t) << LOGN;
    auto k = lde_columns_kernel<LOGN, CPB, THREADS>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    unsigned blocks = (unsigned)((n_cols + CPB - 1) / CPB);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(THREADS), lds, st, values, coeffs, lde, n_cols, rate_bits, tw_fwd, tw_inv, tw_log,
                       coset_scale, from_coeffs);
    return hipGetLastError();
}

hipError_t launch_lde_columns(const gl_t* values, gl_t* coeffs, gl_t* lde, size_t n_cols, unsigned log_n, unsigned rate_bits,
                              const gl_t* tw_fwd, const gl_t* tw_inv, unsigned tw_log, const gl_t* coset_scale, int from_coeffs,
                              hipStream_t st) {
#define CASE(L, CPB, T) \
    case L: return launch_lde_t<L, CPB, T>(values, coeffs, lde, n_cols, rate_bits, tw_fwd, tw_inv, tw_log, coset_scale, from_coeffs, st);
    switch (log_n) {
        CASE(1, 64, 256) CASE(2, 64, 256) CASE(3, 64, 256) CASE(4, 32, 256) CASE(5, 16, 256) CASE(6, 8, 256) CASE(7, 4, 256)
        CASE(8, 2, 256) CASE(9, 1, 256) CASE(10, 1, 512) CASE(11, 1, 1024) CASE(12, 1, 1024) CASE(13, 1, 1024)
        default: return hipErrorInvalidValue;
    }
#undef CASE
}

// ---------------------------------------------------------------- global-memory NTT (few long vectors)
// One workgroup per vector; natural in, natural out.  mode bits: 1 = inverse (uses tw_inv and scales by n^-1),
// pre_scale / post_scale (nullable): element-wise multipliers (coset shift powers) applied before / after.
__global__ __launch_bounds__(1024) void ntt_global_kernel(gl_t* data, size_t vec_stride, unsigned log_n, const gl_t* __restrict__ tw,
                                                           unsigned tw_log, const gl_t* __restrict__ pre_scale,
                                                           const gl_t* __restrict__ post_scale, gl_t final_mul) { STARKHIP_PRIO_ENTRY
    gl_t* a = data + (size_t)blockIdx.x * vec_stride;
    const size_t n = (size_t)1 << log_n;
    const int tid = threadIdx.x, nt = blockDim.x;
    // bit-reversal permutation (+ pre-scale)
    for (size_t i = tid; i < n; i += nt) {
        size_t j = gl_bitrev((uint32_t)i, log_n);
        if (pre_scale) {
            if (i < j) {
                gl_t x = gl_mul(a[i], pre_scale[i]), y = gl_mul(a[j], pre_scale[j]);
                a[i] = y;
                a[j] = x;
            } else if (i == j) {
                a[i] = gl_mul(a[i], pre_scale[i]);
            }
        } else if (i < j) {
            gl_t x = a[i];
            a[i] = a[j];
            a[j] = x;
        }
    }
    __syncthreads();
    for (unsigned st = 0; st < log_n; st++) {
        const size_t half = (size_t)1 << st;
        for (size_t b = tid; b < n / 2; b += nt) {
            size_t j = b & (half - 1);
            size_t i0 = ((b >> st) << (st + 1)) + j, i1 = i0 + half;
            gl_t w = tw[j << (tw_log - 1 - st)];
            gl_t u = a[i0], v = gl_mul(a[i1], w);
            a[i0] = gl_add(u, v);
            a[i1] = gl_sub(u, v);
        }
        __syncthreads();
    }
    if (post_scale || final_mul != 1) {
        for (size_t i = tid; i < n; i += nt) {
            gl_t v = gl_mul(a[i], final_mul);
            if (post_scale) v = gl_mul(v, post_scale[i]);
            a[i] = v;
        }
    }
}

hipError_t launch_ntt_global(gl_t* data, size_t n_vecs, size_t vec_stride, unsigned log_n, const gl_t* tw, unsigned tw_log,
                             const gl_t* pre_scale, const gl_t* post_scale, gl_t final_mul, hipStream_t st) {
    if (log_n == 0 || n_vecs == 0) return hipSuccess;
    unsigned threads = 1024;
    while (threads > 64 && threads > (1u << log_n) / 2) threads >>= 1;
    hipLaunchKernelGGL(ntt_global_kernel, dim3((unsigned)n_vecs), dim3(threads), 0, st, data, vec_stride, log_n, tw, tw_log, pre_scale,
                       post_scale, final_mul);
    return hipGetLastError();
}

hipError_t launch_fill_powers(gl_t* out, gl_t base, gl_t mult, size_t count, hipStream_t st) {
    if (!count) return hipSuccess;
    hipLaunchKernelGGL(fill_powers_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, out, base, mult, count);
    return hipGetLastError();
}
hipError_t launch_fill_coset_scale(gl_t* out, unsigned log_n, unsigned rate_bits, hipStream_t st) {
    size_t count = (size_t)1 << (log_n + rate_bits);
    hipLaunchKernelGGL(fill_coset_scale_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, out, log_n, rate_bits);
    return hipGetLastError();
}
hipError_t launch_transpose(const gl_t* in, gl_t* out, size_t rows, size_t cols, hipStream_t st) {
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, st, in, out, rows, cols);
    return hipGetLastError();
}

}  // namespace starkhip
