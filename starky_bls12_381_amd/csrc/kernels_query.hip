// Query phase of FRI (plonky2 fri_prover_query_rounds) assembled on the device: every
// query round's leaves and Merkle paths are written straight into the proof blob's query section
// (layout: include/starkhip.h), so the host does one device-to-host copy instead of rebuilding 84 x 590 KB rows.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace starkhip {

// digest buffer layout: level 0 (n_leaves nodes) followed by level 1, ...; node offset of level l:
__device__ __forceinline__ size_t level_off_dev(size_t n_leaves, unsigned l) { return 2 * n_leaves - (2 * n_leaves >> l); }

// out[q * stride + off + c] = mat[c * N + phys(bitrev(x_q))]  for the coset-major matrix `mat`; x_q = tree leaf index
__global__ void query_leaf_colmajor_kernel(const gl_t* __restrict__ mat, size_t n_cols, unsigned log_n, unsigned rate_bits,
                                           const uint32_t* __restrict__ xs, gl_t* __restrict__ out, size_t stride, size_t off) { STARKHIP_PRIO_ENTRY
    const size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (c >= n_cols) return;
    const size_t q = blockIdx.y;
    const unsigned log_N = log_n + rate_bits;
    const size_t N = (size_t)1 << log_N;
    const size_t i = gl_bitrev(xs[q], log_N);  // natural point index of leaf x
    const size_t phys = ((i & (((size_t)1 << rate_bits) - 1)) << log_n) + (i >> rate_bits);
    out[q * stride + off + c] = mat[c * N + phys];
}

// out[q * stride + off + e] = rows[(x_q >> shift) * width + e]
__global__ void query_leaf_rows_kernel(const gl_t* __restrict__ rows, size_t width, const uint32_t* __restrict__ xs, unsigned shift,
                                       gl_t* __restrict__ out, size_t stride, size_t off) { STARKHIP_PRIO_ENTRY
    const size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (e >= width) return;
    const size_t q = blockIdx.y;
    out[q * stride + off + e] = rows[(size_t)(xs[q] >> shift) * width + e];
}

// siblings bottom-up: out[q * stride + off + 4 l + e] = digest(level l, ((x_q >> shift) >> l) ^ 1)[e], l < depth
__global__ void query_path_kernel(const gl_t* __restrict__ digests, size_t n_leaves, unsigned depth, const uint32_t* __restrict__ xs,
                                  unsigned shift, gl_t* __restrict__ out, size_t stride, size_t off) { STARKHIP_PRIO_ENTRY
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 4 * depth) return;
    const size_t q = blockIdx.y;
    const unsigned l = t >> 2, e = t & 3;
    const size_t idx = ((size_t)(xs[q] >> shift) >> l) ^ 1;
    out[q * stride + off + t] = digests[4 * (level_off_dev(n_leaves, l) + idx) + e];
}

static inline unsigned nb(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

hipError_t launch_query_leaf_colmajor(const gl_t* mat, size_t n_cols, unsigned log_n, unsigned rate_bits, const uint32_t* xs, size_t n_queries,
                                      gl_t* out, size_t stride, size_t off, hipStream_t st) {
    hipLaunchKernelGGL(query_leaf_colmajor_kernel, dim3(nb(n_cols, 256), (unsigned)n_queries), dim3(256), 0, st, mat, n_cols, log_n, rate_bits, xs,
                       out, stride, off);
    return hipGetLastError();
}
hipError_t launch_query_leaf_rows(const gl_t* rows, size_t width, const uint32_t* xs, unsigned shift, size_t n_queries, gl_t* out, size_t stride,
                                  size_t off, hipStream_t st) {
    hipLaunchKernelGGL(query_leaf_rows_kernel, dim3(nb(width, 64), (unsigned)n_queries), dim3(64), 0, st, rows, width, xs, shift, out, stride, off);
    return hipGetLastError();
}
hipError_t launch_query_path(const gl_t* digests, size_t n_leaves, unsigned depth, const uint32_t* xs, unsigned shift, size_t n_queries, gl_t* out,
                             size_t stride, size_t off, hipStream_t st) {
    if (depth == 0) return hipSuccess;
    hipLaunchKernelGGL(query_path_kernel, dim3(nb(4 * depth, 64), (unsigned)n_queries), dim3(64), 0, st, digests, n_leaves, depth, xs, shift, out,
                       stride, off);
    return hipGetLastError();
}

}  // namespace starkhip
