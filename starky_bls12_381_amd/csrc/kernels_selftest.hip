// Field-arithmetic self-test kernel: applies one of the lazy-reduction helpers of gl_dev.h / poseidon_dev.h to arrays of
// operand pairs and returns canonical results, so that the tests can drive them with boundary values (0, p - 1, p,
// 2^64 - 1, ...) the prover's data hits only with probability ~2^-32 per operation.
#include <hip/hip_runtime.h>

#include "gl_dev.h"
#include "kernels.h"
#include "poseidon_dev.h"

namespace starkhip {

// op: 0 mul_nc(a, b)   1 mad_nc(a, b, a ^ b)   2 add_nn   3 sub_nn   4 add_nc(a, canon(b))   5 sub_nc(a, canon(b))
//     6 combine_lohi_nc(a & (2^44 - 1), b & (2^44 - 1))   7 reduce128_nc(hi = a, lo = b)   8 canon(a)
//     100 + e: mul_pow2_nn(a, e), 0 <= e < 96
template <int E>
__device__ __forceinline__ gl_t pow2_case(gl_t x, int e) {
    if (e == E) return gl_mul_pow2_nn(x, E);
    if constexpr (E + 1 < 96) return pow2_case<E + 1>(x, e);
    return 0;
}

__global__ void field_ops_kernel(int op, const gl_t* __restrict__ a, const gl_t* __restrict__ b, gl_t* __restrict__ out, size_t n) { STARKHIP_PRIO_ENTRY
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const gl_t x = a[i], y = b[i];
    gl_t r = 0;
    switch (op) {
        case 0: r = gl_mul_nc(x, y); break;
        case 1: r = gl_mad_nc(x, y, x ^ y); break;
        case 2: r = gl_add_nn(x, y); break;
        case 3: r = gl_sub_nn(x, y); break;
        case 4: r = gl_add_nc(x, gl_canon(y)); break;
        case 5: r = gl_sub_nc(x, gl_canon(y)); break;
        case 6: r = combine_lohi_nc(x & 0xFFFFFFFFFFFull, y & 0xFFFFFFFFFFFull); break;
        case 7: r = gl_reduce128_nc(x, y); break;
        case 8: r = x; break;
        case 9: r = gl_mad_nc_ub(x, b[0], y); break;  // wave-uniform multiplicand: a * b[0] + b[i]
        default: r = pow2_case<0>(x, op - 100); break;
    }
    out[i] = gl_canon(r);
}

hipError_t launch_field_ops(int op, const gl_t* a, const gl_t* b, gl_t* out, size_t n, hipStream_t st) {
    if (!((op >= 0 && op <= 9) || (op >= 100 && op < 196))) return hipErrorInvalidValue;
    hipLaunchKernelGGL(field_ops_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, op, a, b, out, n);
    return hipGetLastError();
}

}  // namespace starkhip
