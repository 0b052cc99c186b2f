// Constraint-quotient evaluation on gfx950: the device twin of starky's compute_quotient_polys
// (SURVEY.md App. A.6), which calls S::eval_packed_generic once per LDE point
// (/root/reference/src/final_exponentiate.rs:907, src/miller_loop.rs:644,
//  src/calc_pairing_precomp.rs:376, src/fp12_mul.rs:58).
//
// The AIR arrives as the flat program of air_ir.h.  One lane owns one coset point; the program
// counter, group headers, constants and public inputs are wave-uniform and come through scalar
// loads, every trace-cell access is a coalesced 512-byte line of the coset-major LDE.  The program
// is cut into `n_chunks` pieces at group boundaries so that (points / 64) x n_chunks waves fill the
// chip; a chunk's partial fold is scaled by alpha^(constraints after the chunk) in the combine
// kernel, which is exact in the field.
#include <hip/hip_runtime.h>

#include "air_ir.h"
#include "kernels.h"

namespace starkhip {

// Per-point tables in the quotient domain's physical order t = s' * n + k  <->  i = k * 2^qdb + s',
// x = 7 * w_size^i:  tab[0][t] = x - g^-1 (z_last), tab[1][t] = L_first(x), tab[2][t] = L_last(x),
// tab[3][t] = 1 / Z_H(x).
__global__ void quotient_tables_kernel(gl_t* tab, unsigned log_n, unsigned qdb) {
    const size_t n = (size_t)1 << log_n, size = n << qdb;
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= size) return;
    size_t sp = t >> log_n, k = t & (n - 1);
    size_t i = (k << qdb) + sp;
    gl_t g = gl_root_of_unity(log_n);
    gl_t x = gl_mul(GL_GENERATOR, gl_pow(gl_root_of_unity(log_n + qdb), i));
    gl_t zh = gl_sub(gl_mul(gl_pow(GL_GENERATOR, n), gl_pow(gl_root_of_unity(qdb), sp)), 1);  // x^n - 1
    tab[t] = gl_sub(x, gl_inv(g));
    tab[size + t] = gl_mul(zh, gl_inv(gl_mul((gl_t)n, gl_sub(x, 1))));
    tab[2 * size + t] = gl_mul(zh, gl_inv(gl_mul((gl_t)n, gl_sub(gl_mul(g, x), 1))));
    tab[3 * size + t] = gl_inv(zh);
}

struct QuotientParams {
    const uint32_t* code;
    const gl_t* consts;
    const gl_t* pis;
    const gl_t* lde;           // [C][N] coset-major
    const gl_t* tab;           // quotient_tables_kernel output
    const uint32_t* chunk_off; // [n_chunks + 1] word offsets into code (group boundaries)
    const gl_t* apow;          // [2][AIR_MAX_GROUP + 1] powers of alpha_0 / alpha_1
    gl_t* partial;             // [n_chunks][2][size]
    gl_t alpha0, alpha1;
    unsigned log_n, rate_bits, qdb;
};

__global__ __launch_bounds__(64) void quotient_eval_kernel(QuotientParams P) {
    const size_t n = (size_t)1 << P.log_n, size = n << P.qdb, N = n << P.rate_bits;
    const unsigned t_raw = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t_raw < size;  // domains smaller than a wave (FP12Mul: 32 points): idle lanes shadow point 0
    const unsigned t = live ? t_raw : 0;
    const unsigned chunk = blockIdx.y;
    const unsigned sp = t >> P.log_n, k = t & (unsigned)(n - 1);
    const unsigned s = sp << (P.rate_bits - P.qdb);  // LDE coset of this quotient point
    const unsigned off_local = s * (unsigned)n + k;
    const unsigned off_next = s * (unsigned)n + ((k + 1) & (unsigned)(n - 1));
    gl_t mask[4];
    mask[0] = 1;
    mask[1] = P.tab[t];
    mask[2] = P.tab[size + t];
    mask[3] = P.tab[2 * size + t];

    const uint32_t* __restrict__ w = P.code + P.chunk_off[chunk];
    const uint32_t* const wend = P.code + P.chunk_off[chunk + 1];
    const gl_t* __restrict__ consts = P.consts;
    const gl_t* __restrict__ pis = P.pis;
    const gl_t* __restrict__ lde = P.lde;
    const gl_t a0 = P.alpha0, a1 = P.alpha1;

    gl_t acc0 = 0, acc1 = 0;
    while (w < wend) {
        const uint32_t gw = *w++;
        const uint32_t kind = (gw >> 4) & 3u, ng = (gw >> 8) & 255u, m = gw >> 16;
        gl_t G = mask[0];
        if (kind == 1) G = mask[1];
        else if (kind == 2) G = mask[2];
        else if (kind == 3) G = mask[3];
        for (uint32_t g = 0; g < ng; g++) {
            const uint32_t ref = *w++;
            const gl_t* colp = lde + (size_t)(ref & REF_COL_MASK) * N;
            gl_t v = colp[(ref & REF_NEXT) ? off_next : off_local];
            if (ref & REF_COMPL) v = gl_sub(1, v);
            G = gl_mul(G, v);
        }
        gl_t t0 = 0, t1 = 0;
        for (uint32_t c = 0; c < m; c++) {
            gl_t body = 0;
            uint32_t tw;
            do {
                tw = *w++;
                const uint32_t nf = tw & 3u, ck = (tw >> 2) & 7u, idx = tw >> 6;
                gl_t v = 1;
                if (nf >= 1) {
                    const uint32_t r0 = *w++;
                    v = (lde + (size_t)(r0 & REF_COL_MASK) * N)[(r0 & REF_NEXT) ? off_next : off_local];
                    if (nf >= 2) {
                        const uint32_t r1 = *w++;
                        v = gl_mul(v, (lde + (size_t)(r1 & REF_COL_MASK) * N)[(r1 & REF_NEXT) ? off_next : off_local]);
                        if (nf >= 3) {
                            const uint32_t r2 = *w++;
                            v = gl_mul(v, (lde + (size_t)(r2 & REF_COL_MASK) * N)[(r2 & REF_NEXT) ? off_next : off_local]);
                        }
                    }
                }
                if (ck == CK_PLUS) body = gl_add(body, v);
                else if (ck == CK_MINUS) body = gl_sub(body, v);
                else if (ck == CK_CONST) body = gl_add(body, gl_mul(v, consts[idx]));
                else if (ck == CK_PI) body = gl_add(body, gl_mul(v, pis[idx]));
                else body = gl_sub(body, gl_mul(v, pis[idx]));
            } while (!(tw & 32u));
            t0 = gl_add(gl_mul(t0, a0), body);
            t1 = gl_add(gl_mul(t1, a1), body);
        }
        acc0 = gl_add(gl_mul(acc0, P.apow[m]), gl_mul(G, t0));
        acc1 = gl_add(gl_mul(acc1, P.apow[(AIR_MAX_GROUP + 1) + m]), gl_mul(G, t1));
    }
    if (live) {
        P.partial[((size_t)chunk * 2 + 0) * size + t] = acc0;
        P.partial[((size_t)chunk * 2 + 1) * size + t] = acc1;
    }
}

// out[j][i] (natural quotient index i) = (sum_p partial[p][j][t] * chunk_scale[p][j]) / Z_H(x_i)
__global__ void quotient_combine_kernel(const gl_t* __restrict__ partial, const gl_t* __restrict__ chunk_scale, unsigned n_chunks,
                                        const gl_t* __restrict__ tab, unsigned log_n, unsigned qdb, gl_t* __restrict__ out) {
    const size_t n = (size_t)1 << log_n, size = n << qdb;
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= size) return;
    size_t sp = t >> log_n, k = t & (n - 1);
    size_t i = (k << qdb) + sp;
    gl_t zhi = tab[3 * size + t];
    for (int j = 0; j < 2; j++) {
        gl_t acc = 0;
        for (unsigned p = 0; p < n_chunks; p++) acc = gl_add(acc, gl_mul(partial[((size_t)p * 2 + j) * size + t], chunk_scale[p * 2 + j]));
        out[(size_t)j * size + i] = gl_mul(acc, zhi);
    }
}

hipError_t launch_quotient_tables(gl_t* tab, unsigned log_n, unsigned qdb, hipStream_t st) {
    size_t size = (size_t)1 << (log_n + qdb);
    hipLaunchKernelGGL(quotient_tables_kernel, dim3((unsigned)((size + 255) / 256)), dim3(256), 0, st, tab, log_n, qdb);
    return hipGetLastError();
}

hipError_t launch_quotient_eval(const uint32_t* code, const gl_t* consts, const gl_t* pis, const gl_t* lde, const gl_t* tab,
                                const uint32_t* chunk_off, unsigned n_chunks, const gl_t* apow, gl_t alpha0, gl_t alpha1, gl_t* partial,
                                unsigned log_n, unsigned rate_bits, unsigned qdb, hipStream_t st) {
    QuotientParams P;
    P.code = code; P.consts = consts; P.pis = pis; P.lde = lde; P.tab = tab; P.chunk_off = chunk_off; P.apow = apow; P.partial = partial;
    P.alpha0 = alpha0; P.alpha1 = alpha1; P.log_n = log_n; P.rate_bits = rate_bits; P.qdb = qdb;
    size_t size = (size_t)1 << (log_n + qdb);
    hipLaunchKernelGGL(quotient_eval_kernel, dim3((unsigned)((size + 63) / 64), n_chunks), dim3(64), 0, st, P);
    return hipGetLastError();
}

hipError_t launch_quotient_combine(const gl_t* partial, const gl_t* chunk_scale, unsigned n_chunks, const gl_t* tab, unsigned log_n,
                                   unsigned qdb, gl_t* out, hipStream_t st) {
    size_t size = (size_t)1 << (log_n + qdb);
    hipLaunchKernelGGL(quotient_combine_kernel, dim3((unsigned)((size + 255) / 256)), dim3(256), 0, st, partial, chunk_scale, n_chunks, tab,
                       log_n, qdb, out);
    return hipGetLastError();
}

}  // namespace starkhip
