// Constraint-quotient evaluation on gfx950: the device twin of starky's compute_quotient_polys
// (SURVEY.md App. A.6), which calls S::eval_packed_generic once per LDE point
// (/root/reference/src/final_exponentiate.rs:907, src/miller_loop.rs:644,
//  src/calc_pairing_precomp.rs:376, src/fp12_mul.rs:58).
//
// The AIR arrives as the op stream of quotient_ops.h (the flat program of air_ir.h in 16-byte ops).
// One lane owns one coset point; ops are wave-uniform and are fetched four at a time with one scalar
// load, two batches ahead; their four trace-cell loads (scalar column base + per-lane 32-bit offset,
// each a coalesced 512-byte line of the coset-major LDE) are issued one batch ahead of the arithmetic,
// so neither the scalar nor the vector memory latency sits on the dependent chain.  Field arithmetic is
// the lazy-reduction form of gl_dev.h.  The program is cut into `n_chunks` pieces at group boundaries so
// that (points / 64) x n_chunks waves fill the chip; a chunk's partial fold is scaled by
// alpha^(constraints after the chunk) in the combine kernel, which is exact in the field.
#include <hip/hip_runtime.h>

#include "air_ir.h"
#include "gl_dev.h"
#include "kernels.h"
#include "quotient_ops.h"

namespace starkhip {

struct alignas(64) QBatch {
    QOp op[QOP_BATCH];
};

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
    const QBatch* ops;          // compile_quotient_ops() output, batches of QOP_BATCH ops
    const uint32_t* chunk_batch;// [n_chunks + 1] first batch of each chunk
    const gl_t* pis;
    const gl_t* lde;            // [C][N] coset-major
    const gl_t* tab;            // quotient_tables_kernel output
    const gl_t* apow;           // [2][AIR_MAX_GROUP + 1] powers of alpha_0 / alpha_1
    gl_t* partial;              // [n_chunks][2][size]
    gl_t alpha0, alpha1;
    unsigned log_n, rate_bits, qdb;
};

// The four cell loads of one batch: scalar base (lde + col * N * 8, SALU) + 32-bit per-lane byte offset.
__device__ __forceinline__ void quotient_issue_loads(const QBatch& b, const char* __restrict__ lde, unsigned col_shift, uint32_t boff_local,
                                                     uint32_t boff_next, gl_t (&x)[QOP_BATCH]) {
#pragma unroll
    for (unsigned i = 0; i < QOP_BATCH; i++) {
        const uint32_t ref = b.op[i].ref;
        const char* base = lde + ((uint64_t)(ref & REF_COL_MASK) << col_shift);
        x[i] = *(const gl_t*)(base + ((ref & REF_NEXT) ? boff_next : boff_local));
    }
}

__global__ __launch_bounds__(64) void quotient_eval_kernel(QuotientParams P) {
    const size_t n = (size_t)1 << P.log_n, size = n << P.qdb;
    const unsigned t_raw = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t_raw < size;  // domains smaller than a wave (FP12Mul: 32 points): idle lanes shadow point 0
    const unsigned t = live ? t_raw : 0;
    const unsigned chunk = blockIdx.y;
    const unsigned sp = t >> P.log_n, k = t & (unsigned)(n - 1);
    const unsigned s = sp << (P.rate_bits - P.qdb);  // LDE coset of this quotient point
    const uint32_t boff_local = (s * (unsigned)n + k) * 8u;                           // < 2^32: N * 8 <= 2^(13+3+3)
    const uint32_t boff_next = (s * (unsigned)n + ((k + 1) & (unsigned)(n - 1))) * 8u;
    const unsigned col_shift = P.log_n + P.rate_bits + 3;
    const gl_t mask_tr = P.tab[t], mask_first = P.tab[size + t], mask_last = P.tab[2 * size + t];

    const QBatch* __restrict__ ops = P.ops;
    const gl_t* __restrict__ pis = P.pis;
    const gl_t* __restrict__ apow = P.apow;
    const char* __restrict__ lde = (const char*)P.lde;
    const gl_t a0 = P.alpha0, a1 = P.alpha1;
    unsigned b = P.chunk_batch[chunk];
    const unsigned b_end = P.chunk_batch[chunk + 1];

    gl_t acc0 = 0, acc1 = 0, t0 = 0, t1 = 0, G = 1, body = 0, v = 1;
    // software pipeline: ops of batch b+2 and cells of batch b+1 are in flight while batch b is evaluated
    QBatch cur = ops[b];
    gl_t x_cur[QOP_BATCH];
    quotient_issue_loads(cur, lde, col_shift, boff_local, boff_next, x_cur);
    QBatch nxt = ops[b + 1];
    for (; b < b_end; b++) {
        gl_t x_nxt[QOP_BATCH];
        quotient_issue_loads(nxt, lde, col_shift, boff_local, boff_next, x_nxt);
        const QBatch nn = ops[b + 2];
#pragma unroll
        for (unsigned i = 0; i < QOP_BATCH; i++) {
            const uint32_t hdr = cur.op[i].hdr;
            const gl_t x = x_cur[i];  // canonical (LDE output)
            switch (hdr & 7u) {
                case QOP_TERM: {
                    gl_t u = (hdr & QOP_NOCELL) ? (gl_t)1 : x;
                    if (hdr & QOP_PREV) u = gl_canon(gl_mul_nc(v, x));
                    const uint32_t ck = (hdr >> QOP_CK_SHIFT) & 7u;
                    if (ck == CK_PLUS) body = gl_add_nc(body, u);
                    else if (ck == CK_MINUS) body = gl_sub_nc(body, u);
                    else {
                        gl_t kk = cur.op[i].k;
                        if (ck != CK_CONST) {
                            kk = pis[hdr >> QOP_IDX_SHIFT];
                            if (ck == CK_NEG_PI) kk = kk ? GL_P - kk : 0;
                        }
                        body = gl_mad_nc(u, kk, body);
                    }
                    if (hdr & QOP_FOLD) {
                        t0 = gl_mad_nc(t0, a0, body);
                        t1 = gl_mad_nc(t1, a1, body);
                        body = 0;
                    }
                    break;
                }
                case QOP_FACTOR:
                    v = (hdr & QOP_PREV) ? gl_mul_nc(v, x) : x;
                    break;
                case QOP_GATE:
                    G = gl_mul_nc(G, (cur.op[i].ref & REF_COMPL) ? gl_sub_nc(1, x) : x);
                    break;
                case QOP_GROUP: {
                    const uint32_t kind = (hdr >> QOP_KIND_SHIFT) & 3u;
                    G = kind == KIND_PLAIN ? (gl_t)1 : kind == KIND_TRANSITION ? mask_tr : kind == KIND_FIRST ? mask_first : mask_last;
                    t0 = 0;
                    t1 = 0;
                    break;
                }
                case QOP_ENDGROUP: {
                    const uint32_t m = hdr >> QOP_IDX_SHIFT;
                    acc0 = gl_mad_nc(acc0, apow[m], gl_mul_nc(G, t0));
                    acc1 = gl_mad_nc(acc1, apow[(AIR_MAX_GROUP + 1) + m], gl_mul_nc(G, t1));
                    break;
                }
                default:
                    break;
            }
        }
        cur = nxt;
        nxt = nn;
#pragma unroll
        for (unsigned i = 0; i < QOP_BATCH; i++) x_cur[i] = x_nxt[i];
    }
    if (live) {
        P.partial[((size_t)chunk * 2 + 0) * size + t] = gl_canon(acc0);
        P.partial[((size_t)chunk * 2 + 1) * size + t] = gl_canon(acc1);
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

hipError_t launch_quotient_eval(const QOp* ops, const uint32_t* chunk_batch, unsigned n_chunks, const gl_t* pis, const gl_t* lde, const gl_t* tab,
                                const gl_t* apow, gl_t alpha0, gl_t alpha1, gl_t* partial, unsigned log_n, unsigned rate_bits, unsigned qdb,
                                hipStream_t st) {
    QuotientParams P;
    P.ops = (const QBatch*)ops; P.chunk_batch = chunk_batch; P.pis = pis; P.lde = lde; P.tab = tab; P.apow = apow; P.partial = partial;
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
