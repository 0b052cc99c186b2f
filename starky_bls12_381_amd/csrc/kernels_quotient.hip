// Constraint-quotient evaluation on gfx950: the device twin of starky's compute_quotient_polys
// (SURVEY.md App. A.6), which calls S::eval_packed_generic once per LDE point
// (/root/reference/src/final_exponentiate.rs:907, src/miller_loop.rs:644,
//  src/calc_pairing_precomp.rs:376, src/fp12_mul.rs:58).
//
// The AIR arrives as the op stream of quotient_ops.h (the flat program of air_ir.h in 16-byte ops).
// One lane owns one coset point; ops are wave-uniform and are fetched four at a time with one scalar load.
//
// Memory: a chunk of the FinalExp program touches each trace column ~15 times, a few hundred ops apart --
// too far for L1/L2 with thousands of waves in flight (the first version fetched 249 GB per launch, 12.9x the
// LDE's 19.3 GB).  So every wave (= workgroup) keeps a cell cache in LDS, `n_slots` x 64 lanes x 8 B, whose
// slot assignment the host computed with Belady's rule (attach_cell_cache).  An op either reads its cell from a
// slot, or takes it from its global load and, if told so, writes it to a slot for later ops.  EVERY op issues
// exactly one global load (ops that need none load column 0, an L1 hit), four batches ahead of its use, so the
// s_waitcnt counts are compile-time constants and the HBM / MALL latency sits behind three batches of work; LDS
// reads are issued one batch ahead.  The batch loop is unrolled four times so both staging rings are registers.
//
// Field arithmetic is the lazy-reduction form of gl_dev.h.  The program is cut into `n_chunks` pieces at group
// boundaries so that (points / 64) x n_chunks waves fill the chip; a chunk's partial fold is scaled by
// alpha^(constraints after the chunk) in the combine kernel, which is exact in the field.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "air_ir.h"
#include "gl_dev.h"
#include "kernels.h"
#include "quotient_ops.h"

namespace starkhip {

struct alignas(64) QBatch {
    QOp op[QOP_BATCH];
};
struct alignas(16) QLoads {
    uint32_t ref[QOP_BATCH];
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
    const QBatch* ops;           // compile_quotient_ops() + attach_cell_cache() output, batches of QOP_BATCH ops
    const QLoads* loads;         // per batch, the four cellrefs to load
    const uint32_t* chunk_batch; // [n_chunks + 1] first batch of each chunk
    const gl_t* pis;
    const gl_t* lde;             // [C][N] coset-major
    const gl_t* tab;             // quotient_tables_kernel output
    const gl_t* apow;            // [2][AIR_MAX_GROUP + 1] powers of alpha_0 / alpha_1
    gl_t* partial;               // [n_chunks][2][size]
    gl_t alpha0, alpha1;
    unsigned log_n, rate_bits, qdb;
};

// per-lane evaluation state
struct QState {
    gl_t acc0, acc1, t0, t1, G, body, v;
};
struct QLane {
    const char* lde;
    const gl_t* pis;
    const gl_t* apow;
    gl_t a0, a1, mask_tr, mask_first, mask_last;
    uint32_t boff_local, boff_next;
    unsigned col_shift;
};

// The four (always issued) cell loads of one batch: scalar base (lde + col * N * 8, SALU) + 32-bit per-lane byte offset.
__device__ __forceinline__ void quotient_issue_loads(const QLoads& r, const QLane& L, gl_t (&x)[QOP_BATCH]) {
#pragma unroll
    for (unsigned i = 0; i < QOP_BATCH; i++) {
        const uint32_t ref = r.ref[i];
        const char* base = L.lde + ((uint64_t)(ref & REF_COL_MASK) << L.col_shift);
        x[i] = *(const gl_t*)(base + ((ref & REF_NEXT) ? L.boff_next : L.boff_local));
    }
}

// LDS reads of the ops of one batch that take their cell from the cache (CACHE = false: the program has no such ops)
template <bool CACHE>
__device__ __forceinline__ void quotient_cache_reads(const QBatch& b, const gl_t* cache_lane, gl_t (&x)[QOP_BATCH]) {
    if constexpr (CACHE) {
#pragma unroll
        for (unsigned i = 0; i < QOP_BATCH; i++) {
            const uint32_t ref = b.op[i].ref;
            if (ref & QREF_FROM_LDS) x[i] = cache_lane[(ref & QREF_SLOT_MASK) << 6];
        }
    }
}

template <bool CACHE>
__device__ __forceinline__ void quotient_eval_batch(const QBatch& cur, const gl_t (&xg)[QOP_BATCH], const gl_t (&xl)[QOP_BATCH], gl_t* cache_lane,
                                                    const QLane& L, QState& S) {
#pragma unroll
    for (unsigned i = 0; i < QOP_BATCH; i++) {
        const uint32_t hdr = cur.op[i].hdr, ref = cur.op[i].ref;
        gl_t x = xg[i];  // canonical (LDE output)
        if constexpr (CACHE) {
            x = (ref & QREF_FROM_LDS) ? xl[i] : xg[i];
            if (ref & QREF_STORE) cache_lane[(ref & QREF_SLOT_MASK) << 6] = x;
        }
        // The kernel is bound by SCALAR instruction issue (decode, compare, branch: one per SIMD every four cycles, the
        // same rate as the vector ALU), so 92 % of the ops -- a term that is +-1 or a constant times one cell -- take a short
        // path of two or three tests; everything else goes through a sequence of independent `if`s, each updating its own
        // part of the state in place (a switch there costs register-to-register copies of the state at the merge points).
        const uint32_t op = hdr & 7u;
        if (hdr & QOP_SIMPLE) {
            if (((hdr >> QOP_CK_SHIFT) & 7u) == CK_CONST) {
                S.body = gl_mad_nc_ub(x, cur.op[i].k, S.body);
            } else {
                // +x and -x on one path: (x ^ m) + c with wave-uniform (m, c) = (0, 0) or (~0, p + 1), i.e. x or p - x
                // (x is canonical; p - 0 = p is a harmless alias of 0 for the single-correction add)
                const bool minus = ((hdr >> QOP_CK_SHIFT) & 7u) == CK_MINUS;
                const uint64_t m = minus ? ~0ull : 0ull, c = minus ? GL_P + 1 : 0ull;
                S.body = gl_add_nc(S.body, (x ^ m) + c);
            }
            if (hdr & QOP_FOLD) {
                S.t0 = gl_mad_nc_ub(S.t0, L.a0, S.body);
                S.t1 = gl_mad_nc_ub(S.t1, L.a1, S.body);
                S.body = 0;
            }
            continue;
        }
        if (op == QOP_TERM) {
            gl_t u = x;  // the common case (92 % of the terms): one cell, no earlier factor
            if (hdr & (QOP_NOCELL | QOP_PREV)) u = (hdr & QOP_NOCELL) ? (gl_t)1 : gl_canon(gl_mul_nc(S.v, x));
            const uint32_t ck = (hdr >> QOP_CK_SHIFT) & 7u;
            if (ck == CK_PLUS) S.body = gl_add_nc(S.body, u);
            else if (ck == CK_MINUS) S.body = gl_sub_nc(S.body, u);
            else {
                gl_t kk = cur.op[i].k;
                if (ck != CK_CONST) {
                    kk = L.pis[hdr >> QOP_IDX_SHIFT];
                    if (ck == CK_NEG_PI) kk = kk ? GL_P - kk : 0;
                }
                S.body = gl_mad_nc_ub(u, kk, S.body);
            }
        }
        if (hdr & QOP_FOLD) {  // only ever set on a TERM: the constraint is complete
            S.t0 = gl_mad_nc_ub(S.t0, L.a0, S.body);
            S.t1 = gl_mad_nc_ub(S.t1, L.a1, S.body);
            S.body = 0;
        }
        if (op == QOP_FACTOR) S.v = (hdr & QOP_PREV) ? gl_mul_nc(S.v, x) : x;
        if (op == QOP_GATE) {
            gl_t g = x;
            if (ref & REF_COMPL) g = gl_sub_nc(1, x);
            S.G = gl_mul_nc(S.G, g);
        }
        if (op == QOP_GROUP) {
            const uint32_t kind = (hdr >> QOP_KIND_SHIFT) & 3u;
            S.G = kind == KIND_PLAIN ? (gl_t)1 : kind == KIND_TRANSITION ? L.mask_tr : kind == KIND_FIRST ? L.mask_first : L.mask_last;
            S.t0 = 0;
            S.t1 = 0;
        }
        if (op == QOP_ENDGROUP) {
            const uint32_t m = hdr >> QOP_IDX_SHIFT;
            S.acc0 = gl_mad_nc(S.acc0, L.apow[m], gl_mul_nc(S.G, S.t0));
            S.acc1 = gl_mad_nc(S.acc1, L.apow[(AIR_MAX_GROUP + 1) + m], gl_mul_nc(S.G, S.t1));
        }
    }
}

template <bool CACHE>
__global__ __launch_bounds__(64) void quotient_eval_kernel(QuotientParams P) {
    extern __shared__ gl_t cell_cache[];  // [n_slots][64]
    const size_t n = (size_t)1 << P.log_n, size = n << P.qdb;
    const unsigned t_raw = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t_raw < size;  // domains smaller than a wave (FP12Mul: 32 points): idle lanes shadow point 0
    const unsigned t = live ? t_raw : 0;
    const unsigned chunk = blockIdx.y;
    const unsigned sp = t >> P.log_n, k = t & (unsigned)(n - 1);
    const unsigned s = sp << (P.rate_bits - P.qdb);  // LDE coset of this quotient point
    QLane L;
    L.lde = (const char*)P.lde;
    L.pis = P.pis;
    L.apow = P.apow;
    L.a0 = P.alpha0;
    L.a1 = P.alpha1;
    L.mask_tr = P.tab[t];
    L.mask_first = P.tab[size + t];
    L.mask_last = P.tab[2 * size + t];
    L.boff_local = (s * (unsigned)n + k) * 8u;  // < 2^32: N * 8 <= 2^(13+3+3)
    L.boff_next = (s * (unsigned)n + ((k + 1) & (unsigned)(n - 1))) * 8u;
    L.col_shift = P.log_n + P.rate_bits + 3;
    gl_t* cache_lane = cell_cache + threadIdx.x;

    const QBatch* __restrict__ ops = P.ops;
    const QLoads* __restrict__ loads = P.loads;
    unsigned b = P.chunk_batch[chunk];
    const unsigned b_end = P.chunk_batch[chunk + 1];  // b_end - b is a multiple of QOP_UNROLL

    QState S = {0, 0, 0, 0, 1, 0, 1};
    // the two staging rings as separately named arrays, so that every index is a constant and they stay in registers
    static_assert(QOP_UNROLL == 4, "the step sequence below is written out for a ring of four");
    gl_t xg0[QOP_BATCH], xg1[QOP_BATCH], xg2[QOP_BATCH], xg3[QOP_BATCH], xl0[QOP_BATCH], xl1[QOP_BATCH];
    quotient_issue_loads(loads[b + 0], L, xg0);
    quotient_issue_loads(loads[b + 1], L, xg1);
    quotient_issue_loads(loads[b + 2], L, xg2);
    quotient_issue_loads(loads[b + 3], L, xg3);
    QLoads rn = loads[b + QOP_UNROLL];
    QBatch cur = ops[b], nxt = ops[b + 1];
#pragma unroll
    for (unsigned i = 0; i < QOP_BATCH; i++) xl0[i] = xl1[i] = 0;
#define QSTEP(U, XG, XL_CUR, XL_NXT)                                                                         \
    {                                                                                                        \
        quotient_cache_reads<CACHE>(nxt, cache_lane, XL_NXT); /* batch b+U+1, before this batch's slot writes */ \
        const QBatch nn = ops[b + (U) + 2];                                                                  \
        quotient_eval_batch<CACHE>(cur, XG, XL_CUR, cache_lane, L, S);                                       \
        quotient_issue_loads(rn, L, XG); /* batch b+U+4 takes over this ring entry */                        \
        rn = loads[b + (U) + QOP_UNROLL + 1];                                                                \
        cur = nxt;                                                                                           \
        nxt = nn;                                                                                            \
    }
    for (; b < b_end; b += QOP_UNROLL) {
        QSTEP(0, xg0, xl0, xl1)
        QSTEP(1, xg1, xl1, xl0)
        QSTEP(2, xg2, xl0, xl1)
        QSTEP(3, xg3, xl1, xl0)
    }
#undef QSTEP
    if (live) {
        P.partial[((size_t)chunk * 2 + 0) * size + t] = gl_canon(S.acc0);
        P.partial[((size_t)chunk * 2 + 1) * size + t] = gl_canon(S.acc1);
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

hipError_t launch_quotient_eval(const QOp* ops, const uint32_t* loads, unsigned n_slots, const uint32_t* chunk_batch, unsigned n_chunks,
                                const gl_t* pis, const gl_t* lde, const gl_t* tab, const gl_t* apow, gl_t alpha0, gl_t alpha1, gl_t* partial,
                                unsigned log_n, unsigned rate_bits, unsigned qdb, hipStream_t st) {
    QuotientParams P;
    P.ops = (const QBatch*)ops; P.loads = (const QLoads*)loads; P.chunk_batch = chunk_batch; P.pis = pis; P.lde = lde; P.tab = tab;
    P.apow = apow; P.partial = partial;
    P.alpha0 = alpha0; P.alpha1 = alpha1; P.log_n = log_n; P.rate_bits = rate_bits; P.qdb = qdb;
    size_t size = (size_t)1 << (log_n + qdb);
    const size_t lds_bytes = (size_t)(n_slots ? n_slots : 1) * 64 * sizeof(gl_t);
    if (n_slots)
        hipLaunchKernelGGL(quotient_eval_kernel<true>, dim3((unsigned)((size + 63) / 64), n_chunks), dim3(64), lds_bytes, st, P);
    else
    {
        static const size_t pad = [] { const char* e = getenv("STARKHIP_QUOTIENT_LDS_PAD"); return e ? (size_t)atol(e) : (size_t)0; }();
        hipLaunchKernelGGL(quotient_eval_kernel<false>, dim3((unsigned)((size + 63) / 64), n_chunks), dim3(64), pad, st, P);
    }
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
