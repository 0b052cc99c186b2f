// Constraint-quotient evaluation on gfx950: the device twin of starky's compute_quotient_polys
// (SURVEY.md App. A.6), which calls S::eval_packed_generic once per LDE point
// (/root/reference/src/final_exponentiate.rs:907, src/miller_loop.rs:644,
//  src/calc_pairing_precomp.rs:376, src/fp12_mul.rs:58).
//
// Two evaluators live here; both produce exactly the reference's fold  acc_j = sum_k mask_k c_k alpha_j^(K-1-k).
//
//  * quotient_tiles_kernel (default): the tiled plan of quotient_plan.h.  Constraints are regrouped by (kind, gates) and by
//    64-column tile, with per-proof alpha weights; a workgroup stages each tile's 64-point slice in LDS once (a producer wave,
//    direct-to-LDS loads) and seven waves run wave-uniform record streams over it: every LDE cell is read from HBM once
//    (FinalExp: 30 GB fetched per launch against 253 GB for the interpreter; 29 ms against 40 ms).  Measured on MI355X:
//    vector and scalar instructions of a SIMD's waves do NOT overlap for this kind of code -- time = (VALU + SALU + LDS
//    instructions) x 4 cycles per SIMD -- so the kernel is written for total instruction count: 12 multiply-adds per record,
//    everything else amortised (details at the kernel).
//  * quotient_eval_kernel (ctx option "quotient_impl" = 1): the op-stream interpreter of round 1, one global load per op.
//    Kept as the second implementation the tests cross-check the first against on the GPU.
//
// Interpreter: the AIR arrives as the op stream of quotient_ops.h (the flat program of air_ir.h in 16-byte ops).
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
#include "quotient_plan.h"

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
__global__ void quotient_tables_kernel(gl_t* tab, unsigned log_n, unsigned qdb) { STARKHIP_PRIO_ENTRY
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
__global__ __launch_bounds__(64) void quotient_eval_kernel(QuotientParams P) { STARKHIP_PRIO_ENTRY
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
                                        const gl_t* __restrict__ tab, unsigned log_n, unsigned qdb, gl_t* __restrict__ out) { STARKHIP_PRIO_ENTRY
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


// =====================================================================================================================
// Tiled evaluator (quotient_plan.h): LDS-staged column tiles, one pass over the LDE.
//
// A workgroup of QT_WAVES waves owns 64 coset points and one chunk of the plan.  For every tile of the chunk it stages
// the 64-point slice of QT_TILE_COLS columns (+ the successor row) in LDS -- double buffered: the next tile's cells are
// requested before the current tile's records are processed and stored afterwards, one barrier per tile -- and each
// wave runs its own stream of wave-uniform records over the staged cells:
//     x = cell (LDS), product of cells, or 1;   S_j[i][l] += x_i * w_j,l     (x = x_1 2^32 + x_0, w_j = sum_l w_j,l 2^(22 l))
// i.e. twelve v_mad_u64_u32 with the weight limb as scalar operand and no carries (each product < 2^54, a piece has at most
// 96 records); at the end of a piece the six sums of each alpha are folded mod p, multiplied by mask * G and added to acc_j.
// Gate cells and factors outside the tile are direct loads; a piece's gate cells are requested when the previous piece ends.
struct QTParams {
    const QTRec* recs;
    const QTStream* streams;          // [n_chunks][QT_WAVES]
    const uint32_t* chunk_tile_off;   // [n_chunks + 1]
    const uint32_t* tile_list;
    const gl_t* lde;                  // [C][N] coset-major
    const gl_t* tab;                  // quotient_tables_kernel output
    gl_t* partial;                    // [n_chunks][2][size]
    unsigned log_n, rate_bits, qdb, n_cols;
    unsigned dbg;  // profiling only: 1 = the producer loads nothing, 2 = the evaluators skip the arithmetic (results are garbage)
};

// (hi:lo) += (xh:xl) as one carry chain (the sums involved stay far below 2^128)
__device__ __forceinline__ void qt_add128(uint64_t& lo, uint64_t& hi, uint64_t xl, uint64_t xh) {
    uint32_t l0 = (uint32_t)lo, l1 = (uint32_t)(lo >> 32), h0 = (uint32_t)hi, h1 = (uint32_t)(hi >> 32);
    asm("v_add_co_u32_e32 %0, vcc, %4, %0\n\t"
        "v_addc_co_u32_e32 %1, vcc, %5, %1, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, %6, %2, vcc\n\t"
        "v_addc_co_u32_e32 %3, vcc, %7, %3, vcc"
        : "+v"(l0), "+v"(l1), "+v"(h0), "+v"(h1)
        : "v"((uint32_t)xl), "v"((uint32_t)(xl >> 32)), "v"((uint32_t)xh), "v"((uint32_t)(xh >> 32))
        : "vcc");
    lo = ((uint64_t)l1 << 32) | l0;
    hi = ((uint64_t)h1 << 32) | h0;
}

// sum_l 2^(22 l) * (S[l] + 2^32 S[3 + l]) mod p, any representative.  Every S < QT_MAX_CHAIN * 2^54 = 0.9375 * 2^64
// (a chain carried across tiles holds up to QT_MAX_CHAIN = 960 products < 2^54; quotient_plan.h has the derivation).
//   U = S0 + S1 2^22 + S2 2^44,  V = S3 + S4 2^22 + S5 2^44  (both < 2^106),  total = U + V 2^32 = sum_k t_k 2^(32 k), k < 5
//   total mod p = (t1:t0) - (t4:t3) + t2 eps          (2^64 = eps, 2^96 = -1, 2^128 = -2^32)
// with the borrow / carry settled as in gl_reduce_words (the subtrahend is < 2^42, so the same bounds hold).
__device__ __forceinline__ gl_t qt_fold_sums(const uint64_t (&S)[6]) {
    // The five 32-bit words t0 .. t4 of the total, by word position, as chains of 32 x 32 + 64 multiply-adds (the shifts by
    // 22, 44, 54 and 76 bits become multiplications by 2^22 / 2^12 at word offsets; every chain stays below (0.9375 + 2^-9) 2^64):
    //   position  0:  S0 + S1_lo 2^22                                            -> (x1 : t0)
    //   position 32:  x1 + S3 + S1_hi 2^22 + S2_lo 2^12 + S4_lo 2^22               -> (y1 : t1)
    //   position 64:  y1 + S2_hi 2^12 + S4_hi 2^22 + S5_lo 2^12                    -> (z1 : t2)
    //   position 96:  z1 + S5_hi 2^12                                              -> (t4 : t3)
    // eleven multiply-adds instead of twelve 64-bit shifts and twenty carry instructions.
    uint32_t k22 = 1u << QT_LIMB_BITS, k12 = 1u << (2 * QT_LIMB_BITS - 32), k1 = 1u;
    asm("" : "+s"(k22), "+s"(k12), "+s"(k1));  // opaque: as known powers of two the products come back as shift + carry chains
    auto lo = [](uint64_t v) { return (uint32_t)v; };
    auto hi = [](uint64_t v) { return (uint32_t)(v >> 32); };
    auto mad = [](uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * b + c; };
    const uint64_t X = mad(lo(S[1]), k22, S[0]);
    uint64_t Y = mad(hi(S[1]), k22, S[3]);
    Y = mad(lo(S[2]), k12, Y);
    Y = mad(lo(S[4]), k22, Y);
    Y = mad(hi(X), k1, Y);
    uint64_t Z = mad(hi(S[2]), k12, 0);
    Z = mad(hi(S[4]), k22, Z);
    Z = mad(lo(S[5]), k12, Z);
    Z = mad(hi(Y), k1, Z);
    uint64_t Wd = mad(hi(S[5]), k12, 0);
    Wd = mad(hi(Z), k1, Wd);
    const uint32_t t0 = lo(X), t1 = lo(Y), t2 = lo(Z), t3 = lo(Wd), t4 = hi(Wd);
    // D = (t1:t0) - (t4:t3), borrow b;  r = D + t2 * eps, carry c;  result r + (c - b) * eps
    uint32_t d0, d1, t;
    uint64_t borrow_mask, carry_mask, scratch_mask, r;
    asm("v_sub_co_u32_e64 %0, %2, %3, %4\n\ts_nop 1\n\tv_subb_co_u32_e64 %1, %2, %5, %6, %2"
        : "=&v"(d0), "=&v"(d1), "=&s"(borrow_mask)
        : "v"(t0), "v"(t3), "v"(t1), "v"(t4));
    const uint64_t D = ((uint64_t)d1 << 32) | d0;
    asm("v_mad_u64_u32 %0, %2, %4, -1, %5\n\t"
        "v_subb_co_u32_e64 %1, %3, 0, 0, %6\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %1, %3, %1, 0, %2\n\t"
        "v_mad_i64_i32 %0, %3, %1, -1, %0"
        : "=&v"(r), "=&v"(t), "=&s"(carry_mask), "=&s"(scratch_mask)
        : "v"(t2), "v"(D), "s"(borrow_mask));
    return r + ((uint64_t)t << 32);
}

#ifdef STARKHIP_QT_PROF  // `make variant NAME=qprof DEFS=-DSTARKHIP_QT_PROF`: per-wave clocks of the tiled evaluator (tools/quotient_wave_prof.py)
// [workgroup][wave][4]: cycles from the wave's start to its end, cycles between arriving at a tile barrier and leaving it, barriers,
// (producer wave only) cycles waiting for its tile loads
__device__ unsigned long long qt_prof[16384 * (QT_WAVES + 1) * 4];
// [chunk][wave][tile < 192]: cycles an evaluator wave of the workgroups with blockIdx.x == 0 worked on each tile (barrier to barrier)
__device__ unsigned long long qt_tile_prof[64 * QT_WAVES * 192];
#define QT_PROF_CLOCK() __builtin_amdgcn_s_memtime()
#endif

template <bool SMALL_N, unsigned DBG>
__global__ __launch_bounds__(64 * (QT_WAVES + 1), 4) void quotient_tiles_kernel(QTParams P) { STARKHIP_PRIO_ENTRY
    __shared__ gl_t tile[2][(QT_TILE_COLS + 1) * QT_TILE_ROWS];  // + the column of ones (QT_ONES_SLOT: constant terms are plain records)
    __shared__ uint32_t rec_ring[QT_WAVES][3][32][4];  // per evaluating wave: three blocks of 16 records (32 x 16 bytes each)
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // 0 .. QT_WAVES - 1 evaluate; SMALL_N: wave QT_WAVES stages the tiles
    const unsigned lane = threadIdx.x & 63u;
    const size_t n = (size_t)1 << P.log_n, size = n << P.qdb;
    const unsigned t_raw = blockIdx.x * 64u + lane;
    const bool live = t_raw < size;  // domains smaller than a wave (FP12Mul: 32 points): idle lanes shadow point 0
    const unsigned t = live ? t_raw : 0;
    const unsigned chunk = blockIdx.y;
    const unsigned sp = t >> P.log_n, k = t & (unsigned)(n - 1);
    const unsigned s = sp << (P.rate_bits - P.qdb);  // LDE coset of this quotient point
    const unsigned k_next = (k + 1) & (unsigned)(n - 1);
    const char* const lde = (const char*)P.lde;
    const unsigned col_shift = P.log_n + P.rate_bits + 3;
    const uint32_t boff_local = (s * (unsigned)n + k) * 8u;  // < 2^32: N * 8 <= 2^(13+3+3)
    const uint32_t boff_next = (s * (unsigned)n + k_next) * 8u;
    const uint32_t* tiles = P.tile_list + P.chunk_tile_off[chunk];
    const unsigned n_tiles = P.chunk_tile_off[chunk + 1] - P.chunk_tile_off[chunk];

    if (SMALL_N && wave == QT_WAVES) {
        // ---- producer wave of the SMALL_N kernels (n < 64: FP12Mul's 16 rows; a wave's lanes span several cosets, so a column is not
        // one contiguous run and goes through registers): tile ti + 1 is staged while the evaluating waves work on tile ti
        for (unsigned buf = 0; buf < 2; buf++) {  // the column of ones of both buffers, once (visible after the first barrier)
            tile[buf][QT_ONES_SLOT * QT_TILE_ROWS + lane] = 1;
            if (lane < QT_TILE_ROWS - 64) tile[buf][QT_ONES_SLOT * QT_TILE_ROWS + 64 + lane] = 1;
        }
        const uint32_t boff_next_last = __builtin_amdgcn_readlane(boff_next, 63);  // successor of the block's last point: row 64
        for (unsigned ti = 0; ti <= n_tiles; ti++) {
            if (ti < n_tiles && !(DBG & 1u)) {
                const uint32_t c0 = tiles[ti] * QT_TILE_COLS;
                gl_t* dst = tile[ti & 1u];
                const uint32_t ecol = c0 + lane;  // row 64: one lane per column
                gl_t ext = 0;
                if (ecol < P.n_cols) ext = *(const gl_t*)(lde + ((uint64_t)ecol << col_shift) + boff_next_last);
#pragma unroll 1
                for (unsigned b = 0; b < QT_TILE_COLS; b += 16) {
                    gl_t v[16];
#pragma unroll
                    for (unsigned i = 0; i < 16; i++) {
                        const uint32_t col = c0 + b + i;
                        v[i] = 0;
                        if (col < P.n_cols) v[i] = *(const gl_t*)(lde + ((uint64_t)col << col_shift) + boff_local);
                    }
#pragma unroll
                    for (unsigned i = 0; i < 16; i++) dst[(b + i) * QT_TILE_ROWS + lane] = v[i];
                }
                dst[lane * QT_TILE_ROWS + 64] = ext;
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");  // the evaluators' reduction barrier
        return;
    }
    // ---- n >= 64 (every AIR but FP12Mul): no producer wave.  Round 5 had seven evaluating waves and a producer per workgroup; with
    // two workgroups on a CU that is 4 + 4 + 4 + 2 evaluating waves on the four SIMDs, the fourth SIMD half idle while the others
    // run at the vector pipe's rate (per-wave clocks: tools/quotient_wave_prof.py; the evaluators of the crowded SIMDs reached every
    // barrier last).  Now EIGHT waves evaluate -- four per SIMD -- and each stages an eighth of the next tile itself: eight
    // global_load_lds_dwordx4 per tile, 33 lanes each -- lanes 0 .. 31 the 64 points of the column (512 contiguous bytes), lane 32
    // the successor of the last point (row 64, the first half of its 16 bytes; contiguous too except in a coset's last block) --
    // issued when the wave enters tile ti, for tile ti + 1, into the other buffer; the wave waits for them before the barrier
    // that ends tile ti.  No vector registers are held; the tile's first column rides in the QT_TILE record.
    const uint32_t boff_next_last = __builtin_amdgcn_readlane(boff_next, 63);
    const uint32_t boff_block = __builtin_amdgcn_readfirstlane(boff_local);  // the 64 points are 512 contiguous bytes
    const uint32_t tile_lds = (uint32_t)(uintptr_t)&tile[0][0];
    auto tile_fill = [&](uint32_t c0, unsigned buf) {  // both wave-uniform
        if (SMALL_N || (DBG & 1u)) return;
        if (lane < 33) {
            const uint32_t off = lane < 32 ? boff_block + lane * 16u : boff_next_last;
#pragma unroll
            for (unsigned i = 0; i < QT_TILE_COLS / QT_WAVES; i++) {
                const uint32_t slot = wave * (QT_TILE_COLS / QT_WAVES) + i, col = c0 + slot;
                if (col < P.n_cols) {
                    const char* base = lde + ((uint64_t)col << col_shift);
                    const uint32_t dst = tile_lds + (buf * (QT_TILE_COLS + 1) + slot) * (uint32_t)(QT_TILE_ROWS * sizeof(gl_t));
                    uint32_t keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep)
                                 : "v"(off), "s"(base), "s"(dst)
                                 : "memory");
                }
            }
        }
    };
    static_assert(QT_TILE_COLS % QT_WAVES == 0, "every wave stages the same number of columns");
    if (!SMALL_N) {
        if (wave == 0)
            for (unsigned buf = 0; buf < 2; buf++) {  // the column of ones of both buffers, once (visible after the first barrier)
                tile[buf][QT_ONES_SLOT * QT_TILE_ROWS + lane] = 1;
                if (lane < QT_TILE_ROWS - 64) tile[buf][QT_ONES_SLOT * QT_TILE_ROWS + 64 + lane] = 1;
            }
        if (n_tiles) tile_fill(tiles[0] * QT_TILE_COLS, 0);
    }

#ifdef STARKHIP_QT_PROF
    const unsigned long long pt_start = QT_PROF_CLOCK();
    unsigned long long pt_bar = 0, pt_last = pt_start;
    unsigned pt_n = 0;
#endif
    const gl_t mask_tr = P.tab[t], mask_first = P.tab[size + t], mask_last = P.tab[2 * size + t];
    // LDS addressing: a record's offset already holds slot * 520 (+ 8 for the next row); the lane adds its row.  Blocks that
    // lie inside one coset with n >= 64 have "next row = lane + 1" (row 64 = successor of the last point).  Otherwise (SMALL_N:
    // n < 64, several cosets or idle lanes in a wave) the next row of a lane is the lane that holds point (sp, k + 1 mod n).
    const uint32_t lds_local = lane * 8u;
    uint32_t lds_next = lds_local;
    if (SMALL_N) {
        const unsigned t_next = sp * (unsigned)n + k_next;  // quotient-domain index of the successor
        const unsigned base = blockIdx.x * 64u;
        const unsigned row = (live && t_next >= base && t_next < base + 64u) ? t_next - base : (live ? 64u : (k_next & 63u));
        lds_next = row * 8u - 8u;  // the record offset carries + 8
    }

    // The record stream is wave-uniform.  Scalar loads would share the LGKM counter with the LDS reads (and return out of
    // order, so that every wait for a staged cell also waits for the record fetch issued just before it: 1 300 cycles per
    // record, measured); per-lane vector loads of the same address hold 8 registers per record in flight for an L2 latency.
    // So every wave streams its records through a private 1 KB ring in LDS, filled 512 bytes (16 records) at a time by a
    // direct-to-LDS load issued 16 .. 32 records ahead of use, and reads a record with two broadcast ds_read_b128.
    const QTStream stream = P.streams[chunk * QT_WAVES + wave];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4* const ring = (u32x4*)&rec_ring[wave][0][0][0];
    const char* const rec_base = (const char*)(P.recs + stream.rec_off);
    // The refill is issued from inline asm: once hipcc sees an LDS-DMA builtin in the loop it waits lgkmcnt(0) / vmcnt(0) at
    // every LDS read (the DMA may alias any of them), i.e. one full LDS round trip per record.  Its completion is waited for
    // explicitly (QT_STEP); hipcc's own vmcnt counts only over-wait because of the extra operation (returns are in order).
    const uint32_t ring_lds = (uint32_t)(uintptr_t)ring;  // LDS byte address of this wave's ring
    auto ring_fill = [&](unsigned block, unsigned slot) {  // records 16 block .. 16 block + 15 -> ring slot (= block % 3)
        if (lane < 32) {
            const char* src = rec_base + (size_t)block * 512u + lane * 16u;
            const uint32_t dst = ring_lds + slot * 512u;
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(src), "s"(dst)
                         : "memory");
        }
    };
    struct Rec {
        u32x4 a, b;  // {ctl, aux, w0[0], w0[1]}, {w0[2], w1[0], w1[1], w1[2]}
    };
    // a record's ring position: byte offset `base` of its group of four + 32 * (index in the group); groups never straddle
    // the ring's end (48 records)
    auto ring_read = [&](uint32_t base, unsigned i, Rec& r) {
        const u32x4* q = (const u32x4*)((const char*)ring + base) + i * 2u;
        r.a = q[0];
        r.b = q[1];
    };
    auto direct = [&](uint32_t col, bool next) -> gl_t {
        const char* base = lde + ((uint64_t)col << col_shift);
        return *(const gl_t*)(base + (next ? boff_next : boff_local));
    };

    gl_t acc0 = 0, acc1 = 0, v = 1;
    uint64_t S0[6] = {0, 0, 0, 0, 0, 0}, S1[6] = {0, 0, 0, 0, 0, 0};
    uint32_t piece_ctl = 0;
    gl_t gate[4] = {0, 0, 0, 0};
    // The descriptor of the piece that starts (a QT_DESC record: QTPiece in the weight words, quotient_plan.h): its gate cells -- and
    // the cells absorbed from other tiles -- are requested now and used when the piece ends.  The record has been in registers for
    // four steps, so nothing here waits (round 5 fetched descriptors from an array of their own: a dependent L2 round trip per piece).
    auto gates_request = [&](const Rec& d) {
        const uint32_t dgate[4] = {d.a.z, d.a.w, d.b.x, d.b.y};
        piece_ctl = __builtin_amdgcn_readfirstlane(d.b.z);
        const uint32_t ng = ((piece_ctl >> 2) & 7u) + ((piece_ctl >> QT_FOREIGN_SHIFT) & 7u);  // gates, then the absorbed pieces' cells (quotient_plan.h)
#pragma unroll
        for (unsigned g = 0; g < 4; g++)
            if (g < ng) {
                const uint32_t ref = __builtin_amdgcn_readfirstlane(dgate[g]);
                gate[g] = direct(ref & REF_COL_MASK, ref & REF_NEXT);
            }
        // (all four slots loaded from vector addresses, without the scalar chains and the branches, was slower: 20.45 against 20.15 ms --
        // the unused slots' loads cost more than the chains; profiles/r06_ab_experiments.txt 3)
    };
    ring_fill(0, 0);
    ring_fill(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // Record pipeline (round 4).  A record is read WHOLE from the ring (two broadcast ds_read_b128: control word, weights) four steps
    // before it is evaluated, into the register set the step that has just finished frees; its cell is requested two steps ahead, the
    // address coming from the control word in that vector register.  So nothing a step needs was requested less than two steps before.
    // 77 % of the records are PLAIN (x = a cell of the tile, accumulate; no flag set) and sit in runs -- the planner sorts nothing for
    // it, monomials of one cell simply dominate.  The length of the run that follows it rides in bits 31:24 of every special record
    // (build_quotient_plan), and a run is evaluated four records at a time by a straight-line block without a single test:
    // per record one cell read, one record read, twelve multiply-adds (the round-3 loop spent ~ 34 issue slots per record -- a
    // control-word read of its own, a v_readfirstlane, two scalar tests and the copies hipcc's flow blocks add at every merge --
    // and scalar instructions cost issue slots like vector ones here).  Everything else takes the generic step.
    Rec W0, W1, W2, W3;                   // records g .. g + 3 (slot U = g % 4 holds record g)
    gl_t x0 = 0, x1 = 0, x2 = 0, x3 = 0;  // their cells
    unsigned g = 0;                       // index of the record being evaluated
    uint32_t b1 = 128;                    // ring offset of the group of four that holds record g + 4 (entry g % 4)
    unsigned next_block = 2, next_slot = 2;
    uint32_t n_plain = 0;                 // plain records known to follow (wave-uniform)
    uint32_t n_pairs = 0;                 // fast pairs known to follow them (degree-2 monomials inside the tile: two records each)
    uint32_t n_dpairs = 0;                // direct pairs known to follow those (second factor outside the tile: a direct load)
    ring_read(0, 0, W0);
    ring_read(0, 1, W1);
    ring_read(0, 2, W2);
    ring_read(0, 3, W3);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // tile 0 is staged
    if (!SMALL_N && n_tiles > 1) tile_fill(tiles[1] * QT_TILE_COLS, 1);

    uint32_t lds_cur = lds_local;  // this lane's row in the tile buffer in use (byte offset from tile[0])
    // `c`: the control word as it came from the ring (a vector register): offset = low half, one add
    auto lds_read = [&](uint32_t c, gl_t& x) {
        uint32_t a = (c & QT_OFF_MASK) + lds_cur;
        if (SMALL_N) a = (c & QT_OFF_MASK) + ((c & QT_NEXT) ? lds_cur - lds_local + lds_next : lds_cur);
        x = *(const gl_t*)((const char*)tile[0] + a);
    };
    lds_read(W0.a.x, x0);
    lds_read(W1.a.x, x1);
    unsigned ti = 0;
#define QT_MADS(W, XV)                                                                                                \
    if (!(DBG & 2u)) {                                                                                                \
        const uint32_t xl_ = (uint32_t)(XV), xh_ = (uint32_t)((XV) >> 32);                                            \
        const uint32_t w0_[3] = {W.a.z, W.a.w, W.b.x}, w1_[3] = {W.b.y, W.b.z, W.b.w};                                \
        _Pragma("unroll") for (int l = 0; l < 3; l++) {                                                               \
            S0[l] += (uint64_t)xl_ * w0_[l];                                                                          \
            S0[3 + l] += (uint64_t)xh_ * w0_[l];                                                                      \
            S1[l] += (uint64_t)xl_ * w1_[l];                                                                          \
            S1[3 + l] += (uint64_t)xh_ * w1_[l];                                                                      \
        }                                                                                                             \
    }
    // the end of every step: record g + 4 replaces record g in its register set
#define QT_ADVANCE(U, W)                                                                                              \
    {                                                                                                                 \
        if ((U) == 0) {                                                                                               \
            if (__builtin_expect(((g + 4u) & 15u) == 0, 0)) {                                                         \
                /* the read-ahead enters a new block: it has landed (requested 16 records ago); refill the slot the   */ \
                /* readers left longest ago.  lgkmcnt(0): every read of that slot has RETURNED -- a ds_read still     */ \
                /* queued behind other waves' LDS traffic would otherwise see the refill                              */ \
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                           \
                ring_fill(next_block, next_slot);                                                                     \
                next_block++;                                                                                         \
                next_slot = next_slot == 2 ? 0 : next_slot + 1;                                                       \
            }                                                                                                         \
        }                                                                                                             \
        ring_read(b1, (U), W);                                                                                        \
        if ((U) == 3) b1 = b1 == 1408 ? 0 : b1 + 128;                                                                 \
        g++;                                                                                                          \
    }
    // one record of a plain run: (W, X) record g and its cell, (WC2, X2) of record g + 2
#define QT_PLAIN(U, W, X, WC2, X2)                                                                                    \
    {                                                                                                                 \
        lds_read(WC2.a.x, X2);                                                                                        \
        QT_MADS(W, X)                                                                                                 \
        QT_ADVANCE(U, W)                                                                                              \
    }
    // a fast pair: record g = the first factor (nothing to do but keep its cell), record g + 1 = the second factor with the weights
#define QT_PAIR_A(U, W, WC2, X2)                                                                                      \
    {                                                                                                                 \
        lds_read(WC2.a.x, X2);                                                                                        \
        QT_ADVANCE(U, W)                                                                                              \
    }
#define QT_PAIR_B(U, W, X, XA, WC2, X2)                                                                               \
    {                                                                                                                 \
        lds_read(WC2.a.x, X2);                                                                                        \
        const gl_t xp_ = gl_mul_nc(XA, X);                                                                            \
        QT_MADS(W, xp_)                                                                                               \
        QT_ADVANCE(U, W)                                                                                              \
    }
    // a direct pair: record g = the first factor (a cell of the tile), record g + 1 = the second factor, column aux, with the weights;
    // both second factors of a turn's two pairs are requested when the turn starts
#define QT_DIRECT_OF(W) direct(__builtin_amdgcn_readfirstlane(W.a.y), __builtin_amdgcn_readfirstlane(W.a.x) & QT_NEXT)
#define QT_DPAIR_B(U, W, D, XA, WC2, X2)                                                                              \
    {                                                                                                                 \
        lds_read(WC2.a.x, X2);                                                                                        \
        const gl_t xp_ = gl_mul_nc(XA, D);                                                                            \
        QT_MADS(W, xp_)                                                                                               \
        QT_ADVANCE(U, W)                                                                                              \
    }
#ifdef STARKHIP_QT_PROF
#define QT_TILE_BARRIER()                                                                                             \
    {                                                                                                                 \
        const unsigned long long pa_ = QT_PROF_CLOCK();                                                               \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                                     \
        const unsigned long long pb_ = QT_PROF_CLOCK();                                                               \
        if (blockIdx.x == 0 && chunk < 64 && pt_n < 192 && lane == 0) qt_tile_prof[(chunk * QT_WAVES + wave) * 192 + pt_n] = pa_ - pt_last; \
        pt_bar += pb_ - pa_;                                                                                          \
        pt_last = pb_;                                                                                                \
        pt_n++;                                                                                                       \
    }
#else
#define QT_TILE_BARRIER() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  /* vmcnt: this wave's share of the next tile */
#endif
    // any record: (W1c, X1) of record g + 1 as well (re-read when the tile changes)
#define QT_GENERIC(U, W, X, WC1, X1, WC2, X2)                                                                         \
    {                                                                                                                 \
        lds_read(WC2.a.x, X2);                                                                                        \
        if (n_plain != 0) { /* the tail of a run (fewer than four left) */                                            \
            n_plain--;                                                                                                \
            QT_MADS(W, X)                                                                                             \
        } else {                                                                                                      \
            const uint32_t ctl = __builtin_amdgcn_readfirstlane(W.a.x);                                               \
            const uint32_t aux = __builtin_amdgcn_readfirstlane(W.a.y);                                               \
            n_plain = ctl >> QT_RUN_SHIFT;                                                                            \
            n_pairs = (ctl & QT_SRC_GLOBAL) ? 0u : (aux & QT_AUX_PAIRS_MASK);                                         \
            n_dpairs = (ctl & QT_SRC_GLOBAL) ? 0u : ((aux >> QT_AUX_DPAIRS_SHIFT) & QT_AUX_DPAIRS_MASK);              \
            gl_t x = X;                                                                                               \
            if ((ctl & QT_ODD_SOURCE) != 0) {                                                                         \
                if (ctl & (QT_TILE | QT_STOP)) {                                                                      \
                    /* every LDS read of this tile has returned; the gate loads of the next piece stay in flight */   \
                    QT_TILE_BARRIER()                                                                                 \
                    ti++;                                                                                             \
                    if ((ctl & QT_STOP) != 0 || ti >= n_tiles) goto stream_done;                                      \
                    lds_cur = lds_local + (ti & 1u) * (uint32_t)((QT_TILE_COLS + 1) * QT_TILE_ROWS * sizeof(gl_t));         \
                    if (!SMALL_N) { /* the tile after this one, into the buffer everybody has just left */         \
                        const uint32_t c0n_ = __builtin_amdgcn_readfirstlane(W.a.z);                                  \
                        if (c0n_ != 0xFFFFFFFFu) tile_fill(c0n_, (ti + 1u) & 1u);                                     \
                    }                                                                                                 \
                    lds_read(WC1.a.x, X1); /* the cells of the next two records live in the new tile */               \
                    lds_read(WC2.a.x, X2);                                                                            \
                    goto advance_##U;                                                                                 \
                }                                                                                                     \
                if (ctl & QT_SRC_ONE) {                                                                               \
                    x = 1;                                                                                            \
                    if ((ctl & QT_NEXT) && !(DBG & 4u)) gates_request(W); /* QT_DESC: a piece starts */              \
                }                                                                                                     \
                if ((ctl & QT_SRC_GLOBAL) && !(DBG & 8u)) {                                                           \
                    if (aux & QT_AUX_SLOT) { /* a cell requested with the piece's gates */                            \
                        const uint32_t sl = aux & 3u;                                                                 \
                        x = sl == 0 ? gate[0] : sl == 1 ? gate[1] : sl == 2 ? gate[2] : gate[3];                      \
                    } else {                                                                                          \
                        x = direct(aux, ctl & QT_NEXT);                                                               \
                    }                                                                                                 \
                }                                                                                                     \
                if (ctl & QT_MULV) x = gl_mul_nc(v, x);                                                               \
                if (ctl & QT_SETV) {                                                                                  \
                    v = x;                                                                                            \
                    goto advance_##U;                                                                                 \
                }                                                                                                     \
            }                                                                                                         \
            QT_MADS(W, x)                                                                                             \
            if ((ctl & QT_END) != 0 && !(DBG & 4u) && !(DBG & 2u)) {                                                  \
                const uint32_t kind = piece_ctl & 3u, ng = (piece_ctl >> 2) & 7u, cm = (piece_ctl >> 5) & 15u;        \
                gl_t G = kind == KIND_PLAIN ? (gl_t)1 : kind == KIND_TRANSITION ? mask_tr : kind == KIND_FIRST ? mask_first : mask_last; \
                _Pragma("unroll") for (unsigned q = 0; q < 4; q++) if (q < ng) {                                      \
                    gl_t gv = gate[q];                                                                                \
                    if (cm & (1u << q)) gv = gl_sub_nc(1, gv);                                                        \
                    G = (q == 0 && kind == KIND_PLAIN) ? gv : gl_mul_nc(G, gv);                                       \
                }                                                                                                     \
                acc0 = gl_mad_nc(G, qt_fold_sums(S0), acc0);                                                          \
                acc1 = gl_mad_nc(G, qt_fold_sums(S1), acc1);                                                          \
                _Pragma("unroll") for (int l = 0; l < 6; l++) S0[l] = S1[l] = 0;                                      \
            }                                                                                                         \
        }                                                                                                             \
    advance_##U:                                                                                                      \
        QT_ADVANCE(U, W)                                                                                              \
    }
    if (n_tiles)
        for (;;) {
            while (n_plain >= 4u) {
                QT_PLAIN(0, W0, x0, W2, x2)
                QT_PLAIN(1, W1, x1, W3, x3)
                QT_PLAIN(2, W2, x2, W0, x0)
                QT_PLAIN(3, W3, x3, W1, x1)
                n_plain -= 4u;
            }
            if (n_plain == 0u)
                while (n_pairs >= 2u) {
                    QT_PAIR_A(0, W0, W2, x2)
                    QT_PAIR_B(1, W1, x1, x0, W3, x3)
                    QT_PAIR_A(2, W2, W0, x0)
                    QT_PAIR_B(3, W3, x3, x2, W1, x1)
                    n_pairs -= 2u;
                }
            if (n_plain == 0u && n_pairs == 0u)
                while (n_dpairs >= 2u) {
                    const gl_t d0_ = QT_DIRECT_OF(W1), d1_ = QT_DIRECT_OF(W3);
                    QT_PAIR_A(0, W0, W2, x2)
                    QT_DPAIR_B(1, W1, d0_, x0, W3, x3)
                    QT_PAIR_A(2, W2, W0, x0)
                    QT_DPAIR_B(3, W3, d1_, x2, W1, x1)
                    n_dpairs -= 2u;
                }
            QT_GENERIC(0, W0, x0, W1, x1, W2, x2)
            while (n_plain >= 4u) {
                QT_PLAIN(1, W1, x1, W3, x3)
                QT_PLAIN(2, W2, x2, W0, x0)
                QT_PLAIN(3, W3, x3, W1, x1)
                QT_PLAIN(0, W0, x0, W2, x2)
                n_plain -= 4u;
            }
            if (n_plain == 0u)
                while (n_pairs >= 2u) {
                    QT_PAIR_A(1, W1, W3, x3)
                    QT_PAIR_B(2, W2, x2, x1, W0, x0)
                    QT_PAIR_A(3, W3, W1, x1)
                    QT_PAIR_B(0, W0, x0, x3, W2, x2)
                    n_pairs -= 2u;
                }
            if (n_plain == 0u && n_pairs == 0u)
                while (n_dpairs >= 2u) {
                    const gl_t d0_ = QT_DIRECT_OF(W2), d1_ = QT_DIRECT_OF(W0);
                    QT_PAIR_A(1, W1, W3, x3)
                    QT_DPAIR_B(2, W2, d0_, x1, W0, x0)
                    QT_PAIR_A(3, W3, W1, x1)
                    QT_DPAIR_B(0, W0, d1_, x3, W2, x2)
                    n_dpairs -= 2u;
                }
            QT_GENERIC(1, W1, x1, W2, x2, W3, x3)
            while (n_plain >= 4u) {
                QT_PLAIN(2, W2, x2, W0, x0)
                QT_PLAIN(3, W3, x3, W1, x1)
                QT_PLAIN(0, W0, x0, W2, x2)
                QT_PLAIN(1, W1, x1, W3, x3)
                n_plain -= 4u;
            }
            if (n_plain == 0u)
                while (n_pairs >= 2u) {
                    QT_PAIR_A(2, W2, W0, x0)
                    QT_PAIR_B(3, W3, x3, x2, W1, x1)
                    QT_PAIR_A(0, W0, W2, x2)
                    QT_PAIR_B(1, W1, x1, x0, W3, x3)
                    n_pairs -= 2u;
                }
            if (n_plain == 0u && n_pairs == 0u)
                while (n_dpairs >= 2u) {
                    const gl_t d0_ = QT_DIRECT_OF(W3), d1_ = QT_DIRECT_OF(W1);
                    QT_PAIR_A(2, W2, W0, x0)
                    QT_DPAIR_B(3, W3, d0_, x2, W1, x1)
                    QT_PAIR_A(0, W0, W2, x2)
                    QT_DPAIR_B(1, W1, d1_, x0, W3, x3)
                    n_dpairs -= 2u;
                }
            QT_GENERIC(2, W2, x2, W3, x3, W0, x0)
            while (n_plain >= 4u) {
                QT_PLAIN(3, W3, x3, W1, x1)
                QT_PLAIN(0, W0, x0, W2, x2)
                QT_PLAIN(1, W1, x1, W3, x3)
                QT_PLAIN(2, W2, x2, W0, x0)
                n_plain -= 4u;
            }
            if (n_plain == 0u)
                while (n_pairs >= 2u) {
                    QT_PAIR_A(3, W3, W1, x1)
                    QT_PAIR_B(0, W0, x0, x3, W2, x2)
                    QT_PAIR_A(1, W1, W3, x3)
                    QT_PAIR_B(2, W2, x2, x1, W0, x0)
                    n_pairs -= 2u;
                }
            if (n_plain == 0u && n_pairs == 0u)
                while (n_dpairs >= 2u) {
                    const gl_t d0_ = QT_DIRECT_OF(W0), d1_ = QT_DIRECT_OF(W2);
                    QT_PAIR_A(3, W3, W1, x1)
                    QT_DPAIR_B(0, W0, d0_, x3, W2, x2)
                    QT_PAIR_A(1, W1, W3, x3)
                    QT_DPAIR_B(2, W2, d1_, x1, W0, x0)
                    n_dpairs -= 2u;
                }
            QT_GENERIC(3, W3, x3, W0, x0, W1, x1)
        }
stream_done:
#undef QT_MADS
#undef QT_ADVANCE
#undef QT_PLAIN
#undef QT_PAIR_A
#undef QT_PAIR_B
#undef QT_DPAIR_B
#undef QT_DIRECT_OF
#undef QT_GENERIC
#undef QT_TILE_BARRIER
    // waves whose stream ended before the chunk's last tile (never by construction) would desynchronise the barrier count
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef STARKHIP_QT_PROF
    if (lane == 0) {
        unsigned long long* o = qt_prof + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (QT_WAVES + 1) + wave) * 4;
        o[0] = QT_PROF_CLOCK() - pt_start;
        o[1] = pt_bar;
        o[2] = pt_n;
        o[3] = 0;
    }
#endif

    // acc of the eight waves -> partial[chunk]
    gl_t* red = tile[0];
    red[(wave * 2 + 0) * 64 + lane] = gl_canon(acc0);
    red[(wave * 2 + 1) * 64 + lane] = gl_canon(acc1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (wave < 2 && live) {
        gl_t sum = 0;
        for (unsigned w = 0; w < QT_WAVES; w++) sum = gl_add(sum, red[(w * 2 + wave) * 64 + lane]);
        P.partial[((size_t)chunk * 2 + wave) * size + t] = sum;
    }
}

// apow[j][e] = alpha_j^e, e < K
__global__ void quotient_alpha_powers_kernel(gl_t* apow, gl_t alpha0, gl_t alpha1, uint32_t K) { STARKHIP_PRIO_ENTRY
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * K) return;
    const uint32_t j = i >= K, e = j ? i - K : i;
    apow[i] = gl_pow(j ? alpha1 : alpha0, e);
}

// per-proof weights of every record: w_j = sum over its terms of coefficient * alpha_j^e, split into three 22-bit limbs
__global__ void quotient_weights_kernel(QTRec* recs, const uint32_t* contrib_off, const QTContrib* contribs, uint32_t n_recs,
                                        const gl_t* apow, uint32_t K, const gl_t* consts, const gl_t* pis) { STARKHIP_PRIO_ENTRY
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_recs) return;
    const uint32_t c0 = contrib_off[r], c1 = contrib_off[r + 1];
    if (c0 == c1) return;  // markers and no-ops keep their zeros, descriptors (QT_DESC) the gate cells their weight words hold
    gl_t w[2] = {0, 0};
    for (uint32_t c = c0; c < c1; c++) {
        const QTContrib t = contribs[c];
        const uint32_t ck = t.coef & 7u, idx = t.coef >> 3;
        for (int j = 0; j < 2; j++) {
            const gl_t a = apow[(size_t)j * K + t.e];
            gl_t term;
            if (ck == CK_PLUS) term = a;
            else if (ck == CK_MINUS) term = gl_neg(a);
            else if (ck == CK_CONST) term = gl_mul(a, consts[idx]);
            else if (ck == CK_PI) term = gl_mul(a, pis[idx]);
            else term = gl_neg(gl_mul(a, pis[idx]));
            w[j] = gl_add(w[j], term);
        }
    }
    const uint32_t M = (1u << QT_LIMB_BITS) - 1;
    for (int j = 0; j < 2; j++) {
        recs[r].w[3 * j + 0] = (uint32_t)(w[j] & M);
        recs[r].w[3 * j + 1] = (uint32_t)((w[j] >> QT_LIMB_BITS) & M);
        recs[r].w[3 * j + 2] = (uint32_t)(w[j] >> (2 * QT_LIMB_BITS));
    }
}

// out[j][i] (natural quotient index i) = (sum_c partial[c][j][t]) / Z_H(x_i)
__global__ void quotient_tiles_combine_kernel(const gl_t* __restrict__ partial, unsigned n_chunks, const gl_t* __restrict__ tab, unsigned log_n,
                                              unsigned qdb, gl_t* __restrict__ out) { STARKHIP_PRIO_ENTRY
    const size_t n = (size_t)1 << log_n, size = n << qdb;
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= size) return;
    size_t sp = t >> log_n, k = t & (n - 1);
    size_t i = (k << qdb) + sp;
    gl_t zhi = tab[3 * size + t];
    for (int j = 0; j < 2; j++) {
        gl_t acc = 0;
        for (unsigned p = 0; p < n_chunks; p++) acc = gl_add(acc, partial[((size_t)p * 2 + j) * size + t]);
        out[(size_t)j * size + i] = gl_mul(acc, zhi);
    }
}

hipError_t launch_quotient_weights(QTRec* recs, const uint32_t* contrib_off, const QTContrib* contribs, uint32_t n_recs, gl_t* apow, uint32_t K,
                                   const gl_t* consts, const gl_t* pis, gl_t alpha0, gl_t alpha1, hipStream_t st) {
    if (K) hipLaunchKernelGGL(quotient_alpha_powers_kernel, dim3((2 * K + 255) / 256), dim3(256), 0, st, apow, alpha0, alpha1, K);
    hipLaunchKernelGGL(quotient_weights_kernel, dim3((n_recs + 255) / 256), dim3(256), 0, st, recs, contrib_off, contribs, n_recs, apow, K, consts, pis);
    return hipGetLastError();
}

hipError_t launch_quotient_tiles(const QTRec* recs, const QTStream* streams, const uint32_t* chunk_tile_off,
                                 const uint32_t* tile_list, unsigned n_chunks, const gl_t* lde, const gl_t* tab, gl_t* partial, unsigned log_n,
                                 unsigned rate_bits, unsigned qdb, unsigned n_cols, unsigned dbg, hipStream_t st) {
    QTParams P;
    P.dbg = dbg;
    P.recs = recs; P.streams = streams; P.chunk_tile_off = chunk_tile_off; P.tile_list = tile_list;
    P.lde = lde; P.tab = tab; P.partial = partial; P.log_n = log_n; P.rate_bits = rate_bits; P.qdb = qdb; P.n_cols = n_cols;
    const size_t size = (size_t)1 << (log_n + qdb);
    const bool small = log_n < 6 || size < 64;
    const dim3 grid((unsigned)((size + 63) / 64), n_chunks), block(64 * (QT_WAVES + (small ? 1 : 0)));  // SMALL_N: + the producer wave
#ifdef STARKHIP_DEBUG  // make DEBUG_KNOBS=1: profiling variants with parts switched off (results are garbage); not in the release library
    switch (dbg) {  // 1 tile loads, 2 arithmetic, 3 both, 4 piece ends, 8 factors outside the tile not loaded
        case 0: break;
        case 1: hipLaunchKernelGGL((quotient_tiles_kernel<false, 1>), grid, block, 0, st, P); return hipGetLastError();
        case 2: hipLaunchKernelGGL((quotient_tiles_kernel<false, 2>), grid, block, 0, st, P); return hipGetLastError();
        case 3: hipLaunchKernelGGL((quotient_tiles_kernel<false, 3>), grid, block, 0, st, P); return hipGetLastError();
        case 8: hipLaunchKernelGGL((quotient_tiles_kernel<false, 8>), grid, block, 0, st, P); return hipGetLastError();
        default: hipLaunchKernelGGL((quotient_tiles_kernel<false, 4>), grid, block, 0, st, P); return hipGetLastError();
    }
#else
    if (dbg) return hipErrorInvalidValue;
#endif
    if (small) hipLaunchKernelGGL((quotient_tiles_kernel<true, 0>), grid, block, 0, st, P);
    else hipLaunchKernelGGL((quotient_tiles_kernel<false, 0>), grid, block, 0, st, P);
    return hipGetLastError();
}

hipError_t launch_quotient_tiles_combine(const gl_t* partial, unsigned n_chunks, const gl_t* tab, unsigned log_n, unsigned qdb, gl_t* out,
                                         hipStream_t st) {
    size_t size = (size_t)1 << (log_n + qdb);
    hipLaunchKernelGGL(quotient_tiles_combine_kernel, dim3((unsigned)((size + 255) / 256)), dim3(256), 0, st, partial, n_chunks, tab, log_n, qdb, out);
    return hipGetLastError();
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
        hipLaunchKernelGGL(quotient_eval_kernel<false>, dim3((unsigned)((size + 63) / 64), n_chunks), dim3(64), 0, st, P);
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

#ifdef STARKHIP_QT_PROF
extern "C" __attribute__((visibility("default"))) int starkhip_debug_qt_prof(unsigned long long* out, size_t n_words) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(starkhip::qt_prof), n_words * 8, 0, hipMemcpyDeviceToHost);
}
extern "C" __attribute__((visibility("default"))) int starkhip_debug_qt_tile_prof(unsigned long long* out, size_t n_words) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(starkhip::qt_tile_prof), n_words * 8, 0, hipMemcpyDeviceToHost);
}
#endif
