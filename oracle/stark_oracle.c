/* TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  PARITY UNPINNED (see oracle_field.h).
 *
 * CPU restatement of starky::prover::prove() as the reference calls it at
 * /root/reference/src/aggregate_proof.rs:59-65 (F = Goldilocks, C = PoseidonGoldilocksConfig,
 * D = 2).  The implementation being restated is NOT under /root/reference: it is the git
 * dependency starky 0.1.2 / plonky2 0.1.4 @ Electron-Labs/plonky2
 * 666f31517353b29b3d847c6e18b26c9be8bf060b (Cargo.lock:1424-1427,2043-2046).  Each function
 * cites the SURVEY.md appendix paragraph (the restated published algorithm) it follows and the
 * reference call site that exercises it.
 *
 * The AIR itself (the constraint list) is consumed as DATA: the flat program described in
 * starky_bls12_381_amd/csrc/air_ir.h, produced by the product's restatement of the
 * reference's add_*_constraints functions.  This file has its own parser/evaluator for it.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "oracle_field.h"

#define EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------ Poseidon
 * SURVEY.md App. A.4 / App. C.  width 12, rate 8, x^7, 4+22+4 rounds; constants from
 * ChaCha8Rng::seed_from_u64(0) (derived here at first use, not copied from a table). */
static fe RC[360];
static int rc_ready = 0;
static const uint64_t MDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static const uint64_t MDS_DIAG0 = 8;

static uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
static void chacha8_block(const uint32_t key[8], uint64_t ctr, uint32_t out[16]) {
    uint32_t s[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
    for (int i = 0; i < 8; i++) s[4 + i] = key[i];
    s[12] = (uint32_t)ctr; s[13] = (uint32_t)(ctr >> 32); s[14] = 0; s[15] = 0;
    uint32_t w[16];
    memcpy(w, s, sizeof w);
#define QR(a, b, c, d)                                                              \
    w[a] += w[b]; w[d] = rotl32(w[d] ^ w[a], 16); w[c] += w[d]; w[b] = rotl32(w[b] ^ w[c], 12); \
    w[a] += w[b]; w[d] = rotl32(w[d] ^ w[a], 8);  w[c] += w[d]; w[b] = rotl32(w[b] ^ w[c], 7);
    for (int r = 0; r < 4; r++) {
        QR(0, 4, 8, 12) QR(1, 5, 9, 13) QR(2, 6, 10, 14) QR(3, 7, 11, 15)
        QR(0, 5, 10, 15) QR(1, 6, 11, 12) QR(2, 7, 8, 13) QR(3, 4, 9, 14)
    }
#undef QR
    for (int i = 0; i < 16; i++) out[i] = w[i] + s[i];
}

static void init_rc(void) {
    if (rc_ready) return;
#pragma omp critical(oracle_rc)
    {
        if (!rc_ready) {
            uint32_t key[8];
            uint64_t st = 0;
            for (int i = 0; i < 8; i++) { /* rand_core seed_from_u64: PCG32 */
                st = st * 6364136223846793005ULL + 11634580027462260723ULL;
                uint32_t xs = (uint32_t)(((st >> 18) ^ st) >> 27);
                uint32_t rot = (uint32_t)(st >> 59);
                key[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
            }
            uint32_t blk[16];
            int pos = 16, n = 0;
            uint64_t ctr = 0;
            while (n < 360) {
                uint32_t lohi[2];
                for (int k = 0; k < 2; k++) {
                    if (pos == 16) { chacha8_block(key, ctr++, blk); pos = 0; }
                    lohi[k] = blk[pos++];
                }
                uint64_t v = (uint64_t)lohi[0] | ((uint64_t)lohi[1] << 32);
                u128 m = (u128)v * OR_P; /* gen_range(0..p): widening multiply, reject on low word */
                if ((uint64_t)m <= OR_P - 1) RC[n++] = (fe)(m >> 64);
            }
            rc_ready = 1;
        }
    }
}

static inline fe sbox7(fe x) {
    fe x2 = f_mul(x, x), x4 = f_mul(x2, x2), x3 = f_mul(x2, x);
    return f_mul(x3, x4);
}

EXPORT void oracle_poseidon_permute(fe s[12]) {
    init_rc();
    for (int r = 0; r < 30; r++) {
        for (int i = 0; i < 12; i++) s[i] = f_add(s[i], RC[12 * r + i]);
        if (r < 4 || r >= 26) {
            for (int i = 0; i < 12; i++) s[i] = sbox7(s[i]);
        } else {
            s[0] = sbox7(s[0]);
        }
        fe t[12];
        for (int row = 0; row < 12; row++) {
            u128 acc = 0; /* sum < 12*41*2^64 < 2^73 */
            for (int i = 0; i < 12; i++) acc += (u128)s[(i + row) % 12] * MDS_CIRC[i];
            if (row == 0) acc += (u128)s[0] * MDS_DIAG0;
            t[row] = f_red128(acc);
        }
        memcpy(s, t, sizeof t);
    }
}

EXPORT void oracle_round_constants(fe out[360]) { init_rc(); memcpy(out, RC, sizeof RC); }

/* hash_no_pad: overwrite-mode sponge (App. A.4) */
EXPORT void oracle_hash_no_pad(const fe* in, size_t len, fe out[4]) {
    fe st[12] = {0};
    for (size_t off = 0; off < len; off += 8) {
        size_t k = len - off < 8 ? len - off : 8;
        for (size_t i = 0; i < k; i++) st[i] = in[off + i];
        oracle_poseidon_permute(st);
    }
    memcpy(out, st, 4 * sizeof(fe));
}
/* hash_or_noop: leaves of <= 4 elements are copied, zero padded (App. A.3) */
static void hash_or_noop(const fe* in, size_t len, fe out[4]) {
    if (len <= 4) {
        for (size_t i = 0; i < 4; i++) out[i] = i < len ? in[i] : 0;
    } else {
        oracle_hash_no_pad(in, len, out);
    }
}
EXPORT void oracle_two_to_one(const fe a[4], const fe b[4], fe out[4]) {
    fe st[12] = {0};
    memcpy(st, a, 32); memcpy(st + 4, b, 32);
    oracle_poseidon_permute(st);
    memcpy(out, st, 32);
}

/* ------------------------------------------------------------------ Challenger (App. A.4) */
typedef struct { fe st[12]; fe in[8]; int nin; fe out[8]; int nout; } challenger;
static void ch_init(challenger* c) { memset(c, 0, sizeof *c); }
static void ch_duplex(challenger* c) {
    for (int i = 0; i < c->nin; i++) c->st[i] = c->in[i];
    c->nin = 0;
    oracle_poseidon_permute(c->st);
    memcpy(c->out, c->st, 8 * sizeof(fe));
    c->nout = 8;
}
static void ch_observe(challenger* c, fe x) {
    c->nout = 0;
    c->in[c->nin++] = x;
    if (c->nin == 8) ch_duplex(c);
}
static void ch_observe_ext(challenger* c, fe2 x) { ch_observe(c, x.a0); ch_observe(c, x.a1); }
static void ch_observe_cap(challenger* c, const fe* cap, int ncap) { for (int i = 0; i < 4 * ncap; i++) ch_observe(c, cap[i]); }
static fe ch_get(challenger* c) {
    if (c->nin > 0 || c->nout == 0) ch_duplex(c);
    return c->out[--c->nout];
}
static fe2 ch_get_ext(challenger* c) { fe a = ch_get(c); fe b = ch_get(c); return e_make(a, b); }

/* ------------------------------------------------------------------ NTT (App. A.2)
 * values[i] = P(w^i), natural order both sides.  Textbook iterative radix-2. */
static void ntt_inplace(fe* a, unsigned logn, fe root) {
    size_t n = (size_t)1 << logn;
    for (size_t i = 0; i < n; i++) {
        size_t j = bitrev((uint32_t)i, logn);
        if (i < j) { fe t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (unsigned s = 1; s <= logn; s++) {
        size_t m = (size_t)1 << s, h = m >> 1;
        fe wm = root;
        for (unsigned k = s; k < logn; k++) wm = f_mul(wm, wm);
        fe* tw = (fe*)malloc(h * sizeof(fe));
        tw[0] = 1;
        for (size_t j = 1; j < h; j++) tw[j] = f_mul(tw[j - 1], wm);
        for (size_t k = 0; k < n; k += m)
            for (size_t j = 0; j < h; j++) {
                fe t = f_mul(tw[j], a[k + j + h]), u = a[k + j];
                a[k + j] = f_add(u, t);
                a[k + j + h] = f_sub(u, t);
            }
        free(tw);
    }
}
EXPORT void oracle_fft(fe* a, unsigned logn) { ntt_inplace(a, logn, f_root(logn)); }
EXPORT void oracle_ifft(fe* a, unsigned logn) {
    size_t n = (size_t)1 << logn;
    ntt_inplace(a, logn, f_inv(f_root(logn)));
    fe ninv = f_inv((fe)n);
    for (size_t i = 0; i < n; i++) a[i] = f_mul(a[i], ninv);
}
/* evaluate on shift * <w>: scale coefficient i by shift^i then fft */
EXPORT void oracle_coset_fft(fe* a, unsigned logn, fe shift) {
    size_t n = (size_t)1 << logn;
    fe s = 1;
    for (size_t i = 0; i < n; i++) { a[i] = f_mul(a[i], s); s = f_mul(s, shift); }
    oracle_fft(a, logn);
}
EXPORT void oracle_coset_ifft(fe* a, unsigned logn, fe shift) {
    size_t n = (size_t)1 << logn;
    oracle_ifft(a, logn);
    fe si = f_inv(shift), s = 1;
    for (size_t i = 0; i < n; i++) { a[i] = f_mul(a[i], s); s = f_mul(s, si); }
}
static void ext_transform(fe2* a, unsigned logn, fe shift, int inverse) {
    size_t n = (size_t)1 << logn;
    fe* t = (fe*)malloc(n * sizeof(fe));
    for (int h = 0; h < 2; h++) {
        for (size_t i = 0; i < n; i++) t[i] = h ? a[i].a1 : a[i].a0;
        if (inverse) oracle_coset_ifft(t, logn, shift); else oracle_coset_fft(t, logn, shift);
        for (size_t i = 0; i < n; i++) { if (h) a[i].a1 = t[i]; else a[i].a0 = t[i]; }
    }
    free(t);
}

/* ------------------------------------------------------------------ LDE of a column batch
 * App. A.3 from_values: coeffs = ifft(col); lde = coset_fft_7(zero-pad(coeffs)).
 * cols: column-major [C][n]; coeffs_out (nullable) column-major [C][n];
 * lde_rows_out row-major [N][C] with NATURAL point index (row i <-> 7*w_N^i). */
EXPORT void oracle_lde_rows(const fe* cols, size_t C, unsigned logn, unsigned rate_bits, fe* coeffs_out, fe* lde_rows_out) {
    size_t n = (size_t)1 << logn, N = n << rate_bits;
#pragma omp parallel
    {
        fe* buf = (fe*)malloc(N * sizeof(fe));
#pragma omp for schedule(dynamic, 16)
        for (size_t c = 0; c < C; c++) {
            memcpy(buf, cols + c * n, n * sizeof(fe));
            oracle_ifft(buf, logn);
            if (coeffs_out) memcpy(coeffs_out + c * n, buf, n * sizeof(fe));
            memset(buf + n, 0, (N - n) * sizeof(fe));
            oracle_coset_fft(buf, logn + rate_bits, 7);
            for (size_t i = 0; i < N; i++) lde_rows_out[i * C + c] = buf[i];
        }
        free(buf);
    }
}

/* ------------------------------------------------------------------ Merkle tree (App. A.3)
 * leaf j = rows[bitrev(j)] (reverse_index_bits_in_place of the transposed LDE).
 * digests: level 0 = N leaf digests, then N/2, ... down to 2^cap_h nodes; all kept. */
typedef struct { unsigned logn, cap_h; fe** lev; } mtree;
static void mtree_build(mtree* t, const fe* rows, size_t width, unsigned logN, unsigned cap_h, int leaves_bitrev) {
    size_t N = (size_t)1 << logN;
    t->logn = logN; t->cap_h = cap_h;
    unsigned nlev = logN - cap_h + 1;
    t->lev = (fe**)calloc(nlev, sizeof(fe*));
    t->lev[0] = (fe*)malloc(N * 4 * sizeof(fe));
#pragma omp parallel for schedule(dynamic, 8)
    for (size_t j = 0; j < N; j++) {
        size_t src = leaves_bitrev ? bitrev((uint32_t)j, logN) : j;
        hash_or_noop(rows + src * width, width, t->lev[0] + 4 * j);
    }
    for (unsigned l = 1; l < nlev; l++) {
        size_t cnt = N >> l;
        t->lev[l] = (fe*)malloc(cnt * 4 * sizeof(fe));
#pragma omp parallel for
        for (size_t j = 0; j < cnt; j++) oracle_two_to_one(t->lev[l - 1] + 8 * j, t->lev[l - 1] + 8 * j + 4, t->lev[l] + 4 * j);
    }
}
static const fe* mtree_cap(const mtree* t) { return t->lev[t->logn - t->cap_h]; }
static void mtree_free(mtree* t) { for (unsigned l = 0; l <= t->logn - t->cap_h; l++) free(t->lev[l]); free(t->lev); }
/* siblings bottom-up, logN - cap_h of them */
static void mtree_path(const mtree* t, size_t idx, fe* out) {
    for (unsigned l = 0; l < t->logn - t->cap_h; l++) { memcpy(out + 4 * l, t->lev[l] + 4 * (idx ^ 1), 32); idx >>= 1; }
}
EXPORT void oracle_merkle_cap(const fe* rows_natural, size_t width, unsigned logN, unsigned cap_h, fe* cap_out) {
    mtree t;
    mtree_build(&t, rows_natural, width, logN, cap_h, 1);
    memcpy(cap_out, mtree_cap(&t), ((size_t)4 << cap_h) * sizeof(fe));
    mtree_free(&t);
}

/* ------------------------------------------------------------------ AIR program (air_ir.h) */
typedef struct {
    uint32_t n_cols, n_pis, degree, n_constraints, n_consts, n_code, n_groups;
    const fe* consts;
    uint32_t* code;
} air_prog;
static int air_parse(const uint64_t* blob, size_t words, air_prog* p) {
    if (words < 8 || blob[0] != 0x3152495F52494153ULL) return -1;
    p->n_cols = (uint32_t)blob[1]; p->n_pis = (uint32_t)blob[2]; p->degree = (uint32_t)blob[3];
    p->n_constraints = (uint32_t)blob[4]; p->n_consts = (uint32_t)blob[5]; p->n_code = (uint32_t)blob[6];
    p->n_groups = (uint32_t)blob[7];
    p->consts = blob + 8;
    const uint64_t* c = blob + 8 + p->n_consts;
    p->code = (uint32_t*)malloc(((size_t)p->n_code + 1) * 4);
    for (uint32_t i = 0; i < p->n_code; i++) p->code[i] = (uint32_t)(c[i / 2] >> (32 * (i & 1)));
    return 0;
}

/* Evaluate every constraint at one point and fold with each alpha exactly as the reference's
 * ConstraintConsumer does (App. A.6): acc = acc*alpha + mask*c_k, constraint by constraint.
 * masks[kind] = {1, z_last, L_first, L_last}.  Deliberately the naive per-constraint fold (no
 * group Horner) so that the product's grouped evaluation is checked against the plain definition.
 * If `each` is non-NULL it receives every c_k*mask (used by the trace checker). */
static void air_eval(const air_prog* p, const fe* local, const fe* next, const fe* pis, const fe masks[4],
                     const fe* alphas, int nalpha, fe* acc, fe* each) {
    const uint32_t* w = p->code;
    uint32_t k = 0;
    for (int j = 0; j < nalpha; j++) acc[j] = 0;
    while ((*w & 15) == 1) {
        uint32_t kind = (*w >> 4) & 3, ng = (*w >> 8) & 255, m = *w >> 16;
        w++;
        fe G = masks[kind];
        for (uint32_t g = 0; g < ng; g++, w++) {
            const fe* row = (*w & (1u << 30)) ? next : local;
            fe v = row[*w & 0xFFFFFF];
            if (*w & (1u << 31)) v = f_sub(1, v);
            G = f_mul(G, v);
        }
        for (uint32_t c = 0; c < m; c++) {
            fe body = 0;
            for (;;) {
                uint32_t t = *w++;
                uint32_t nf = t & 3, ck = (t >> 2) & 7, idx = t >> 6;
                fe v = 1;
                for (uint32_t f = 0; f < nf; f++, w++) {
                    const fe* row = (*w & (1u << 30)) ? next : local;
                    v = f_mul(v, row[*w & 0xFFFFFF]);
                }
                switch (ck) {
                    case 0: body = f_add(body, v); break;
                    case 1: body = f_sub(body, v); break;
                    case 2: body = f_add(body, f_mul(v, p->consts[idx])); break;
                    case 3: body = f_add(body, f_mul(v, pis[idx])); break;
                    default: body = f_sub(body, f_mul(v, pis[idx])); break;
                }
                if (t & 32) break;
            }
            fe val = f_mul(G, body);
            if (each) each[k] = val;
            k++;
            for (int j = 0; j < nalpha; j++) acc[j] = f_add(f_mul(acc[j], alphas[j]), val);
        }
    }
}

/* Every constraint's value mask(kind)*c_k on ONE arbitrary frame (local, next, public inputs need not come from a trace).
 * tests/test_constraint_schedule_cpu.py compares these with the polynomials extracted from the reference's Rust source,
 * evaluated in Python on the same random frame: that pins this evaluator's reading of the program to the reference. */
EXPORT int oracle_eval_frame(const uint64_t* air_blob, size_t air_words, const fe* local, const fe* next, const fe* pis,
                             const fe masks[4], fe* each_out) {
    air_prog p;
    if (air_parse(air_blob, air_words, &p)) return -1;
    fe acc;
    fe alpha = 1;
    air_eval(&p, local, next, pis, masks, &alpha, 1, &acc, each_out);
    free(p.code);
    return 0;
}

/* Check that every constraint vanishes on every row of a trace:
 * plain on all rows (next row wraps), transition on rows 0..n-2, first on row 0, last on row n-1.
 * trace row-major [n][C].  Returns number of violations; first violation reported in out3 =
 * {constraint index, row, value}. */
EXPORT long oracle_check_trace(const uint64_t* air_blob, size_t air_words, const fe* trace_rows, size_t n, const fe* pis, uint64_t out3[3]) {
    air_prog p;
    if (air_parse(air_blob, air_words, &p)) return -1;
    long bad = 0;
    int reported = 0;
    /* kind per constraint */
    uint8_t* kinds = (uint8_t*)malloc(p.n_constraints);
    {
        const uint32_t* w = p.code; uint32_t k = 0;
        while ((*w & 15) == 1) {
            uint32_t kind = (*w >> 4) & 3, ng = (*w >> 8) & 255, m = *w >> 16;
            w += 1 + ng;
            for (uint32_t c = 0; c < m; c++) { kinds[k++] = (uint8_t)kind; for (;;) { uint32_t t = *w++; w += t & 3; if (t & 32) break; } }
        }
    }
#pragma omp parallel
    {
        fe* each = (fe*)malloc((size_t)p.n_constraints * sizeof(fe));
#pragma omp for schedule(dynamic, 4)
        for (size_t r = 0; r < n; r++) {
            fe masks[4] = {1, 1, 1, 1}, acc[1], alpha = 0;
            air_eval(&p, trace_rows + r * p.n_cols, trace_rows + ((r + 1) % n) * p.n_cols, pis, masks, &alpha, 1, acc, each);
            for (uint32_t k = 0; k < p.n_constraints; k++) {
                int active = kinds[k] == 0 || (kinds[k] == 1 && r + 1 < n) || (kinds[k] == 2 && r == 0) || (kinds[k] == 3 && r + 1 == n);
                if (active && each[k] != 0) {
#pragma omp critical(oracle_report)
                    { bad++; if (!reported || (uint64_t)k < out3[0]) { reported = 1; out3[0] = k; out3[1] = r; out3[2] = each[k]; } }
                }
            }
        }
        free(each);
    }
    free(kinds); free(p.code);
    return bad;
}

/* ------------------------------------------------------------------ prove() */
typedef struct {
    uint32_t security_bits, num_challenges, rate_bits, cap_height, proof_of_work_bits, arity_bits, final_poly_bits, num_query_rounds;
} oracle_config;

static fe2 horner_base_at_ext(const fe* coeffs, size_t n, fe2 z) {
    fe2 acc = e_make(0, 0);
    for (size_t i = n; i-- > 0;) acc = e_add(e_mul(acc, z), e_base(coeffs[i]));
    return acc;
}

typedef struct { uint64_t* d; size_t n, cap; } wbuf;
static void wb_push(wbuf* b, const uint64_t* src, size_t k) {
    if (b->n + k > b->cap) { while (b->n + k > b->cap) b->cap = b->cap ? b->cap * 2 : 1024; b->d = (uint64_t*)realloc(b->d, b->cap * 8); }
    memcpy(b->d + b->n, src, k * 8); b->n += k;
}
static void wb_push1(wbuf* b, uint64_t x) { wb_push(b, &x, 1); }

/* error codes shared with include/starkhip.h */
#define ERR_QUOTIENT_NOT_DIVISIBLE (-1)
/* Test hook (tests/test_gpu_airs.py): substitute zeta = 7 w_n^(k - 1), a point of the coset the product keeps its trace values on, so that
 * the product's branch for that case (probability 2^-115 in a real transcript) can be compared byte for byte.  0 = off. */
static long g_zeta_on_coset = 0;
EXPORT void oracle_set_zeta_on_coset(long k_plus_1) { g_zeta_on_coset = k_plus_1; }
#define ERR_ZETA_IN_SUBGROUP (-2)
#define ERR_BAD_SHAPE (-3)

/* trace_cols: column-major [C][n].  pow_override: UINT64_MAX => search smallest nonce.
 * Output: malloc'd proof blob in the canonical order of SURVEY.md App. A.9 (layout documented in
 * include/starkhip.h).  fri_mul_x: reserved switch (App. A.11 item 1), must be 0. */
EXPORT int oracle_prove(const uint64_t* air_blob, size_t air_words, const oracle_config* cfg, const fe* trace_cols, uint32_t n_rows,
                        const fe* pis, uint64_t pow_override, uint64_t** out, size_t* out_words) {
    air_prog P;
    if (air_parse(air_blob, air_words, &P)) return ERR_BAD_SHAPE;
    unsigned logn = 0;
    while (((size_t)1 << logn) < n_rows) logn++;
    if (((size_t)1 << logn) != n_rows) return ERR_BAD_SHAPE;
    const size_t n = n_rows, C = P.n_cols;
    const unsigned r = cfg->rate_bits, capH = cfg->cap_height, logN = logn + r;
    const size_t N = n << r, ncap = (size_t)1 << capH;
    const int nch = (int)cfg->num_challenges;
    /* A.8 fri_params: ConstantArityBits(arity_bits, final_poly_bits) */
    unsigned arities[16]; int L = 0;
    { unsigned db = logn; while (db > cfg->final_poly_bits && db + r - cfg->arity_bits >= capH) { arities[L++] = cfg->arity_bits; db -= cfg->arity_bits; } }
    { unsigned tot = 0; for (int i = 0; i < L; i++) tot += arities[i]; if (tot > logn + r - capH || logN < capH) return ERR_BAD_SHAPE; }
    /* A.6 quotient geometry */
    const unsigned factor = P.degree > 1 ? P.degree - 1 : 1;
    unsigned qdb = 0; while ((1u << qdb) < factor) qdb++;
    if (qdb > r) return ERR_BAD_SHAPE;
    const size_t Q = (size_t)factor * nch;

    /* ---- A.3 trace commit */
    fe* coeffs = (fe*)malloc(C * n * sizeof(fe));
    fe* lde = (fe*)malloc(C * N * sizeof(fe));
    oracle_lde_rows(trace_cols, C, logn, r, coeffs, lde);
    mtree ttree; mtree_build(&ttree, lde, C, logN, capH, 1);
    challenger ch; ch_init(&ch);
    ch_observe_cap(&ch, mtree_cap(&ttree), (int)ncap); /* A.5.2; public inputs are not observed */
    fe alphas[8]; for (int j = 0; j < nch; j++) alphas[j] = ch_get(&ch);

    /* ---- A.6 quotient */
    const size_t size = n << qdb, step = (size_t)1 << (r - qdb), next_step = (size_t)1 << qdb;
    fe* qvals = (fe*)malloc((size_t)nch * size * sizeof(fe)); /* [j][i] */
    {
        fe wsize = f_root(logn + qdb), g_inv = f_inv(f_root(logn));
        fe seven_n = f_pow(7, n), ninv = f_inv((fe)n), g = f_root(logn);
#pragma omp parallel for schedule(dynamic, 8)
        for (size_t i = 0; i < size; i++) {
            fe x = f_mul(7, f_pow(wsize, i));
            fe zh = f_sub(f_mul(seven_n, f_pow(f_root(qdb), i % next_step)), 1); /* x^n - 1 */
            fe masks[4];
            masks[0] = 1;
            masks[1] = f_sub(x, g_inv);
            masks[2] = f_mul(zh, f_inv(f_mul((fe)n, f_sub(x, 1))));            /* L_first */
            masks[3] = f_mul(zh, f_inv(f_mul((fe)n, f_sub(f_mul(g, x), 1)))); /* L_last */
            (void)ninv;
            fe acc[8];
            air_eval(&P, lde + (i * step) * C, lde + (((i + next_step) % size) * step) * C, pis, masks, alphas, nch, acc, NULL);
            fe zhi = f_inv(zh);
            for (int j = 0; j < nch; j++) qvals[(size_t)j * size + i] = f_mul(acc[j], zhi);
        }
    }
    fe* qchunks = (fe*)malloc(Q * n * sizeof(fe)); /* column-major [Q][n], order [alpha0: c0..cf-1, alpha1: ...] */
    int rc = 0;
    for (int j = 0; j < nch; j++) {
        fe* qv = qvals + (size_t)j * size;
        oracle_coset_ifft(qv, logn + qdb, 7);
        for (size_t i = (size_t)factor * n; i < size; i++) if (qv[i] != 0) rc = ERR_QUOTIENT_NOT_DIVISIBLE;
        memcpy(qchunks + (size_t)j * factor * n, qv, (size_t)factor * n * sizeof(fe));
    }
    free(qvals);
    if (rc) { free(coeffs); free(lde); free(qchunks); mtree_free(&ttree); free(P.code); return rc; }
    /* quotient commit: from_coeffs (already coefficients) */
    fe* qlde = (fe*)malloc(Q * N * sizeof(fe)); /* row-major [N][Q] */
    {
        fe* buf = (fe*)malloc(N * sizeof(fe));
        for (size_t q = 0; q < Q; q++) {
            memcpy(buf, qchunks + q * n, n * sizeof(fe)); memset(buf + n, 0, (N - n) * sizeof(fe));
            oracle_coset_fft(buf, logN, 7);
            for (size_t i = 0; i < N; i++) qlde[i * Q + q] = buf[i];
        }
        free(buf);
    }
    mtree qtree; mtree_build(&qtree, qlde, Q, logN, capH, 1);
    ch_observe_cap(&ch, mtree_cap(&qtree), (int)ncap);
    fe2 zeta = ch_get_ext(&ch);
    if (g_zeta_on_coset > 0) zeta = e_make(f_mul(7, f_pow(f_root(logn), (uint64_t)(g_zeta_on_coset - 1) & (n - 1))), 0);  /* test hook, below */
    if (e_eq(e_pow(zeta, n), e_make(1, 0))) rc = ERR_ZETA_IN_SUBGROUP;

    /* ---- A.7 openings */
    fe2 gz = e_mulb(zeta, f_root(logn));
    fe2* op_local = (fe2*)malloc(C * sizeof(fe2)); fe2* op_next = (fe2*)malloc(C * sizeof(fe2)); fe2* op_q = (fe2*)malloc(Q * sizeof(fe2));
#pragma omp parallel for schedule(static)
    for (size_t c = 0; c < C; c++) { op_local[c] = horner_base_at_ext(coeffs + c * n, n, zeta); op_next[c] = horner_base_at_ext(coeffs + c * n, n, gz); }
    for (size_t q = 0; q < Q; q++) op_q[q] = horner_base_at_ext(qchunks + q * n, n, zeta);
    for (size_t c = 0; c < C; c++) ch_observe_ext(&ch, op_local[c]);
    for (size_t q = 0; q < Q; q++) ch_observe_ext(&ch, op_q[q]);
    for (size_t c = 0; c < C; c++) ch_observe_ext(&ch, op_next[c]);

    /* ---- A.8 prove_openings */
    fe2 alpha = ch_get_ext(&ch);
    fe2* fin = (fe2*)calloc(N, sizeof(fe2)); /* final poly coefficients, zero padded to N */
    {
        /* batch0: trace[0..C) ++ quotient[0..Q) at zeta; batch1: trace[0..C) at g*zeta */
        fe2* F0 = (fe2*)calloc(n, sizeof(fe2)); fe2* F1 = (fe2*)calloc(n, sizeof(fe2));
        fe2* apw = (fe2*)malloc((C + Q) * sizeof(fe2));
        apw[0] = e_make(1, 0); for (size_t j = 1; j < C + Q; j++) apw[j] = e_mul(apw[j - 1], alpha);
#pragma omp parallel for schedule(static)
        for (size_t k = 0; k < n; k++) {
            fe2 s = e_make(0, 0);
            for (size_t j = 0; j < C; j++) s = e_add(s, e_mulb(apw[j], coeffs[j * n + k]));
            F1[k] = s;
            for (size_t q = 0; q < Q; q++) s = e_add(s, e_mulb(apw[C + q], qchunks[q * n + k]));
            F0[k] = s;
        }
        /* divide_by_linear: (F - F(z))/(X - z), then pad with one zero coefficient back to n */
        fe2* Fb[2] = {F0, F1}; fe2 zs[2] = {zeta, gz};
        fe2 shiftC = e_pow(alpha, C); /* alpha^{|batch1|} */
        for (int b = 0; b < 2; b++) {
            fe2* quo = (fe2*)calloc(n, sizeof(fe2));
            fe2 carry = e_make(0, 0);
            for (size_t k = n; k-- > 1;) { carry = e_add(Fb[b][k], e_mul(carry, zs[b])); quo[k - 1] = carry; }
            for (size_t k = 0; k < n; k++) fin[k] = (b == 0) ? quo[k] : e_add(e_mul(fin[k], shiftC), quo[k]);
            free(quo);
        }
        free(F0); free(F1); free(apw);
    }
    /* lde(rate) + coset_fft(7) */
    fe2* fvals = (fe2*)malloc(N * sizeof(fe2));
    memcpy(fvals, fin, N * sizeof(fe2));
    ext_transform(fvals, logN, 7, 0);

    /* commit phase */
    mtree ftrees[16]; fe* fleaves[16]; size_t flen[16];
    size_t cur_len = N; unsigned cur_log = logN; fe shift = 7;
    fe2* coefs = fin; /* length cur_len */
    wbuf caps = {0};
    for (int l = 0; l < L; l++) {
        unsigned ab = arities[l]; size_t ar = (size_t)1 << ab, nleaf = cur_len >> ab;
        fe* leaves = (fe*)malloc(nleaf * ar * 2 * sizeof(fe));
        for (size_t j = 0; j < cur_len; j++) { /* reverse_index_bits then chunk by arity */
            fe2 v = fvals[bitrev((uint32_t)j, cur_log)];
            leaves[2 * j] = v.a0; leaves[2 * j + 1] = v.a1;
        }
        mtree_build(&ftrees[l], leaves, ar * 2, cur_log - ab, capH, 0);
        fleaves[l] = leaves; flen[l] = nleaf;
        ch_observe_cap(&ch, mtree_cap(&ftrees[l]), (int)ncap);
        wb_push(&caps, mtree_cap(&ftrees[l]), 4 * ncap);
        fe2 beta = ch_get_ext(&ch);
        size_t nl = cur_len >> ab;
        fe2* nc = (fe2*)malloc(nl * sizeof(fe2));
        for (size_t k = 0; k < nl; k++) { /* reduce_with_powers(chunk, beta) */
            fe2 s = e_make(0, 0);
            for (size_t i = ar; i-- > 0;) s = e_add(e_mul(s, beta), coefs[k * ar + i]);
            nc[k] = s;
        }
        if (coefs != fin) free(coefs);
        coefs = nc; cur_len = nl; cur_log -= ab;
        shift = f_pow(shift, ar);
        free(fvals);
        fvals = (fe2*)malloc(cur_len * sizeof(fe2));
        memcpy(fvals, coefs, cur_len * sizeof(fe2));
        ext_transform(fvals, cur_log, shift, 0);
    }
    size_t fplen = cur_len >> r;
    for (size_t k = fplen; k < cur_len; k++) if (coefs[k].a0 || coefs[k].a1) rc = rc ? rc : ERR_QUOTIENT_NOT_DIVISIBLE; /* "should always be zero" */
    for (size_t k = 0; k < fplen; k++) ch_observe_ext(&ch, coefs[k]);

    /* PoW: smallest w such that observe(w); get_challenge() has >= pow_bits leading zeros */
    uint64_t pow_w = pow_override;
    if (pow_override == UINT64_MAX && cfg->proof_of_work_bits == 0) pow_w = 0;
    else if (pow_override == UINT64_MAX) {
        for (uint64_t w = 0;; w++) { challenger c2 = ch; ch_observe(&c2, w); if ((ch_get(&c2) >> (64 - cfg->proof_of_work_bits)) == 0) { pow_w = w; break; } }
    }
    ch_observe(&ch, pow_w);
    (void)ch_get(&ch);

    /* ---- serialise (layout: include/starkhip.h) */
    wbuf o = {0};
    uint64_t hdr[16] = {0x3130304652505353ULL /* "SSPRF001" */, C, Q, logn, r, capH, (uint64_t)L, cfg->num_query_rounds, fplen, P.n_pis, cfg->arity_bits, (uint64_t)nch, 0, 0, 0, 0};
    wb_push(&o, hdr, 16);
    wb_push(&o, mtree_cap(&ttree), 4 * ncap);
    wb_push(&o, mtree_cap(&qtree), 4 * ncap);
    wb_push(&o, (uint64_t*)op_local, 2 * C);
    wb_push(&o, (uint64_t*)op_next, 2 * C);
    wb_push(&o, (uint64_t*)op_q, 2 * Q);
    wb_push(&o, caps.d, caps.n);
    fe sib[64 * 4];
    for (uint32_t qr = 0; qr < cfg->num_query_rounds; qr++) {
        size_t x = (size_t)(ch_get(&ch) % N);
        wb_push(&o, lde + (size_t)bitrev((uint32_t)x, logN) * C, C);
        mtree_path(&ttree, x, sib); wb_push(&o, sib, 4 * (logN - capH));
        wb_push(&o, qlde + (size_t)bitrev((uint32_t)x, logN) * Q, Q);
        mtree_path(&qtree, x, sib); wb_push(&o, sib, 4 * (logN - capH));
        for (int l = 0; l < L; l++) {
            size_t ar = (size_t)1 << arities[l];
            x >>= arities[l];
            wb_push(&o, fleaves[l] + x * ar * 2, ar * 2);
            mtree_path(&ftrees[l], x, sib); wb_push(&o, sib, 4 * (ftrees[l].logn - capH));
        }
    }
    wb_push(&o, (uint64_t*)coefs, 2 * fplen);
    wb_push1(&o, pow_w);
    wb_push(&o, pis, P.n_pis);
    *out = o.d; *out_words = o.n;

    for (int l = 0; l < L; l++) { mtree_free(&ftrees[l]); free(fleaves[l]); }
    (void)flen;
    if (coefs != fin) free(coefs);
    free(fin); free(fvals); free(caps.d);
    free(op_local); free(op_next); free(op_q);
    mtree_free(&ttree); mtree_free(&qtree);
    free(coeffs); free(lde); free(qchunks); free(qlde); free(P.code);
    return rc;
}

/* cpu_baseline helper for bench.py: the quotient phase's inner loop (every constraint, both alphas) at
 * n_points points taken from `rows` (row-major [n_points + 1][C]; point i uses rows i and i + 1).
 * Returns a checksum so the work cannot be optimised away. */
EXPORT fe oracle_bench_quotient(const uint64_t* air_blob, size_t air_words, const fe* rows, size_t n_points, const fe* pis) {
    air_prog p;
    if (air_parse(air_blob, air_words, &p)) return 0;
    fe sum = 0;
#pragma omp parallel
    {
        fe local = 0;
#pragma omp for schedule(dynamic, 4)
        for (size_t i = 0; i < n_points; i++) {
            fe masks[4] = {1, 3, 5, 7}, alphas[2] = {0x1234567, 0x7654321}, acc[2];
            air_eval(&p, rows + i * p.n_cols, rows + (i + 1) * p.n_cols, pis, masks, alphas, 2, acc, NULL);
            local = f_add(local, f_add(acc[0], acc[1]));
        }
#pragma omp critical(oracle_bench)
        sum = f_add(sum, local);
    }
    free(p.code);
    return sum;
}

EXPORT void oracle_free(void* p) { free(p); }
EXPORT fe oracle_mul(fe a, fe b) { return f_mul(a, b); }
EXPORT fe oracle_mul_slow(fe a, fe b) { return f_mul_slow(a, b); }
/* size of the OpenMP team of the calls that follow (a caller that knows its CPU quota: a container may show 256 hardware threads and
 * be entitled to 16 of them) */
EXPORT void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
EXPORT int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
