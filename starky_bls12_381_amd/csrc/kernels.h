// Launchers of the gfx950 kernels (definitions in kernels_*.hip).  All launches are asynchronous on `st`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gl.h"

// First statement of every kernel except the lane-form leaf hash: its waves raise their issue priority (s_setprio 2).  A lane-form group
// of four saturates the SIMDs' issue slots for a third of a second; whatever else is resident beside it in that time -- the LDE and the
// quotient of the proofs that are not in the group, Merkle levels, openings, the small AIRs' commitments -- would get the slots the
// arbiter's round robin leaves them and crawl (an LDE measured 255 ms in flight against 17 alone).  At a raised priority those waves
// issue first and the hash waves take every other slot: the same work, but the proofs outside the group reach THEIR commitment sooner
// and the next group is ready when this one ends.  Measured on 16 hardware queues, 48 proofs from operands: 7.02 - 7.08 proofs/s without,
// 7.57 - 7.58 with (profiles/r04_ab_experiments.txt 19; on HIP's default of four queues, where bench.py ran until the end of round 4,
// the same build measured 5.79 against 5.90: kernels of different proofs waited behind each other in shared queues anyway).
// -DSTARKHIP_NO_PRIO builds without it.
#ifndef STARKHIP_NO_PRIO
#define STARKHIP_PRIO_ENTRY __builtin_amdgcn_s_setprio(2);
#else
#define STARKHIP_PRIO_ENTRY
#endif

namespace starkhip {
struct QOp;  // quotient_ops.h
struct QTRec;  // quotient_plan.h
struct QTStream;
struct QTContrib;
}

namespace starkhip {

// kernels_ntt.hip
hipError_t launch_fill_powers(gl_t* out, gl_t base, gl_t mult, size_t count, hipStream_t st);  // out[i] = base * mult^i
hipError_t launch_fill_coset_scale(gl_t* out, unsigned log_n, unsigned rate_bits, hipStream_t st);
hipError_t launch_transpose(const gl_t* in, gl_t* out, size_t rows, size_t cols, hipStream_t st);
// compact trace log (trace_log.h) -> column-major values[col][row]; `values` must be zeroed
hipError_t launch_expand_trace(const uint32_t* words, const uint32_t* offsets, size_t n_records, gl_t* values, size_t n_rows, hipStream_t st);
hipError_t launch_zero_cells(const uint32_t* col_row, size_t n_cells, gl_t* values, size_t n_rows, hipStream_t st);
// from_coeffs == 0: `values` holds evaluations on the subgroup (PolynomialBatch::from_values);
// from_coeffs != 0: `values` already holds coefficients (PolynomialBatch::from_coeffs), coeffs_out unused.
// 2^1 .. 2^7 rows (radix-2 in LDS, kernels_ntt.hip)
hipError_t launch_lde_columns(const gl_t* values, gl_t* coeffs, gl_t* lde, size_t n_cols, unsigned log_n, unsigned rate_bits,
                              const gl_t* tw_fwd, const gl_t* tw_inv, unsigned tw_log, const gl_t* coset_scale, int from_coeffs,
                              hipStream_t st);
// 2^8 .. 2^13 rows (register radix-16 Stockham passes, kernels_lde.hip) with its own host-built tables
bool lde_v2_supported(unsigned log_n);
size_t lde_v2_tw_words(unsigned log_n);
size_t lde_v2_oh_words(unsigned log_n, unsigned rate_bits);  // closed-form tables of unit-vector columns (0: shape without them)
hipError_t lde_v2_upload_tables(unsigned log_n, unsigned rate_bits, gl_t* d_tw_fwd, gl_t* d_tw_inv, gl_t* d_cs, gl_t* d_oh, hipStream_t st);
hipError_t launch_lde_columns_v2(const gl_t* values, gl_t* coeffs, gl_t* lde, size_t n_cols, unsigned log_n, unsigned rate_bits,
                                 const gl_t* tw_fwd, const gl_t* tw_inv, const gl_t* cs, const gl_t* oh, int from_coeffs, hipStream_t st);
// 2^13 rows, values -> LDE only (no coefficient output): the wave-resident kernel (kernels_lde.hip; tools/lde_wave_model.py)
bool lde_wave_supported(unsigned log_n);
size_t lde_wave_table_words(unsigned rate_bits);
hipError_t lde_wave_upload_tables(unsigned rate_bits, gl_t* d_tab, hipStream_t st);
hipError_t launch_lde_columns_wave(const gl_t* values, gl_t* lde, size_t n_cols, unsigned rate_bits, const gl_t* d_tab, const gl_t* oh, unsigned* next,
                                   hipStream_t st);
hipError_t launch_ntt_global(gl_t* data, size_t n_vecs, size_t vec_stride, unsigned log_n, const gl_t* tw, unsigned tw_log,
                             const gl_t* pre_scale, const gl_t* post_scale, gl_t final_mul, hipStream_t st);

// kernels_hash.hip
// up to LEAF_HASH_MAX_BATCH matrices of one shape hashed by one launch (the trace commitments of proofs of the same AIR)
static const unsigned LEAF_HASH_MAX_BATCH = 64;
struct LeafHashBatch {
    const gl_t* mat[LEAF_HASH_MAX_BATCH];
    gl_t* digests[LEAF_HASH_MAX_BATCH];
};
hipError_t launch_leaf_hash_multi(const LeafHashBatch& B, unsigned count, size_t n_cols, unsigned log_n, unsigned rate_bits, hipStream_t st);
hipError_t launch_leaf_hash(const gl_t* mat, size_t n_cols, unsigned log_n, unsigned rate_bits, gl_t* digests, hipStream_t st);
// the same digests from the row form (16 lanes per leaf): shorter chain per leaf, 4x the lane-instructions -- for a lone commitment of few leaves
// the same digests from the lane form (one lane per leaf): fewest instructions per permutation, but 1/4 of the waves -- for big commitments when several are in flight
hipError_t launch_leaf_hash_lane(const gl_t* mat, size_t n_cols, unsigned log_n, unsigned rate_bits, gl_t* digests, hipStream_t st);
// the same digests from the pair form (two lanes per leaf, one 256-register wave per SIMD at 32 768 leaves): a LONE big commitment
hipError_t launch_leaf_hash_pair(const gl_t* mat, size_t n_cols, unsigned log_n, unsigned rate_bits, gl_t* digests, hipStream_t st);
hipError_t launch_leaf_hash_row(const gl_t* mat, size_t n_cols, unsigned log_n, unsigned rate_bits, gl_t* digests, hipStream_t st);
hipError_t launch_leaf_hash_rows(const gl_t* rows, size_t width, size_t n_leaves, gl_t* digests, hipStream_t st);
hipError_t launch_merkle_levels(gl_t* digests, unsigned log_leaves, unsigned cap_h, hipStream_t st);
hipError_t launch_permute_batch(gl_t* states, size_t n, hipStream_t st);
// kernels_selftest.hip
// CPU replay of the leaf-hash kernel's merged-partial-round tables against the plain permutation; mismatching states out of n
int quad_merged_tables_selfcheck(unsigned n);
int merged_fours_selfcheck(unsigned n);   // poseidon_host.cpp: the same for the four-round merges of the lane and pair forms
hipError_t launch_field_ops(int op, const gl_t* a, const gl_t* b, gl_t* out, size_t n, hipStream_t st);
hipError_t launch_pow_grind(const gl_t* base_state, int pos, unsigned pow_bits, uint64_t start, uint64_t count, unsigned long long* best,
                            hipStream_t st);

// kernels_quotient.hip
hipError_t launch_quotient_tables(gl_t* tab, unsigned log_n, unsigned qdb, hipStream_t st);
hipError_t launch_quotient_eval(const QOp* ops, const uint32_t* loads, unsigned n_slots, const uint32_t* chunk_batch, unsigned n_chunks,
                                const gl_t* pis, const gl_t* lde, const gl_t* tab, const gl_t* apow, gl_t alpha0, gl_t alpha1, gl_t* partial,
                                unsigned log_n, unsigned rate_bits, unsigned qdb, hipStream_t st);
hipError_t launch_quotient_combine(const gl_t* partial, const gl_t* chunk_scale, unsigned n_chunks, const gl_t* tab, unsigned log_n,
                                   unsigned qdb, gl_t* out, hipStream_t st);

// tiled evaluator (quotient_plan.h): per-proof record weights, the LDS-tiled pass, the sum over chunks / Z_H
hipError_t launch_quotient_weights(QTRec* recs, const uint32_t* contrib_off, const QTContrib* contribs, uint32_t n_recs, gl_t* apow, uint32_t K,
                                   const gl_t* consts, const gl_t* pis, gl_t alpha0, gl_t alpha1, hipStream_t st);
hipError_t launch_quotient_tiles(const QTRec* recs, const QTStream* streams, const uint32_t* chunk_tile_off,
                                 const uint32_t* tile_list, unsigned n_chunks, const gl_t* lde, const gl_t* tab, gl_t* partial, unsigned log_n,
                                 unsigned rate_bits, unsigned qdb, unsigned n_cols, unsigned dbg, hipStream_t st);
hipError_t launch_quotient_tiles_combine(const gl_t* partial, unsigned n_chunks, const gl_t* tab, unsigned log_n, unsigned qdb, gl_t* out,
                                         hipStream_t st);

// kernels_fri.hip
hipError_t launch_ext_powers(gl2_t* out, gl2_t base, size_t count, hipStream_t st);
// weights of the evaluation at z (and, rotated by one, at w_n z) from values on coset 0 of the LDE; scale = (z^n - 7^n) / (n 7^n)
hipError_t launch_coset_weights(gl2_t* wz, gl2_t* wgz, gl2_t z, gl2_t scale, unsigned log_n, hipStream_t st);
// vectors of n words, `stride` words apart (coefficients: stride = n; coset 0 of an LDE: stride = 2^rate n)
hipError_t launch_openings(const gl_t* coeffs, size_t stride, size_t n_polys, size_t n, const gl2_t* zpow, const gl2_t* gzpow, gl2_t* out_z,
                           gl2_t* out_gz, hipStream_t st);
hipError_t launch_fri_combine(const gl_t* coeffs, size_t stride, size_t n_polys, size_t n, const gl2_t* apow, size_t polys_per_chunk,
                              size_t n_chunks, gl2_t* partial, hipStream_t st);
hipError_t launch_ext_reduce(const gl2_t* partial, size_t n_chunks, size_t n, gl_t* out, hipStream_t st);  // two vectors of n words
hipError_t launch_fri_leaves(const gl_t* vals, unsigned log_len, unsigned arity_bits, gl_t* rows, hipStream_t st);
hipError_t launch_fri_fold(const gl_t* in, size_t len, unsigned arity_bits, gl2_t beta, gl_t* out, hipStream_t st);

// kernels_query.hip: query-round leaves and Merkle paths written in proof-blob layout (stride = words per query round)
hipError_t launch_query_leaf_colmajor(const gl_t* mat, size_t n_cols, unsigned log_n, unsigned rate_bits, const uint32_t* xs, size_t n_queries,
                                      gl_t* out, size_t stride, size_t off, hipStream_t st);
hipError_t launch_query_leaf_rows(const gl_t* rows, size_t width, const uint32_t* xs, unsigned shift, size_t n_queries, gl_t* out, size_t stride,
                                  size_t off, hipStream_t st);
hipError_t launch_query_path(const gl_t* digests, size_t n_leaves, unsigned depth, const uint32_t* xs, unsigned shift, size_t n_queries, gl_t* out,
                             size_t stride, size_t off, hipStream_t st);

}  // namespace starkhip
