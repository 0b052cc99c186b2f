/* libstarkhip -- C ABI of the MI355X-native STARK prover for the four BLS12-381 AIRs of
 * Electron-Labs/starky_bls12_381.
 *
 * Drop-in boundary: each entry point replaces one call the reference makes into the
 * (un-vendored) starky / plonky2 crates or into its own AIR modules:
 *
 *   starkhip_prove            <- starky::prover::prove::<F, C, S, D>(stark, &config, trace_poly_values,
 *                                &public_inputs, &mut timing)         src/aggregate_proof.rs:59-65,
 *                                                                     :105-111, :138-144, :169-175
 *   starkhip_verify           <- starky::verifier::verify_stark_proof src/aggregate_proof.rs:67,113,146,177
 *   starkhip_config_standard_fast <- StarkConfig::standard_fast_config()  src/aggregate_proof.rs:32,76,122,155
 *   starkhip_air_*            <- the associated consts S::COLUMNS / S::PUBLIC_INPUTS / constraint_degree()
 *                                src/fp12_mul.rs:21-27,142-144; src/final_exponentiate.rs:1362-1364 ...
 *   starkhip_trace_*          <- S::generate_trace(..) + trace_rows_to_poly_values
 *                                src/fp12_mul.rs:44-48, src/final_exponentiate.rs:240-279,
 *                                src/miller_loop.rs, src/calc_pairing_precomp.rs:150-348;
 *                                src/aggregate_proof.rs:57,104,137,168
 *   starkhip_trace_ecc_aggregate <- ECCAggStark::generate_trace + the public inputs of ec_aggregate_main
 *                                src/ecc_aggregate.rs:39-84, src/aggregate_proof.rs:181-221
 *   starkhip_native_*         <- crate::native (Fp12 mul, final_exponentiate, miller_loop,
 *                                calc_pairing_precomp)               src/native.rs:1201-1468
 *
 * Because `S: Stark` is a compile-time generic in the reference, the AIR is selected by id.
 * All field elements are canonical Goldilocks values (< 2^64 - 2^32 + 1) in little-endian
 * uint64_t.  Fp elements are 12 little-endian u32 limbs (src/fp.rs:1).
 *
 * Threading: one in-flight prove per context; different contexts (GPUs) may run concurrently.
 * Ownership: the caller owns every input for the duration of the call; buffers returned through
 * `uint64_t** out` are owned by the library until starkhip_free().
 */
#ifndef STARKHIP_H
#define STARKHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint32_t security_bits;      /* 100 */
    uint32_t num_challenges;     /* 2   */
    uint32_t rate_bits;          /* 1 (2 for PairingPrecomp / FinalExp) */
    uint32_t cap_height;         /* 4   */
    uint32_t proof_of_work_bits; /* 16  */
    uint32_t arity_bits;         /* ConstantArityBits(4, 5) */
    uint32_t final_poly_bits;
    uint32_t num_query_rounds;   /* 84  */
} starkhip_config_t;

typedef enum {
    STARKHIP_AIR_FP12_MUL = 0,
    STARKHIP_AIR_PAIRING_PRECOMP = 1,
    STARKHIP_AIR_MILLER_LOOP = 2,
    STARKHIP_AIR_FINAL_EXP = 3,
    STARKHIP_AIR_ECC_AGGREGATE = 4, /* ECCAggStark: sum of the 512 sync-committee keys whose bit is set (src/ecc_aggregate.rs) */
    STARKHIP_AIR_TEST_FIBONACCI = 100 /* 2-column toy AIR used by the unit tests */
} starkhip_air_t;

enum {
    STARKHIP_OK = 0,
    STARKHIP_ERR_QUOTIENT_NOT_DIVISIBLE = -1, /* "Quotient has failed, the vanishing polynomial is not divisible by Z_H" */
    STARKHIP_ERR_ZETA_IN_SUBGROUP = -2, /* the challenge zeta lies in the trace subgroup H: starky's "Opening point is in the subgroup" (probability
                                          2^-115 per proof, deterministic for the input it hits).  zeta on the coset 7 H, where the trace polynomials'
                                          values are kept, is proven as the reference proves it (the opening is then the value itself) */
    STARKHIP_ERR_BAD_SHAPE = -3,
    STARKHIP_ERR_HIP = -4,
    STARKHIP_ERR_OOM = -5,
    STARKHIP_ERR_NO_DEVICE = -6,
    STARKHIP_ERR_VERIFY = -7,
    STARKHIP_ERR_BAD_AIR = -8
};

#define STARKHIP_POW_SEARCH UINT64_MAX

/* --- configuration ------------------------------------------------------------------- */
void starkhip_config_standard_fast(starkhip_config_t* cfg);
/* the per-AIR config the reference builds (rate_bits override for PairingPrecomp / FinalExp) */
int starkhip_config_for_air(starkhip_air_t air, starkhip_config_t* cfg);

/* --- AIR metadata -------------------------------------------------------------------- */
int starkhip_air_columns(starkhip_air_t air);
int starkhip_air_public_inputs(starkhip_air_t air);
int starkhip_air_constraint_degree(starkhip_air_t air);
int starkhip_air_num_constraints(starkhip_air_t air);
int starkhip_air_default_rows(starkhip_air_t air);
/* serialised constraint program (format: starky_bls12_381_amd/csrc/air_ir.h); library-owned */
int starkhip_air_program(starkhip_air_t air, const uint64_t** blob, size_t* words);
/* What one call of S::eval_packed_generic leaves in the ConstraintConsumer (the reference folds
 * acc_j = acc_j * alpha_j + mask(kind) * c_k constraint by constraint; e.g. src/final_exponentiate.rs:907-1136) on ONE
 * frame over the quadratic extension — the host evaluator the verifier uses.  All field arguments are (a0, a1) pairs:
 * local / next hold starkhip_air_columns() pairs, masks = {1, z_last, L_first, L_last} as the consumer applies them to
 * constraint / constraint_transition / constraint_first_row / constraint_last_row, acc_out n_alpha pairs.  public_inputs
 * are base-field.  Host only. */
int starkhip_air_eval_frame(starkhip_air_t air, const uint64_t* local, const uint64_t* next, const uint64_t* public_inputs,
                            const uint64_t masks[8], const uint64_t* alphas, int n_alpha, uint64_t* acc_out);

/* --- natives + trace generation (host) ----------------------------------------------- */
/* inputs are u32 limb arrays: Fp = 12, Fp2 = 24, Fp12 = 144 limbs.
 * trace is written row-major [n_rows][columns]; public_inputs has starkhip_air_public_inputs() entries. */
int starkhip_trace_fp12_mul(const uint32_t x[144], const uint32_t y[144], uint64_t* trace, size_t n_rows, uint64_t* public_inputs);
int starkhip_trace_final_exp(const uint32_t x[144], uint64_t* trace, size_t n_rows, uint64_t* public_inputs);
int starkhip_trace_miller_loop(const uint32_t px[12], const uint32_t py[12], const uint32_t qx[24], const uint32_t qy[24],
                               const uint32_t qz[24], uint64_t* trace, size_t n_rows, uint64_t* public_inputs);
int starkhip_trace_pairing_precomp(const uint32_t qx[24], const uint32_t qy[24], const uint32_t qz[24], uint64_t* trace,
                                   size_t n_rows, uint64_t* public_inputs);
/* points: 512 affine G1 points as [x(12 limbs), y(12 limbs)]; bits: 512 bytes (0 / 1).  The aggregate is computed here and
 * written to the last 24 public inputs.  At least one of the first two bits must be set; consecutive operands must differ in x. */
int starkhip_trace_ecc_aggregate(const uint32_t* points, const uint8_t* bits, uint64_t* trace, size_t n_rows, uint64_t* public_inputs);
int starkhip_native_g1_aggregate(const uint32_t* points, const uint8_t* bits, uint32_t out[24]);
int starkhip_trace_fibonacci(uint64_t x0, uint64_t x1, uint64_t* trace, size_t n_rows, uint64_t* public_inputs);
int starkhip_native_fp12_mul(const uint32_t x[144], const uint32_t y[144], uint32_t out[144]);
int starkhip_native_final_exponentiate(const uint32_t x[144], uint32_t out[144]);
int starkhip_native_miller_loop(const uint32_t px[12], const uint32_t py[12], const uint32_t qx[24], const uint32_t qy[24],
                                const uint32_t qz[24], uint32_t out[144]);
int starkhip_native_pairing_precomp(const uint32_t qx[24], const uint32_t qy[24], const uint32_t qz[24], uint32_t out[68 * 72]);

/* --- prover (GPU) -------------------------------------------------------------------- */
int starkhip_init(int device_ordinal, void** ctx);
void starkhip_shutdown(void* ctx);

/* trace_layout: 0 = row-major [n_rows][C] (what generate_trace returns), 1 = column-major [C][n_rows]
 * (what trace_rows_to_poly_values returns).  trace_on_device != 0: `trace` is a device pointer
 * (already resident in HBM; the benchmark path).  pow_witness: STARKHIP_POW_SEARCH = smallest valid
 * nonce, otherwise use the given one.  *proof is a blob in the layout below.
 * Shapes: n_rows a power of two, 2 <= n_rows <= 8192 (the reference's largest trace; one LDS image per column),
 * n_pis and n_cols as the AIR declares (n_cols is checked before the buffer is touched: it is read as n_rows x n_cols words), num_challenges = 2, and rate_bits large enough for the AIR's
 * constraint degree (2^rate_bits >= degree - 1); anything else is STARKHIP_ERR_BAD_SHAPE before any GPU work. */
int starkhip_prove(void* ctx, starkhip_air_t air, const starkhip_config_t* cfg, const uint64_t* trace, size_t n_rows, size_t n_cols,
                   int trace_layout, int trace_on_device, const uint64_t* public_inputs, size_t n_pis, uint64_t pow_witness,
                   uint64_t** proof, size_t* proof_words);

/* The same proof from the LITERAL argument of starky's prove(): `trace_poly_values: Vec<PolynomialValues<F>>`
 * (src/aggregate_proof.rs:57-65, :104-111, :137-144, :168-175) is one heap allocation per column -- 73 527 of them for FinalExp.
 * `columns` is a table of n_cols host pointers, each to n_rows canonical words (PolynomialValues<GoldilocksField>::values; the field
 * type is repr(transparent) over u64).  The library gathers the scattered columns through the context's page-locked staging, half
 * by half under the copies, so a binding at prove() itself needs no 4.8 GB host-side repack.  Byte-identical to starkhip_prove on
 * the same matrix; shapes and errors as starkhip_prove (a NULL column pointer: STARKHIP_ERR_BAD_SHAPE). */
int starkhip_prove_columns(void* ctx, starkhip_air_t air, const starkhip_config_t* cfg, const uint64_t* const* columns, size_t n_rows, size_t n_cols,
                           const uint64_t* public_inputs, size_t n_pis, uint64_t pow_witness, uint64_t** proof, size_t* proof_words);

/* Tuning knobs of a context (defaults are the measured best; tests and profiling tools use them to reach the other code
 * paths): "quotient_impl" 0 = tiled evaluator / 1 = op-stream interpreter, "quotient_chunks" (0 = automatic),
 * "quotient_waves", "quotient_slots", "lde_closed_forms" (1 = constant and unit-vector trace columns take their closed-form LDE
 * instead of five transforms, 0 = every column is transformed; same bytes either way), "host_commit_leaves" (default 64: trace commitments of at most this many leaves -- FP12Mul's 32 -- are hashed by host threads with the
 * challenger's permutation, 0.8 us against 5.6 us per permutation of a lone GPU wave; 0 = never; only with "leaf_hash_form" 0), "leaf_hash_form" (0 = a context on its own
 * hashes commitments of <= 4096 leaves in the row form -- 16 lanes per leaf, the shortest chain per leaf --, those of >= 32 768 leaves in
 * the pair form -- two lanes per leaf, one 256-register wave per SIMD -- and the ones between in the quad form; 1 = quad always; 2 = row always;
 * 3 = lane form, one lane per leaf: what a pool with five or more big contexts uses for groups of big commitments, slower than the pair form for
 * one commitment alone; 4 = pair always; same digests; a pool's commitments always go through its scheduler),
 * "lde_impl" (1 = 8192-row traces through lde_columns_v2_kernel instead of the wave-resident kernel; same bytes), "zeta_on_coset"
 * (TESTS ONLY: k + 1 substitutes zeta = 7 w_n^k for the transcript's challenge, a point of the coset the trace values are kept on -- the
 * branch a real transcript takes with probability 2^-115; the result is compared with the oracle under the same substitution and is not a
 * proof the verifier accepts; 0 = off).  Unknown name or value out of range: STARKHIP_ERR_BAD_SHAPE. */
int starkhip_set_option(void* ctx, const char* name, long value);

/* --- compact traces: on-device trace expansion -----------------------------------------------
 * Between starkhip_trace_log_begin and _end the calling thread's ONE starkhip_trace_* call records its writes as runs
 * (a limb vector + the rows it repeats on) instead of filling n_rows x C cells; its `trace` argument is ignored and may
 * be NULL, public inputs are produced as usual.  FinalExp: ~150 MB of records instead of 4.8 GB of rows.
 * starkhip_prove_compact uploads the records and expands them on the device straight into the column-major matrix;
 * the proof is bit-identical to the one from the dense trace.  A log is immutable after _end and may be proven any
 * number of times, from any thread; release it with starkhip_trace_log_free. */
int starkhip_trace_log_begin(void** log);
/* Host threads ONE recording generator call may use (process-wide, default 1; returns the previous value).  Only
 * starkhip_trace_final_exp uses more than one so far: the reference's generate_trace (src/final_exponentiate.rs:240-279) fills
 * its 32 ops one after the other, but once the 32 native results are known every op's rows are independent.  The recorded
 * trace -- and the proof -- does not depend on the setting.  A driver that records many traces at once keeps it at 1. */
int starkhip_trace_set_threads(int n);
int starkhip_trace_log_end(void* log);
void starkhip_trace_log_free(void* log);
int starkhip_trace_log_info(const void* log, size_t* n_rows, size_t* n_cols, size_t* n_records, size_t* n_words);
/* A finished log from explicit cell writes, (row, col, value) triples applied in order through the recorder's `set` (values < 2^32;
 * a zero clears as the fillers' "selector(b) = 0" does).  For tests of the recorder's corner cases -- the generators are the
 * product's way to make a log.  Release with starkhip_trace_log_free. */
int starkhip_trace_log_from_writes(size_t n_rows, size_t n_cols, const uint64_t* writes, size_t n_writes, void** log);
/* The two traces of the fillers' one overwriting idiom ("selector = 1 on rows a..b", then "selector(b) = 0") in a finished log:
 * records whose run the clear took back to ZERO rows (a one-row run cleared again: the record stays in the log and stands for no
 * cell -- kernels_trace.hip skips it) and cells cleared inside a longer run (zeroed after the expansion).  Tests. */
int starkhip_trace_log_overwrites(const void* log, size_t* empty_runs, size_t* late_zeros);
/* CPU replay into a row-major matrix (tests): *conflicts = cells that two records wrote with different values (must be 0) */
int starkhip_trace_log_expand_host(const void* log, uint64_t* trace_rowmajor, size_t* conflicts);
int starkhip_prove_compact(void* ctx, starkhip_air_t air, const starkhip_config_t* cfg, const void* log, const uint64_t* public_inputs,
                           size_t n_pis, uint64_t pow_witness, uint64_t** proof, size_t* proof_words);

/* --- proof pool: submit / wait ------------------------------------------------------------------------------------
 * The reference's caller proves one AIR after the other on one thread (src/aggregate_proof.rs:304-370: pp1, ml1, pp2, ml2,
 * fp12_mul, final_exp; `aggregate_proof` :402-414) and lets rayon fill the cores inside each prove().  On a GPU several
 * proofs in flight are what fills the chip, so the library schedules them itself: a pool owns `big_contexts` prover contexts
 * for the 8192-row AIRs (FinalExp, ECCAgg: 19.6 GB of buffers each) and `small_contexts` for the others, one host thread
 * per context, `generator_threads` host threads that record traces (starkhip_pool_submit_witness), and a commitment
 * scheduler: the trace commitments of small proofs that arrive together are hashed by ONE merged launch (and, by option, a
 * FinalExp-class commitment -- a one-shot grid that owns the chip -- never shares the chip with a small one).  Proofs are byte-identical to
 * starkhip_prove's.  submit returns at once with a ticket; wait blocks until that proof is done and hands it over
 * (starkhip_free), exactly once per ticket, from any thread.  Inputs of submit / submit_compact (trace, log, public inputs)
 * stay the caller's and must stay valid until the ticket has been waited for; submit_witness copies its operands.
 * Hardware queues: starkhip_pool_create sets GPU_MAX_HW_QUEUES=16 unless the variable exists, but the HIP runtime reads it only when
 * the process first uses HIP -- a process that has used HIP before it creates its first pool exports the variable itself, earlier
 * (with HIP's default of 4 queues the same pool is 6 % slower: INTEGRATION.md). */
typedef struct {
    int device;
    unsigned big_contexts;      /* 0 = default (3).  Five or more: the trace commitments of these proofs go out in groups of up to four in the
                                   lane form of the leaf hash (6.3 against 5.65 proofs/s on one MI355X; 19.6 GB of HBM per context,
                                   starkhip_pool_reservation) */
    unsigned small_contexts;    /* 0 = default (16) */
    unsigned generator_threads; /* recordings under way at once; 0 = default (a quarter of the CPUs the process may use -- its
                                   cgroup quota or affinity mask --, 3 .. 12) */
    unsigned trace_threads;     /* host threads ONE recording may use; 0 = automatic (FinalExp-class: 3/4 of the CPU budget, at most 16;
                                   the others at most 4) */
    unsigned commit_policy;     /* 0 = default: small commitments that arrive together share one launch; 1 = in addition a FinalExp-class
                                   commitment never shares the chip with a small one; 2 = no commitment scheduling (every context
                                   launches its own; for A/B measurements) */
    unsigned stream_priority;   /* 0 = all contexts alike; 1 = the last wave of FinalExp-class proofs (no more of them waiting than there are
                                   contexts) on high-priority streams; 2 = the small contexts, 3 = all FinalExp-class proofs (measurements) */
    unsigned warm_up;           /* != 0: starkhip_pool_create returns when every context has allocated what the BLS pipeline's AIRs of its class
                                   need (FinalExp on the big contexts; MillerLoop, PairingPrecomp, FP12Mul on the small ones: tables, constraint
                                   plans, buffers, upload staging), so that no proof pays for -- or stalls the device with -- allocations;
                                   2: the same without the page-locked upload staging: for a caller whose traces are already column-major
                                   device memory (starkhip_pool_submit with trace_on_device) */
    float gather_ms;            /* how long a merged launch waits for small proofs that have started but not reached their commitment; 0 = default (25) */
} starkhip_pool_config_t;
typedef struct {
    float phase_ms[11];   /* as starkhip_last_timings */
    float kernel_ms[3];   /* as starkhip_last_kernel_timings */
    float host_ms[2];     /* as starkhip_last_host_timings */
    double t_submit, t_generate_start, t_generate_end, t_prove_start, t_done; /* seconds since the pool was created */
    int leaf_hash_form;       /* how the trace commitment went out: 0 quad form, 1 row form, 2 one grid merged with other proofs' commitments
                                 (quad form), 3 lane form, 4 hashed by host threads (at most "host_commit_leaves" leaves), 5 pair form; kernel_ms[1] is
                                 that kernel's own duration on its launch stream (form 4: the host's time) */
    unsigned leaf_hash_group; /* commitments that were launched side by side with it (itself included) */
} starkhip_ticket_info_t;
typedef struct {
    unsigned long big_commit_launches, small_commit_launches, small_commit_requests, max_merged_commitments;
} starkhip_pool_stats_t;
int starkhip_pool_create(const starkhip_pool_config_t* cfg, void** pool);
void starkhip_pool_destroy(void* pool); /* runs what is queued to the end first */
/* as starkhip_prove / starkhip_prove_compact */
int starkhip_pool_submit(void* pool, starkhip_air_t air, const starkhip_config_t* cfg, const uint64_t* trace, size_t n_rows, size_t n_cols,
                         int trace_layout, int trace_on_device, const uint64_t* public_inputs, size_t n_pis, uint64_t pow_witness,
                         uint64_t* ticket);
/* as starkhip_prove_columns; the pointer TABLE is copied before this returns, the columns themselves (and the public inputs) stay the
 * caller's until the ticket has been waited for */
int starkhip_pool_submit_columns(void* pool, starkhip_air_t air, const starkhip_config_t* cfg, const uint64_t* const* columns, size_t n_rows,
                                 size_t n_cols, const uint64_t* public_inputs, size_t n_pis, uint64_t pow_witness, uint64_t* ticket);
int starkhip_pool_submit_compact(void* pool, starkhip_air_t air, const starkhip_config_t* cfg, const void* log, const uint64_t* public_inputs,
                                 size_t n_pis, uint64_t pow_witness, uint64_t* ticket);
/* generate_trace + prove (src/aggregate_proof.rs:23-179, one driver each) from the driver's operands as u32 limbs, packed:
 *   FP12_MUL x[144] y[144];  FINAL_EXP x[144];  MILLER_LOOP px[12] py[12] qx[24] qy[24] qz[24];  PAIRING_PRECOMP qx[24] qy[24] qz[24];
 *   ECC_AGGREGATE points[512][24] then 512 words of 0 / 1;  TEST_FIBONACCI x0 (lo, hi) x1 (lo, hi).
 * The trace is recorded on a generator thread of the pool (default rows of the AIR); cfg == NULL: starkhip_config_for_air.
 * The public inputs are the tail of the proof blob. */
int starkhip_pool_submit_witness(void* pool, starkhip_air_t air, const starkhip_config_t* cfg, const uint32_t* operands, size_t n_limbs,
                                 uint64_t pow_witness, uint64_t* ticket);
/* returns the proof's status (what starkhip_prove would have returned); info may be NULL */
int starkhip_pool_wait(void* pool, uint64_t ticket, uint64_t** proof, size_t* proof_words, starkhip_ticket_info_t* info);
int starkhip_pool_stats(void* pool, starkhip_pool_stats_t* out);
/* What the pool has reserved (a warmed pool: everything its proofs will ever need; read it between proofs): device memory of all
 * contexts, page-locked upload staging, and the largest context of each class.  A FinalExp-class context holds the LDE (19.3 GB; the trace
 * columns wait for it inside that buffer, row-major uploads and recordings are staged in it before the LDE kernel writes it, and
 * coefficients are not kept) and ~ 0.3 GB of small buffers. */
typedef struct {
    uint64_t device_bytes, pinned_host_bytes, big_context_device_bytes, small_context_device_bytes;
    unsigned big_contexts, small_contexts;
} starkhip_pool_reservation_t;
int starkhip_pool_reservation(void* pool, starkhip_pool_reservation_t* out);
/* The host side of a pool: the CPUs it plans with -- the process's budget (cgroup quota, else affinity mask, divided by
 * LOCAL_WORLD_SIZE under torch.distributed.run and by the number of pools of a multi-device handle: starkhip_cpu_budget is the
 * process's figure before that last division) --, its generator threads, the threads one recording of each class may use, and its
 * prover threads (one per context; they launch kernels, hash the Fiat-Shamir transcript and gather uploads).  A scaling curve that
 * bends for want of host CPUs shows here: a pool whose budget is 2 records one FinalExp trace at a time on one thread (212 ms). */
typedef struct {
    unsigned cpu_budget, generator_threads, trace_threads_big, trace_threads_small, prover_threads;
    int device;
    unsigned pools_on_device; /* pools of the same multi-device handle on this pool's device: 1, unless the handle was given one ordinal
                               * several times -- the rehearsal of N devices on one card (the tests do it); such a handle's figures are not a
                               * measurement of N devices, and starkhip_multipool_create says so once on stderr */
} starkhip_pool_host_info_t;
int starkhip_pool_host_info(void* pool, starkhip_pool_host_info_t* out);
unsigned starkhip_cpu_budget(void);
/* Where the pools' host CPU time goes, process-wide and cumulative (seconds of CPU, not of wall): [0] recording traces (the generator
 * threads inside starkhip_trace_* and the helper threads of multi-threaded recordings), [1] the context threads inside prove() (kernel
 * launches, the challenger's Fiat-Shamir sponge, gathering uploads, the FRI batches' host arithmetic), [2] the part of [1] spent INSIDE the waits for the device (an event wait that sleeps costs next to nothing).  What a process's
 * total CPU time holds beyond these two is the HIP runtime's own threads and the caller. */
void starkhip_host_cpu_seconds(double out[3]);
/* Proof blobs of a warmed pool are recycled page-locked buffers (the final device-to-host copy of 21 .. 69 MB runs at PCIe rate and
 * touches no fresh pages); starkhip_free() hands them back.  Process-wide counters: [0] blobs held, [1] of them with a caller,
 * [2] bytes held, [3] proofs served from them, [4] proofs served by malloc (no idle blob that fits).  STARKHIP_PINNED_PROOFS=0 in
 * the environment of starkhip_pool_create turns the reservation off. */
void starkhip_proof_blob_stats(uint64_t out[5]);

/* 0: GPU_MAX_HW_QUEUES was in the environment before the HIP runtime came up (the library's 16 or the caller's own value); 1: the
 * runtime was already initialised when the first pool / multi-device handle of this process set it, so the pools run on HIP's default
 * of 4 hardware queues (a line on stderr says so once).  Detected without touching HIP (the runtime's open /dev/kfd). */
int starkhip_hw_queues_status(void);

/* --- one caller, many devices -------------------------------------------------------------------------------------
 * The reference's caller is ONE process: generate_aggregate_proof issues its six proves from one thread (src/aggregate_proof.rs:304-370,
 * `aggregate_proof` :402-414).  A multi-device handle lets that caller use a node of GPUs as it is -- no process group, no collective
 * (the proofs are independent; operands reach every pool through host memory): one proof pool per entry of
 * `devices` (an ordinal may repeat: two pools on one card), each configured by `cfg` (cfg->device is ignored), the process's CPU budget
 * split between them.  Jobs are placed longest processing time first: `slot` = -1 lets the library choose -- a FinalExp-class job goes
 * to the pool with the fewest of them open, any job to the pool with the least outstanding cost (starkhip_air_cost), ties to the lowest
 * slot -- and `slot` >= 0 names the pool (required for a trace in device memory: it lives on one device).  submit_witness_batch places a
 * whole batch in order of decreasing cost (ties in the caller's order): BASELINE configs[3] (one signature's six proofs on six
 * devices: each pool gets one) and configs[4] (48 proofs on 8: every device gets a FinalExp proof first, then two MillerLoop, ...).
 * Tickets are the handle's own (they carry the slot); everything else -- proofs, ownership, errors, ticket info -- is as starkhip_pool_*.
 * starkhip_multipool_pool hands out the pool of a slot for starkhip_pool_stats / _reservation (not for submit / wait / destroy). */
int starkhip_multipool_create(const int* devices, size_t n_devices, const starkhip_pool_config_t* cfg, void** mpool);
void starkhip_multipool_destroy(void* mpool); /* runs what is queued to the end first */
size_t starkhip_multipool_size(const void* mpool);
void* starkhip_multipool_pool(void* mpool, size_t slot);
int starkhip_multipool_device(const void* mpool, size_t slot);
int starkhip_multipool_submit(void* mpool, int slot, starkhip_air_t air, const starkhip_config_t* cfg, const uint64_t* trace, size_t n_rows,
                              size_t n_cols, int trace_layout, int trace_on_device, const uint64_t* public_inputs, size_t n_pis,
                              uint64_t pow_witness, uint64_t* ticket);
int starkhip_multipool_submit_columns(void* mpool, int slot, starkhip_air_t air, const starkhip_config_t* cfg, const uint64_t* const* columns,
                                      size_t n_rows, size_t n_cols, const uint64_t* public_inputs, size_t n_pis, uint64_t pow_witness,
                                      uint64_t* ticket);
int starkhip_multipool_submit_compact(void* mpool, int slot, starkhip_air_t air, const starkhip_config_t* cfg, const void* log,
                                      const uint64_t* public_inputs, size_t n_pis, uint64_t pow_witness, uint64_t* ticket);
int starkhip_multipool_submit_witness(void* mpool, int slot, starkhip_air_t air, const starkhip_config_t* cfg, const uint32_t* operands,
                                      size_t n_limbs, uint64_t pow_witness, uint64_t* ticket);
/* operands[i] / n_limbs[i] as starkhip_pool_submit_witness takes them; tickets[i] = 0 and rcs[i] (may be NULL) != 0 for a job that was
 * refused; returns the first failure */
int starkhip_multipool_submit_witness_batch(void* mpool, size_t n_jobs, const starkhip_air_t* airs, const uint32_t* const* operands,
                                            const size_t* n_limbs, uint64_t pow_witness, uint64_t* tickets, int* rcs);
int starkhip_multipool_ticket_slot(const void* mpool, uint64_t ticket); /* -1: not a ticket of this handle */
int starkhip_multipool_wait(void* mpool, uint64_t ticket, uint64_t** proof, size_t* proof_words, starkhip_ticket_info_t* info);
/* the placement rule alone (no GPU needed): the slot each job of a batch gets on n_pools idle pools; starkhip_air_cost is the relative
 * single-GPU proving cost the rule uses (FinalExp 92, MillerLoop 12.5, PairingPrecomp 4.5, ECCAgg 3, FP12Mul 0.22) */
int starkhip_plan_lpt(size_t n_jobs, const starkhip_air_t* airs, size_t n_pools, int* slots);
double starkhip_air_cost(starkhip_air_t air);

/* per-phase device timings of the last prove on this ctx, milliseconds (HIP events):
 * [0] upload/transpose [1] ifft+lde [2] trace leaf hash + merkle [3] quotient [4] quotient commit
 * [5] openings [6] fri combine [7] fri commit [8] pow [9] queries [10] total */
#define STARKHIP_N_PHASES 11
int starkhip_last_timings(void* ctx, float ms[STARKHIP_N_PHASES]);
/* durations (ms, HIP events on the launch stream) of the three heavy kernels of the last prove:
 * [0] the trace's LDE kernel, the sum of its launches (lde_columns_wave_kernel for 8192 rows, lde_columns_v2_kernel for the other sizes
 * from 2^8 on and with "lde_impl" = 1) [1] the trace commitment's leaf hash in the form it went out in (leaf_hash_kernel / _row_kernel /
 * _lane_kernel / _pair_kernel / merged) [2] quotient_tiles_kernel */
int starkhip_last_kernel_timings(void* ctx, float ms[3]);
/* host time (ms, wall) inside the last prove: [0] Fiat-Shamir hashing -- the challenger's sequential Poseidon sponge over the
 * caps, the 2 (2 C + Q) opening words and the FRI data, on the proof's critical path (it falls inside the device phases
 * "fri_combine" and "fri_commit" above) -- [1] the other host arithmetic (the two divisions by X - z of the FRI batches) */
int starkhip_last_host_timings(void* ctx, float ms[2]);

/* Page-locked, reusable host memory for traces (starkhip_trace_* take any pointer).  Measured on FinalExp (4.8 GB of
 * rows): the upload itself already runs at link speed from pageable memory (86 ms, 56 GB/s) and stays there; what a
 * long-lived buffer saves is the first-touch page faulting of a fresh 4.8 GB allocation per trace (host generation
 * 815 -> 315 ms).  Needs an initialised context; release with starkhip_host_free. */
int starkhip_host_alloc(void* ctx, size_t bytes, void** out);
void starkhip_host_free(void* p);

/* --- kernel-level entry points (parity tests / micro-benchmarks) ---------------------- */
/* values column-major [C][n] (host) -> coeffs [C][n] and LDE [C][N] in NATURAL point order i <-> 7*w_N^i */
int starkhip_lde_batch(void* ctx, const uint64_t* values, size_t n_cols, unsigned log_n, unsigned rate_bits, uint64_t* coeffs_out,
                       uint64_t* lde_out);
/* Merkle cap of the matrix whose leaf j is the row bitrev(j) of an LDE given column-major natural order [C][N] */
int starkhip_merkle_cap(void* ctx, const uint64_t* lde_colmajor, size_t n_cols, unsigned log_N, unsigned cap_height, uint64_t* cap_out);
int starkhip_poseidon_permute_batch(void* ctx, uint64_t* states, size_t n_states);
/* micro-benchmark: the values -> LDE kernel prove() uses for this shape on n_cols synthetic columns already in HBM, `const_per_64` of
 * every 64 of them constant (these take a closed form; a FinalExp trace has 11 in 64), `reps` launches between two HIP events; average
 * milliseconds per launch.  const_per_64 + 256: unit vectors instead of constants.  device_values != NULL: the caller's own column-major
 * matrix in device memory (n_cols x 2^log_n words) instead of the synthetic one.  reps == 0: one launch with no warm-up launch in front
 * of it.  each_ms (may be NULL): min(reps, 16) durations, launch by launch */
int starkhip_lde_bench(void* ctx, size_t n_cols, unsigned log_n, unsigned rate_bits, unsigned reps, unsigned const_per_64, const uint64_t* device_values,
                       float* ms_per_launch, float* each_ms);
/* a finished log through the device's expansion kernels (csrc/kernels_trace.hip) into a host matrix, COLUMN-major [C][n_rows] */
int starkhip_trace_log_expand_device(void* ctx, const void* log, uint64_t* trace_colmajor);
/* device field arithmetic under test: out[i] = canonical(op(a[i], b[i])) with the lazy-reduction helpers the kernels use
 * (op codes: starky_bls12_381_amd/csrc/kernels_selftest.hip); lets the tests feed boundary operands */
int starkhip_field_ops_batch(void* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
/* CPU check (no GPU needed) of the constant tables the leaf-hash kernels use for their merged partial rounds -- three at a time in the
 * quad form, four at a time in the lane and pair forms: replays both formulations on n_states inputs against the plain permutation;
 * returns the number of mismatches (0 = good), -1 if a sum of the four-round merge would not fit its 64-bit accumulator */
int starkhip_selfcheck_hash_tables(unsigned n_states);
/* The launch plan of a trace's LDE (no GPU needed).  The trace columns wait for the LDE inside the buffer the LDE is written to, as
 * its last n_cols * n words, so the LDE goes out in several launches, each overwriting only parked columns that an earlier launch
 * has read; the last one reads a copy of its columns (csrc/lde_ranges.h).  Writes up to `cap` launches as triples
 * {first column, end column, 1 if it reads the copy} and returns how many the plan has. */
size_t starkhip_lde_launch_ranges(size_t n_cols, unsigned rate_bits, uint64_t* triples, size_t cap);
/* CPU check (no GPU needed) of the tiled constraint plan the quotient kernel executes (csrc/quotient_plan.h): builds the plan
 * of `air` with `want_chunks` chunks, derives the per-proof weights from random alphas / public inputs and replays the
 * record streams on one random frame with the kernel's own accumulators; the result must equal the plain fold
 * acc = acc * alpha + mask * c_k over all constraints.  stats = {chunks, supergroups, pieces, records, LDS cell records,
 * direct loads, tiles, term contributions}.  0 = equal, STARKHIP_ERR_VERIFY = different, BAD_SHAPE = malformed plan. */
int starkhip_quotient_plan_check(starkhip_air_t air, unsigned want_chunks, uint64_t seed, uint64_t stats[8]);
/* host-side permutation (the one the Fiat-Shamir challenger uses) */
void starkhip_poseidon_permute_host(uint64_t state[12]);
/* n chained host permutations; which = 0: the challenger's tuned permutation, 1: the portable loop */
void starkhip_poseidon_permute_host_many(uint64_t state[12], size_t n, int which);

/* --- verifier (CPU) ------------------------------------------------------------------- */
int starkhip_verify(starkhip_air_t air, const starkhip_config_t* cfg, const uint64_t* proof, size_t proof_words);

void starkhip_free(void* p);
const char* starkhip_error_string(int code);

/* Proof blob, uint64 words (canonical field order of StarkProofWithPublicInputs):
 *   [0..16)  header: magic "SSPRF001", C, Q, degree_bits, rate_bits, cap_height, L (FRI layers),
 *            num_query_rounds, final_poly_len, n_pis, arity_bits, num_challenges, 0,0,0,0
 *   trace_cap[2^cap_h][4]; quotient_polys_cap[2^cap_h][4]
 *   openings.local_values[C][2]; openings.next_values[C][2]; openings.quotient_polys[Q][2]
 *   commit_phase_merkle_caps[L][2^cap_h][4]
 *   query_round_proofs[num_query_rounds]:
 *       trace leaf[C], trace siblings[log N - cap_h][4], quotient leaf[Q], quotient siblings[log N - cap_h][4],
 *       steps[L]: evals[2^arity][2], siblings[log(N / arity^(l+1)) - cap_h][4]
 *   final_poly[final_poly_len][2]; pow_witness; public_inputs[n_pis]
 */
#define STARKHIP_PROOF_MAGIC 0x3130304652505353ULL

/* Offsets (in uint64 words) of every field of the blob above, so that a caller fills the fields of
 * StarkProofWithPublicInputs (src/aggregate_proof.rs:59, consumed at :67 and :435-439) without parsing the layout by hand.
 * q_* are offsets inside one query round (each round is query_round_words long, round r starts at
 * off_query_rounds + r * query_round_words); siblings are 4 words each. */
typedef struct {
    size_t n_columns, n_quotient_polys, degree_bits, rate_bits, cap_height, n_fri_layers, n_query_rounds, final_poly_len,
        n_public_inputs, arity_bits;
    size_t off_trace_cap, off_quotient_cap, off_local_values, off_next_values, off_quotient_openings, off_fri_caps, off_query_rounds,
        query_round_words, off_final_poly, off_pow_witness, off_public_inputs, total_words;
    size_t q_trace_leaf, q_trace_siblings, q_quotient_leaf, q_quotient_siblings, initial_sibling_count;
    size_t q_step_evals[16], q_step_siblings[16], step_sibling_count[16];
} starkhip_proof_layout_t;
/* STARKHIP_ERR_BAD_SHAPE when the blob's header is not a proof header or disagrees with proof_words */
int starkhip_proof_layout(const uint64_t* proof, size_t proof_words, starkhip_proof_layout_t* out);

#ifdef __cplusplus
}
#endif
#endif
