// extern "C" surface of libstarkhip.so (declared in include/starkhip.h).
#include <dirent.h>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <new>

#include "air_eval.h"
#include "airs.h"
#include "blob_arena.h"
#include "kernels.h"
#include "lde_ranges.h"
#include "trace_log.h"

#include <atomic>
#include "poseidon.h"
#include "prover.h"
#include "proof.h"
#include "quotient_plan.h"

using namespace starkhip;

namespace starkhip {
TraceLog*& armed_trace_log() {
    static thread_local TraceLog* armed = nullptr;
    return armed;
}
static std::atomic<int> g_trace_threads(1);
static thread_local int tl_trace_threads = 0;  // a pool's generator thread sets its own share (scheduler.cpp)
int trace_threads() { return tl_trace_threads > 0 ? tl_trace_threads : g_trace_threads.load(); }
void set_thread_trace_threads(int n) { tl_trace_threads = n < 0 ? 0 : n > 64 ? 64 : n; }
}  // namespace starkhip

extern "C" {

void starkhip_config_standard_fast(starkhip_config_t* cfg) {
    // starky StarkConfig::standard_fast_config(): security 100, 2 challenges, FRI rate_bits 1, cap_height 4,
    // proof_of_work_bits 16, ConstantArityBits(4, 5), 84 query rounds
    cfg->security_bits = 100;
    cfg->num_challenges = 2;
    cfg->rate_bits = 1;
    cfg->cap_height = 4;
    cfg->proof_of_work_bits = 16;
    cfg->arity_bits = 4;
    cfg->final_poly_bits = 5;
    cfg->num_query_rounds = 84;
}

int starkhip_config_for_air(starkhip_air_t air, starkhip_config_t* cfg) {
    starkhip_config_standard_fast(cfg);
    switch (air) {
        case STARKHIP_AIR_PAIRING_PRECOMP:  // src/aggregate_proof.rs:32-33
        case STARKHIP_AIR_ECC_AGGREGATE:    // src/aggregate_proof.rs:186-187
        case STARKHIP_AIR_FINAL_EXP:        // src/aggregate_proof.rs:155-156
            cfg->rate_bits = 2;
            return STARKHIP_OK;
        case STARKHIP_AIR_MILLER_LOOP:  // :76
        case STARKHIP_AIR_FP12_MUL:     // :122
        case STARKHIP_AIR_TEST_FIBONACCI:
            return STARKHIP_OK;
        default:
            return STARKHIP_ERR_BAD_AIR;
    }
}

int starkhip_air_columns(starkhip_air_t air) { const AirInfo* a = air_get(air); return a ? (int)a->cols : STARKHIP_ERR_BAD_AIR; }
int starkhip_air_public_inputs(starkhip_air_t air) { const AirInfo* a = air_get(air); return a ? (int)a->pis : STARKHIP_ERR_BAD_AIR; }
int starkhip_air_constraint_degree(starkhip_air_t air) { const AirInfo* a = air_get(air); return a ? (int)a->degree : STARKHIP_ERR_BAD_AIR; }
int starkhip_air_num_constraints(starkhip_air_t air) { const AirInfo* a = air_get(air); return a ? (int)a->prog.n_constraints : STARKHIP_ERR_BAD_AIR; }
int starkhip_air_default_rows(starkhip_air_t air) { const AirInfo* a = air_get(air); return a ? (int)a->default_rows : STARKHIP_ERR_BAD_AIR; }
int starkhip_air_program(starkhip_air_t air, const uint64_t** blob, size_t* words) {
    const AirInfo* a = air_get(air);
    if (!a) return STARKHIP_ERR_BAD_AIR;
    *blob = a->blob.data();
    *words = a->blob.size();
    return STARKHIP_OK;
}

int starkhip_air_eval_frame(starkhip_air_t air, const uint64_t* local, const uint64_t* next, const uint64_t* public_inputs,
                            const uint64_t masks[8], const uint64_t* alphas, int n_alpha, uint64_t* acc_out) {
    const AirInfo* a = air_get(air);
    if (!a) return STARKHIP_ERR_BAD_AIR;
    if (!local || !next || !masks || !alphas || !acc_out || n_alpha < 1 || (a->pis && !public_inputs)) return STARKHIP_ERR_BAD_SHAPE;
    try {
        std::vector<gl2_t> l(a->cols), n(a->cols), al(n_alpha), acc(n_alpha);
        for (size_t i = 0; i < a->cols; i++) {
            l[i] = gl2_make(local[2 * i], local[2 * i + 1]);
            n[i] = gl2_make(next[2 * i], next[2 * i + 1]);
        }
        gl2_t mk[4];
        for (int i = 0; i < 4; i++) mk[i] = gl2_make(masks[2 * i], masks[2 * i + 1]);
        for (int j = 0; j < n_alpha; j++) al[j] = gl2_make(alphas[2 * j], alphas[2 * j + 1]);
        air_eval_folded<ExtOps>(a->prog, l.data(), n.data(), public_inputs, mk, al.data(), n_alpha, acc.data());
        for (int j = 0; j < n_alpha; j++) {
            acc_out[2 * j] = acc[j].a0;
            acc_out[2 * j + 1] = acc[j].a1;
        }
    } catch (const std::bad_alloc&) {
        return STARKHIP_ERR_OOM;
    }
    return STARKHIP_OK;
}

int starkhip_quotient_plan_check(starkhip_air_t air, unsigned want_chunks, uint64_t seed, uint64_t stats[8]) {
    const AirInfo* a = air_get(air);
    if (!a) return STARKHIP_ERR_BAD_AIR;
    try {
        QTPlan Q = build_quotient_plan(a->prog, want_chunks);
        uint64_t s = seed;
        auto rnd = [&]() {  // splitmix64, reduced
            s += 0x9E3779B97F4A7C15ULL;
            uint64_t z = s;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            return gl_from_u64((z ^ (z >> 31)) % GL_P);
        };
        std::vector<gl_t> local(a->cols), next(a->cols), pis(a->pis ? a->pis : 1);
        for (auto& v : local) v = rnd();
        for (auto& v : next) v = rnd();
        for (auto& v : pis) v = rnd();
        gl_t masks[4] = {1, rnd(), rnd(), rnd()}, alphas[2] = {rnd(), rnd()}, want[2], got[2];
        air_eval_folded<BaseOps>(a->prog, local.data(), next.data(), pis.data(), masks, alphas, 2, want);
        quotient_plan_weights_host(a->prog, Q, alphas, pis.data());
        const bool ok = quotient_plan_eval_host(Q, local.data(), next.data(), masks, got);
        if (stats) {
            stats[0] = Q.n_chunks; stats[1] = Q.n_supergroups; stats[2] = Q.n_pieces; stats[3] = Q.recs.size();
            stats[4] = Q.n_cell_records; stats[5] = Q.n_direct_loads; stats[6] = Q.tile_list.size(); stats[7] = Q.contribs.size();
            if (seed == 0xBA1A) { stats[6] = Q.cost_sum_max; stats[7] = Q.cost_sum_mean; }  // balance figures (tools)
            if (seed == 0xBA1B) { stats[4] = Q.n_piece_ends; stats[5] = Q.tile_phases; stats[6] = Q.rec_sum_max; stats[7] = Q.rec_sum_total; }
            if (seed == 0xBA1C) {  // the announcements (tools): fast pairs, direct pairs, and the direct second factors there are
                uint64_t fast = 0, dpairs = 0, direct_b = 0, direct_b_end = 0;
                for (const QTRec& r : Q.recs) {
                    if ((r.ctl & QT_SPECIAL) != 0 && !(r.ctl & QT_SRC_GLOBAL) && !(r.ctl & QT_STOP)) {
                        fast += r.aux & QT_AUX_PAIRS_MASK;
                        dpairs += (r.aux >> QT_AUX_DPAIRS_SHIFT) & QT_AUX_DPAIRS_MASK;
                    }
                    if ((r.ctl & (QT_SRC_GLOBAL | QT_MULV | QT_SETV)) == (QT_SRC_GLOBAL | QT_MULV) && !(r.aux & QT_AUX_SLOT)) (r.ctl & QT_END ? direct_b_end : direct_b)++;
                }
                stats[4] = fast; stats[5] = dpairs; stats[6] = direct_b; stats[7] = direct_b_end;
            }
        }
        if (!ok) return STARKHIP_ERR_BAD_SHAPE;
        return (want[0] == got[0] && want[1] == got[1]) ? STARKHIP_OK : STARKHIP_ERR_VERIFY;
    } catch (const std::bad_alloc&) {
        return STARKHIP_ERR_OOM;
    } catch (const std::exception&) {
        return STARKHIP_ERR_BAD_SHAPE;
    }
}

int starkhip_init(int device_ordinal, void** ctx) {
    Ctx* c = nullptr;
    int rc = ctx_create(device_ordinal, &c);
    *ctx = c;
    return rc;
}
void starkhip_shutdown(void* ctx) { ctx_destroy((Ctx*)ctx); }

int starkhip_prove(void* ctx, starkhip_air_t air, const starkhip_config_t* cfg, const uint64_t* trace, size_t n_rows, size_t n_cols,
                   int trace_layout, int trace_on_device, const uint64_t* public_inputs, size_t n_pis, uint64_t pow_witness, uint64_t** proof,
                   size_t* proof_words) {
    if (!ctx) return STARKHIP_ERR_NO_DEVICE;
    const AirInfo* a = air_get(air);
    if (!a) return STARKHIP_ERR_BAD_AIR;
    if (!cfg || !trace || !proof || !proof_words || (n_pis && !public_inputs)) return STARKHIP_ERR_BAD_SHAPE;
    if (trace_layout != 0 && trace_layout != 1) return STARKHIP_ERR_BAD_SHAPE;
    if (n_cols != a->cols) return STARKHIP_ERR_BAD_SHAPE;  // the caller's buffer is read as n_rows x columns words
    try {
        return prove((Ctx*)ctx, *a, *cfg, trace, n_rows, trace_layout, trace_on_device, public_inputs, n_pis, pow_witness, proof, proof_words);
    } catch (const std::bad_alloc&) {
        return STARKHIP_ERR_OOM;
    } catch (const std::exception&) {
        return STARKHIP_ERR_BAD_SHAPE;
    }
}

int starkhip_prove_columns(void* ctx, starkhip_air_t air, const starkhip_config_t* cfg, const uint64_t* const* columns, size_t n_rows, size_t n_cols,
                           const uint64_t* public_inputs, size_t n_pis, uint64_t pow_witness, uint64_t** proof, size_t* proof_words) {
    if (!ctx) return STARKHIP_ERR_NO_DEVICE;
    const AirInfo* a = air_get(air);
    if (!a) return STARKHIP_ERR_BAD_AIR;
    if (!cfg || !columns || !proof || !proof_words || (n_pis && !public_inputs)) return STARKHIP_ERR_BAD_SHAPE;
    if (n_cols != a->cols) return STARKHIP_ERR_BAD_SHAPE;  // the table is read as `columns` pointers of n_rows words each
    for (size_t i = 0; i < n_cols; i++)
        if (!columns[i]) return STARKHIP_ERR_BAD_SHAPE;
    try {
        return prove((Ctx*)ctx, *a, *cfg, (const uint64_t*)columns, n_rows, 3, 0, public_inputs, n_pis, pow_witness, proof, proof_words);
    } catch (const std::bad_alloc&) {
        return STARKHIP_ERR_OOM;
    } catch (const std::exception&) {
        return STARKHIP_ERR_BAD_SHAPE;
    }
}

int starkhip_set_option(void* ctx, const char* name, long value) {
    if (!ctx) return STARKHIP_ERR_NO_DEVICE;
    return ctx_set_option((Ctx*)ctx, name, value);
}

int starkhip_proof_layout(const uint64_t* proof, size_t proof_words, starkhip_proof_layout_t* out) {
    if (!proof || !out) return STARKHIP_ERR_BAD_SHAPE;
    ProofLayout pl;
    if (!pl.read_header(proof, proof_words)) return STARKHIP_ERR_BAD_SHAPE;
    memset(out, 0, sizeof *out);
    out->n_columns = pl.C; out->n_quotient_polys = pl.Q; out->degree_bits = pl.log_n; out->rate_bits = pl.rate_bits;
    out->cap_height = pl.cap_h; out->n_fri_layers = pl.L; out->n_query_rounds = pl.n_queries; out->final_poly_len = pl.final_len;
    out->n_public_inputs = pl.n_pis; out->arity_bits = pl.arity_bits;
    out->off_trace_cap = pl.off_trace_cap; out->off_quotient_cap = pl.off_quot_cap; out->off_local_values = pl.off_local;
    out->off_next_values = pl.off_next; out->off_quotient_openings = pl.off_quot_open; out->off_fri_caps = pl.off_fri_caps;
    out->off_query_rounds = pl.off_queries; out->query_round_words = pl.query_words; out->off_final_poly = pl.off_final;
    out->off_pow_witness = pl.off_pow; out->off_public_inputs = pl.off_pis; out->total_words = pl.total;
    const size_t d0 = pl.log_N - pl.cap_h;
    size_t o = 0;
    out->q_trace_leaf = o; o += pl.C;
    out->q_trace_siblings = o; o += 4 * d0;
    out->q_quotient_leaf = o; o += pl.Q;
    out->q_quotient_siblings = o; o += 4 * d0;
    for (size_t l = 0; l < pl.L && l < 16; l++) {
        out->q_step_evals[l] = o; o += 2 * ((size_t)1 << pl.arity_bits);
        out->q_step_siblings[l] = o; o += 4 * pl.layer_depth[l];
        out->step_sibling_count[l] = pl.layer_depth[l];
    }
    out->initial_sibling_count = d0;
    return STARKHIP_OK;
}

// ---- compact traces (trace_log.h)
int starkhip_trace_set_threads(int n) {
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    return g_trace_threads.exchange(n);
}
int starkhip_trace_log_begin(void** log) {
    if (!log) return STARKHIP_ERR_BAD_SHAPE;
    *log = nullptr;
    if (armed_trace_log()) return STARKHIP_ERR_BAD_SHAPE;  // already recording on this thread
    TraceLog* l = new (std::nothrow) TraceLog();
    if (!l) return STARKHIP_ERR_OOM;
    armed_trace_log() = l;
    *log = l;
    return STARKHIP_OK;
}
int starkhip_trace_log_end(void* log) {
    if (!log || armed_trace_log() != (TraceLog*)log) return STARKHIP_ERR_BAD_SHAPE;
    armed_trace_log() = nullptr;
    TraceLog* l = (TraceLog*)log;
    l->open.clear();
    l->open.shrink_to_fit();
    return l->rows ? STARKHIP_OK : STARKHIP_ERR_BAD_SHAPE;  // no generator ran in between
}
void starkhip_trace_log_free(void* log) {
    if (log && armed_trace_log() == (TraceLog*)log) armed_trace_log() = nullptr;
    delete (TraceLog*)log;
}
int starkhip_trace_log_info(const void* log, size_t* n_rows, size_t* n_cols, size_t* n_records, size_t* n_words) {
    if (!log) return STARKHIP_ERR_BAD_SHAPE;
    const TraceLog* l = (const TraceLog*)log;
    if (n_rows) *n_rows = l->rows;
    if (n_cols) *n_cols = l->cols;
    if (n_records) *n_records = l->total_records();
    if (n_words) *n_words = l->total_words();
    return STARKHIP_OK;
}
int starkhip_trace_log_from_writes(size_t n_rows, size_t n_cols, const uint64_t* writes, size_t n_writes, void** log) {
    if (!log || (n_writes && !writes) || !n_rows || !n_cols) return STARKHIP_ERR_BAD_SHAPE;
    *log = nullptr;
    TraceLog* l = new (std::nothrow) TraceLog();
    if (!l) return STARKHIP_ERR_OOM;
    try {
        l->reset(n_rows, n_cols);
        for (size_t i = 0; i < n_writes; i++) l->set((size_t)writes[3 * i], (size_t)writes[3 * i + 1], writes[3 * i + 2]);
    } catch (const std::bad_alloc&) {
        delete l;
        return STARKHIP_ERR_OOM;
    } catch (const std::exception&) {  // a write outside the trace, a value of more than 32 bits
        delete l;
        return STARKHIP_ERR_BAD_SHAPE;
    }
    l->open.clear();
    l->open.shrink_to_fit();
    *log = l;
    return STARKHIP_OK;
}
int starkhip_trace_log_overwrites(const void* log, size_t* empty_runs, size_t* late_zeros) {
    if (!log) return STARKHIP_ERR_BAD_SHAPE;
    const TraceLog* l = (const TraceLog*)log;
    size_t empty = 0;
    l->for_each_part([&](const TraceLog& part) {
        for (uint32_t off : part.offsets) empty += part.words[off - part.base + 2] == 0;
    });
    if (empty_runs) *empty_runs = empty;
    if (late_zeros) *late_zeros = l->total_late_zeros() / 2;
    return STARKHIP_OK;
}
int starkhip_trace_log_expand_host(const void* log, uint64_t* trace_rowmajor, size_t* conflicts) {
    if (!log || !trace_rowmajor) return STARKHIP_ERR_BAD_SHAPE;
    const TraceLog* l = (const TraceLog*)log;
    memset(trace_rowmajor, 0, l->rows * l->cols * sizeof(uint64_t));
    size_t bad = 0;
    l->for_each_part([&](const TraceLog& part) {
        for (uint32_t off : part.offsets) {
            const uint32_t* r = &part.words[off - part.base];
            for (uint32_t k = 0; k < r[2]; k++)
                for (uint32_t i = 0; i < r[3]; i++) {
                    uint64_t& cell = trace_rowmajor[(size_t)(r[1] + k) * l->cols + r[0] + i];
                    if (cell && cell != r[4 + i]) bad++;  // two records disagree: parallel expansion would be order dependent
                    cell = r[4 + i];
                }
        }
    });
    l->for_each_part([&](const TraceLog& part) {
        for (size_t i = 0; i + 1 < part.late_zeros.size(); i += 2)
            trace_rowmajor[(size_t)part.late_zeros[i + 1] * l->cols + part.late_zeros[i]] = 0;
    });
    if (conflicts) *conflicts = bad;
    return STARKHIP_OK;
}
int starkhip_prove_compact(void* ctx, starkhip_air_t air, const starkhip_config_t* cfg, const void* log, const uint64_t* public_inputs,
                           size_t n_pis, uint64_t pow_witness, uint64_t** proof, size_t* proof_words) {
    if (!ctx) return STARKHIP_ERR_NO_DEVICE;
    const AirInfo* a = air_get(air);
    if (!a) return STARKHIP_ERR_BAD_AIR;
    const TraceLog* l = (const TraceLog*)log;
    if (!l || armed_trace_log() == l || l->cols != a->cols || !l->rows) return STARKHIP_ERR_BAD_SHAPE;
    if (!cfg || !proof || !proof_words || (n_pis && !public_inputs)) return STARKHIP_ERR_BAD_SHAPE;
    try {
        return prove((Ctx*)ctx, *a, *cfg, (const uint64_t*)l, l->rows, /*layout: compact log*/ 2, 0, public_inputs, n_pis, pow_witness, proof,
                     proof_words);
    } catch (const std::bad_alloc&) {
        return STARKHIP_ERR_OOM;
    } catch (const std::exception&) {
        return STARKHIP_ERR_BAD_SHAPE;
    }
}

// ---- proof pool (scheduler.cpp)
// One hardware queue per in-flight proof: with HIP's default of 4, streams share queues and kernels of different proofs serialise behind
// each other (-6 % on the FinalExp pool).  The runtime reads GPU_MAX_HW_QUEUES ONCE, when the process first uses HIP, so setting it here
// works only in a process that has not.  Whether it has is visible without touching HIP: initialising the runtime opens the compute
// driver's device node, /dev/kfd.  0 = the variable was in the environment before the runtime came up (ours or the caller's own);
// 1 = the runtime was already up when the first pool set it: the pools run on HIP's default, and the caller should export the variable
// itself, earlier (bench.py does).
static std::atomic<int> g_hw_queues_late(-1);
static bool hip_runtime_is_up() {
    DIR* d = opendir("/proc/self/fd");  // every descriptor, whatever its number (a server may hold thousands)
    if (!d) return false;
    bool up = false;
    char link[300], target[256];
    while (const dirent* e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        snprintf(link, sizeof link, "/proc/self/fd/%s", e->d_name);
        const ssize_t n = readlink(link, target, sizeof target - 1);
        if (n <= 0) continue;
        target[n] = 0;
        if (strcmp(target, "/dev/kfd") == 0) {
            up = true;
            break;
        }
    }
    closedir(d);
    return up;
}
static void ask_for_hw_queues() {
    static std::once_flag once;  // decided with the first pool, by one thread (setenv beside another thread's getenv is a race)
    std::call_once(once, [] {
        int late = 0;
        if (!getenv("GPU_MAX_HW_QUEUES")) {
            late = hip_runtime_is_up() ? 1 : 0;
            setenv("GPU_MAX_HW_QUEUES", "16", 0);
            if (late)
                fprintf(stderr, "starkhip: the HIP runtime was initialised before the first pool could set GPU_MAX_HW_QUEUES=16; pools run on HIP's "
                                "default of 4 hardware queues (about 6 %% slower). Export GPU_MAX_HW_QUEUES=16 before the process first uses HIP.\n");
        }
        g_hw_queues_late.store(late);
    });
}
int starkhip_hw_queues_status(void) { return std::max(0, g_hw_queues_late.load()); }

int starkhip_pool_create(const starkhip_pool_config_t* cfg, void** pool) {
    if (!cfg || !pool) return STARKHIP_ERR_BAD_SHAPE;
    *pool = nullptr;
    ask_for_hw_queues();
    Pool* p = nullptr;
    try {
        const int rc = pool_create(*cfg, &p);
        *pool = p;
        return rc;
    } catch (const std::bad_alloc&) {
        return STARKHIP_ERR_OOM;
    } catch (const std::exception&) {
        return STARKHIP_ERR_HIP;
    }
}
void starkhip_pool_destroy(void* pool) { pool_destroy((Pool*)pool); }
int starkhip_pool_submit(void* pool, starkhip_air_t air, const starkhip_config_t* cfg, const uint64_t* trace, size_t n_rows, size_t n_cols,
                         int trace_layout, int trace_on_device, const uint64_t* public_inputs, size_t n_pis, uint64_t pow_witness,
                         uint64_t* ticket) {
    if (!pool) return STARKHIP_ERR_NO_DEVICE;
    return pool_submit((Pool*)pool, air, cfg, trace, n_rows, n_cols, trace_layout, trace_on_device, public_inputs, n_pis, pow_witness, ticket);
}
int starkhip_pool_submit_columns(void* pool, starkhip_air_t air, const starkhip_config_t* cfg, const uint64_t* const* columns, size_t n_rows,
                                 size_t n_cols, const uint64_t* public_inputs, size_t n_pis, uint64_t pow_witness, uint64_t* ticket) {
    if (!pool) return STARKHIP_ERR_NO_DEVICE;
    return pool_submit_columns((Pool*)pool, air, cfg, columns, n_rows, n_cols, public_inputs, n_pis, pow_witness, ticket);
}
int starkhip_pool_submit_compact(void* pool, starkhip_air_t air, const starkhip_config_t* cfg, const void* log, const uint64_t* public_inputs,
                                 size_t n_pis, uint64_t pow_witness, uint64_t* ticket) {
    if (!pool) return STARKHIP_ERR_NO_DEVICE;
    if (log && armed_trace_log() == (const TraceLog*)log) return STARKHIP_ERR_BAD_SHAPE;
    return pool_submit_compact((Pool*)pool, air, cfg, log, public_inputs, n_pis, pow_witness, ticket);
}
int starkhip_pool_submit_witness(void* pool, starkhip_air_t air, const starkhip_config_t* cfg, const uint32_t* operands, size_t n_limbs,
                                 uint64_t pow_witness, uint64_t* ticket) {
    if (!pool) return STARKHIP_ERR_NO_DEVICE;
    try {
        return pool_submit_witness((Pool*)pool, air, cfg, operands, n_limbs, pow_witness, ticket);
    } catch (const std::bad_alloc&) {
        return STARKHIP_ERR_OOM;
    }
}
int starkhip_pool_wait(void* pool, uint64_t ticket, uint64_t** proof, size_t* proof_words, starkhip_ticket_info_t* info) {
    if (!pool) return STARKHIP_ERR_NO_DEVICE;
    return pool_wait((Pool*)pool, ticket, proof, proof_words, info);
}
int starkhip_pool_reservation(void* pool, starkhip_pool_reservation_t* out) {
    if (!pool || !out) return STARKHIP_ERR_BAD_SHAPE;
    return pool_reservation((Pool*)pool, out);
}
int starkhip_pool_host_info(void* pool, starkhip_pool_host_info_t* out) {
    if (!pool || !out) return STARKHIP_ERR_BAD_SHAPE;
    return pool_host_info((Pool*)pool, out);
}
unsigned starkhip_cpu_budget(void) { return cpu_budget(); }
void starkhip_host_cpu_seconds(double out[3]) {
    if (out) host_cpu_seconds(out);
}
int starkhip_pool_stats(void* pool, starkhip_pool_stats_t* out) {
    if (!pool || !out) return STARKHIP_ERR_BAD_SHAPE;
    return pool_stats((Pool*)pool, out);
}

// ---- a pool per device behind one handle (scheduler.cpp)
int starkhip_multipool_create(const int* devices, size_t n_devices, const starkhip_pool_config_t* cfg, void** mpool) {
    if (!devices || !n_devices || !cfg || !mpool) return STARKHIP_ERR_BAD_SHAPE;
    *mpool = nullptr;
    ask_for_hw_queues();
    MultiPool* mp = nullptr;
    try {
        const int rc = multipool_create(devices, n_devices, *cfg, &mp);
        *mpool = mp;
        return rc;
    } catch (const std::bad_alloc&) {
        return STARKHIP_ERR_OOM;
    } catch (const std::exception&) {
        return STARKHIP_ERR_HIP;
    }
}
void starkhip_multipool_destroy(void* mpool) { multipool_destroy((MultiPool*)mpool); }
size_t starkhip_multipool_size(const void* mpool) { return mpool ? multipool_size((const MultiPool*)mpool) : 0; }
void* starkhip_multipool_pool(void* mpool, size_t slot) { return mpool ? (void*)multipool_pool((MultiPool*)mpool, slot) : nullptr; }
int starkhip_multipool_device(const void* mpool, size_t slot) { return mpool ? multipool_device((const MultiPool*)mpool, slot) : -1; }
int starkhip_multipool_submit(void* mpool, int slot, starkhip_air_t air, const starkhip_config_t* cfg, const uint64_t* trace, size_t n_rows,
                              size_t n_cols, int trace_layout, int trace_on_device, const uint64_t* public_inputs, size_t n_pis,
                              uint64_t pow_witness, uint64_t* ticket) {
    if (!mpool) return STARKHIP_ERR_NO_DEVICE;
    return multipool_submit((MultiPool*)mpool, slot, air, cfg, trace, n_rows, n_cols, trace_layout, trace_on_device, public_inputs, n_pis, pow_witness,
                            ticket);
}
int starkhip_multipool_submit_columns(void* mpool, int slot, starkhip_air_t air, const starkhip_config_t* cfg, const uint64_t* const* columns,
                                      size_t n_rows, size_t n_cols, const uint64_t* public_inputs, size_t n_pis, uint64_t pow_witness,
                                      uint64_t* ticket) {
    if (!mpool) return STARKHIP_ERR_NO_DEVICE;
    return multipool_submit_columns((MultiPool*)mpool, slot, air, cfg, columns, n_rows, n_cols, public_inputs, n_pis, pow_witness, ticket);
}
int starkhip_multipool_submit_compact(void* mpool, int slot, starkhip_air_t air, const starkhip_config_t* cfg, const void* log,
                                      const uint64_t* public_inputs, size_t n_pis, uint64_t pow_witness, uint64_t* ticket) {
    if (!mpool) return STARKHIP_ERR_NO_DEVICE;
    if (log && armed_trace_log() == (const TraceLog*)log) return STARKHIP_ERR_BAD_SHAPE;
    return multipool_submit_compact((MultiPool*)mpool, slot, air, cfg, log, public_inputs, n_pis, pow_witness, ticket);
}
int starkhip_multipool_submit_witness(void* mpool, int slot, starkhip_air_t air, const starkhip_config_t* cfg, const uint32_t* operands,
                                      size_t n_limbs, uint64_t pow_witness, uint64_t* ticket) {
    if (!mpool) return STARKHIP_ERR_NO_DEVICE;
    try {
        return multipool_submit_witness((MultiPool*)mpool, slot, air, cfg, operands, n_limbs, pow_witness, ticket);
    } catch (const std::bad_alloc&) {
        return STARKHIP_ERR_OOM;
    }
}
int starkhip_multipool_submit_witness_batch(void* mpool, size_t n_jobs, const starkhip_air_t* airs, const uint32_t* const* operands,
                                            const size_t* n_limbs, uint64_t pow_witness, uint64_t* tickets, int* rcs) {
    if (!mpool) return STARKHIP_ERR_NO_DEVICE;
    static_assert(sizeof(starkhip_air_t) == sizeof(int), "air ids travel as int");
    try {
        return multipool_submit_witness_batch((MultiPool*)mpool, n_jobs, (const int*)airs, operands, n_limbs, pow_witness, tickets, rcs);
    } catch (const std::bad_alloc&) {
        return STARKHIP_ERR_OOM;
    }
}
int starkhip_multipool_ticket_slot(const void* mpool, uint64_t ticket) { return mpool ? multipool_ticket_slot((const MultiPool*)mpool, ticket) : -1; }
int starkhip_multipool_wait(void* mpool, uint64_t ticket, uint64_t** proof, size_t* proof_words, starkhip_ticket_info_t* info) {
    if (!mpool) return STARKHIP_ERR_NO_DEVICE;
    return multipool_wait((MultiPool*)mpool, ticket, proof, proof_words, info);
}
int starkhip_plan_lpt(size_t n_jobs, const starkhip_air_t* airs, size_t n_pools, int* slots) {
    if ((n_jobs && (!airs || !slots)) || !n_pools) return STARKHIP_ERR_BAD_SHAPE;
    plan_lpt(n_jobs, (const int*)airs, n_pools, slots);
    return STARKHIP_OK;
}
double starkhip_air_cost(starkhip_air_t air) { return air_cost((int)air); }

int starkhip_last_timings(void* ctx, float ms[STARKHIP_N_PHASES]) {
    if (!ctx) return STARKHIP_ERR_NO_DEVICE;
    memcpy(ms, ctx_timings((Ctx*)ctx), sizeof(float) * STARKHIP_N_PHASES);
    return STARKHIP_OK;
}

int starkhip_last_kernel_timings(void* ctx, float ms[3]) {
    if (!ctx) return STARKHIP_ERR_NO_DEVICE;
    memcpy(ms, ctx_kernel_timings((Ctx*)ctx), sizeof(float) * 3);
    return STARKHIP_OK;
}

int starkhip_last_host_timings(void* ctx, float ms[2]) {
    if (!ctx) return STARKHIP_ERR_NO_DEVICE;
    memcpy(ms, ctx_host_timings((Ctx*)ctx), sizeof(float) * 2);
    return STARKHIP_OK;
}

int starkhip_host_alloc(void* ctx, size_t bytes, void** out) {
    if (!out) return STARKHIP_ERR_BAD_SHAPE;
    *out = nullptr;
    if (!ctx) return STARKHIP_ERR_NO_DEVICE;
    return host_alloc((Ctx*)ctx, bytes, out);
}
void starkhip_host_free(void* p) { host_free(p); }

int starkhip_lde_batch(void* ctx, const uint64_t* values, size_t n_cols, unsigned log_n, unsigned rate_bits, uint64_t* coeffs_out,
                       uint64_t* lde_out) {
    if (!ctx) return STARKHIP_ERR_NO_DEVICE;
    return lde_batch((Ctx*)ctx, values, n_cols, log_n, rate_bits, coeffs_out, lde_out);
}
int starkhip_merkle_cap(void* ctx, const uint64_t* lde_colmajor, size_t n_cols, unsigned log_N, unsigned cap_height, uint64_t* cap_out) {
    if (!ctx) return STARKHIP_ERR_NO_DEVICE;
    return merkle_cap((Ctx*)ctx, lde_colmajor, n_cols, log_N, cap_height, cap_out);
}
int starkhip_lde_bench(void* ctx, size_t n_cols, unsigned log_n, unsigned rate_bits, unsigned reps, unsigned const_per_64, const uint64_t* device_values, float* ms_per_launch, float* each_ms) {
    if (!ctx) return STARKHIP_ERR_NO_DEVICE;
    if (!ms_per_launch) return STARKHIP_ERR_BAD_SHAPE;
    return lde_bench((Ctx*)ctx, n_cols, log_n, rate_bits, reps, const_per_64, device_values, ms_per_launch, each_ms);
}
int starkhip_trace_log_expand_device(void* ctx, const void* log, uint64_t* trace_colmajor) {
    if (!ctx) return STARKHIP_ERR_NO_DEVICE;
    if (!log || !trace_colmajor || armed_trace_log() == (const TraceLog*)log) return STARKHIP_ERR_BAD_SHAPE;
    try {
        return expand_log((Ctx*)ctx, (const TraceLog*)log, trace_colmajor);
    } catch (const std::bad_alloc&) {
        return STARKHIP_ERR_OOM;
    }
}
int starkhip_poseidon_permute_batch(void* ctx, uint64_t* states, size_t n_states) {
    if (!ctx) return STARKHIP_ERR_NO_DEVICE;
    return permute_batch((Ctx*)ctx, states, n_states);
}
int starkhip_field_ops_batch(void* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    if (!ctx) return STARKHIP_ERR_NO_DEVICE;
    return field_ops((Ctx*)ctx, op, a, b, out, n);
}
int starkhip_selfcheck_hash_tables(unsigned n_states) {
    const int fours = merged_fours_selfcheck(n_states);   // the lane and pair forms' tables (poseidon_host.cpp)
    if (fours < 0) return -1;
    return quad_merged_tables_selfcheck(n_states) + fours;
}
size_t starkhip_lde_launch_ranges(size_t n_cols, unsigned rate_bits, uint64_t* triples, size_t cap) {
    const std::vector<LdeLaunch> plan = lde_launch_plan(n_cols, rate_bits);
    for (size_t i = 0; i < plan.size() && i < cap; i++) {
        triples[3 * i] = plan[i].a;
        triples[3 * i + 1] = plan[i].b;
        triples[3 * i + 2] = plan[i].from_copy ? 1 : 0;
    }
    return plan.size();
}
void starkhip_poseidon_permute_host(uint64_t state[12]) { poseidon_permute_host(state); }
/* n chained permutations with the challenger's host permutation (which = 0) or the portable reference loop (which = 1);
 * lets the tests compare the two and the benchmark report the host hashing rate */
void starkhip_poseidon_permute_host_many(uint64_t state[12], size_t n, int which) {
    for (size_t i = 0; i < n; i++) {
        if (which == 0) poseidon_permute_host(state);
        else poseidon_permute(state);
    }
}

int starkhip_verify(starkhip_air_t air, const starkhip_config_t* cfg, const uint64_t* proof, size_t proof_words) {
    const AirInfo* a = air_get(air);
    if (!a) return STARKHIP_ERR_BAD_AIR;
    return verify_proof(*a, *cfg, proof, proof_words);
}

void starkhip_proof_blob_stats(uint64_t out[5]) {
    const starkhip::BlobArenaStats st = starkhip::blob_arena_stats();
    out[0] = st.blobs; out[1] = st.busy; out[2] = st.bytes; out[3] = st.taken; out[4] = st.missed;
}
void starkhip_free(void* p) { starkhip::blob_free(p); }  // a proof blob may be a recycled page-locked one (blob_arena.h)

const char* starkhip_error_string(int code) {
    switch (code) {
        case STARKHIP_OK: return "ok";
        case STARKHIP_ERR_QUOTIENT_NOT_DIVISIBLE: return "Quotient has failed, the vanishing polynomial is not divisible by Z_H";
        case STARKHIP_ERR_ZETA_IN_SUBGROUP: return "Opening point is in the subgroup";
        case STARKHIP_ERR_BAD_SHAPE: return "bad shape (columns / public inputs / rows / config do not match the AIR)";
        case STARKHIP_ERR_HIP: return "HIP runtime error";
        case STARKHIP_ERR_OOM: return "out of device memory";
        case STARKHIP_ERR_NO_DEVICE: return "no such HIP device / context";
        case STARKHIP_ERR_VERIFY: return "proof rejected";
        case STARKHIP_ERR_BAD_AIR: return "unknown or unbuildable AIR id";
        default: return "unknown error";
    }
}

}  // extern "C"
