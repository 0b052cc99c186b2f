// extern "C" surface of libstarkhip.so (declared in include/starkhip.h).
#include <stdlib.h>
#include <string.h>

#include <new>

#include "air_eval.h"
#include "airs.h"
#include "kernels.h"
#include "trace_log.h"
#include "poseidon.h"
#include "prover.h"

using namespace starkhip;

namespace starkhip {
TraceLog*& armed_trace_log() {
    static thread_local TraceLog* armed = nullptr;
    return armed;
}
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

int starkhip_init(int device_ordinal, void** ctx) {
    Ctx* c = nullptr;
    int rc = ctx_create(device_ordinal, &c);
    *ctx = c;
    return rc;
}
void starkhip_shutdown(void* ctx) { ctx_destroy((Ctx*)ctx); }

int starkhip_prove(void* ctx, starkhip_air_t air, const starkhip_config_t* cfg, const uint64_t* trace, size_t n_rows, int trace_layout,
                   int trace_on_device, const uint64_t* public_inputs, size_t n_pis, uint64_t pow_witness, uint64_t** proof,
                   size_t* proof_words) {
    if (!ctx) return STARKHIP_ERR_NO_DEVICE;
    const AirInfo* a = air_get(air);
    if (!a) return STARKHIP_ERR_BAD_AIR;
    if (trace_layout != 0 && trace_layout != 1) return STARKHIP_ERR_BAD_SHAPE;
    return prove((Ctx*)ctx, *a, *cfg, trace, n_rows, trace_layout, trace_on_device, public_inputs, n_pis, pow_witness, proof, proof_words);
}

// ---- compact traces (trace_log.h)
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
    if (n_records) *n_records = l->offsets.size();
    if (n_words) *n_words = l->words.size();
    return STARKHIP_OK;
}
int starkhip_trace_log_expand_host(const void* log, uint64_t* trace_rowmajor, size_t* conflicts) {
    if (!log || !trace_rowmajor) return STARKHIP_ERR_BAD_SHAPE;
    const TraceLog* l = (const TraceLog*)log;
    memset(trace_rowmajor, 0, l->rows * l->cols * sizeof(uint64_t));
    size_t bad = 0;
    for (uint32_t off : l->offsets) {
        const uint32_t* r = &l->words[off];
        for (uint32_t k = 0; k < r[2]; k++)
            for (uint32_t i = 0; i < r[3]; i++) {
                uint64_t& cell = trace_rowmajor[(size_t)(r[1] + k) * l->cols + r[0] + i];
                if (cell && cell != r[4 + i]) bad++;  // two records disagree: parallel expansion would be order dependent
                cell = r[4 + i];
            }
    }
    for (size_t i = 0; i + 1 < l->late_zeros.size(); i += 2) trace_rowmajor[(size_t)l->late_zeros[i + 1] * l->cols + l->late_zeros[i]] = 0;
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
    return prove((Ctx*)ctx, *a, *cfg, (const uint64_t*)l, l->rows, /*layout: compact log*/ 2, 0, public_inputs, n_pis, pow_witness, proof, proof_words);
}

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
int starkhip_poseidon_permute_batch(void* ctx, uint64_t* states, size_t n_states) {
    if (!ctx) return STARKHIP_ERR_NO_DEVICE;
    return permute_batch((Ctx*)ctx, states, n_states);
}
int starkhip_field_ops_batch(void* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    if (!ctx) return STARKHIP_ERR_NO_DEVICE;
    return field_ops((Ctx*)ctx, op, a, b, out, n);
}
int starkhip_selfcheck_hash_tables(unsigned n_states) { return quad_merged_tables_selfcheck(n_states); }
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

void starkhip_free(void* p) { free(p); }

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
