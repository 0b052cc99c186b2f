// Internal interface between the C ABI (capi.cpp) and the GPU prover driver (prover.hip).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "airs.h"

namespace starkhip {

struct Ctx;
int ctx_create(int device, Ctx** out, int priority = 0);
void ctx_destroy(Ctx* c);
const float* ctx_timings(Ctx* c);
const float* ctx_kernel_timings(Ctx* c);
const float* ctx_host_timings(Ctx* c);
void ctx_commit_info(Ctx* c, int* form, unsigned* group);  // how the last proof's trace commitment went out (HashService::Timing)
int ctx_set_option(Ctx* c, const char* name, long value);
class HashService;  // scheduler.h
void ctx_attach_hash_service(Ctx* c, HashService* hs);  // trace commitments of this context go through the pool's scheduler
bool ctx_has_hash_service(Ctx* c);
int ctx_set_urgent(Ctx* c, bool urgent);
// tables, plan, work buffers, upload staging and `proof_blobs` page-locked proof blobs (blob_arena.h) for proofs of `air`,
// allocated now (a pool's warm-up)
int ctx_reserve(Ctx* c, const AirInfo& air, const starkhip_config_t& cfg, size_t log_bytes, unsigned proof_blobs = 0, bool device_traces = false);
void ctx_hash_request_reset(Ctx* c);
bool ctx_hash_requested(Ctx* c);  // the current / last prove() reached its trace commitment

struct Pool;
int pool_create(const starkhip_pool_config_t& cfg, Pool** out, unsigned cpu_share = 1);  // cpu_share: pools that split this process's CPUs
void pool_destroy(Pool* p);
int pool_submit(Pool* p, int air, const starkhip_config_t* cfg, const uint64_t* trace, size_t n_rows, size_t n_cols, int layout, int on_device,
                const uint64_t* pis, size_t n_pis, uint64_t pow, uint64_t* ticket);
int pool_submit_columns(Pool* p, int air, const starkhip_config_t* cfg, const uint64_t* const* columns, size_t n_rows, size_t n_cols,
                        const uint64_t* pis, size_t n_pis, uint64_t pow, uint64_t* ticket);
int pool_submit_compact(Pool* p, int air, const starkhip_config_t* cfg, const void* log, const uint64_t* pis, size_t n_pis, uint64_t pow,
                        uint64_t* ticket);
int pool_submit_witness(Pool* p, int air, const starkhip_config_t* cfg, const uint32_t* operands, size_t n_limbs, uint64_t pow, uint64_t* ticket);
int pool_wait(Pool* p, uint64_t ticket, uint64_t** proof, size_t* words, starkhip_ticket_info_t* info);
int pool_stats(Pool* p, starkhip_pool_stats_t* out);
int pool_reservation(Pool* p, starkhip_pool_reservation_t* out);
int pool_host_info(Pool* p, starkhip_pool_host_info_t* out);
unsigned cpu_budget();  // scheduler.cpp: CPUs this process may really use
void host_cpu_seconds(double out[3]);
// a pool per device behind one handle (scheduler.cpp): placement by outstanding cost, longest job first
struct MultiPool;
double air_cost(int air);
int multipool_create(const int* devices, size_t n, const starkhip_pool_config_t& cfg, MultiPool** out);
void multipool_destroy(MultiPool* mp);
size_t multipool_size(const MultiPool* mp);
Pool* multipool_pool(MultiPool* mp, size_t slot);
int multipool_device(const MultiPool* mp, size_t slot);
int multipool_submit(MultiPool* mp, int slot, int air, const starkhip_config_t* cfg, const uint64_t* trace, size_t n_rows, size_t n_cols, int layout,
                     int on_device, const uint64_t* pis, size_t n_pis, uint64_t pow, uint64_t* ticket);
int multipool_submit_columns(MultiPool* mp, int slot, int air, const starkhip_config_t* cfg, const uint64_t* const* columns, size_t n_rows, size_t n_cols,
                             const uint64_t* pis, size_t n_pis, uint64_t pow, uint64_t* ticket);
int multipool_submit_compact(MultiPool* mp, int slot, int air, const starkhip_config_t* cfg, const void* log, const uint64_t* pis, size_t n_pis,
                             uint64_t pow, uint64_t* ticket);
int multipool_submit_witness(MultiPool* mp, int slot, int air, const starkhip_config_t* cfg, const uint32_t* operands, size_t n_limbs, uint64_t pow,
                             uint64_t* ticket);
int multipool_submit_witness_batch(MultiPool* mp, size_t n, const int* airs, const uint32_t* const* operands, const size_t* n_limbs, uint64_t pow,
                                   uint64_t* tickets, int* rcs);
int multipool_ticket_slot(const MultiPool* mp, uint64_t ticket);
int multipool_wait(MultiPool* mp, uint64_t ticket, uint64_t** proof, size_t* words, starkhip_ticket_info_t* info);
void plan_lpt(size_t n, const int* airs, size_t n_pools, int* slots);
size_t ctx_device_bytes(Ctx* c);  // device memory this context holds (work buffers, tables, plans)
size_t ctx_pinned_bytes(Ctx* c);  // page-locked host memory it holds (upload staging; proof blobs are counted by starkhip_proof_blob_stats)

int prove(Ctx* c, const AirInfo& air, const starkhip_config_t& cfg, const uint64_t* trace, size_t n_rows, int layout, int on_device,
          const uint64_t* pis, size_t n_pis, uint64_t pow_witness, uint64_t** proof_out, size_t* proof_words);
int lde_batch(Ctx* c, const uint64_t* values, size_t n_cols, unsigned log_n, unsigned rate_bits, uint64_t* coeffs_out, uint64_t* lde_out);
int merkle_cap(Ctx* c, const uint64_t* lde_natural, size_t n_cols, unsigned log_N, unsigned cap_h, uint64_t* cap_out);
int permute_batch(Ctx* c, uint64_t* states, size_t n);
struct TraceLog;
int expand_log(Ctx* c, const TraceLog* log, uint64_t* out_colmajor);
int lde_bench(Ctx* c, size_t n_cols, unsigned log_n, unsigned rate_bits, unsigned reps, unsigned const_per_64, const uint64_t* device_values, float* ms_out, float* each_ms = nullptr);
int field_ops(Ctx* c, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n);
int host_alloc(Ctx* c, size_t bytes, void** out);
void host_free(void* p);

int verify_proof(const AirInfo& air, const starkhip_config_t& cfg, const uint64_t* proof, size_t words);

}  // namespace starkhip
