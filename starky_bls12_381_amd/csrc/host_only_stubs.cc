// Stand-ins for the GPU entry points, linked ONLY into the sanitizer builds of the host code (make asan / tsan-test).  Not part of
// libstarkhip.so.  By default every device path answers "no device", as the real library does on a box without a GPU.
// With STARKHIP_FAKE_DEVICE=1 in the environment the stand-ins PRETEND instead: contexts can be created, prove() sleeps a few
// milliseconds, asks the pool's commitment scheduler for its "commitment" (so HashService runs its real gather / merge / launch
// logic against no-op HIP calls) and returns a blob that ends in the public inputs.  That lets ThreadSanitizer and
// AddressSanitizer run the proof pool's threads -- generators, context workers, the commitment scheduler, shutdown -- on a CPU
// (tests/tsan_pool_main.cpp); it proves nothing about proofs.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <set>
#include <thread>

#include "blob_arena.h"
#include "kernels.h"
#include "prover.h"
#include "scheduler.h"
#include "trace_log.h"

namespace starkhip {
std::atomic<uint64_t> g_wait_cpu_ns(0);

static bool fake_device() {
    const char* e = getenv("STARKHIP_FAKE_DEVICE");
    return e && *e == '1';
}

struct Ctx {
    int device = 0;
    HashService* hs = nullptr;
    bool hash_requested = false, urgent = false;
    float timings[STARKHIP_N_PHASES] = {0}, ktimings[3] = {0}, htimings[2] = {0};
    std::set<int> blob_airs;
};
int ctx_create(int device, Ctx** out, int) {
    if (!fake_device() || device < 0 || device >= 8) {  // the pretended node has eight devices
        *out = nullptr;
        return STARKHIP_ERR_NO_DEVICE;
    }
    *out = new Ctx();
    (*out)->device = device;
    return STARKHIP_OK;
}
void ctx_destroy(Ctx* c) {
    if (c) blob_arena_drop(c);
    delete c;
}
size_t ctx_device_bytes(Ctx*) { return 0; }
size_t ctx_pinned_bytes(Ctx*) { return 0; }
const float* ctx_timings(Ctx* c) { return c->timings; }
const float* ctx_kernel_timings(Ctx* c) { return c->ktimings; }
const float* ctx_host_timings(Ctx* c) { return c->htimings; }
void ctx_commit_info(Ctx*, int* form, unsigned* group) { *form = 0; *group = 1; }
int ctx_set_option(Ctx*, const char*, long) { return STARKHIP_ERR_NO_DEVICE; }
void ctx_attach_hash_service(Ctx* c, HashService* hs) { c->hs = hs; }
bool ctx_has_hash_service(Ctx* c) { return c->hs != nullptr; }
void ctx_hash_request_reset(Ctx* c) { c->hash_requested = false; }
bool ctx_hash_requested(Ctx* c) { return c->hash_requested; }
int ctx_set_urgent(Ctx* c, bool urgent) { c->urgent = urgent; return STARKHIP_OK; }
int ctx_reserve(Ctx* c, const AirInfo& air, const starkhip_config_t&, size_t, unsigned proof_blobs, bool) {
    if (proof_blobs && !c->blob_airs.count(air.id)) {  // the fake proofs are tiny; what is exercised is the arena's bookkeeping
        if (blob_arena_add(c, 64 + air.prog.n_pis * 8, proof_blobs) != 0) return STARKHIP_ERR_OOM;
        c->blob_airs.insert(air.id);
    }
    return STARKHIP_OK;
}

int prove(Ctx* c, const AirInfo& air, const starkhip_config_t& cfg, const uint64_t* trace, size_t n_rows, int layout, int, const uint64_t* pis,
          size_t n_pis, uint64_t pow_witness, uint64_t** proof_out, size_t* proof_words) {
    if (!fake_device()) return STARKHIP_ERR_NO_DEVICE;
    if (n_pis != air.prog.n_pis) return STARKHIP_ERR_BAD_SHAPE;              // before the "commitment", like the real one
    if (pow_witness == 0xBAD) return STARKHIP_ERR_BAD_SHAPE;                  // a job that fails before its commitment (tests)
    if (layout == 2 && ((const TraceLog*)trace)->rows != n_rows) return STARKHIP_ERR_BAD_SHAPE;
    std::this_thread::sleep_for(std::chrono::milliseconds(air.cols > 50000 ? 3 : 1));  // "upload + LDE"
    unsigned log_n = 0;
    while (((size_t)1 << log_n) < n_rows) log_n++;
    if (c->hs) {
        c->hash_requested = true;
        static gl_t dummy[8];
        if (c->hs->hash(dummy, air.cols, log_n, cfg.rate_bits, dummy, nullptr, nullptr, nullptr, !HashService::is_big(log_n, cfg.rate_bits), c->urgent) != hipSuccess)
            return STARKHIP_ERR_HIP;
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(2));  // "the rest of the proof"
    if (pow_witness == 0xBAD2) return STARKHIP_ERR_QUOTIENT_NOT_DIVISIBLE;     // a job that fails after its commitment
    uint64_t* out = blob_alloc((4 + n_pis) * 8);
    if (!out) return STARKHIP_ERR_OOM;
    out[0] = 0xFA4EULL; out[1] = (uint64_t)air.id; out[2] = n_rows; out[3] = (uint64_t)c->urgent | ((uint64_t)c->device << 8);
    if (n_pis) memcpy(out + 4, pis, n_pis * 8);
    *proof_out = out;
    *proof_words = 4 + n_pis;
    c->timings[STARKHIP_N_PHASES - 1] = 3.0f;
    return STARKHIP_OK;
}
int lde_batch(Ctx*, const uint64_t*, size_t, unsigned, unsigned, uint64_t*, uint64_t*) { return STARKHIP_ERR_NO_DEVICE; }
int merkle_cap(Ctx*, const uint64_t*, size_t, unsigned, unsigned, uint64_t*) { return STARKHIP_ERR_NO_DEVICE; }
int permute_batch(Ctx*, uint64_t*, size_t) { return STARKHIP_ERR_NO_DEVICE; }
int expand_log(Ctx*, const TraceLog*, uint64_t*) { return STARKHIP_ERR_NO_DEVICE; }
int lde_bench(Ctx*, size_t, unsigned, unsigned, unsigned, unsigned, const uint64_t*, float*, float*) { return STARKHIP_ERR_NO_DEVICE; }
int field_ops(Ctx*, int, const uint64_t*, const uint64_t*, uint64_t*, size_t) { return STARKHIP_ERR_NO_DEVICE; }
int host_alloc(Ctx*, size_t, void**) { return STARKHIP_ERR_NO_DEVICE; }
void host_free(void*) {}
int quad_merged_tables_selfcheck(unsigned) { return 0; }  // the real one is compiled with the HIP sources

hipError_t event_wait_sleeping(hipEvent_t) { return hipSuccess; }  // prover.hip's sleeping wait: the fake device is always done

// the two launches the commitment scheduler makes
static std::atomic<unsigned long> g_fake_launches(0), g_fake_merged(0);
hipError_t launch_leaf_hash(const gl_t*, size_t, unsigned, unsigned, gl_t*, hipStream_t) { g_fake_launches++; return hipSuccess; }
hipError_t launch_leaf_hash_row(const gl_t*, size_t, unsigned, unsigned, gl_t*, hipStream_t) { g_fake_launches++; return hipSuccess; }
hipError_t launch_leaf_hash_lane(const gl_t*, size_t, unsigned, unsigned, gl_t*, hipStream_t) { g_fake_launches++; return hipSuccess; }
hipError_t launch_leaf_hash_pair(const gl_t*, size_t, unsigned, unsigned, gl_t*, hipStream_t) { g_fake_launches++; return hipSuccess; }
hipError_t launch_leaf_hash_multi(const LeafHashBatch&, unsigned count, size_t, unsigned, unsigned, hipStream_t) {
    g_fake_launches++;
    g_fake_merged += count;
    return hipSuccess;
}
}  // namespace starkhip

// ---- no-op HIP runtime: only what scheduler.cpp calls (the sanitizer builds do not link libamdhip64)
extern "C" {
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)malloc(1); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)malloc(1); return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 1; *greatest = -1; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free((void*)s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { *p = malloc(bytes ? bytes : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
}
