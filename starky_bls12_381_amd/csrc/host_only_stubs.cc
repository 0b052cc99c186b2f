// Stand-ins for the GPU entry points, linked ONLY into the sanitizer build of the host code (make asan): every device
// path answers "no device", as the real library does on a box without a GPU.  Not part of libstarkhip.so.
#include "prover.h"

namespace starkhip {
struct Ctx {};
int ctx_create(int, Ctx** out) { *out = nullptr; return STARKHIP_ERR_NO_DEVICE; }
void ctx_destroy(Ctx*) {}
const float* ctx_timings(Ctx*) { static float z[STARKHIP_N_PHASES] = {0}; return z; }
const float* ctx_kernel_timings(Ctx*) { static float z[3] = {0}; return z; }
int ctx_set_option(Ctx*, const char*, long) { return STARKHIP_ERR_NO_DEVICE; }
int prove(Ctx*, const AirInfo&, const starkhip_config_t&, const uint64_t*, size_t, int, int, const uint64_t*, size_t, uint64_t, uint64_t**, size_t*) {
    return STARKHIP_ERR_NO_DEVICE;
}
int lde_batch(Ctx*, const uint64_t*, size_t, unsigned, unsigned, uint64_t*, uint64_t*) { return STARKHIP_ERR_NO_DEVICE; }
int merkle_cap(Ctx*, const uint64_t*, size_t, unsigned, unsigned, uint64_t*) { return STARKHIP_ERR_NO_DEVICE; }
int permute_batch(Ctx*, uint64_t*, size_t) { return STARKHIP_ERR_NO_DEVICE; }
int field_ops(Ctx*, int, const uint64_t*, const uint64_t*, uint64_t*, size_t) { return STARKHIP_ERR_NO_DEVICE; }
int host_alloc(Ctx*, size_t, void**) { return STARKHIP_ERR_NO_DEVICE; }
void host_free(void*) {}
int quad_merged_tables_selfcheck(unsigned) { return 0; }  // the real one is compiled with the HIP sources
}  // namespace starkhip
