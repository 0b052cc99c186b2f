// Placeholders for AIRs that are not restated yet: an empty program makes air_get() return nullptr.
#include "airs.h"
namespace starkhip {
AirProgram build_air_pairing_precomp() { return AirProgram(); }
}  // namespace starkhip

extern "C" {
int starkhip_trace_pairing_precomp(const uint32_t*, const uint32_t*, const uint32_t*, uint64_t*, size_t, uint64_t*) { return STARKHIP_ERR_BAD_AIR; }
}
