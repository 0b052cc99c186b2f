// AIR registry: builds each constraint program once, on first use.
#include "airs.h"

#include <mutex>

namespace starkhip {

namespace {
struct Slot {
    int id;
    const char* name;
    uint32_t default_rows;
    AirProgram (*build)();
    std::once_flag once;
    AirInfo info;
    bool ok = false;
};

// default_rows: the sizes the reference instantiates, /root/reference/src/aggregate_proof.rs:34,77,123,157
Slot g_slots[] = {
    {STARKHIP_AIR_FP12_MUL, "FP12MulStark", 16, build_air_fp12_mul},
    {STARKHIP_AIR_PAIRING_PRECOMP, "PairingPrecompStark", 1024, build_air_pairing_precomp},
    {STARKHIP_AIR_MILLER_LOOP, "MillerLoopStark", 1024, build_air_miller_loop},
    {STARKHIP_AIR_FINAL_EXP, "FinalExponentiateStark", 8192, build_air_final_exp},
    {STARKHIP_AIR_ECC_AGGREGATE, "ECCAggStark", 8192, build_air_ecc_aggregate},  // src/aggregate_proof.rs:188-189
    {STARKHIP_AIR_TEST_FIBONACCI, "TestFibonacci", 64, build_air_fibonacci},
};
}  // namespace

const AirInfo* air_get(int id) {
    for (auto& s : g_slots) {
        if (s.id != id) continue;
        std::call_once(s.once, [&s]() {
            try {
                s.info.prog = s.build();
                s.info.id = s.id;
                s.info.name = s.name;
                s.info.cols = s.info.prog.n_cols;
                s.info.pis = s.info.prog.n_pis;
                s.info.degree = s.info.prog.degree;
                s.info.default_rows = s.default_rows;
                s.info.blob = s.info.prog.serialize();
                s.ok = s.info.prog.n_constraints > 0;
            } catch (const std::exception& e) {
                fprintf(stderr, "starkhip: building AIR %s failed: %s\n", s.name, e.what());
                s.ok = false;
            }
        });
        return s.ok ? &s.info : nullptr;
    }
    return nullptr;
}

}  // namespace starkhip
