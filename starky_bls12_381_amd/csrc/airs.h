// Registry of the AIRs this library can prove (one per `impl Stark for ...` in the reference).
#pragma once
#include <vector>

#include "../../include/starkhip.h"
#include "air_ir.h"

namespace starkhip {

struct AirInfo {
    int id;
    const char* name;
    uint32_t cols, pis, degree, default_rows;
    AirProgram prog;
    std::vector<uint64_t> blob;  // prog.serialize()
};

// nullptr for an unknown id.  Programs are built on first use and cached for the process lifetime.
const AirInfo* air_get(int id);

// builders (air_*.cpp)
AirProgram build_air_fibonacci();
AirProgram build_air_fp12_mul();
AirProgram build_air_final_exp();
AirProgram build_air_miller_loop();
AirProgram build_air_pairing_precomp();
AirProgram build_air_ecc_aggregate();

}  // namespace starkhip
