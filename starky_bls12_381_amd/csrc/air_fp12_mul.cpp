// FP12MulStark: one Fp12 multiplication (60285 columns x 16 rows, constraint degree 3).
// Restates /root/reference/src/fp12_mul.rs: generate_trace (:44-48), eval_packed_generic (:58-99),
// constraint_degree (:142-144); public inputs as built by fp12_mul_main, src/aggregate_proof.rs:125-135.
#include "airs.h"
#include "gadgets.h"

namespace starkhip {
using namespace lay;
using namespace lay_fp12mul;

AirProgram build_air_fp12_mul() {
    AirBuilder b(COLUMNS, PUBLIC_INPUTS, 3);
    CS cs(b);
    const Expr sel = cs.L(FP12_MUL_SELECTOR_OFFSET);
    for (size_t i = 0; i < 144; i++) {
        cs.c(sel * (cs.L(FP12_MUL_X_INPUT_OFFSET + i) - b.PI(PIS_INPUT_X_OFFSET + i)));
        cs.c(sel * (cs.L(FP12_MUL_Y_INPUT_OFFSET + i) - b.PI(PIS_INPUT_Y_OFFSET + i)));
    }
    const size_t RR = FP_SINGLE_REDUCE_TOTAL + RANGE_CHECK_TOTAL;
    for (size_t i = 0; i < 12; i++)
        for (size_t j = 0; j < 6; j++)
            for (size_t k = 0; k < 2; k++) {
                const size_t xy = k == 0 ? FP12_MUL_X_CALC_OFFSET + FP6_ADDITION_TOTAL : FP12_MUL_Y_CALC_OFFSET + FP6_ADDITION_TOTAL + FP6_SUBTRACTION_TOTAL;
                const size_t off = xy + RR * j + FP_SINGLE_REDUCED_OFFSET + i;
                cs.c(sel * (cs.L(off) - b.PI(PIS_OUTPUT_OFFSET + k * 72 + j * 12 + i)));
            }
    add_fp12_multiplication_constraints(cs, 0, CS::one());
    return b.finish();
}

}  // namespace starkhip

using namespace starkhip;

extern "C" int starkhip_trace_fp12_mul(const uint32_t x[144], const uint32_t y[144], uint64_t* trace, size_t n_rows, uint64_t* public_inputs) {
    if (n_rows < 12 || (n_rows & (n_rows - 1))) return STARKHIP_ERR_BAD_SHAPE;
    try {
        bls::Fp12 X = bls::Fp12::from_limbs(x), Y = bls::Fp12::from_limbs(y);
        Trace t = open_trace(trace, n_rows, lay_fp12mul::COLUMNS);
        fill_trace_fp12_multiplication(t, X, Y, 0, 11, 0);
        bls::Fp12 Z = X * Y;
        uint32_t z[144];
        Z.to_limbs(z);
        for (int i = 0; i < 144; i++) {
            public_inputs[i] = x[i];
            public_inputs[144 + i] = y[i];
            public_inputs[288 + i] = z[i];
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "starkhip_trace_fp12_mul: %s\n", e.what());
        return STARKHIP_ERR_BAD_SHAPE;
    }
    return STARKHIP_OK;
}
