// Internal helpers for the composite gadgets (Fp6 / Fp12): where an Fp2 value lives inside a sub-gadget
// block, and the recurring families of copy constraints that wire one sub-gadget's output into the next
// one's input.  The reference writes each of these loops out by hand (e.g. src/fp6.rs:905-1553); the
// ORDER inside each loop body and the orientation (a - b) are preserved exactly.
#pragma once
#include "gadgets.h"

namespace starkhip {
namespace wire {
using namespace lay;

static const size_t RR = FP_SINGLE_REDUCE_TOTAL + RANGE_CHECK_TOTAL;

// location of the two Fp halves of an Fp2 value
struct Loc2 {
    size_t c0, c1;
};
inline Loc2 raw(size_t col) { return {col, col + 12}; }                                                      // 24 consecutive input limbs
inline Loc2 mul_out(size_t t) { return {t + Z1_REDUCE_OFFSET + REDUCED_OFFSET, t + Z2_REDUCE_OFFSET + REDUCED_OFFSET}; }  // fp2 mul result
inline Loc2 addred_out(size_t t) { return {t + FP2_ADDITION_TOTAL + FP_SINGLE_REDUCED_OFFSET, t + FP2_ADDITION_TOTAL + RR + FP_SINGLE_REDUCED_OFFSET}; }
inline Loc2 subred_out(size_t t) {
    return {t + FP2_ADDITION_TOTAL + FP2_SUBTRACTION_TOTAL + FP_SINGLE_REDUCED_OFFSET, t + FP2_ADDITION_TOTAL + FP2_SUBTRACTION_TOTAL + RR + FP_SINGLE_REDUCED_OFFSET};
}
inline Loc2 nr_out(size_t t) {
    return {t + FP2_NON_RESIDUE_MUL_Z0_REDUCE_OFFSET + FP_SINGLE_REDUCED_OFFSET, t + FP2_NON_RESIDUE_MUL_Z1_REDUCE_OFFSET + FP_SINGLE_REDUCED_OFFSET};
}
inline Loc2 fp2fp_out(size_t t) { return {t + X0_Y_REDUCE_OFFSET + REDUCED_OFFSET, t + X1_Y_REDUCE_OFFSET + REDUCED_OFFSET}; }  // fp2 x fp result

// fp2-mul block t fed from two 24-limb runs; `gadget_first`: (T.X[i] - x[i]) else (x[i] - T.X[i]).  i < 24, X then Y per i.
inline void mul_in24(CS& cs, const Expr& bs, size_t t, size_t xcol, size_t ycol, bool gadget_first) {
    const size_t sel = t + FP2_FP2_SELECTOR_OFFSET, X = t + FP2_FP2_X_INPUT_OFFSET, Y = t + FP2_FP2_Y_INPUT_OFFSET;
    if (gadget_first) cs.links(false, bs, 24, {{sel, X, xcol}, {sel, Y, ycol}});
    else cs.links(false, bs, 24, {{sel, xcol, X}, {sel, ycol, Y}});
}
// fp2-mul block t fed from two Loc2 values, i < 12: x.c0, x.c1, y.c0, y.c1 per i.
inline void mul_in(CS& cs, const Expr& bs, size_t t, Loc2 x, Loc2 y, bool gadget_first) {
    const size_t sel = t + FP2_FP2_SELECTOR_OFFSET, X = t + FP2_FP2_X_INPUT_OFFSET, Y = t + FP2_FP2_Y_INPUT_OFFSET;
    if (gadget_first) cs.links(false, bs, 12, {{sel, X, x.c0}, {sel, X + 12, x.c1}, {sel, Y, y.c0}, {sel, Y + 12, y.c1}});
    else cs.links(false, bs, 12, {{sel, x.c0, X}, {sel, x.c1, X + 12}, {sel, y.c0, Y}, {sel, y.c1, Y + 12}});
}
// addition-with-reduction block t: per i: A0.X - x.c0, A1.X - x.c1, A0.Y - y.c0, A1.Y - y.c1
inline void add_in(CS& cs, const Expr& bs, size_t t, Loc2 x, Loc2 y) {
    const size_t a0 = t + FP2_ADDITION_0_OFFSET, a1 = t + FP2_ADDITION_1_OFFSET;
    cs.links(false, bs, 12, {{a0 + FP_ADDITION_CHECK_OFFSET, a0 + FP_ADDITION_X_OFFSET, x.c0}, {a1 + FP_ADDITION_CHECK_OFFSET, a1 + FP_ADDITION_X_OFFSET, x.c1},
                             {a0 + FP_ADDITION_CHECK_OFFSET, a0 + FP_ADDITION_Y_OFFSET, y.c0}, {a1 + FP_ADDITION_CHECK_OFFSET, a1 + FP_ADDITION_Y_OFFSET, y.c1}});
}
// subtraction-with-reduction block t: per i: A0.X - x.c0, A1.X - x.c1, S0.Y - y.c0, S1.Y - y.c1
inline void sub_in(CS& cs, const Expr& bs, size_t t, Loc2 x, Loc2 y) {
    const size_t a0 = t + FP2_ADDITION_0_OFFSET, a1 = t + FP2_ADDITION_1_OFFSET;
    const size_t s0 = t + FP2_ADDITION_TOTAL + FP2_SUBTRACTION_0_OFFSET, s1 = t + FP2_ADDITION_TOTAL + FP2_SUBTRACTION_1_OFFSET;
    cs.links(false, bs, 12, {{a0 + FP_ADDITION_CHECK_OFFSET, a0 + FP_ADDITION_X_OFFSET, x.c0}, {a1 + FP_ADDITION_CHECK_OFFSET, a1 + FP_ADDITION_X_OFFSET, x.c1},
                             {s0 + FP_SUBTRACTION_CHECK_OFFSET, s0 + FP_SUBTRACTION_Y_OFFSET, y.c0}, {s1 + FP_SUBTRACTION_CHECK_OFFSET, s1 + FP_SUBTRACTION_Y_OFFSET, y.c1}});
}
// non-residue block t: per i: IN[i] - x.c0, IN[i + 12] - x.c1
inline void nr_in(CS& cs, const Expr& bs, size_t t, Loc2 x) {
    const size_t chk = t + FP2_NON_RESIDUE_MUL_CHECK_OFFSET, in = t + FP2_NON_RESIDUE_MUL_INPUT_OFFSET;
    cs.links(false, bs, 12, {{chk, in, x.c0}, {chk, in + 12, x.c1}});
}

// "per Fp half" orderings used by src/fp2.rs add_fp4_sq_constraints and src/fp12.rs add_cyclotomic_sq_constraints:
// per i: A0.X - x.c0, A0.Y - y.c0, A1.X - x.c1, A1.Y - y.c1
inline void add_in_alt(CS& cs, const Expr& bs, size_t t, Loc2 x, Loc2 y) {
    const size_t a0 = t + FP2_ADDITION_0_OFFSET, a1 = t + FP2_ADDITION_1_OFFSET;
    cs.links(false, bs, 12, {{a0 + FP_ADDITION_CHECK_OFFSET, a0 + FP_ADDITION_X_OFFSET, x.c0}, {a0 + FP_ADDITION_CHECK_OFFSET, a0 + FP_ADDITION_Y_OFFSET, y.c0},
                             {a1 + FP_ADDITION_CHECK_OFFSET, a1 + FP_ADDITION_X_OFFSET, x.c1}, {a1 + FP_ADDITION_CHECK_OFFSET, a1 + FP_ADDITION_Y_OFFSET, y.c1}});
}
// per i: A0.X - x.c0, S0.Y - y.c0, A1.X - x.c1, S1.Y - y.c1
inline void sub_in_alt(CS& cs, const Expr& bs, size_t t, Loc2 x, Loc2 y) {
    const size_t a0 = t + FP2_ADDITION_0_OFFSET, a1 = t + FP2_ADDITION_1_OFFSET;
    const size_t s0 = t + FP2_ADDITION_TOTAL + FP2_SUBTRACTION_0_OFFSET, s1 = t + FP2_ADDITION_TOTAL + FP2_SUBTRACTION_1_OFFSET;
    cs.links(false, bs, 12, {{a0 + FP_ADDITION_CHECK_OFFSET, a0 + FP_ADDITION_X_OFFSET, x.c0}, {s0 + FP_SUBTRACTION_CHECK_OFFSET, s0 + FP_SUBTRACTION_Y_OFFSET, y.c0},
                             {a1 + FP_ADDITION_CHECK_OFFSET, a1 + FP_ADDITION_X_OFFSET, x.c1}, {s1 + FP_SUBTRACTION_CHECK_OFFSET, s1 + FP_SUBTRACTION_Y_OFFSET, y.c1}});
}

// ---- Fp6-level: location of the i-th Fp (i < 6) of a value / of the i-th Fp add / sub block
typedef size_t (*Loc6)(size_t base, size_t i);
inline size_t raw6(size_t col, size_t i) { return col + 12 * i; }
inline size_t add6_block(size_t t, size_t i) {
    const size_t f2[3] = {FP6_ADDITION_0_OFFSET, FP6_ADDITION_1_OFFSET, FP6_ADDITION_2_OFFSET};
    return t + f2[i / 2] + (i % 2 ? FP2_ADDITION_1_OFFSET : FP2_ADDITION_0_OFFSET);
}
inline size_t sub6_block(size_t t, size_t i) {  // inside a subtraction-with-reduction block
    const size_t f2[3] = {FP6_SUBTRACTION_0_OFFSET, FP6_SUBTRACTION_1_OFFSET, FP6_SUBTRACTION_2_OFFSET};
    return t + FP6_ADDITION_TOTAL + f2[i / 2] + (i % 2 ? FP2_SUBTRACTION_1_OFFSET : FP2_SUBTRACTION_0_OFFSET);
}
inline size_t addred6_out(size_t t, size_t i) { return t + FP6_ADDITION_TOTAL + RR * i + FP_SINGLE_REDUCED_OFFSET; }
inline size_t subred6_out(size_t t, size_t i) { return t + FP6_ADDITION_TOTAL + FP6_SUBTRACTION_TOTAL + RR * i + FP_SINGLE_REDUCED_OFFSET; }
inline size_t fp6mul_out(size_t t, size_t i) {
    const size_t blk[3] = {FP6_MUL_X_CALC_OFFSET, FP6_MUL_Y_CALC_OFFSET, FP6_MUL_Z_CALC_OFFSET};
    return t + blk[i / 2] + FP2_ADDITION_TOTAL + RR * (i % 2) + FP_SINGLE_REDUCED_OFFSET;
}
inline size_t nr6_out(size_t t, size_t i) {
    if (i == 0) return t + FP6_NON_RESIDUE_MUL_C2 + FP2_NON_RESIDUE_MUL_Z0_REDUCE_OFFSET + FP_SINGLE_REDUCED_OFFSET;
    if (i == 1) return t + FP6_NON_RESIDUE_MUL_C2 + FP2_NON_RESIDUE_MUL_Z1_REDUCE_OFFSET + FP_SINGLE_REDUCED_OFFSET;
    return t + FP6_NON_RESIDUE_MUL_INPUT_OFFSET + (i - 2) * 12;
}
inline size_t m01_out(size_t t, size_t i) {
    const size_t blk[3] = {MULTIPLY_BY_01_X_CALC_OFFSET + FP2_ADDITION_TOTAL, MULTIPLY_BY_01_Y_CALC_OFFSET + FP2_ADDITION_TOTAL + FP2_SUBTRACTION_TOTAL,
                           MULTIPLY_BY_01_Z_CALC_OFFSET + FP2_ADDITION_TOTAL};
    return t + blk[i / 2] + RR * (i % 2) + FP_SINGLE_REDUCED_OFFSET;
}
inline size_t m1_out(size_t t, size_t i) {
    switch (i) {
        case 0: return t + MULTIPLY_BY_1_X_CALC_OFFSET + FP2_NON_RESIDUE_MUL_Z0_REDUCE_OFFSET + FP_SINGLE_REDUCED_OFFSET;
        case 1: return t + MULTIPLY_BY_1_X_CALC_OFFSET + FP2_NON_RESIDUE_MUL_Z1_REDUCE_OFFSET + FP_SINGLE_REDUCED_OFFSET;
        case 2: return t + MULTIPLY_BY_1_Y_CALC_OFFSET + Z1_REDUCE_OFFSET + REDUCED_OFFSET;
        case 3: return t + MULTIPLY_BY_1_Y_CALC_OFFSET + Z2_REDUCE_OFFSET + REDUCED_OFFSET;
        case 4: return t + MULTIPLY_BY_1_Z_CALC_OFFSET + Z1_REDUCE_OFFSET + REDUCED_OFFSET;
        default: return t + MULTIPLY_BY_1_Z_CALC_OFFSET + Z2_REDUCE_OFFSET + REDUCED_OFFSET;
    }
}
// Fp6 addition-with-reduction block t: for each Fp i < 6, for each limb j: A_i.X[j] - x_i[j], A_i.Y[j] - y_i[j]
inline void add6_in(CS& cs, const Expr& bs, size_t t, Loc6 xf, size_t xb, Loc6 yf, size_t yb) {
    for (size_t i = 0; i < 6; i++) {
        const size_t a = add6_block(t, i);
        cs.links(false, bs, 12, {{a + FP_ADDITION_CHECK_OFFSET, a + FP_ADDITION_X_OFFSET, xf(xb, i)}, {a + FP_ADDITION_CHECK_OFFSET, a + FP_ADDITION_Y_OFFSET, yf(yb, i)}});
    }
}
// Fp6 subtraction-with-reduction block t: for each Fp i, limb j: A_i.X[j] - x_i[j], S_i.Y[j] - y_i[j]
inline void sub6_in(CS& cs, const Expr& bs, size_t t, Loc6 xf, size_t xb, Loc6 yf, size_t yb) {
    for (size_t i = 0; i < 6; i++) {
        const size_t a = add6_block(t, i), s = sub6_block(t, i);
        cs.links(false, bs, 12, {{a + FP_ADDITION_CHECK_OFFSET, a + FP_ADDITION_X_OFFSET, xf(xb, i)}, {s + FP_SUBTRACTION_CHECK_OFFSET, s + FP_SUBTRACTION_Y_OFFSET, yf(yb, i)}});
    }
}

}  // namespace wire
}  // namespace starkhip
