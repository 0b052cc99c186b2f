// Fp2 gadgets: trace fillers and constraint emitters.  Restates the fill_* / packed add_*_constraints
// halves of /root/reference/src/fp2.rs (line references on each function).
#include "gadgets.h"

namespace starkhip {
using namespace lay;
using namespace bls;

static const size_t RR = FP_SINGLE_REDUCE_TOTAL + RANGE_CHECK_TOTAL;  // one "reduce + range check" slot

static L24 add24(const L24& a, const L24& b) {
    L24 s, c;
    add_u32_slices(a, b, s, c);
    return s;
}
static L24 sub24(const L24& a, const L24& b) {
    L24 d, br;
    sub_u32_slices(a, b, d, br);
    return d;
}
static L12 add12(const L12& a, const L12& b) {  // BigUint sum truncated to 12 limbs (callers guarantee it fits)
    L12 s, c;
    add_u32_slices_12(a, b, s, c);
    return s;
}
static L12 sub12(const L12& a, const L12& b) {
    L12 d, br;
    sub_u32_slices_12(a, b, d, br);
    return d;
}

// ------------------------------------------------------------------ fillers
void fill_trace_addition_fp2(Trace& t, const Fp2& x, const Fp2& y, size_t row, size_t col) {  // fp2.rs:187-199
    fill_trace_addition_fp(t, x.c[0].l, y.c[0].l, row, col + FP2_ADDITION_0_OFFSET);
    fill_trace_addition_fp(t, x.c[1].l, y.c[1].l, row, col + FP2_ADDITION_1_OFFSET);
}
void fill_trace_subtraction_fp2(Trace& t, const Fp2& x, const Fp2& y, size_t row, size_t col) {  // fp2.rs:202-214
    fill_trace_subtraction_fp(t, x.c[0].l, y.c[0].l, row, col + FP2_SUBTRACTION_0_OFFSET);
    fill_trace_subtraction_fp(t, x.c[1].l, y.c[1].l, row, col + FP2_SUBTRACTION_1_OFFSET);
}
void fill_trace_negate_fp2(Trace& t, const Fp2& x, size_t row, size_t col) {  // fp2.rs:232-243
    fill_trace_addition_fp2(t, x, -x, row, col);
}
// fp2.rs:246-319.  Additions / subtractions live on row start_row + 11 (where the long multiplication finishes);
// the range checks are filled on start_row only (App. B.4 item 14).
void generate_trace_fp2_mul(Trace& t, const Fp2& x, const Fp2& y, size_t start_row, size_t end_row, size_t col) {
    for (size_t i = start_row; i <= end_row; i++) {
        t.at(i, col + FP2_FP2_SELECTOR_OFFSET) = 1;
        t.put(i, col + FP2_FP2_X_INPUT_OFFSET, x);
        t.put(i, col + FP2_FP2_Y_INPUT_OFFSET, y);
    }
    t.at(end_row, col + FP2_FP2_SELECTOR_OFFSET) = 0;
    fill_multiplication_trace_no_mod_reduction(t, x.c[0].l, y.c[0].l, start_row, end_row, col + X_0_Y_0_MULTIPLICATION_OFFSET);
    fill_multiplication_trace_no_mod_reduction(t, x.c[1].l, y.c[1].l, start_row, end_row, col + X_1_Y_1_MULTIPLICATION_OFFSET);
    L24 x0y0 = mul_wide(x.c[0].l, y.c[0].l);
    fill_addition_trace(t, x0y0, modulus_sq_limbs(), start_row + 11, col + Z1_ADD_MODULUS_OFFSET);
    L24 x0y0_p2 = add24(x0y0, modulus_sq_limbs());
    L24 x1y1 = mul_wide(x.c[1].l, y.c[1].l);
    fill_subtraction_trace(t, x0y0_p2, x1y1, start_row + 11, col + Z1_SUBTRACTION_OFFSET);
    L12 rem = fill_reduction_trace(t, sub24(x0y0_p2, x1y1), start_row, end_row, col + Z1_REDUCE_OFFSET);
    fill_range_check_trace(t, rem, start_row, col + Z1_RANGECHECK_OFFSET);
    fill_multiplication_trace_no_mod_reduction(t, x.c[0].l, y.c[1].l, start_row, end_row, col + X_0_Y_1_MULTIPLICATION_OFFSET);
    fill_multiplication_trace_no_mod_reduction(t, x.c[1].l, y.c[0].l, start_row, end_row, col + X_1_Y_0_MULTIPLICATION_OFFSET);
    L24 x0y1 = mul_wide(x.c[0].l, y.c[1].l), x1y0 = mul_wide(x.c[1].l, y.c[0].l);
    fill_addition_trace(t, x0y1, x1y0, start_row + 11, col + Z2_ADDITION_OFFSET);
    rem = fill_reduction_trace(t, add24(x0y1, x1y0), start_row, end_row, col + Z2_REDUCE_OFFSET);
    fill_range_check_trace(t, rem, start_row, col + Z2_RANGECHECK_OFFSET);
}
void fill_trace_fp2_fp_mul(Trace& t, const Fp2& x, const Fp& y, size_t start_row, size_t end_row, size_t col) {  // fp2.rs:322-341
    for (size_t i = start_row; i <= end_row; i++) {
        t.at(i, col + FP2_FP_MUL_SELECTOR_OFFSET) = 1;
        t.put(i, col + FP2_FP_X_INPUT_OFFSET, x);
        t.put(i, col + FP2_FP_Y_INPUT_OFFSET, y.l);
    }
    t.at(end_row, col + FP2_FP_MUL_SELECTOR_OFFSET) = 0;
    fill_multiplication_trace_no_mod_reduction(t, x.c[0].l, y.l, start_row, end_row, col + X0_Y_MULTIPLICATION_OFFSET);
    L12 rem = fill_reduction_trace(t, mul_wide(x.c[0].l, y.l), start_row, end_row, col + X0_Y_REDUCE_OFFSET);
    fill_range_check_trace(t, rem, start_row, col + X0_Y_RANGECHECK_OFFSET);
    fill_multiplication_trace_no_mod_reduction(t, x.c[1].l, y.l, start_row, end_row, col + X1_Y_MULTIPLICATION_OFFSET);
    rem = fill_reduction_trace(t, mul_wide(x.c[1].l, y.l), start_row, end_row, col + X1_Y_REDUCE_OFFSET);
    fill_range_check_trace(t, rem, start_row, col + X1_Y_RANGECHECK_OFFSET);
}
void fill_trace_subtraction_with_reduction(Trace& t, const Fp2& x, const Fp2& y, size_t row, size_t col) {  // fp2.rs:344-367
    Fp2 pp{Fp(MODULUS), Fp(MODULUS)};
    fill_trace_addition_fp2(t, x, pp, row, col);
    Fp2 xm{Fp(add12(x.c[0].l, MODULUS)), Fp(add12(x.c[1].l, MODULUS))};
    fill_trace_subtraction_fp2(t, xm, y, row, col + FP2_ADDITION_TOTAL);
    L12 d0 = sub12(xm.c[0].l, y.c[0].l), d1 = sub12(xm.c[1].l, y.c[1].l);
    const size_t base = col + FP2_ADDITION_TOTAL + FP2_SUBTRACTION_TOTAL;
    L12 rem = fill_trace_reduce_single(t, d0, row, base);
    fill_range_check_trace(t, rem, row, base + FP_SINGLE_REDUCE_TOTAL);
    rem = fill_trace_reduce_single(t, d1, row, base + RR);
    fill_range_check_trace(t, rem, row, base + RR + FP_SINGLE_REDUCE_TOTAL);
}
void fill_multiply_by_b_trace(Trace& t, const Fp2& x, size_t start_row, size_t end_row, size_t col) {  // fp2.rs:370-404
    for (size_t i = start_row; i <= end_row; i++) {
        t.at(i, col + MULTIPLY_B_SELECTOR_OFFSET) = 1;
        t.put(i, col + MULTIPLY_B_X_OFFSET, x);
    }
    t.at(end_row, col + MULTIPLY_B_SELECTOR_OFFSET) = 0;
    const L12 four = Fp::from_u32(4).l;
    fill_multiplication_trace_no_mod_reduction(t, x.c[0].l, four, start_row, end_row, col + MULTIPLY_B_X0_B_MUL_OFFSET);
    fill_multiplication_trace_no_mod_reduction(t, x.c[1].l, four, start_row, end_row, col + MULTIPLY_B_X1_B_MUL_OFFSET);
    L24 x0y = mul_wide(x.c[0].l, four), x1y = mul_wide(x.c[1].l, four);
    fill_addition_trace(t, x0y, modulus_sq_limbs(), start_row + 11, col + MULTIPLY_B_ADD_MODSQ_OFFSET);
    L24 x0y_p2 = add24(x0y, modulus_sq_limbs());
    fill_subtraction_trace(t, x0y_p2, x1y, start_row + 11, col + MULTIPLY_B_SUB_OFFSET);
    L12 rem = fill_reduction_trace(t, sub24(x0y_p2, x1y), start_row, end_row, col + MULTIPLY_B_Z0_REDUCE_OFFSET);
    fill_range_check_trace(t, rem, start_row, col + MULTIPLY_B_Z0_RANGECHECK_OFFSET);
    fill_addition_trace(t, x0y, x1y, start_row + 11, col + MULTIPLY_B_ADD_OFFSET);
    rem = fill_reduction_trace(t, add24(x0y, x1y), start_row, end_row, col + MULTIPLY_B_Z1_REDUCE_OFFSET);
    fill_range_check_trace(t, rem, start_row, col + MULTIPLY_B_Z1_RANGECHECK_OFFSET);
}
void fill_trace_addition_with_reduction(Trace& t, const Fp2& x, const Fp2& y, size_t row, size_t col) {  // fp2.rs:407-422
    fill_trace_addition_fp2(t, x, y, row, col);
    L12 s0 = add12(x.c[0].l, y.c[0].l), s1 = add12(x.c[1].l, y.c[1].l);
    L12 rem = fill_trace_reduce_single(t, s0, row, col + FP2_ADDITION_TOTAL);
    fill_range_check_trace(t, rem, row, col + FP2_ADDITION_TOTAL + FP_SINGLE_REDUCE_TOTAL);
    rem = fill_trace_reduce_single(t, s1, row, col + FP2_ADDITION_TOTAL + RR);
    fill_range_check_trace(t, rem, row, col + FP2_ADDITION_TOTAL + RR + FP_SINGLE_REDUCE_TOTAL);
}
void fill_trace_non_residue_multiplication(Trace& t, const Fp2& x, size_t row, size_t col) {  // fp2.rs:425-447
    t.at(row, col + FP2_NON_RESIDUE_MUL_CHECK_OFFSET) = 1;
    t.put(row, col + FP2_NON_RESIDUE_MUL_INPUT_OFFSET, x);
    fill_trace_addition_fp(t, x.c[0].l, MODULUS, row, col + FP2_NON_RESIDUE_MUL_C0_C1_SUB_OFFSET);
    L12 xm = add12(x.c[0].l, MODULUS);
    fill_trace_subtraction_fp(t, xm, x.c[1].l, row, col + FP2_NON_RESIDUE_MUL_C0_C1_SUB_OFFSET + FP_ADDITION_TOTAL);
    L12 rem = fill_trace_reduce_single(t, sub12(xm, x.c[1].l), row, col + FP2_NON_RESIDUE_MUL_Z0_REDUCE_OFFSET);
    fill_range_check_trace(t, rem, row, col + FP2_NON_RESIDUE_MUL_Z0_RANGECHECK_OFFSET);
    fill_trace_addition_fp(t, x.c[0].l, x.c[1].l, row, col + FP2_NON_RESIDUE_MUL_C0_C1_ADD_OFFSET);
    rem = fill_trace_reduce_single(t, add12(x.c[0].l, x.c[1].l), row, col + FP2_NON_RESIDUE_MUL_Z1_REDUCE_OFFSET);
    fill_range_check_trace(t, rem, row, col + FP2_NON_RESIDUE_MUL_Z1_RANGECHECK_OFFSET);
}
void fill_trace_fp4_sq(Trace& t, const Fp2& x, const Fp2& y, size_t start_row, size_t end_row, size_t col) {  // fp2.rs:450-494
    {
        RowSpan rows_(t, end_row - start_row + 1);
        t.put(start_row, col + FP4_SQ_INPUT_X_OFFSET, x);
        t.put(start_row, col + FP4_SQ_INPUT_Y_OFFSET, y);
        t.at(start_row, col + FP4_SQ_SELECTOR_OFFSET) = 1;
    }
    t.at(end_row, col + FP4_SQ_SELECTOR_OFFSET) = 0;
    Fp2 t0 = x * x;
    generate_trace_fp2_mul(t, x, x, start_row, end_row, col + FP4_SQ_T0_CALC_OFFSET);
    Fp2 t1 = y * y;
    generate_trace_fp2_mul(t, y, y, start_row, end_row, col + FP4_SQ_T1_CALC_OFFSET);
    Fp2 t2 = t1.mul_by_nonresidue();
    { RowSpan rows_(t, end_row - start_row + 1); fill_trace_non_residue_multiplication(t, t1, start_row, col + FP4_SQ_T2_CALC_OFFSET); }
    { RowSpan rows_(t, end_row - start_row + 1); fill_trace_addition_with_reduction(t, t2, t0, start_row, col + FP4_SQ_X_CALC_OFFSET); }
    Fp2 t3 = x + y;
    { RowSpan rows_(t, end_row - start_row + 1); fill_trace_addition_with_reduction(t, x, y, start_row, col + FP4_SQ_T3_CALC_OFFSET); }
    Fp2 t4 = t3 * t3;
    generate_trace_fp2_mul(t, t3, t3, start_row, end_row, col + FP4_SQ_T4_CALC_OFFSET);
    Fp2 t5 = t4 - t0;
    { RowSpan rows_(t, end_row - start_row + 1); fill_trace_subtraction_with_reduction(t, t4, t0, start_row, col + FP4_SQ_T5_CALC_OFFSET); }
    { RowSpan rows_(t, end_row - start_row + 1); fill_trace_subtraction_with_reduction(t, t5, t1, start_row, col + FP4_SQ_Y_CALC_OFFSET); }
}
void fill_trace_fp2_forbenius_map(Trace& t, const Fp2& x, size_t pow, size_t start_row, size_t end_row, size_t col) {  // fp2.rs:497-521
    const size_t div = pow / 2, rem = pow % 2;
    {
        RowSpan rows_(t, end_row - start_row + 1);
        t.put(start_row, col + FP2_FORBENIUS_MAP_INPUT_OFFSET, x);
        t.at(start_row, col + FP2_FORBENIUS_MAP_SELECTOR_OFFSET) = 1;
        t.at(start_row, col + FP2_FORBENIUS_MAP_POW_OFFSET) = pow;
        t.at(start_row, col + FP2_FORBENIUS_MAP_DIV_OFFSET) = div;
        t.at(start_row, col + FP2_FORBENIUS_MAP_REM_OFFSET) = rem;
    }
    t.at(end_row, col + FP2_FORBENIUS_MAP_SELECTOR_OFFSET) = 0;
    const Fp& coef = FP2_FROBENIUS_COEFF[rem];
    fill_multiplication_trace_no_mod_reduction(t, x.c[1].l, coef.l, start_row, end_row, col + FP2_FORBENIUS_MAP_T0_CALC_OFFSET);
    t.at(start_row + 11, col + FP2_FORBENIUS_MAP_MUL_RES_ROW) = 1;
    L12 res = fill_reduction_trace(t, mul_wide(x.c[1].l, coef.l), start_row, end_row, col + FP2_FORBENIUS_MAP_T0_CALC_OFFSET + FP_MULTIPLICATION_TOTAL_COLUMNS);
    for (size_t row = start_row; row <= end_row; row++)
        fill_range_check_trace(t, res, row, col + FP2_FORBENIUS_MAP_T0_CALC_OFFSET + FP_MULTIPLICATION_TOTAL_COLUMNS + REDUCTION_TOTAL);
}

// ------------------------------------------------------------------ constraints
void add_addition_fp2_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp2.rs:524-541
    add_addition_fp_constraints(cs, sc + FP2_ADDITION_0_OFFSET, bs);
    add_addition_fp_constraints(cs, sc + FP2_ADDITION_1_OFFSET, bs);
}
void add_subtraction_fp2_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp2.rs:558-575
    add_subtraction_fp_constraints(cs, sc + FP2_SUBTRACTION_0_OFFSET, bs);
    add_subtraction_fp_constraints(cs, sc + FP2_SUBTRACTION_1_OFFSET, bs);
}
void add_fp2_single_multiply_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp2.rs:592-609
    add_fp_single_multiply_constraints(cs, sc + FP2_MULTIPLY_SINGLE_0_OFFSET, bs);
    add_fp_single_multiply_constraints(cs, sc + FP2_MULTIPLY_SINGLE_1_OFFSET, bs);
}
void add_negate_fp2_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp2.rs:626-659
    add_addition_fp2_constraints(cs, sc, bs);
    const size_t a0 = sc + FP2_ADDITION_0_OFFSET, a1 = sc + FP2_ADDITION_1_OFFSET;
    for (size_t i = 0; i < 12; i++) {
        cs.c(bs * cs.L(a0 + FP_ADDITION_CHECK_OFFSET) * (cs.L(a0 + FP_ADDITION_SUM_OFFSET + i) - CS::K(MODULUS[i])));
        cs.c(bs * cs.L(a1 + FP_ADDITION_CHECK_OFFSET) * (cs.L(a1 + FP_ADDITION_SUM_OFFSET + i) - CS::K(MODULUS[i])));
    }
}
void add_fp2_mul_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp2.rs:697-927
    const size_t sel = sc + FP2_FP2_SELECTOR_OFFSET, X = sc + FP2_FP2_X_INPUT_OFFSET, Y = sc + FP2_FP2_Y_INPUT_OFFSET;
    const size_t m00 = sc + X_0_Y_0_MULTIPLICATION_OFFSET, m11 = sc + X_1_Y_1_MULTIPLICATION_OFFSET;
    const size_t m01 = sc + X_0_Y_1_MULTIPLICATION_OFFSET, m10 = sc + X_1_Y_0_MULTIPLICATION_OFFSET;
    const Expr g = bs * cs.L(sel);
    for (size_t i = 0; i < 24; i++) {
        cs.ct(g * (cs.L(X + i) - cs.N(X + i)));
        cs.ct(g * (cs.L(Y + i) - cs.N(Y + i)));
    }
    cs.links(false, bs, 12, {{sel, m00 + X_INPUT_OFFSET, X}, {sel, m00 + Y_INPUT_OFFSET, Y},
                             {sel, m01 + X_INPUT_OFFSET, X}, {sel, m01 + Y_INPUT_OFFSET, Y + 12},
                             {sel, m10 + X_INPUT_OFFSET, X + 12}, {sel, m10 + Y_INPUT_OFFSET, Y},
                             {sel, m11 + X_INPUT_OFFSET, X + 12}, {sel, m11 + Y_INPUT_OFFSET, Y + 12}});
    add_multiplication_constraints(cs, m00, bs);
    add_multiplication_constraints(cs, m11, bs);
    const size_t zadd = sc + Z1_ADD_MODULUS_OFFSET, zsub = sc + Z1_SUBTRACTION_OFFSET, zred = sc + Z1_REDUCE_OFFSET;
    cs.link(true, bs * cs.L(zadd + ADDITION_CHECK_OFFSET), zadd + ADDITION_X_OFFSET, m00 + SUM_OFFSET, 24);
    cs.link_const(true, bs * cs.L(zadd + ADDITION_CHECK_OFFSET), zadd + ADDITION_Y_OFFSET, modulus_sq_limbs().data(), 24);
    add_addition_constraints(cs, zadd, bs);
    cs.link(true, bs * cs.L(zsub + SUBTRACTION_CHECK_OFFSET), zsub + SUBTRACTION_X_OFFSET, zadd + ADDITION_SUM_OFFSET, 24);
    cs.link(true, bs * cs.L(zsub + SUBTRACTION_CHECK_OFFSET), zsub + SUBTRACTION_Y_OFFSET, m11 + SUM_OFFSET, 24);
    add_subtraction_constraints(cs, zsub, bs);
    cs.link(true, bs * cs.L(zsub + SUBTRACTION_CHECK_OFFSET), zsub + SUBTRACTION_DIFF_OFFSET, zred + REDUCE_X_OFFSET, 24);
    add_reduce_constraints(cs, zred, sel, bs);
    add_range_check_constraints(cs, sc + Z1_RANGECHECK_OFFSET, bs);
    add_multiplication_constraints(cs, m01, bs);
    add_multiplication_constraints(cs, m10, bs);
    const size_t z2add = sc + Z2_ADDITION_OFFSET, z2red = sc + Z2_REDUCE_OFFSET;
    cs.link(true, bs * cs.L(z2add + ADDITION_CHECK_OFFSET), z2add + ADDITION_X_OFFSET, m01 + SUM_OFFSET, 24);
    cs.link(true, bs * cs.L(z2add + ADDITION_CHECK_OFFSET), z2add + ADDITION_Y_OFFSET, m10 + SUM_OFFSET, 24);
    add_addition_constraints(cs, z2add, bs);
    cs.link(true, bs * cs.L(z2add + ADDITION_CHECK_OFFSET), z2add + ADDITION_SUM_OFFSET, z2red + REDUCE_X_OFFSET, 24);
    add_reduce_constraints(cs, z2red, sel, bs);
    add_range_check_constraints(cs, sc + Z2_RANGECHECK_OFFSET, bs);
}
void add_fp2_fp_mul_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp2.rs:1089-1172
    const size_t sel = sc + FP2_FP_MUL_SELECTOR_OFFSET, X = sc + FP2_FP_X_INPUT_OFFSET, Y = sc + FP2_FP_Y_INPUT_OFFSET;
    const size_t m0 = sc + X0_Y_MULTIPLICATION_OFFSET, m1 = sc + X1_Y_MULTIPLICATION_OFFSET;
    const Expr g = bs * cs.L(sel);
    for (size_t i = 0; i < 12; i++) {
        for (size_t j = 0; j < 2; j++) cs.ct(g * (cs.L(X + j * 12 + i) - cs.N(X + j * 12 + i)));
        cs.ct(g * (cs.L(Y + i) - cs.N(Y + i)));
    }
    cs.links(true, bs, 12, {{sel, X, m0 + X_INPUT_OFFSET}, {sel, X + 12, m1 + X_INPUT_OFFSET}, {sel, Y, m0 + Y_INPUT_OFFSET}, {sel, Y, m1 + Y_INPUT_OFFSET}});
    add_multiplication_constraints(cs, m0, bs);
    const size_t r0 = sc + X0_Y_REDUCE_OFFSET, r1 = sc + X1_Y_REDUCE_OFFSET;
    cs.link(false, bs * cs.L(r0 + REDUCTION_ADDITION_OFFSET + ADDITION_CHECK_OFFSET), r0 + REDUCTION_ADDITION_OFFSET + ADDITION_SUM_OFFSET, m0 + SUM_OFFSET, 24);
    add_reduce_constraints(cs, r0, sel, bs);
    add_range_check_constraints(cs, sc + X0_Y_RANGECHECK_OFFSET, bs);
    add_multiplication_constraints(cs, m1, bs);
    cs.link(false, bs * cs.L(r1 + REDUCTION_ADDITION_OFFSET + ADDITION_CHECK_OFFSET), r1 + REDUCTION_ADDITION_OFFSET + ADDITION_SUM_OFFSET, m1 + SUM_OFFSET, 24);
    add_reduce_constraints(cs, r1, sel, bs);
    add_range_check_constraints(cs, sc + X1_Y_RANGECHECK_OFFSET, bs);
}
void add_multiply_by_b_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp2.rs:1251-1376
    const size_t sel = sc + MULTIPLY_B_SELECTOR_OFFSET, X = sc + MULTIPLY_B_X_OFFSET;
    const size_t m0 = sc + MULTIPLY_B_X0_B_MUL_OFFSET, m1 = sc + MULTIPLY_B_X1_B_MUL_OFFSET;
    const Expr g = bs * cs.L(sel);
    cs.keep(true, g, X, 24);
    for (size_t i = 0; i < 12; i++) {
        cs.c(g * (cs.L(X + i) - cs.L(m0 + X_INPUT_OFFSET + i)));
        cs.c(g * (cs.L(X + 12 + i) - cs.L(m1 + X_INPUT_OFFSET + i)));
        if (i == 0) {
            cs.c(g * (cs.L(m0 + Y_INPUT_OFFSET + i) - CS::K(4)));
            cs.c(g * (cs.L(m1 + Y_INPUT_OFFSET + i) - CS::K(4)));
        } else {
            cs.c(g * cs.L(m0 + Y_INPUT_OFFSET + i));
            cs.c(g * cs.L(m1 + Y_INPUT_OFFSET + i));
        }
    }
    add_multiplication_constraints(cs, m0, bs);
    add_multiplication_constraints(cs, m1, bs);
    const size_t am = sc + MULTIPLY_B_ADD_MODSQ_OFFSET, sb = sc + MULTIPLY_B_SUB_OFFSET, ad = sc + MULTIPLY_B_ADD_OFFSET;
    const size_t z0 = sc + MULTIPLY_B_Z0_REDUCE_OFFSET, z1 = sc + MULTIPLY_B_Z1_REDUCE_OFFSET;
    const L24& p2 = modulus_sq_limbs();
    for (size_t i = 0; i < 24; i++) {
        cs.c(bs * cs.L(am + ADDITION_CHECK_OFFSET) * (cs.L(am + ADDITION_X_OFFSET + i) - cs.L(m0 + SUM_OFFSET + i)));
        cs.c(bs * cs.L(am + ADDITION_CHECK_OFFSET) * (cs.L(am + ADDITION_Y_OFFSET + i) - CS::K(p2[i])));
        cs.c(bs * cs.L(sb + SUBTRACTION_CHECK_OFFSET) * (cs.L(sb + SUBTRACTION_X_OFFSET + i) - cs.L(am + ADDITION_SUM_OFFSET + i)));
        cs.c(bs * cs.L(sb + SUBTRACTION_CHECK_OFFSET) * (cs.L(sb + SUBTRACTION_Y_OFFSET + i) - cs.L(m1 + SUM_OFFSET + i)));
        cs.c(bs * cs.L(ad + ADDITION_CHECK_OFFSET) * (cs.L(ad + ADDITION_X_OFFSET + i) - cs.L(m0 + SUM_OFFSET + i)));
        cs.c(bs * cs.L(ad + ADDITION_CHECK_OFFSET) * (cs.L(ad + ADDITION_Y_OFFSET + i) - cs.L(m1 + SUM_OFFSET + i)));
    }
    add_addition_constraints(cs, am, bs);
    add_subtraction_constraints(cs, sb, bs);
    add_addition_constraints(cs, ad, bs);
    cs.links(false, bs, 24, {{sb + SUBTRACTION_CHECK_OFFSET, z0 + REDUCE_X_OFFSET, sb + SUBTRACTION_DIFF_OFFSET},
                             {ad + ADDITION_CHECK_OFFSET, z1 + REDUCE_X_OFFSET, ad + ADDITION_SUM_OFFSET}});
    add_reduce_constraints(cs, z0, sel, bs);
    add_range_check_constraints(cs, sc + MULTIPLY_B_Z0_RANGECHECK_OFFSET, bs);
    add_reduce_constraints(cs, z1, sel, bs);
    add_range_check_constraints(cs, sc + MULTIPLY_B_Z1_RANGECHECK_OFFSET, bs);
}
void add_subtraction_with_reduction_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp2.rs:1499-1573
    const size_t a0 = sc + FP2_ADDITION_0_OFFSET, a1 = sc + FP2_ADDITION_1_OFFSET;
    const size_t s0 = sc + FP2_ADDITION_TOTAL + FP2_SUBTRACTION_0_OFFSET, s1 = sc + FP2_ADDITION_TOTAL + FP2_SUBTRACTION_1_OFFSET;
    const size_t red = sc + FP2_ADDITION_TOTAL + FP2_SUBTRACTION_TOTAL;
    for (size_t i = 0; i < 12; i++) {
        cs.c(bs * cs.L(a0 + FP_ADDITION_CHECK_OFFSET) * (cs.L(a0 + FP_ADDITION_Y_OFFSET + i) - CS::K(MODULUS[i])));
        cs.c(bs * cs.L(a1 + FP_ADDITION_CHECK_OFFSET) * (cs.L(a1 + FP_ADDITION_Y_OFFSET + i) - CS::K(MODULUS[i])));
    }
    add_addition_fp2_constraints(cs, sc, bs);
    cs.links(false, bs, 12, {{s0 + FP_SUBTRACTION_CHECK_OFFSET, s0 + FP_SUBTRACTION_X_OFFSET, a0 + FP_ADDITION_SUM_OFFSET},
                             {s1 + FP_SUBTRACTION_CHECK_OFFSET, s1 + FP_SUBTRACTION_X_OFFSET, a1 + FP_ADDITION_SUM_OFFSET}});
    add_subtraction_fp2_constraints(cs, sc + FP2_ADDITION_TOTAL, bs);
    cs.link(false, bs * cs.L(s0 + FP_SUBTRACTION_CHECK_OFFSET), s0 + FP_SUBTRACTION_DIFF_OFFSET, red + FP_SINGLE_REDUCE_X_OFFSET, 12);
    add_fp_reduce_single_constraints(cs, red, bs);
    add_range_check_constraints(cs, red + FP_SINGLE_REDUCE_TOTAL, bs);
    cs.link(false, bs * cs.L(s1 + FP_SUBTRACTION_CHECK_OFFSET), s1 + FP_SUBTRACTION_DIFF_OFFSET, red + RR + FP_SINGLE_REDUCE_X_OFFSET, 12);
    add_fp_reduce_single_constraints(cs, red + RR, bs);
    add_range_check_constraints(cs, red + RR + FP_SINGLE_REDUCE_TOTAL, bs);
}
void add_addition_with_reduction_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp2.rs:1635-1675
    const size_t a0 = sc + FP2_ADDITION_0_OFFSET, a1 = sc + FP2_ADDITION_1_OFFSET, red = sc + FP2_ADDITION_TOTAL;
    add_addition_fp2_constraints(cs, sc, bs);
    cs.link(false, bs * cs.L(a0 + FP_ADDITION_CHECK_OFFSET), a0 + FP_ADDITION_SUM_OFFSET, red + FP_SINGLE_REDUCE_X_OFFSET, 12);
    add_fp_reduce_single_constraints(cs, red, bs);
    add_range_check_constraints(cs, red + FP_SINGLE_REDUCE_TOTAL, bs);
    cs.link(false, bs * cs.L(a1 + FP_ADDITION_CHECK_OFFSET), a1 + FP_ADDITION_SUM_OFFSET, red + RR + FP_SINGLE_REDUCE_X_OFFSET, 12);
    add_fp_reduce_single_constraints(cs, red + RR, bs);
    add_range_check_constraints(cs, red + RR + FP_SINGLE_REDUCE_TOTAL, bs);
}
// fp2.rs:1713-1792.  The last link is gated by the column AFTER the c0+c1 addition block (first column of the
// following reduce block), exactly as the reference writes it (App. B.4 item 3).
void add_non_residue_multiplication_constraints(CS& cs, size_t sc, const Expr& bs) {
    const size_t in = sc + FP2_NON_RESIDUE_MUL_INPUT_OFFSET, ad = sc + FP2_NON_RESIDUE_MUL_C0_C1_SUB_OFFSET, sb = ad + FP_ADDITION_TOTAL;
    const size_t ad2 = sc + FP2_NON_RESIDUE_MUL_C0_C1_ADD_OFFSET;
    for (size_t i = 0; i < 12; i++) {
        cs.c(bs * cs.L(ad + FP_ADDITION_CHECK_OFFSET) * (cs.L(ad + FP_ADDITION_X_OFFSET + i) - cs.L(in + i)));
        cs.c(bs * cs.L(ad + FP_ADDITION_CHECK_OFFSET) * (cs.L(ad + FP_ADDITION_Y_OFFSET + i) - CS::K(MODULUS[i])));
    }
    add_addition_fp_constraints(cs, ad, bs);
    cs.links(false, bs, 12, {{sb + FP_SUBTRACTION_CHECK_OFFSET, sb + FP_SUBTRACTION_X_OFFSET, ad + FP_ADDITION_SUM_OFFSET},
                             {sb + FP_SUBTRACTION_CHECK_OFFSET, sb + FP_SUBTRACTION_Y_OFFSET, in + 12}});
    add_subtraction_fp_constraints(cs, sb, bs);
    cs.link(false, bs * cs.L(sb + FP_SUBTRACTION_CHECK_OFFSET), sb + FP_SUBTRACTION_DIFF_OFFSET, sc + FP2_NON_RESIDUE_MUL_Z0_REDUCE_OFFSET + FP_SINGLE_REDUCE_X_OFFSET, 12);
    add_fp_reduce_single_constraints(cs, sc + FP2_NON_RESIDUE_MUL_Z0_REDUCE_OFFSET, bs);
    add_range_check_constraints(cs, sc + FP2_NON_RESIDUE_MUL_Z0_RANGECHECK_OFFSET, bs);
    cs.links(false, bs, 12, {{ad2 + FP_ADDITION_CHECK_OFFSET, ad2 + FP_ADDITION_X_OFFSET, in}, {ad2 + FP_ADDITION_CHECK_OFFSET, ad2 + FP_ADDITION_Y_OFFSET, in + 12}});
    add_addition_fp_constraints(cs, ad2, bs);
    cs.link(false, bs * cs.L(ad2 + FP_ADDITION_TOTAL + FP_ADDITION_CHECK_OFFSET), ad2 + FP_ADDITION_SUM_OFFSET,
            sc + FP2_NON_RESIDUE_MUL_Z1_REDUCE_OFFSET + FP_SINGLE_REDUCE_X_OFFSET, 12);
    add_fp_reduce_single_constraints(cs, sc + FP2_NON_RESIDUE_MUL_Z1_REDUCE_OFFSET, bs);
    add_range_check_constraints(cs, sc + FP2_NON_RESIDUE_MUL_Z1_RANGECHECK_OFFSET, bs);
}
void add_fp4_sq_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp2.rs:1873-2088
    const size_t sel = sc + FP4_SQ_SELECTOR_OFFSET, X = sc + FP4_SQ_INPUT_X_OFFSET, Y = sc + FP4_SQ_INPUT_Y_OFFSET;
    const size_t t0 = sc + FP4_SQ_T0_CALC_OFFSET, t1 = sc + FP4_SQ_T1_CALC_OFFSET, t2 = sc + FP4_SQ_T2_CALC_OFFSET, xc = sc + FP4_SQ_X_CALC_OFFSET;
    const size_t t3 = sc + FP4_SQ_T3_CALC_OFFSET, t4 = sc + FP4_SQ_T4_CALC_OFFSET, t5 = sc + FP4_SQ_T5_CALC_OFFSET, yc = sc + FP4_SQ_Y_CALC_OFFSET;
    for (size_t i = 0; i < 24; i++) {
        cs.ct(bs * cs.L(sel) * (cs.L(X + i) - cs.N(X + i)));
        cs.ct(bs * cs.L(sel) * (cs.L(Y + i) - cs.N(Y + i)));
    }
    cs.links(false, bs, 24, {{t0 + FP2_FP2_SELECTOR_OFFSET, t0 + FP2_FP2_X_INPUT_OFFSET, X}, {t0 + FP2_FP2_SELECTOR_OFFSET, t0 + FP2_FP2_Y_INPUT_OFFSET, X}});
    add_fp2_mul_constraints(cs, t0, bs);
    cs.links(false, bs, 24, {{t1 + FP2_FP2_SELECTOR_OFFSET, t1 + FP2_FP2_X_INPUT_OFFSET, Y}, {t1 + FP2_FP2_SELECTOR_OFFSET, t1 + FP2_FP2_Y_INPUT_OFFSET, Y}});
    add_fp2_mul_constraints(cs, t1, bs);
    const size_t nrc = t2 + FP2_NON_RESIDUE_MUL_CHECK_OFFSET, nri = t2 + FP2_NON_RESIDUE_MUL_INPUT_OFFSET;
    cs.links(false, bs, 12, {{nrc, nri, t1 + Z1_REDUCE_OFFSET + REDUCED_OFFSET}, {nrc, nri + 12, t1 + Z2_REDUCE_OFFSET + REDUCED_OFFSET}});
    add_non_residue_multiplication_constraints(cs, t2, bs);
    {
        const size_t a0 = xc + FP2_ADDITION_0_OFFSET, a1 = xc + FP2_ADDITION_1_OFFSET;
        cs.links(false, bs, 12, {{a0 + FP_ADDITION_CHECK_OFFSET, a0 + FP_ADDITION_X_OFFSET, t2 + FP2_NON_RESIDUE_MUL_Z0_REDUCE_OFFSET + FP_SINGLE_REDUCED_OFFSET},
                                 {a0 + FP_ADDITION_CHECK_OFFSET, a0 + FP_ADDITION_Y_OFFSET, t0 + Z1_REDUCE_OFFSET + REDUCED_OFFSET},
                                 {a1 + FP_ADDITION_CHECK_OFFSET, a1 + FP_ADDITION_X_OFFSET, t2 + FP2_NON_RESIDUE_MUL_Z1_REDUCE_OFFSET + FP_SINGLE_REDUCED_OFFSET},
                                 {a1 + FP_ADDITION_CHECK_OFFSET, a1 + FP_ADDITION_Y_OFFSET, t0 + Z2_REDUCE_OFFSET + REDUCED_OFFSET}});
    }
    add_addition_with_reduction_constraints(cs, xc, bs);
    {
        const size_t a0 = t3 + FP2_ADDITION_0_OFFSET, a1 = t3 + FP2_ADDITION_1_OFFSET;
        cs.links(false, bs, 12, {{a0 + FP_ADDITION_CHECK_OFFSET, a0 + FP_ADDITION_X_OFFSET, X}, {a0 + FP_ADDITION_CHECK_OFFSET, a0 + FP_ADDITION_Y_OFFSET, Y},
                                 {a1 + FP_ADDITION_CHECK_OFFSET, a1 + FP_ADDITION_X_OFFSET, X + 12}, {a1 + FP_ADDITION_CHECK_OFFSET, a1 + FP_ADDITION_Y_OFFSET, Y + 12}});
    }
    add_addition_with_reduction_constraints(cs, t3, bs);
    {
        const size_t s4 = t4 + FP2_FP2_SELECTOR_OFFSET, r0 = t3 + FP2_ADDITION_TOTAL + FP_SINGLE_REDUCED_OFFSET, r1 = t3 + FP2_ADDITION_TOTAL + RR + FP_SINGLE_REDUCED_OFFSET;
        cs.links(false, bs, 12, {{s4, t4 + FP2_FP2_X_INPUT_OFFSET, r0}, {s4, t4 + FP2_FP2_X_INPUT_OFFSET + 12, r1},
                                 {s4, t4 + FP2_FP2_Y_INPUT_OFFSET, r0}, {s4, t4 + FP2_FP2_Y_INPUT_OFFSET + 12, r1}});
    }
    add_fp2_mul_constraints(cs, t4, bs);
    {
        const size_t a0 = t5 + FP2_ADDITION_0_OFFSET, a1 = t5 + FP2_ADDITION_1_OFFSET;
        const size_t s0 = t5 + FP2_ADDITION_TOTAL + FP2_SUBTRACTION_0_OFFSET, s1 = t5 + FP2_ADDITION_TOTAL + FP2_SUBTRACTION_1_OFFSET;
        cs.links(false, bs, 12, {{a0 + FP_ADDITION_CHECK_OFFSET, a0 + FP_ADDITION_X_OFFSET, t4 + Z1_REDUCE_OFFSET + REDUCED_OFFSET},
                                 {s0 + FP_SUBTRACTION_CHECK_OFFSET, s0 + FP_SUBTRACTION_Y_OFFSET, t0 + Z1_REDUCE_OFFSET + REDUCED_OFFSET},
                                 {a1 + FP_ADDITION_CHECK_OFFSET, a1 + FP_ADDITION_X_OFFSET, t4 + Z2_REDUCE_OFFSET + REDUCED_OFFSET},
                                 {s1 + FP_SUBTRACTION_CHECK_OFFSET, s1 + FP_SUBTRACTION_Y_OFFSET, t0 + Z2_REDUCE_OFFSET + REDUCED_OFFSET}});
    }
    add_subtraction_with_reduction_constraints(cs, t5, bs);
    {
        const size_t a0 = yc + FP2_ADDITION_0_OFFSET, a1 = yc + FP2_ADDITION_1_OFFSET;
        const size_t s0 = yc + FP2_ADDITION_TOTAL + FP2_SUBTRACTION_0_OFFSET, s1 = yc + FP2_ADDITION_TOTAL + FP2_SUBTRACTION_1_OFFSET;
        const size_t r5 = t5 + FP2_ADDITION_TOTAL + FP2_SUBTRACTION_TOTAL;
        cs.links(false, bs, 12, {{a0 + FP_ADDITION_CHECK_OFFSET, a0 + FP_ADDITION_X_OFFSET, r5 + FP_SINGLE_REDUCED_OFFSET},
                                 {s0 + FP_SUBTRACTION_CHECK_OFFSET, s0 + FP_SUBTRACTION_Y_OFFSET, t1 + Z1_REDUCE_OFFSET + REDUCED_OFFSET},
                                 {a1 + FP_ADDITION_CHECK_OFFSET, a1 + FP_ADDITION_X_OFFSET, r5 + RR + FP_SINGLE_REDUCED_OFFSET},
                                 {s1 + FP_SUBTRACTION_CHECK_OFFSET, s1 + FP_SUBTRACTION_Y_OFFSET, t1 + Z2_REDUCE_OFFSET + REDUCED_OFFSET}});
    }
    add_subtraction_with_reduction_constraints(cs, yc, bs);
}
void add_fp2_forbenius_map_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp2.rs:2271-2337
    const size_t sel = sc + FP2_FORBENIUS_MAP_SELECTOR_OFFSET, in = sc + FP2_FORBENIUS_MAP_INPUT_OFFSET, t0 = sc + FP2_FORBENIUS_MAP_T0_CALC_OFFSET;
    const size_t powc = sc + FP2_FORBENIUS_MAP_POW_OFFSET, divc = sc + FP2_FORBENIUS_MAP_DIV_OFFSET, remc = sc + FP2_FORBENIUS_MAP_REM_OFFSET;
    cs.keep(true, bs * cs.L(sel), in, 24);
    cs.ct(bs * cs.L(sel) * (cs.L(powc) - cs.N(powc)));
    cs.c(bs * cs.L(sel) * (cs.L(divc) * CS::K(2) + cs.L(remc) - cs.L(powc)));
    const Expr bit = cs.L(remc);
    for (size_t i = 0; i < 12; i++) {
        Expr y = (CS::one() - bit) * CS::K(FP2_FROBENIUS_COEFF[0].l[i]) + bit * CS::K(FP2_FROBENIUS_COEFF[1].l[i]);
        cs.c(bs * cs.L(t0 + MULTIPLICATION_SELECTOR_OFFSET) * (cs.L(t0 + X_INPUT_OFFSET + i) - cs.L(in + 12 + i)));
        cs.c(bs * cs.L(t0 + MULTIPLICATION_SELECTOR_OFFSET) * (cs.L(t0 + Y_INPUT_OFFSET + i) - y));
    }
    add_multiplication_constraints(cs, t0, bs);
    const size_t red = t0 + FP_MULTIPLICATION_TOTAL_COLUMNS;
    cs.link(false, bs * cs.L(sc + FP2_FORBENIUS_MAP_MUL_RES_ROW), t0 + SUM_OFFSET, red + REDUCE_X_OFFSET, 24);
    add_reduce_constraints(cs, red, sel, bs);
    add_range_check_constraints(cs, red + REDUCTION_TOTAL, bs);
}

}  // namespace starkhip
