// Fp6 gadgets: trace fillers and constraint emitters.  Restates the fill_* / packed add_*_constraints
// halves of /root/reference/src/fp6.rs (line references on each function).
#include "gadgets.h"
#include "wiring.h"

namespace starkhip {
using namespace lay;
using namespace bls;
using namespace wire;

static L12 add12(const L12& a, const L12& b) {
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
void fill_trace_addition_fp6(Trace& t, const Fp6& x, const Fp6& y, size_t row, size_t col) {  // fp6.rs:124-131
    fill_trace_addition_fp2(t, x.c2(0), y.c2(0), row, col + FP6_ADDITION_0_OFFSET);
    fill_trace_addition_fp2(t, x.c2(1), y.c2(1), row, col + FP6_ADDITION_1_OFFSET);
    fill_trace_addition_fp2(t, x.c2(2), y.c2(2), row, col + FP6_ADDITION_2_OFFSET);
}
void fill_trace_addition_with_reduction_fp6(Trace& t, const Fp6& x, const Fp6& y, size_t row, size_t col) {  // fp6.rs:134-146
    fill_trace_addition_fp6(t, x, y, row, col);
    for (size_t i = 0; i < 6; i++) {
        L12 rem = fill_trace_reduce_single(t, add12(x.c[i].l, y.c[i].l), row, col + FP6_ADDITION_TOTAL + RR * i);
        fill_range_check_trace(t, rem, row, col + FP6_ADDITION_TOTAL + RR * i + FP_SINGLE_REDUCE_TOTAL);
    }
}
void fill_trace_subtraction_fp6(Trace& t, const Fp6& x, const Fp6& y, size_t row, size_t col) {  // fp6.rs:175-182
    fill_trace_subtraction_fp2(t, x.c2(0), y.c2(0), row, col + FP6_SUBTRACTION_0_OFFSET);
    fill_trace_subtraction_fp2(t, x.c2(1), y.c2(1), row, col + FP6_SUBTRACTION_1_OFFSET);
    fill_trace_subtraction_fp2(t, x.c2(2), y.c2(2), row, col + FP6_SUBTRACTION_2_OFFSET);
}
void fill_trace_subtraction_with_reduction_fp6(Trace& t, const Fp6& x, const Fp6& y, size_t row, size_t col) {  // fp6.rs:149-172
    Fp6 pp, xm;
    for (int i = 0; i < 6; i++) {
        pp.c[i] = Fp(MODULUS);
        xm.c[i] = Fp(add12(MODULUS, x.c[i].l));
    }
    fill_trace_addition_fp6(t, x, pp, row, col);
    fill_trace_subtraction_fp6(t, xm, y, row, col + FP6_ADDITION_TOTAL);
    for (size_t i = 0; i < 6; i++) {
        const size_t base = col + FP6_ADDITION_TOTAL + FP6_SUBTRACTION_TOTAL + RR * i;
        L12 rem = fill_trace_reduce_single(t, sub12(xm.c[i].l, y.c[i].l), row, base);
        fill_range_check_trace(t, rem, row, base + FP_SINGLE_REDUCE_TOTAL);
    }
}
void fill_trace_negate_fp6(Trace& t, const Fp6& x, size_t row, size_t col) { fill_trace_addition_fp6(t, x, -x, row, col); }  // fp6.rs:185-195
void fill_trace_non_residue_multiplication_fp6(Trace& t, const Fp6& x, size_t row, size_t col) {  // fp6.rs:198-208
    t.at(row, col + FP6_NON_RESIDUE_MUL_CHECK_OFFSET) = 1;
    t.put(row, col + FP6_NON_RESIDUE_MUL_INPUT_OFFSET, x);
    fill_trace_non_residue_multiplication(t, x.c2(2), row, col + FP6_NON_RESIDUE_MUL_C2);
}

// per-row replicated single-row gadgets
static void rows_addred(Trace& t, const Fp2& a, const Fp2& b, size_t r0, size_t r1, size_t col) {
    { RowSpan rows_(t, r1 - r0 + 1); fill_trace_addition_with_reduction(t, a, b, r0, col); }
}
static void rows_subred(Trace& t, const Fp2& a, const Fp2& b, size_t r0, size_t r1, size_t col) {
    { RowSpan rows_(t, r1 - r0 + 1); fill_trace_subtraction_with_reduction(t, a, b, r0, col); }
}
static void rows_nr(Trace& t, const Fp2& a, size_t r0, size_t r1, size_t col) {
    { RowSpan rows_(t, r1 - r0 + 1); fill_trace_non_residue_multiplication(t, a, r0, col); }
}

void fill_trace_fp6_multiplication(Trace& t, const Fp6& x, const Fp6& y, size_t r0_, size_t r1_, size_t col) {  // fp6.rs:211-309
    {
        RowSpan rows_(t, r1_ - r0_ + 1);
        t.put(r0_, col + FP6_MUL_X_INPUT_OFFSET, x);
        t.put(r0_, col + FP6_MUL_Y_INPUT_OFFSET, y);
        t.at(r0_, col + FP6_MUL_SELECTOR_OFFSET) = 1;
    }
    t.at(r1_, col + FP6_MUL_SELECTOR_OFFSET) = 0;
    const Fp2 c0 = x.c2(0), c1 = x.c2(1), c2 = x.c2(2), r0 = y.c2(0), r1 = y.c2(1), r2 = y.c2(2);
    Fp2 t0 = c0 * r0;
    generate_trace_fp2_mul(t, c0, r0, r0_, r1_, col + FP6_MUL_T0_CALC_OFFSET);
    Fp2 t1 = c1 * r1;
    generate_trace_fp2_mul(t, c1, r1, r0_, r1_, col + FP6_MUL_T1_CALC_OFFSET);
    Fp2 t2 = c2 * r2;
    generate_trace_fp2_mul(t, c2, r2, r0_, r1_, col + FP6_MUL_T2_CALC_OFFSET);
    Fp2 t3 = c1 + c2;
    rows_addred(t, c1, c2, r0_, r1_, col + FP6_MUL_T3_CALC_OFFSET);
    Fp2 t4 = r1 + r2;
    rows_addred(t, r1, r2, r0_, r1_, col + FP6_MUL_T4_CALC_OFFSET);
    Fp2 t5 = t3 * t4;
    generate_trace_fp2_mul(t, t3, t4, r0_, r1_, col + FP6_MUL_T5_CALC_OFFSET);
    Fp2 t6 = t5 - t1;
    rows_subred(t, t5, t1, r0_, r1_, col + FP6_MUL_T6_CALC_OFFSET);
    Fp2 t7 = t6 - t2;
    rows_subred(t, t6, t2, r0_, r1_, col + FP6_MUL_T7_CALC_OFFSET);
    Fp2 t8 = t7.mul_by_nonresidue();
    rows_nr(t, t7, r0_, r1_, col + FP6_MUL_T8_CALC_OFFSET);
    rows_addred(t, t8, t0, r0_, r1_, col + FP6_MUL_X_CALC_OFFSET);
    Fp2 t9 = c0 + c1;
    rows_addred(t, c0, c1, r0_, r1_, col + FP6_MUL_T9_CALC_OFFSET);
    Fp2 t10 = r0 + r1;
    rows_addred(t, r0, r1, r0_, r1_, col + FP6_MUL_T10_CALC_OFFSET);
    Fp2 t11 = t9 * t10;
    generate_trace_fp2_mul(t, t9, t10, r0_, r1_, col + FP6_MUL_T11_CALC_OFFSET);
    Fp2 t12 = t11 - t0;
    rows_subred(t, t11, t0, r0_, r1_, col + FP6_MUL_T12_CALC_OFFSET);
    Fp2 t13 = t12 - t1;
    rows_subred(t, t12, t1, r0_, r1_, col + FP6_MUL_T13_CALC_OFFSET);
    Fp2 t14 = t2.mul_by_nonresidue();
    rows_nr(t, t2, r0_, r1_, col + FP6_MUL_T14_CALC_OFFSET);
    rows_addred(t, t13, t14, r0_, r1_, col + FP6_MUL_Y_CALC_OFFSET);
    Fp2 t15 = c0 + c2;
    rows_addred(t, c0, c2, r0_, r1_, col + FP6_MUL_T15_CALC_OFFSET);
    Fp2 t16 = r0 + r2;
    rows_addred(t, r0, r2, r0_, r1_, col + FP6_MUL_T16_CALC_OFFSET);
    Fp2 t17 = t15 * t16;
    generate_trace_fp2_mul(t, t15, t16, r0_, r1_, col + FP6_MUL_T17_CALC_OFFSET);
    Fp2 t18 = t17 - t0;
    rows_subred(t, t17, t0, r0_, r1_, col + FP6_MUL_T18_CALC_OFFSET);
    Fp2 t19 = t18 - t2;
    rows_subred(t, t18, t2, r0_, r1_, col + FP6_MUL_T19_CALC_OFFSET);
    rows_addred(t, t19, t1, r0_, r1_, col + FP6_MUL_Z_CALC_OFFSET);
}
void fill_trace_multiply_by_1(Trace& t, const Fp6& x, const Fp2& b1, size_t r0, size_t r1, size_t col) {  // fp6.rs:312-340
    {
        RowSpan rows_(t, r1 - r0 + 1);
        t.put(r0, col + MULTIPLY_BY_1_INPUT_OFFSET, x);
        t.put(r0, col + MULTIPLY_BY_1_B1_OFFSET, b1);
        t.at(r0, col + MULTIPLY_BY_1_SELECTOR_OFFSET) = 1;
    }
    t.at(r1, col + MULTIPLY_BY_1_SELECTOR_OFFSET) = 0;
    const Fp2 c0 = x.c2(0), c1 = x.c2(1), c2 = x.c2(2);
    Fp2 t0 = c2 * b1;
    generate_trace_fp2_mul(t, c2, b1, r0, r1, col + MULTIPLY_BY_1_T0_CALC_OFFSET);
    rows_nr(t, t0, r0, r1, col + MULTIPLY_BY_1_X_CALC_OFFSET);
    generate_trace_fp2_mul(t, c0, b1, r0, r1, col + MULTIPLY_BY_1_Y_CALC_OFFSET);
    generate_trace_fp2_mul(t, c1, b1, r0, r1, col + MULTIPLY_BY_1_Z_CALC_OFFSET);
}
void fill_trace_multiply_by_01(Trace& t, const Fp6& x, const Fp2& b0, const Fp2& b1, size_t r0, size_t r1, size_t col) {  // fp6.rs:343-406
    {
        RowSpan rows_(t, r1 - r0 + 1);
        t.put(r0, col + MULTIPLY_BY_01_INPUT_OFFSET, x);
        t.put(r0, col + MULTIPLY_BY_01_B0_OFFSET, b0);
        t.put(r0, col + MULTIPLY_BY_01_B1_OFFSET, b1);
        t.at(r0, col + MULTIPLY_BY_01_SELECTOR_OFFSET) = 1;
    }
    t.at(r1, col + MULTIPLY_BY_01_SELECTOR_OFFSET) = 0;
    const Fp2 c0 = x.c2(0), c1 = x.c2(1), c2 = x.c2(2);
    Fp2 t0 = c0 * b0;
    generate_trace_fp2_mul(t, c0, b0, r0, r1, col + MULTIPLY_BY_01_T0_CALC_OFFSET);
    Fp2 t1 = c1 * b1;
    generate_trace_fp2_mul(t, c1, b1, r0, r1, col + MULTIPLY_BY_01_T1_CALC_OFFSET);
    Fp2 t2 = c2 * b1;
    generate_trace_fp2_mul(t, c2, b1, r0, r1, col + MULTIPLY_BY_01_T2_CALC_OFFSET);
    Fp2 t3 = t2.mul_by_nonresidue();
    rows_nr(t, t2, r0, r1, col + MULTIPLY_BY_01_T3_CALC_OFFSET);
    rows_addred(t, t3, t0, r0, r1, col + MULTIPLY_BY_01_X_CALC_OFFSET);
    Fp2 t4 = b0 + b1;
    rows_addred(t, b0, b1, r0, r1, col + MULTIPLY_BY_01_T4_CALC_OFFSET);
    Fp2 t5 = c0 + c1;
    rows_addred(t, c0, c1, r0, r1, col + MULTIPLY_BY_01_T5_CALC_OFFSET);
    Fp2 t6 = t4 * t5;
    generate_trace_fp2_mul(t, t4, t5, r0, r1, col + MULTIPLY_BY_01_T6_CALC_OFFSET);
    Fp2 t7 = t6 - t0;
    rows_subred(t, t6, t0, r0, r1, col + MULTIPLY_BY_01_T7_CALC_OFFSET);
    rows_subred(t, t7, t1, r0, r1, col + MULTIPLY_BY_01_Y_CALC_OFFSET);
    Fp2 t8 = c2 * b0;
    generate_trace_fp2_mul(t, c2, b0, r0, r1, col + MULTIPLY_BY_01_T8_CALC_OFFSET);
    rows_addred(t, t8, t1, r0, r1, col + MULTIPLY_BY_01_Z_CALC_OFFSET);
}
void fill_trace_fp6_forbenius_map(Trace& t, const Fp6& x, size_t pow, size_t r0, size_t r1, size_t col) {  // fp6.rs:409-441
    const size_t div = pow / 6, rem = pow % 6;
    {
        RowSpan rows_(t, r1 - r0 + 1);
        t.put(r0, col + FP6_FORBENIUS_MAP_INPUT_OFFSET, x);
        t.at(r0, col + FP6_FORBENIUS_MAP_SELECTOR_OFFSET) = 1;
        t.at(r0, col + FP6_FORBENIUS_MAP_POW_OFFSET) = pow;
        t.at(r0, col + FP6_FORBENIUS_MAP_DIV_OFFSET) = div;
        t.at(r0, col + FP6_FORBENIUS_MAP_REM_OFFSET) = rem;
        t.at(r0, col + FP6_FORBENIUS_MAP_BIT0_OFFSET) = rem & 1;
        t.at(r0, col + FP6_FORBENIUS_MAP_BIT1_OFFSET) = (rem >> 1) & 1;
        t.at(r0, col + FP6_FORBENIUS_MAP_BIT2_OFFSET) = rem >> 2;
    }
    t.at(r1, col + FP6_FORBENIUS_MAP_SELECTOR_OFFSET) = 0;
    const Fp2 c0 = x.c2(0), c1 = x.c2(1), c2 = x.c2(2);
    fill_trace_fp2_forbenius_map(t, c0, pow, r0, r1, col + FP6_FORBENIUS_MAP_X_CALC_OFFSET);
    Fp2 t0 = c1.forbenius_map(pow);
    fill_trace_fp2_forbenius_map(t, c1, pow, r0, r1, col + FP6_FORBENIUS_MAP_T0_CALC_OFFSET);
    generate_trace_fp2_mul(t, t0, fp6_frobenius_coeff_1()[pow % 6], r0, r1, col + FP6_FORBENIUS_MAP_Y_CALC_OFFSET);
    Fp2 t1 = c2.forbenius_map(pow);
    fill_trace_fp2_forbenius_map(t, c2, pow, r0, r1, col + FP6_FORBENIUS_MAP_T1_CALC_OFFSET);
    generate_trace_fp2_mul(t, t1, fp6_frobenius_coeff_2()[pow % 6], r0, r1, col + FP6_FORBENIUS_MAP_Z_CALC_OFFSET);
}

// ------------------------------------------------------------------ constraints
void add_addition_fp6_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp6.rs:444-460
    add_addition_fp2_constraints(cs, sc + FP6_ADDITION_0_OFFSET, bs);
    add_addition_fp2_constraints(cs, sc + FP6_ADDITION_1_OFFSET, bs);
    add_addition_fp2_constraints(cs, sc + FP6_ADDITION_2_OFFSET, bs);
}
static size_t fp6_add_fp_block(size_t j) {  // j-th Fp addition block inside an Fp6 addition
    static const size_t fp2o[3] = {FP6_ADDITION_0_OFFSET, FP6_ADDITION_1_OFFSET, FP6_ADDITION_2_OFFSET};
    return fp2o[j / 2] + (j % 2 == 0 ? FP2_ADDITION_0_OFFSET : FP2_ADDITION_1_OFFSET);
}
void add_addition_with_reduction_constraints_fp6(CS& cs, size_t sc, const Expr& bs) {  // fp6.rs:480-523
    add_addition_fp6_constraints(cs, sc, bs);
    for (size_t j = 0; j < 6; j++) {
        const size_t a = sc + fp6_add_fp_block(j), red = sc + FP6_ADDITION_TOTAL + RR * j;
        cs.link(false, bs * cs.L(a + FP_ADDITION_CHECK_OFFSET), a + FP_ADDITION_SUM_OFFSET, red + FP_SINGLE_REDUCE_X_OFFSET, 12);
        add_fp_reduce_single_constraints(cs, red, bs);
        add_range_check_constraints(cs, red + FP_SINGLE_REDUCE_TOTAL, bs);
    }
}
void add_subtraction_fp6_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp6.rs:564-580
    add_subtraction_fp2_constraints(cs, sc + FP6_SUBTRACTION_0_OFFSET, bs);
    add_subtraction_fp2_constraints(cs, sc + FP6_SUBTRACTION_1_OFFSET, bs);
    add_subtraction_fp2_constraints(cs, sc + FP6_SUBTRACTION_2_OFFSET, bs);
}
void add_negate_fp6_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp6.rs:600-643
    add_addition_fp6_constraints(cs, sc, bs);
    for (size_t i = 0; i < 12; i++)
        for (size_t j = 0; j < 6; j++) {  // j = 2 * (fp2 index) + (fp index), same nesting order as the reference
            const size_t a = sc + fp6_add_fp_block(j);
            cs.c(bs * cs.L(a + FP_ADDITION_CHECK_OFFSET) * (cs.L(a + FP_ADDITION_SUM_OFFSET + i) - CS::K(MODULUS[i])));
        }
}
// fp6.rs:687-765.  For every j the "Y == p" and "X == sum" links are emitted for BOTH Fp halves of the enclosing
// Fp2 block, i.e. each of them appears twice overall (App. B.4 item 4).
void add_subtraction_with_reduction_constraints_fp6(CS& cs, size_t sc, const Expr& bs) {
    static const size_t fp2a[3] = {FP6_ADDITION_0_OFFSET, FP6_ADDITION_1_OFFSET, FP6_ADDITION_2_OFFSET};
    static const size_t fp2s[3] = {FP6_SUBTRACTION_0_OFFSET, FP6_SUBTRACTION_1_OFFSET, FP6_SUBTRACTION_2_OFFSET};
    add_addition_fp6_constraints(cs, sc, bs);
    add_subtraction_fp6_constraints(cs, sc + FP6_ADDITION_TOTAL, bs);
    for (size_t j = 0; j < 6; j++) {
        const size_t a0 = sc + fp2a[j / 2] + FP2_ADDITION_0_OFFSET, a1 = sc + fp2a[j / 2] + FP2_ADDITION_1_OFFSET;
        const size_t s0 = sc + FP6_ADDITION_TOTAL + fp2s[j / 2] + FP2_SUBTRACTION_0_OFFSET, s1 = sc + FP6_ADDITION_TOTAL + fp2s[j / 2] + FP2_SUBTRACTION_1_OFFSET;
        const size_t sj = (j % 2 == 0) ? s0 : s1;
        const size_t red = sc + FP6_ADDITION_TOTAL + FP6_SUBTRACTION_TOTAL + RR * j;
        for (size_t i = 0; i < 12; i++) {
            cs.c(bs * cs.L(a0 + FP_ADDITION_CHECK_OFFSET) * (cs.L(a0 + FP_ADDITION_Y_OFFSET + i) - CS::K(MODULUS[i])));
            cs.c(bs * cs.L(a1 + FP_ADDITION_CHECK_OFFSET) * (cs.L(a1 + FP_ADDITION_Y_OFFSET + i) - CS::K(MODULUS[i])));
        }
        cs.links(false, bs, 12, {{s0 + FP_SUBTRACTION_CHECK_OFFSET, s0 + FP_SUBTRACTION_X_OFFSET, a0 + FP_ADDITION_SUM_OFFSET},
                                 {s1 + FP_SUBTRACTION_CHECK_OFFSET, s1 + FP_SUBTRACTION_X_OFFSET, a1 + FP_ADDITION_SUM_OFFSET}});
        cs.link(false, bs * cs.L(sj + FP_SUBTRACTION_CHECK_OFFSET), sj + FP_SUBTRACTION_DIFF_OFFSET, red + FP_SINGLE_REDUCE_X_OFFSET, 12);
        add_fp_reduce_single_constraints(cs, red, bs);
        add_range_check_constraints(cs, red + FP_SINGLE_REDUCE_TOTAL, bs);
    }
}
void add_non_residue_multiplication_fp6_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp6.rs:832-856
    cs.link(false, bs * cs.L(sc + FP6_NON_RESIDUE_MUL_CHECK_OFFSET), sc + FP6_NON_RESIDUE_MUL_INPUT_OFFSET + 48,
            sc + FP6_NON_RESIDUE_MUL_C2 + FP2_NON_RESIDUE_MUL_INPUT_OFFSET, 24);
    add_non_residue_multiplication_constraints(cs, sc + FP6_NON_RESIDUE_MUL_C2, bs);
}
void add_fp6_multiplication_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp6.rs:882-1553
    const size_t sel = sc + FP6_MUL_SELECTOR_OFFSET, X = sc + FP6_MUL_X_INPUT_OFFSET, Y = sc + FP6_MUL_Y_INPUT_OFFSET;
    auto T = [&](size_t off) { return sc + off; };
    const size_t t0 = T(FP6_MUL_T0_CALC_OFFSET), t1 = T(FP6_MUL_T1_CALC_OFFSET), t2 = T(FP6_MUL_T2_CALC_OFFSET), t3 = T(FP6_MUL_T3_CALC_OFFSET);
    const size_t t4 = T(FP6_MUL_T4_CALC_OFFSET), t5 = T(FP6_MUL_T5_CALC_OFFSET), t6 = T(FP6_MUL_T6_CALC_OFFSET), t7 = T(FP6_MUL_T7_CALC_OFFSET);
    const size_t t8 = T(FP6_MUL_T8_CALC_OFFSET), xc = T(FP6_MUL_X_CALC_OFFSET), t9 = T(FP6_MUL_T9_CALC_OFFSET), t10 = T(FP6_MUL_T10_CALC_OFFSET);
    const size_t t11 = T(FP6_MUL_T11_CALC_OFFSET), t12 = T(FP6_MUL_T12_CALC_OFFSET), t13 = T(FP6_MUL_T13_CALC_OFFSET), t14 = T(FP6_MUL_T14_CALC_OFFSET);
    const size_t yc = T(FP6_MUL_Y_CALC_OFFSET), t15 = T(FP6_MUL_T15_CALC_OFFSET), t16 = T(FP6_MUL_T16_CALC_OFFSET), t17 = T(FP6_MUL_T17_CALC_OFFSET);
    const size_t t18 = T(FP6_MUL_T18_CALC_OFFSET), t19 = T(FP6_MUL_T19_CALC_OFFSET), zc = T(FP6_MUL_Z_CALC_OFFSET);
    for (size_t i = 0; i < 72; i++) {
        cs.ct(bs * cs.L(sel) * (cs.L(X + i) - cs.N(X + i)));
        cs.ct(bs * cs.L(sel) * (cs.L(Y + i) - cs.N(Y + i)));
    }
    mul_in24(cs, bs, t0, X, Y, false);
    add_fp2_mul_constraints(cs, t0, bs);
    mul_in24(cs, bs, t1, X + 24, Y + 24, false);
    add_fp2_mul_constraints(cs, t1, bs);
    mul_in24(cs, bs, t2, X + 48, Y + 48, false);
    add_fp2_mul_constraints(cs, t2, bs);
    add_in(cs, bs, t3, raw(X + 24), raw(X + 48));
    add_addition_with_reduction_constraints(cs, t3, bs);
    add_in(cs, bs, t4, raw(Y + 24), raw(Y + 48));
    add_addition_with_reduction_constraints(cs, t4, bs);
    mul_in(cs, bs, t5, addred_out(t3), addred_out(t4), false);
    add_fp2_mul_constraints(cs, t5, bs);
    sub_in(cs, bs, t6, mul_out(t5), mul_out(t1));
    add_subtraction_with_reduction_constraints(cs, t6, bs);
    sub_in(cs, bs, t7, subred_out(t6), mul_out(t2));
    add_subtraction_with_reduction_constraints(cs, t7, bs);
    nr_in(cs, bs, t8, subred_out(t7));
    add_non_residue_multiplication_constraints(cs, t8, bs);
    add_in(cs, bs, xc, nr_out(t8), mul_out(t0));
    add_addition_with_reduction_constraints(cs, xc, bs);
    add_in(cs, bs, t9, raw(X), raw(X + 24));
    add_addition_with_reduction_constraints(cs, t9, bs);
    add_in(cs, bs, t10, raw(Y), raw(Y + 24));
    add_addition_with_reduction_constraints(cs, t10, bs);
    mul_in(cs, bs, t11, addred_out(t9), addred_out(t10), false);
    add_fp2_mul_constraints(cs, t11, bs);
    sub_in(cs, bs, t12, mul_out(t11), mul_out(t0));
    add_subtraction_with_reduction_constraints(cs, t12, bs);
    sub_in(cs, bs, t13, subred_out(t12), mul_out(t1));
    add_subtraction_with_reduction_constraints(cs, t13, bs);
    nr_in(cs, bs, t14, mul_out(t2));
    add_non_residue_multiplication_constraints(cs, t14, bs);
    add_in(cs, bs, yc, subred_out(t13), nr_out(t14));
    add_addition_with_reduction_constraints(cs, yc, bs);
    add_in(cs, bs, t15, raw(X), raw(X + 48));
    add_addition_with_reduction_constraints(cs, t15, bs);
    add_in(cs, bs, t16, raw(Y), raw(Y + 48));
    add_addition_with_reduction_constraints(cs, t16, bs);
    mul_in(cs, bs, t17, addred_out(t15), addred_out(t16), false);
    add_fp2_mul_constraints(cs, t17, bs);
    sub_in(cs, bs, t18, mul_out(t17), mul_out(t0));
    add_subtraction_with_reduction_constraints(cs, t18, bs);
    sub_in(cs, bs, t19, subred_out(t18), mul_out(t2));
    add_subtraction_with_reduction_constraints(cs, t19, bs);
    add_in(cs, bs, zc, subred_out(t19), mul_out(t1));
    add_addition_with_reduction_constraints(cs, zc, bs);
}
void add_multiply_by_1_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp6.rs:2122-2218
    const size_t sel = sc + MULTIPLY_BY_1_SELECTOR_OFFSET, in = sc + MULTIPLY_BY_1_INPUT_OFFSET, b1 = sc + MULTIPLY_BY_1_B1_OFFSET;
    const size_t t0 = sc + MULTIPLY_BY_1_T0_CALC_OFFSET, xc = sc + MULTIPLY_BY_1_X_CALC_OFFSET, yc = sc + MULTIPLY_BY_1_Y_CALC_OFFSET, zc = sc + MULTIPLY_BY_1_Z_CALC_OFFSET;
    for (size_t i = 0; i < 24; i++) {
        for (size_t j = 0; j < 3; j++) cs.ct(bs * cs.L(sel) * (cs.L(in + j * 24 + i) - cs.N(in + j * 24 + i)));
        cs.ct(bs * cs.L(sel) * (cs.L(b1 + i) - cs.N(b1 + i)));
    }
    mul_in24(cs, bs, t0, in + 48, b1, true);
    add_fp2_mul_constraints(cs, t0, bs);
    nr_in(cs, bs, xc, mul_out(t0));
    add_non_residue_multiplication_constraints(cs, xc, bs);
    mul_in24(cs, bs, yc, in, b1, true);
    add_fp2_mul_constraints(cs, yc, bs);
    mul_in24(cs, bs, zc, in + 24, b1, true);
    add_fp2_mul_constraints(cs, zc, bs);
}
void add_multiply_by_01_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp6.rs:2314-2635
    const size_t sel = sc + MULTIPLY_BY_01_SELECTOR_OFFSET, in = sc + MULTIPLY_BY_01_INPUT_OFFSET, b0 = sc + MULTIPLY_BY_01_B0_OFFSET, b1 = sc + MULTIPLY_BY_01_B1_OFFSET;
    const size_t t0 = sc + MULTIPLY_BY_01_T0_CALC_OFFSET, t1 = sc + MULTIPLY_BY_01_T1_CALC_OFFSET, t2 = sc + MULTIPLY_BY_01_T2_CALC_OFFSET, t3 = sc + MULTIPLY_BY_01_T3_CALC_OFFSET;
    const size_t xc = sc + MULTIPLY_BY_01_X_CALC_OFFSET, t4 = sc + MULTIPLY_BY_01_T4_CALC_OFFSET, t5 = sc + MULTIPLY_BY_01_T5_CALC_OFFSET, t6 = sc + MULTIPLY_BY_01_T6_CALC_OFFSET;
    const size_t t7 = sc + MULTIPLY_BY_01_T7_CALC_OFFSET, yc = sc + MULTIPLY_BY_01_Y_CALC_OFFSET, t8 = sc + MULTIPLY_BY_01_T8_CALC_OFFSET, zc = sc + MULTIPLY_BY_01_Z_CALC_OFFSET;
    for (size_t i = 0; i < 24; i++) {
        for (size_t j = 0; j < 3; j++) cs.ct(bs * cs.L(sel) * (cs.L(in + j * 24 + i) - cs.N(in + j * 24 + i)));
        cs.ct(bs * cs.L(sel) * (cs.L(b0 + i) - cs.N(b0 + i)));
        cs.ct(bs * cs.L(sel) * (cs.L(b1 + i) - cs.N(b1 + i)));
    }
    mul_in24(cs, bs, t0, in, b0, true);
    add_fp2_mul_constraints(cs, t0, bs);
    mul_in24(cs, bs, t1, in + 24, b1, true);
    add_fp2_mul_constraints(cs, t1, bs);
    mul_in24(cs, bs, t2, in + 48, b1, true);
    add_fp2_mul_constraints(cs, t2, bs);
    nr_in(cs, bs, t3, mul_out(t2));
    add_non_residue_multiplication_constraints(cs, t3, bs);
    add_in(cs, bs, xc, nr_out(t3), mul_out(t0));
    add_addition_with_reduction_constraints(cs, xc, bs);
    add_in(cs, bs, t4, raw(b0), raw(b1));
    add_addition_with_reduction_constraints(cs, t4, bs);
    add_in(cs, bs, t5, raw(in), raw(in + 24));
    add_addition_with_reduction_constraints(cs, t5, bs);
    mul_in(cs, bs, t6, addred_out(t4), addred_out(t5), false);
    add_fp2_mul_constraints(cs, t6, bs);
    sub_in(cs, bs, t7, mul_out(t6), mul_out(t0));
    add_subtraction_with_reduction_constraints(cs, t7, bs);
    sub_in(cs, bs, yc, subred_out(t7), mul_out(t1));
    add_subtraction_with_reduction_constraints(cs, yc, bs);
    mul_in24(cs, bs, t8, in + 48, b0, true);
    add_fp2_mul_constraints(cs, t8, bs);
    add_in(cs, bs, zc, mul_out(t8), mul_out(t1));
    add_addition_with_reduction_constraints(cs, zc, bs);
}
// fp6.rs:2941-3106.  The coefficient multiplexer only covers table entries 0..3 (bit0, bit1), App. B.4 item 12.
void add_fp6_forbenius_map_constraints(CS& cs, size_t sc, const Expr& bs) {
    const size_t sel = sc + FP6_FORBENIUS_MAP_SELECTOR_OFFSET, in = sc + FP6_FORBENIUS_MAP_INPUT_OFFSET, powc = sc + FP6_FORBENIUS_MAP_POW_OFFSET;
    const size_t xc = sc + FP6_FORBENIUS_MAP_X_CALC_OFFSET, t0 = sc + FP6_FORBENIUS_MAP_T0_CALC_OFFSET, yc = sc + FP6_FORBENIUS_MAP_Y_CALC_OFFSET;
    const size_t t1 = sc + FP6_FORBENIUS_MAP_T1_CALC_OFFSET, zc = sc + FP6_FORBENIUS_MAP_Z_CALC_OFFSET;
    cs.keep(true, bs * cs.L(sel), in, 72);
    cs.ct(bs * cs.L(sel) * (cs.L(powc) - cs.N(powc)));
    cs.c(bs * cs.L(sel) * (cs.L(sc + FP6_FORBENIUS_MAP_DIV_OFFSET) * CS::K(6) + cs.L(sc + FP6_FORBENIUS_MAP_REM_OFFSET) - cs.L(powc)));
    const Expr bit0 = cs.L(sc + FP6_FORBENIUS_MAP_BIT0_OFFSET), bit1 = cs.L(sc + FP6_FORBENIUS_MAP_BIT1_OFFSET), bit2 = cs.L(sc + FP6_FORBENIUS_MAP_BIT2_OFFSET);
    cs.c(bs * cs.L(sel) * (bit0 + bit1 * CS::K(2) + bit2 * CS::K(4) - cs.L(sc + FP6_FORBENIUS_MAP_REM_OFFSET)));
    auto limb = [](const Fp2& v, size_t i) { return (uint64_t)(i < 12 ? v.c[0].l[i] : v.c[1].l[i - 12]); };
    auto mux = [&](const Fp2* tab, size_t i) {
        const Expr one = CS::one();
        return (one - bit0) * (one - bit1) * CS::K(limb(tab[0], i)) + bit0 * (one - bit1) * CS::K(limb(tab[1], i)) +
               (one - bit0) * bit1 * CS::K(limb(tab[2], i)) + bit0 * bit1 * CS::K(limb(tab[3], i));
    };
    auto sub_frob = [&](size_t blk, size_t in_off) {
        const size_t s = blk + FP2_FORBENIUS_MAP_SELECTOR_OFFSET;
        cs.c(bs * cs.L(s) * (cs.L(blk + FP2_FORBENIUS_MAP_POW_OFFSET) - cs.L(powc)));
        cs.link(false, bs * cs.L(s), blk + FP2_FORBENIUS_MAP_INPUT_OFFSET, in + in_off, 24);
        add_fp2_forbenius_map_constraints(cs, blk, bs);
    };
    auto coef_mul = [&](size_t mulblk, size_t frob, const Fp2* tab) {
        const size_t s = mulblk + FP2_FP2_SELECTOR_OFFSET, X = mulblk + FP2_FP2_X_INPUT_OFFSET, Y = mulblk + FP2_FP2_Y_INPUT_OFFSET;
        const size_t red = frob + FP2_FORBENIUS_MAP_T0_CALC_OFFSET + FP_MULTIPLICATION_TOTAL_COLUMNS + REDUCED_OFFSET;
        for (size_t i = 0; i < 12; i++) {
            cs.c(bs * cs.L(s) * (cs.L(X + i) - cs.L(frob + FP2_FORBENIUS_MAP_INPUT_OFFSET + i)));
            cs.c(bs * cs.L(s) * (cs.L(X + i + 12) - cs.L(red + i)));
            cs.c(bs * cs.L(s) * (cs.L(Y + i) - mux(tab, i)));
            cs.c(bs * cs.L(s) * (cs.L(Y + i + 12) - mux(tab, i + 12)));
        }
        add_fp2_mul_constraints(cs, mulblk, bs);
    };
    sub_frob(xc, 0);
    sub_frob(t0, 24);
    coef_mul(yc, t0, fp6_frobenius_coeff_1());
    sub_frob(t1, 48);
    coef_mul(zc, t1, fp6_frobenius_coeff_2());
}

}  // namespace starkhip
