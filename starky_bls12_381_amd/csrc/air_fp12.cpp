// Fp12 gadgets: trace fillers and constraint emitters.  Restates the fill_* / packed add_*_constraints
// halves of /root/reference/src/fp12.rs (line references on each function).
#include "gadgets.h"
#include "wiring.h"

namespace starkhip {
using namespace lay;
using namespace bls;
using namespace wire;

static void rows_addred6(Trace& t, const Fp6& a, const Fp6& b, size_t r0, size_t r1, size_t col) {
    { RowSpan rows_(t, r1 - r0 + 1); fill_trace_addition_with_reduction_fp6(t, a, b, r0, col); }
}
static void rows_subred6(Trace& t, const Fp6& a, const Fp6& b, size_t r0, size_t r1, size_t col) {
    { RowSpan rows_(t, r1 - r0 + 1); fill_trace_subtraction_with_reduction_fp6(t, a, b, r0, col); }
}
static void rows_nr6(Trace& t, const Fp6& a, size_t r0, size_t r1, size_t col) {
    { RowSpan rows_(t, r1 - r0 + 1); fill_trace_non_residue_multiplication_fp6(t, a, r0, col); }
}
static void rows_addred(Trace& t, const Fp2& a, const Fp2& b, size_t r0, size_t r1, size_t col) {
    { RowSpan rows_(t, r1 - r0 + 1); fill_trace_addition_with_reduction(t, a, b, r0, col); }
}
static void rows_subred(Trace& t, const Fp2& a, const Fp2& b, size_t r0, size_t r1, size_t col) {
    { RowSpan rows_(t, r1 - r0 + 1); fill_trace_subtraction_with_reduction(t, a, b, r0, col); }
}

// ------------------------------------------------------------------ fillers
void fill_trace_multiply_by_014(Trace& t, const Fp12& x, const Fp2& o0, const Fp2& o1, const Fp2& o4, size_t r0, size_t r1, size_t col) {  // fp12.rs:132-183
    {
        RowSpan rows_(t, r1 - r0 + 1);
        t.put(r0, col + MULTIPLY_BY_014_INPUT_OFFSET, x);
        t.put(r0, col + MULTIPLY_BY_014_O0_OFFSET, o0);
        t.put(r0, col + MULTIPLY_BY_014_O1_OFFSET, o1);
        t.put(r0, col + MULTIPLY_BY_014_O4_OFFSET, o4);
        t.at(r0, col + MULTIPLY_BY_014_SELECTOR_OFFSET) = 1;
    }
    t.at(r1, col + MULTIPLY_BY_014_SELECTOR_OFFSET) = 0;
    const Fp6 c0 = x.c6(0), c1 = x.c6(1);
    Fp6 t0 = c0.multiply_by_01(o0, o1);
    fill_trace_multiply_by_01(t, c0, o0, o1, r0, r1, col + MULTIPLY_BY_014_T0_CALC_OFFSET);
    Fp6 t1 = c1.multiply_by_1(o4);
    fill_trace_multiply_by_1(t, c1, o4, r0, r1, col + MULTIPLY_BY_014_T1_CALC_OFFSET);
    Fp6 t2 = mul_by_nonresidue(t1);
    rows_nr6(t, t1, r0, r1, col + MULTIPLY_BY_014_T2_CALC_OFFSET);
    rows_addred6(t, t2, t0, r0, r1, col + MULTIPLY_BY_014_X_CALC_OFFSET);
    Fp6 t3 = c0 + c1;
    rows_addred6(t, c0, c1, r0, r1, col + MULTIPLY_BY_014_T3_CALC_OFFSET);
    Fp2 t4 = o1 + o4;
    rows_addred(t, o1, o4, r0, r1, col + MULTIPLY_BY_014_T4_CALC_OFFSET);
    Fp6 t5 = t3.multiply_by_01(o0, t4);
    fill_trace_multiply_by_01(t, t3, o0, t4, r0, r1, col + MULTIPLY_BY_014_T5_CALC_OFFSET);
    Fp6 t6 = t5 - t0;
    rows_subred6(t, t5, t0, r0, r1, col + MULTIPLY_BY_014_T6_CALC_OFFSET);
    rows_subred6(t, t6, t1, r0, r1, col + MULTIPLY_BY_014_Y_CALC_OFFSET);
}
void fill_trace_fp12_multiplication(Trace& t, const Fp12& x, const Fp12& y, size_t r0_, size_t r1_, size_t col) {  // fp12.rs:186-231
    {
        RowSpan rows_(t, r1_ - r0_ + 1);
        t.put(r0_, col + FP12_MUL_X_INPUT_OFFSET, x);
        t.put(r0_, col + FP12_MUL_Y_INPUT_OFFSET, y);
        t.at(r0_, col + FP12_MUL_SELECTOR_OFFSET) = 1;
    }
    t.at(r1_, col + FP12_MUL_SELECTOR_OFFSET) = 0;
    const Fp6 c0 = x.c6(0), c1 = x.c6(1), r0 = y.c6(0), r1 = y.c6(1);
    Fp6 t0 = c0 * r0;
    fill_trace_fp6_multiplication(t, c0, r0, r0_, r1_, col + FP12_MUL_T0_CALC_OFFSET);
    Fp6 t1 = c1 * r1;
    fill_trace_fp6_multiplication(t, c1, r1, r0_, r1_, col + FP12_MUL_T1_CALC_OFFSET);
    Fp6 t2 = mul_by_nonresidue(t1);
    rows_nr6(t, t1, r0_, r1_, col + FP12_MUL_T2_CALC_OFFSET);
    rows_addred6(t, t0, t2, r0_, r1_, col + FP12_MUL_X_CALC_OFFSET);
    Fp6 t3 = c0 + c1;
    rows_addred6(t, c0, c1, r0_, r1_, col + FP12_MUL_T3_CALC_OFFSET);
    Fp6 t4 = r0 + r1;
    rows_addred6(t, r0, r1, r0_, r1_, col + FP12_MUL_T4_CALC_OFFSET);
    Fp6 t5 = t3 * t4;
    fill_trace_fp6_multiplication(t, t3, t4, r0_, r1_, col + FP12_MUL_T5_CALC_OFFSET);
    Fp6 t6 = t5 - t0;
    rows_subred6(t, t5, t0, r0_, r1_, col + FP12_MUL_T6_CALC_OFFSET);
    rows_subred6(t, t6, t1, r0_, r1_, col + FP12_MUL_Y_CALC_OFFSET);
}
void fill_trace_cyclotomic_sq(Trace& t, const Fp12& x, size_t r0, size_t r1, size_t col) {  // fp12.rs:234-330
    {
        RowSpan rows_(t, r1 - r0 + 1);
        t.put(r0, col + CYCLOTOMIC_SQ_INPUT_OFFSET, x);
        t.at(r0, col + CYCLOTOMIC_SQ_SELECTOR_OFFSET) = 1;
    }
    t.at(r1, col + CYCLOTOMIC_SQ_SELECTOR_OFFSET) = 0;
    const Fp2 c0c0 = x.c2(0), c0c1 = x.c2(1), c0c2 = x.c2(2), c1c0 = x.c2(3), c1c1 = x.c2(4), c1c2 = x.c2(5);
    const Fp two = Fp::from_u32(2);
    Fp2 t00, t01, t10, t11, t20, t21;
    fp4_square(c0c0, c1c1, t00, t01);
    fill_trace_fp4_sq(t, c0c0, c1c1, r0, r1, col + CYCLOTOMIC_SQ_T0_CALC_OFFSET);
    fp4_square(c1c0, c0c2, t10, t11);
    fill_trace_fp4_sq(t, c1c0, c0c2, r0, r1, col + CYCLOTOMIC_SQ_T1_CALC_OFFSET);
    fp4_square(c0c1, c1c2, t20, t21);
    fill_trace_fp4_sq(t, c0c1, c1c2, r0, r1, col + CYCLOTOMIC_SQ_T2_CALC_OFFSET);
    Fp2 t3 = t21.mul_by_nonresidue();
    { RowSpan rows_(t, r1 - r0 + 1); fill_trace_non_residue_multiplication(t, t21, r0, col + CYCLOTOMIC_SQ_T3_CALC_OFFSET); }
    // three "(t - c) * 2 + t" legs, then three "(t + c) * 2 + t" legs
    auto sub_leg = [&](const Fp2& tv, const Fp2& cv, size_t o_sub, size_t o_mul, size_t o_out) {
        Fp2 d = tv - cv;
        rows_subred(t, tv, cv, r0, r1, col + o_sub);
        Fp2 m = d * two;
        fill_trace_fp2_fp_mul(t, d, two, r0, r1, col + o_mul);
        rows_addred(t, m, tv, r0, r1, col + o_out);
    };
    auto add_leg = [&](const Fp2& tv, const Fp2& cv, size_t o_add, size_t o_mul, size_t o_out) {
        Fp2 s = tv + cv;
        rows_addred(t, tv, cv, r0, r1, col + o_add);
        Fp2 m = s * two;
        fill_trace_fp2_fp_mul(t, s, two, r0, r1, col + o_mul);
        rows_addred(t, m, tv, r0, r1, col + o_out);
    };
    sub_leg(t00, c0c0, CYCLOTOMIC_SQ_T4_CALC_OFFSET, CYCLOTOMIC_SQ_T5_CALC_OFFSET, CYCLOTOMIC_SQ_C0_CALC_OFFSET);
    sub_leg(t10, c0c1, CYCLOTOMIC_SQ_T6_CALC_OFFSET, CYCLOTOMIC_SQ_T7_CALC_OFFSET, CYCLOTOMIC_SQ_C1_CALC_OFFSET);
    sub_leg(t20, c0c2, CYCLOTOMIC_SQ_T8_CALC_OFFSET, CYCLOTOMIC_SQ_T9_CALC_OFFSET, CYCLOTOMIC_SQ_C2_CALC_OFFSET);
    add_leg(t3, c1c0, CYCLOTOMIC_SQ_T10_CALC_OFFSET, CYCLOTOMIC_SQ_T11_CALC_OFFSET, CYCLOTOMIC_SQ_C3_CALC_OFFSET);
    add_leg(t01, c1c1, CYCLOTOMIC_SQ_T12_CALC_OFFSET, CYCLOTOMIC_SQ_T13_CALC_OFFSET, CYCLOTOMIC_SQ_C4_CALC_OFFSET);
    add_leg(t11, c1c2, CYCLOTOMIC_SQ_T14_CALC_OFFSET, CYCLOTOMIC_SQ_T15_CALC_OFFSET, CYCLOTOMIC_SQ_C5_CALC_OFFSET);
}
// fp12.rs:333-374.  70 steps of 12 rows: square every step; when the current bit of |x| is set, the NEXT step
// multiplies by the input instead (bitone).  Row start_row + 840 carries the result.
// Steps j0 .. j1-1 of the 70 twelve-row steps (the whole gadget: 0, 70).  The part with j0 == 0 also writes what spans all
// 841 rows (input, selector, start row), the part with j1 == 70 the result row; a later part recomputes the running value z
// natively up to its first step (a few Fp12 operations), so disjoint parts can be filled by different threads.
void fill_trace_cyclotomic_exp_steps(Trace& t, const Fp12& x, size_t start_row, size_t end_row, size_t col, size_t j0, size_t j1) {
    if (end_row + 1 - start_row != 70 * 12 + 1) throw std::runtime_error("fill_trace_cyclotomic_exp: needs 841 rows");
    if (j0 == 0) {
        {
            RowSpan rows_(t, end_row - start_row + 1);
            t.put(start_row, col + INPUT_OFFSET, x);
            t.at(start_row, col + CYCLOTOMIC_EXP_SELECTOR_OFFSET) = 1;
        }
        t.at(end_row, col + CYCLOTOMIC_EXP_SELECTOR_OFFSET) = 0;
        t.at(start_row, col + CYCLOTOMIC_EXP_START_ROW) = 1;
    }
    Fp12 z = Fp12::one();
    int i = 63;
    bool bitone = false;
    for (size_t j = 0; j < j1; j++) {
        const size_t s_row = start_row + j * 12, e_row = s_row + 11;
        if (j >= j0) {
            {
                RowSpan rows_(t, e_row - s_row + 1);
                if (bitone) t.at(s_row, col + BIT1_SELECTOR_OFFSET) = 1;
                t.put(s_row, col + Z_OFFSET, z);
            }
            t.at(s_row, col + FIRST_ROW_SELECTOR_OFFSET) = 1;
        }
        if (bitone) {
            if (j >= j0) fill_trace_fp12_multiplication(t, z, x, s_row, e_row, col + Z_MUL_INPUT_OFFSET);
            z = z * x;
        } else {
            if (j >= j0) fill_trace_cyclotomic_sq(t, z, s_row, e_row, col + Z_CYCLOTOMIC_SQ_OFFSET);
            z = z.cyclotomic_square();
        }
        if (((BLS_X >> i) & 1) && !bitone) {
            bitone = true;
        } else if (j < 69) {
            i -= 1;
            bitone = false;
        }
    }
    if (j1 == 70) {
        t.at(start_row + 70 * 12, col + RES_ROW_SELECTOR_OFFSET) = 1;
        t.put(start_row + 70 * 12, col + Z_OFFSET, z);
    }
}
void fill_trace_cyclotomic_exp(Trace& t, const Fp12& x, size_t start_row, size_t end_row, size_t col) {
    fill_trace_cyclotomic_exp_steps(t, x, start_row, end_row, col, 0, 70);
}
void fill_trace_fp12_forbenius_map(Trace& t, const Fp12& x, size_t pow, size_t r0, size_t r1, size_t col) {  // fp12.rs:377-409
    const size_t div = pow / 12, rem = pow % 12;
    {
        RowSpan rows_(t, r1 - r0 + 1);
        t.put(r0, col + FP12_FORBENIUS_MAP_INPUT_OFFSET, x);
        t.at(r0, col + FP12_FORBENIUS_MAP_SELECTOR_OFFSET) = 1;
        t.at(r0, col + FP12_FORBENIUS_MAP_POW_OFFSET) = pow;
        t.at(r0, col + FP12_FORBENIUS_MAP_DIV_OFFSET) = div;
        t.at(r0, col + FP12_FORBENIUS_MAP_REM_OFFSET) = rem;
        t.at(r0, col + FP12_FORBENIUS_MAP_BIT0_OFFSET) = rem & 1;
        t.at(r0, col + FP12_FORBENIUS_MAP_BIT1_OFFSET) = (rem >> 1) & 1;
        t.at(r0, col + FP12_FORBENIUS_MAP_BIT2_OFFSET) = (rem >> 2) & 1;
        t.at(r0, col + FP12_FORBENIUS_MAP_BIT3_OFFSET) = rem >> 3;
    }
    t.at(r1, col + FP12_FORBENIUS_MAP_SELECTOR_OFFSET) = 0;
    const Fp6 r0v = x.c6(0), r1v = x.c6(1);
    fill_trace_fp6_forbenius_map(t, r0v, pow, r0, r1, col + FP12_FORBENIUS_MAP_R0_CALC_OFFSET);
    Fp6 c = r1v.forbenius_map(pow);
    fill_trace_fp6_forbenius_map(t, r1v, pow, r0, r1, col + FP12_FORBENIUS_MAP_C0C1C2_CALC_OFFSET);
    const Fp2 coeff = fp12_frobenius_coeff()[pow % 12];
    generate_trace_fp2_mul(t, c.c2(0), coeff, r0, r1, col + FP12_FORBENIUS_MAP_C0_CALC_OFFSET);
    generate_trace_fp2_mul(t, c.c2(1), coeff, r0, r1, col + FP12_FORBENIUS_MAP_C1_CALC_OFFSET);
    generate_trace_fp2_mul(t, c.c2(2), coeff, r0, r1, col + FP12_FORBENIUS_MAP_C2_CALC_OFFSET);
}
void fill_trace_fp12_conjugate(Trace& t, const Fp12& x, size_t row, size_t col) {  // fp12.rs:412-422
    t.put(row, col + FP12_CONJUGATE_INPUT_OFFSET, x);
    Fp12 conj = x.conjugate();
    t.put(row, col + FP12_CONJUGATE_OUTPUT_OFFSET, conj);
    fill_trace_addition_fp6(t, x.c6(1), conj.c6(1), row, col + FP12_CONJUGATE_ADDITIION_OFFSET);
}

// ------------------------------------------------------------------ constraints
void add_multiply_by_014_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp12.rs:427-740
    const size_t sel = sc + MULTIPLY_BY_014_SELECTOR_OFFSET, in = sc + MULTIPLY_BY_014_INPUT_OFFSET;
    const size_t o0 = sc + MULTIPLY_BY_014_O0_OFFSET, o1 = sc + MULTIPLY_BY_014_O1_OFFSET, o4 = sc + MULTIPLY_BY_014_O4_OFFSET;
    const size_t t0 = sc + MULTIPLY_BY_014_T0_CALC_OFFSET, t1 = sc + MULTIPLY_BY_014_T1_CALC_OFFSET, t2 = sc + MULTIPLY_BY_014_T2_CALC_OFFSET;
    const size_t xc = sc + MULTIPLY_BY_014_X_CALC_OFFSET, t3 = sc + MULTIPLY_BY_014_T3_CALC_OFFSET, t4 = sc + MULTIPLY_BY_014_T4_CALC_OFFSET;
    const size_t t5 = sc + MULTIPLY_BY_014_T5_CALC_OFFSET, t6 = sc + MULTIPLY_BY_014_T6_CALC_OFFSET, yc = sc + MULTIPLY_BY_014_Y_CALC_OFFSET;
    for (size_t i = 0; i < 12; i++) {
        for (size_t j = 0; j < 12; j++) cs.ct(bs * cs.L(sel) * (cs.L(in + j * 12 + i) - cs.N(in + j * 12 + i)));
        for (size_t j = 0; j < 2; j++) {
            cs.ct(bs * cs.L(sel) * (cs.L(o0 + j * 12 + i) - cs.N(o0 + j * 12 + i)));
            cs.ct(bs * cs.L(sel) * (cs.L(o1 + j * 12 + i) - cs.N(o1 + j * 12 + i)));
            cs.ct(bs * cs.L(sel) * (cs.L(o4 + j * 12 + i) - cs.N(o4 + j * 12 + i)));
        }
    }
    auto m01_inputs = [&](size_t blk, Loc6 xf, size_t xb, size_t b0col, Loc6 b1f, size_t b1b) {
        const Expr g = bs * cs.L(blk + MULTIPLY_BY_01_SELECTOR_OFFSET);
        for (size_t i = 0; i < 12; i++) {
            for (size_t j = 0; j < 6; j++) cs.c(g * (cs.L(blk + MULTIPLY_BY_01_INPUT_OFFSET + j * 12 + i) - cs.L(xf(xb, j) + i)));
            for (size_t j = 0; j < 2; j++) {
                cs.c(g * (cs.L(blk + MULTIPLY_BY_01_B0_OFFSET + j * 12 + i) - cs.L(b0col + j * 12 + i)));
                cs.c(g * (cs.L(blk + MULTIPLY_BY_01_B1_OFFSET + j * 12 + i) - cs.L(b1f(b1b, j) + i)));
            }
        }
    };
    m01_inputs(t0, raw6, in, o0, raw6, o1);
    add_multiply_by_01_constraints(cs, t0, bs);
    {
        const Expr g = bs * cs.L(t1 + MULTIPLY_BY_1_SELECTOR_OFFSET);
        for (size_t i = 0; i < 12; i++) {
            for (size_t j = 0; j < 6; j++) cs.c(g * (cs.L(t1 + MULTIPLY_BY_1_INPUT_OFFSET + j * 12 + i) - cs.L(in + j * 12 + i + 72)));
            for (size_t j = 0; j < 2; j++) cs.c(g * (cs.L(t1 + MULTIPLY_BY_1_B1_OFFSET + j * 12 + i) - cs.L(o4 + j * 12 + i)));
        }
    }
    add_multiply_by_1_constraints(cs, t1, bs);
    {
        const Expr g = bs * cs.L(t2 + FP6_NON_RESIDUE_MUL_CHECK_OFFSET);
        const size_t nin = t2 + FP6_NON_RESIDUE_MUL_INPUT_OFFSET;
        for (size_t j = 0; j < 2; j++)
            for (size_t i = 0; i < 12; i++) {
                cs.c(g * (cs.L(nin + i + j * 12) - cs.L(m1_out(t1, j) + i)));
                cs.c(g * (cs.L(nin + i + j * 12 + 24) - cs.L(m1_out(t1, 2 + j) + i)));
                cs.c(g * (cs.L(nin + i + j * 12 + 48) - cs.L(m1_out(t1, 4 + j) + i)));
            }
    }
    add_non_residue_multiplication_fp6_constraints(cs, t2, bs);
    add6_in(cs, bs, xc, nr6_out, t2, m01_out, t0);
    add_addition_with_reduction_constraints_fp6(cs, xc, bs);
    add6_in(cs, bs, t3, raw6, in, raw6, in + 72);
    add_addition_with_reduction_constraints_fp6(cs, t3, bs);
    for (size_t j = 0; j < 2; j++) {
        const size_t a = t4 + (j ? FP2_ADDITION_1_OFFSET : FP2_ADDITION_0_OFFSET);
        cs.links(false, bs, 12, {{a + FP_ADDITION_CHECK_OFFSET, a + FP_ADDITION_X_OFFSET, o1 + j * 12}, {a + FP_ADDITION_CHECK_OFFSET, a + FP_ADDITION_Y_OFFSET, o4 + j * 12}});
    }
    add_addition_with_reduction_constraints(cs, t4, bs);
    m01_inputs(t5, [](size_t b, size_t i) { return addred6_out(b, i); }, t3, o0,
               [](size_t b, size_t j) { return b + FP2_ADDITION_TOTAL + RR * j + FP_SINGLE_REDUCED_OFFSET; }, t4);
    add_multiply_by_01_constraints(cs, t5, bs);
    sub6_in(cs, bs, t6, m01_out, t5, m01_out, t0);
    add_subtraction_with_reduction_constraints_fp6(cs, t6, bs);
    sub6_in(cs, bs, yc, subred6_out, t6, m1_out, t1);
    add_subtraction_with_reduction_constraints_fp6(cs, yc, bs);
}
void add_fp12_multiplication_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp12.rs:1025-1254
    const size_t sel = sc + FP12_MUL_SELECTOR_OFFSET, X = sc + FP12_MUL_X_INPUT_OFFSET, Y = sc + FP12_MUL_Y_INPUT_OFFSET;
    const size_t t0 = sc + FP12_MUL_T0_CALC_OFFSET, t1 = sc + FP12_MUL_T1_CALC_OFFSET, t2 = sc + FP12_MUL_T2_CALC_OFFSET, xc = sc + FP12_MUL_X_CALC_OFFSET;
    const size_t t3 = sc + FP12_MUL_T3_CALC_OFFSET, t4 = sc + FP12_MUL_T4_CALC_OFFSET, t5 = sc + FP12_MUL_T5_CALC_OFFSET, t6 = sc + FP12_MUL_T6_CALC_OFFSET;
    const size_t yc = sc + FP12_MUL_Y_CALC_OFFSET;
    for (size_t i = 0; i < 144; i++) {
        cs.ct(bs * cs.L(sel) * (cs.L(X + i) - cs.N(X + i)));
        cs.ct(bs * cs.L(sel) * (cs.L(Y + i) - cs.N(Y + i)));
    }
    cs.links(false, bs, 72, {{t0 + FP6_MUL_SELECTOR_OFFSET, t0 + FP6_MUL_X_INPUT_OFFSET, X}, {t0 + FP6_MUL_SELECTOR_OFFSET, t0 + FP6_MUL_Y_INPUT_OFFSET, Y}});
    add_fp6_multiplication_constraints(cs, t0, bs);
    cs.links(false, bs, 72, {{t1 + FP6_MUL_SELECTOR_OFFSET, t1 + FP6_MUL_X_INPUT_OFFSET, X + 72}, {t1 + FP6_MUL_SELECTOR_OFFSET, t1 + FP6_MUL_Y_INPUT_OFFSET, Y + 72}});
    add_fp6_multiplication_constraints(cs, t1, bs);
    for (size_t i = 0; i < 6; i++)
        cs.link(false, bs * cs.L(t2 + FP6_NON_RESIDUE_MUL_CHECK_OFFSET), t2 + FP6_NON_RESIDUE_MUL_INPUT_OFFSET + i * 12, fp6mul_out(t1, i), 12);
    add_non_residue_multiplication_fp6_constraints(cs, t2, bs);
    add6_in(cs, bs, xc, fp6mul_out, t0, nr6_out, t2);
    add_addition_with_reduction_constraints_fp6(cs, xc, bs);
    add6_in(cs, bs, t3, raw6, X, raw6, X + 72);
    add_addition_with_reduction_constraints_fp6(cs, t3, bs);
    add6_in(cs, bs, t4, raw6, Y, raw6, Y + 72);
    add_addition_with_reduction_constraints_fp6(cs, t4, bs);
    for (size_t i = 0; i < 6; i++)
        cs.links(false, bs, 12, {{t5 + FP6_MUL_SELECTOR_OFFSET, t5 + FP6_MUL_X_INPUT_OFFSET + i * 12, addred6_out(t3, i)},
                                 {t5 + FP6_MUL_SELECTOR_OFFSET, t5 + FP6_MUL_Y_INPUT_OFFSET + i * 12, addred6_out(t4, i)}});
    add_fp6_multiplication_constraints(cs, t5, bs);
    sub6_in(cs, bs, t6, fp6mul_out, t5, fp6mul_out, t0);
    add_subtraction_with_reduction_constraints_fp6(cs, t6, bs);
    sub6_in(cs, bs, yc, subred6_out, t6, fp6mul_out, t1);
    add_subtraction_with_reduction_constraints_fp6(cs, yc, bs);
}
void add_cyclotomic_sq_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp12.rs:1556-2076
    const size_t sel = sc + CYCLOTOMIC_SQ_SELECTOR_OFFSET, in = sc + CYCLOTOMIC_SQ_INPUT_OFFSET;
    const size_t t0 = sc + CYCLOTOMIC_SQ_T0_CALC_OFFSET, t1 = sc + CYCLOTOMIC_SQ_T1_CALC_OFFSET, t2 = sc + CYCLOTOMIC_SQ_T2_CALC_OFFSET, t3 = sc + CYCLOTOMIC_SQ_T3_CALC_OFFSET;
    cs.keep(true, bs * cs.L(sel), in, 144);
    auto fp4_in = [&](size_t blk, size_t xoff, size_t yoff) {
        cs.links(false, bs, 24, {{blk + FP4_SQ_SELECTOR_OFFSET, blk + FP4_SQ_INPUT_X_OFFSET, in + xoff}, {blk + FP4_SQ_SELECTOR_OFFSET, blk + FP4_SQ_INPUT_Y_OFFSET, in + yoff}});
        add_fp4_sq_constraints(cs, blk, bs);
    };
    fp4_in(t0, 0, 24 * 4);
    fp4_in(t1, 24 * 3, 24 * 2);
    fp4_in(t2, 24 * 1, 24 * 5);
    auto f4x = [](size_t blk) { return addred_out(blk + FP4_SQ_X_CALC_OFFSET); };
    auto f4y = [](size_t blk) { return subred_out(blk + FP4_SQ_Y_CALC_OFFSET); };
    nr_in(cs, bs, t3, f4y(t2));
    add_non_residue_multiplication_constraints(cs, t3, bs);
    // x2 multiplication block fed from an Fp2 location: per i: X[i] - v.c0[i], X[12 + i] - v.c1[i], Y[i] - (i == 0 ? 2 : 0)
    auto times_two = [&](size_t blk, Loc2 v) {
        const Expr g = bs * cs.L(blk + FP2_FP_MUL_SELECTOR_OFFSET);
        for (size_t i = 0; i < 12; i++) {
            cs.c(g * (cs.L(blk + FP2_FP_X_INPUT_OFFSET + i) - cs.L(v.c0 + i)));
            cs.c(g * (cs.L(blk + FP2_FP_X_INPUT_OFFSET + 12 + i) - cs.L(v.c1 + i)));
            cs.c(g * (cs.L(blk + FP2_FP_Y_INPUT_OFFSET + i) - CS::K(i == 0 ? 2 : 0)));
        }
        add_fp2_fp_mul_constraints(cs, blk, bs);
    };
    auto sub_leg = [&](Loc2 tv, size_t coff, size_t o_sub, size_t o_mul, size_t o_out) {
        sub_in_alt(cs, bs, sc + o_sub, tv, raw(in + coff));
        add_subtraction_with_reduction_constraints(cs, sc + o_sub, bs);
        times_two(sc + o_mul, subred_out(sc + o_sub));
        add_in_alt(cs, bs, sc + o_out, fp2fp_out(sc + o_mul), tv);
        add_addition_with_reduction_constraints(cs, sc + o_out, bs);
    };
    auto add_leg = [&](Loc2 tv, size_t coff, size_t o_add, size_t o_mul, size_t o_out) {
        add_in_alt(cs, bs, sc + o_add, tv, raw(in + coff));
        add_addition_with_reduction_constraints(cs, sc + o_add, bs);
        times_two(sc + o_mul, addred_out(sc + o_add));
        add_in_alt(cs, bs, sc + o_out, fp2fp_out(sc + o_mul), tv);
        add_addition_with_reduction_constraints(cs, sc + o_out, bs);
    };
    sub_leg(f4x(t0), 0, CYCLOTOMIC_SQ_T4_CALC_OFFSET, CYCLOTOMIC_SQ_T5_CALC_OFFSET, CYCLOTOMIC_SQ_C0_CALC_OFFSET);
    sub_leg(f4x(t1), 24, CYCLOTOMIC_SQ_T6_CALC_OFFSET, CYCLOTOMIC_SQ_T7_CALC_OFFSET, CYCLOTOMIC_SQ_C1_CALC_OFFSET);
    sub_leg(f4x(t2), 48, CYCLOTOMIC_SQ_T8_CALC_OFFSET, CYCLOTOMIC_SQ_T9_CALC_OFFSET, CYCLOTOMIC_SQ_C2_CALC_OFFSET);
    add_leg(nr_out(t3), 72, CYCLOTOMIC_SQ_T10_CALC_OFFSET, CYCLOTOMIC_SQ_T11_CALC_OFFSET, CYCLOTOMIC_SQ_C3_CALC_OFFSET);
    add_leg(f4y(t0), 96, CYCLOTOMIC_SQ_T12_CALC_OFFSET, CYCLOTOMIC_SQ_T13_CALC_OFFSET, CYCLOTOMIC_SQ_C4_CALC_OFFSET);
    add_leg(f4y(t1), 120, CYCLOTOMIC_SQ_T14_CALC_OFFSET, CYCLOTOMIC_SQ_T15_CALC_OFFSET, CYCLOTOMIC_SQ_C5_CALC_OFFSET);
}
// fp12.rs:2480-2617.  bit1 / bit0 fold the optional op selector in; they are handed down as the sub-gadgets' selector.
void add_cyclotomic_exp_constraints(CS& cs, size_t sc, const Expr& op) {
    static const size_t CSQ_C[6] = {CYCLOTOMIC_SQ_C0_CALC_OFFSET, CYCLOTOMIC_SQ_C1_CALC_OFFSET, CYCLOTOMIC_SQ_C2_CALC_OFFSET,
                                    CYCLOTOMIC_SQ_C3_CALC_OFFSET, CYCLOTOMIC_SQ_C4_CALC_OFFSET, CYCLOTOMIC_SQ_C5_CALC_OFFSET};
    const size_t sel = sc + CYCLOTOMIC_EXP_SELECTOR_OFFSET, in = sc + INPUT_OFFSET, Z = sc + Z_OFFSET;
    const size_t sq = sc + Z_CYCLOTOMIC_SQ_OFFSET, mul = sc + Z_MUL_INPUT_OFFSET;
    cs.keep(true, op * cs.L(sel), in, 144);
    for (size_t i = 0; i < 144; i++) cs.c(op * cs.L(sc + CYCLOTOMIC_EXP_START_ROW) * (cs.L(Z + i) - CS::K(i == 0 ? 1 : 0)));
    const Expr bit1 = cs.L(sc + BIT1_SELECTOR_OFFSET) * op;
    const Expr bit0 = (CS::one() - cs.L(sc + BIT1_SELECTOR_OFFSET)) * op;
    const Expr nfirst = cs.N(sc + FIRST_ROW_SELECTOR_OFFSET);
    for (size_t i = 0; i < 12; i++)
        for (size_t j = 0; j < 6; j++)
            for (size_t k = 0; k < 2; k++)
                cs.ct(bit0 * cs.L(sel) * nfirst * (cs.N(Z + j * 24 + k * 12 + i) - cs.L(sq + CSQ_C[j] + FP2_ADDITION_TOTAL + RR * k + FP_SINGLE_REDUCED_OFFSET + i)));
    for (size_t i = 0; i < 12; i++)
        for (size_t j = 0; j < 6; j++) {
            cs.ct(bit1 * cs.L(sel) * nfirst * (cs.N(Z + j * 12 + i) - cs.L(addred6_out(mul + FP12_MUL_X_CALC_OFFSET, j) + i)));
            cs.ct(bit1 * cs.L(sel) * nfirst * (cs.N(Z + j * 12 + i + 72) - cs.L(subred6_out(mul + FP12_MUL_Y_CALC_OFFSET, j) + i)));
        }
    cs.link(false, bit0 * cs.L(sq + CYCLOTOMIC_SQ_SELECTOR_OFFSET), sq + CYCLOTOMIC_SQ_INPUT_OFFSET, Z, 144);
    add_cyclotomic_sq_constraints(cs, sq, bit0);
    cs.links(false, bit1, 144, {{mul + FP12_MUL_SELECTOR_OFFSET, mul + FP12_MUL_X_INPUT_OFFSET, Z}, {mul + FP12_MUL_SELECTOR_OFFSET, mul + FP12_MUL_Y_INPUT_OFFSET, in}});
    add_fp12_multiplication_constraints(cs, mul, bit1);
    const Expr nres = cs.N(sc + RES_ROW_SELECTOR_OFFSET);
    for (size_t i = 0; i < 12; i++)
        for (size_t j = 0; j < 6; j++)
            for (size_t k = 0; k < 2; k++)
                cs.ct(op * cs.L(sel) * nres * (cs.N(Z + j * 24 + k * 12 + i) - cs.L(sq + CSQ_C[j] + FP2_ADDITION_TOTAL + RR * k + FP_SINGLE_REDUCED_OFFSET + i)));
}
// fp12.rs:2747-2871.  The multiplexer only covers table entries 0..6 (bit0..bit2); bit3 is decomposed but unused (App. B.4 item 12).
void add_fp12_forbenius_map_constraints(CS& cs, size_t sc, const Expr& bs) {
    const size_t sel = sc + FP12_FORBENIUS_MAP_SELECTOR_OFFSET, in = sc + FP12_FORBENIUS_MAP_INPUT_OFFSET, powc = sc + FP12_FORBENIUS_MAP_POW_OFFSET;
    const size_t r0 = sc + FP12_FORBENIUS_MAP_R0_CALC_OFFSET, cc = sc + FP12_FORBENIUS_MAP_C0C1C2_CALC_OFFSET;
    cs.keep(true, bs * cs.L(sel), in, 144);
    cs.ct(bs * cs.L(sel) * (cs.L(powc) - cs.N(powc)));
    cs.c(bs * cs.L(sel) * (cs.L(sc + FP12_FORBENIUS_MAP_DIV_OFFSET) * CS::K(12) + cs.L(sc + FP12_FORBENIUS_MAP_REM_OFFSET) - cs.L(powc)));
    const Expr b0 = cs.L(sc + FP12_FORBENIUS_MAP_BIT0_OFFSET), b1 = cs.L(sc + FP12_FORBENIUS_MAP_BIT1_OFFSET), b2 = cs.L(sc + FP12_FORBENIUS_MAP_BIT2_OFFSET),
               b3 = cs.L(sc + FP12_FORBENIUS_MAP_BIT3_OFFSET);
    cs.c(bs * cs.L(sel) * (b0 + b1 * CS::K(2) + b2 * CS::K(4) + b3 * CS::K(8) - cs.L(sc + FP12_FORBENIUS_MAP_REM_OFFSET)));
    const Fp2* tab = fp12_frobenius_coeff();
    auto limb = [&](size_t e, size_t i) { return (uint64_t)(i < 12 ? tab[e].c[0].l[i] : tab[e].c[1].l[i - 12]); };
    auto y = [&](size_t i) {
        const Expr one = CS::one();
        return (one - b0) * (one - b1) * (one - b2) * CS::K(limb(0, i)) + b0 * (one - b1) * (one - b2) * CS::K(limb(1, i)) +
               (one - b0) * b1 * (one - b2) * CS::K(limb(2, i)) + b0 * b1 * (one - b2) * CS::K(limb(3, i)) +
               (one - b0) * (one - b1) * b2 * CS::K(limb(4, i)) + b0 * (one - b1) * b2 * CS::K(limb(5, i)) + (one - b0) * b1 * b2 * CS::K(limb(6, i));
    };
    auto sub6 = [&](size_t blk, size_t in_off) {
        const Expr g = bs * cs.L(blk + FP6_FORBENIUS_MAP_SELECTOR_OFFSET);
        cs.c(g * (cs.L(blk + FP6_FORBENIUS_MAP_POW_OFFSET) - cs.L(powc)));
        cs.link(false, g, blk + FP6_FORBENIUS_MAP_INPUT_OFFSET, in + in_off, 72);
        add_fp6_forbenius_map_constraints(cs, blk, bs);
    };
    sub6(r0, 0);
    sub6(cc, 72);
    auto coef_mul = [&](size_t mulblk, size_t src0, size_t src1) {
        const Expr g = bs * cs.L(mulblk + FP2_FP2_SELECTOR_OFFSET);
        for (size_t i = 0; i < 12; i++)
            for (size_t j = 0; j < 2; j++) {
                cs.c(g * (cs.L(mulblk + FP2_FP2_X_INPUT_OFFSET + j * 12 + i) - cs.L((j == 0 ? src0 : src1) + i)));
                cs.c(g * (cs.L(mulblk + FP2_FP2_Y_INPUT_OFFSET + j * 12 + i) - y(j * 12 + i)));
            }
        add_fp2_mul_constraints(cs, mulblk, bs);
    };
    const size_t fx = cc + FP6_FORBENIUS_MAP_X_CALC_OFFSET;
    coef_mul(sc + FP12_FORBENIUS_MAP_C0_CALC_OFFSET, fx + FP2_FORBENIUS_MAP_INPUT_OFFSET,
             fx + FP2_FORBENIUS_MAP_T0_CALC_OFFSET + FP_MULTIPLICATION_TOTAL_COLUMNS + REDUCED_OFFSET);
    const Loc2 fy = mul_out(cc + FP6_FORBENIUS_MAP_Y_CALC_OFFSET), fz = mul_out(cc + FP6_FORBENIUS_MAP_Z_CALC_OFFSET);
    coef_mul(sc + FP12_FORBENIUS_MAP_C1_CALC_OFFSET, fy.c0, fy.c1);
    coef_mul(sc + FP12_FORBENIUS_MAP_C2_CALC_OFFSET, fz.c0, fz.c1);
}
void add_fp12_conjugate_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp12.rs:3068-3097
    const size_t add = sc + FP12_CONJUGATE_ADDITIION_OFFSET;
    for (size_t i = 0; i < 12; i++)
        for (size_t jk = 0; jk < 6; jk++) {
            const size_t a = add6_block(add, jk);
            cs.c(bs * cs.L(a + FP_ADDITION_CHECK_OFFSET) * (cs.L(a + FP_ADDITION_X_OFFSET + i) - cs.L(sc + FP12_CONJUGATE_INPUT_OFFSET + 72 + jk * 12 + i)));
            cs.c(bs * cs.L(a + FP_ADDITION_CHECK_OFFSET) * (cs.L(a + FP_ADDITION_Y_OFFSET + i) - cs.L(sc + FP12_CONJUGATE_OUTPUT_OFFSET + 72 + jk * 12 + i)));
        }
    add_negate_fp6_constraints(cs, add, bs);
}

}  // namespace starkhip
