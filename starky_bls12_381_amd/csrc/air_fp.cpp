// Fp gadgets: trace fillers and constraint emitters.  Restates the fill_* / packed add_*_constraints
// halves of /root/reference/src/fp.rs (line references on each function).
#include "gadgets.h"

namespace starkhip {
using namespace lay;
using namespace bls;

static const uint64_t TWO32 = 1ULL << 32;

const L12& modulus_limbs() { return MODULUS; }
const L24& modulus_sq_limbs() {
    static const L24 v = mul_wide(MODULUS, MODULUS);
    return v;
}
const L12& range_check_offset() {  // 2^382 - p
    static const L12 v = [] {
        L12 two382, out, br;
        two382.fill(0);
        two382[11] = 1u << 30;
        sub_u32_slices_12(two382, MODULUS, out, br);
        return out;
    }();
    return v;
}

// ------------------------------------------------------------------ fillers
void fill_addition_trace(Trace& t, const L24& x, const L24& y, size_t row, size_t col) {  // fp.rs:185-201
    L24 s, c;
    add_u32_slices(x, y, s, c);
    t.at(row, col + ADDITION_CHECK_OFFSET) = 1;
    t.put(row, col + ADDITION_X_OFFSET, x);
    t.put(row, col + ADDITION_Y_OFFSET, y);
    t.put(row, col + ADDITION_SUM_OFFSET, s);
    t.put(row, col + ADDITION_CARRY_OFFSET, c);
}
void fill_trace_addition_fp(Trace& t, const L12& x, const L12& y, size_t row, size_t col) {  // fp.rs:204-220
    L12 s, c;
    add_u32_slices_12(x, y, s, c);
    t.at(row, col + FP_ADDITION_CHECK_OFFSET) = 1;
    t.put(row, col + FP_ADDITION_X_OFFSET, x);
    t.put(row, col + FP_ADDITION_Y_OFFSET, y);
    t.put(row, col + FP_ADDITION_SUM_OFFSET, s);
    t.put(row, col + FP_ADDITION_CARRY_OFFSET, c);
}
void fill_trace_negate_fp(Trace& t, const L12& x, size_t row, size_t col) {  // fp.rs:223-234
    fill_trace_addition_fp(t, x, (-Fp(x)).l, row, col);
}
void fill_subtraction_trace(Trace& t, const L24& x, const L24& y, size_t row, size_t col) {  // fp.rs:237-253
    L24 d, b;
    sub_u32_slices(x, y, d, b);
    t.at(row, col + SUBTRACTION_CHECK_OFFSET) = 1;
    t.put(row, col + SUBTRACTION_X_OFFSET, x);
    t.put(row, col + SUBTRACTION_Y_OFFSET, y);
    t.put(row, col + SUBTRACTION_DIFF_OFFSET, d);
    t.put(row, col + SUBTRACTION_BORROW_OFFSET, b);
}
void fill_trace_subtraction_fp(Trace& t, const L12& x, const L12& y, size_t row, size_t col) {  // fp.rs:256-272
    L12 d, b;
    sub_u32_slices_12(x, y, d, b);
    t.at(row, col + FP_SUBTRACTION_CHECK_OFFSET) = 1;
    t.put(row, col + FP_SUBTRACTION_X_OFFSET, x);
    t.put(row, col + FP_SUBTRACTION_Y_OFFSET, y);
    t.put(row, col + FP_SUBTRACTION_DIFF_OFFSET, d);
    t.put(row, col + FP_SUBTRACTION_BORROW_OFFSET, b);
}
void fill_trace_multiply_single_fp(Trace& t, const L12& x, uint32_t y, size_t row, size_t col) {  // fp.rs:275-291
    L12 s, c;
    mul_u32_slice_u32(x, y, s, c);
    t.at(row, col + FP_MULTIPLY_SINGLE_CHECK_OFFSET) = 1;
    t.put(row, col + FP_MULTIPLY_SINGLE_X_OFFSET, x);
    t.at(row, col + FP_MULTIPLY_SINGLE_Y_OFFSET) = y;
    t.put(row, col + FP_MULTIPLY_SINGLE_SUM_OFFSET, s);
    t.put(row, col + FP_MULTIPLY_SINGLE_CARRY_OFFSET, c);
}
L12 fill_trace_reduce_single(Trace& t, const L12& x, size_t row, size_t col) {  // fp.rs:294-312
    L12 div, rem;
    div_rem_modulus(widen(x), div, rem);
    const uint32_t d = div[0];
    fill_trace_multiply_single_fp(t, MODULUS, d, row, col + FP_SINGLE_REDUCE_MULTIPLICATION_OFFSET);
    t.put(row, col + FP_SINGLE_REDUCE_X_OFFSET, x);
    L12 dm, carries;
    mul_u32_slice_u32(MODULUS, d, dm, carries);  // d * p fits 12 limbs
    t.put(row, col + FP_SINGLE_REDUCED_OFFSET, rem);
    fill_trace_addition_fp(t, dm, rem, row, col + FP_SINGLE_REDUCTION_ADDITION_OFFSET);
    return rem;
}
void fill_range_check_trace(Trace& t, const L12& x, size_t row, size_t col) {  // fp.rs:315-330
    L12 s, c;
    add_u32_slices_12(x, range_check_offset(), s, c);
    t.at(row, col + RANGE_CHECK_SELECTOR_OFFSET) = 1;
    t.put(row, col + RANGE_CHECK_SUM_OFFSET, s);
    t.put(row, col + RANGE_CHECK_SUM_CARRY_OFFSET, c);
    for (int i = 0; i < 32; i++) t.at(row, col + RANGE_CHECK_BIT_DECOMP_OFFSET + i) = (s[11] >> i) & 1;
}
// fp.rs:333-383.  `selector` is a u32 doubled once per row (wraps to 0 after 32 rows);
// inputs and selector bits are written on start_row..=end_row, the product rows only on 12 rows.
void fill_multiplication_trace_no_mod_reduction(Trace& t, const L12& x, const L12& y, size_t start_row, size_t end_row, size_t col) {
    uint32_t selector = 1;
    t.at(start_row, col + MULTIPLICATION_FIRST_ROW_OFFSET) = 1;
    { RowSpan rows_(t, 11); t.at(start_row, col + MULTIPLICATION_SELECTOR_OFFSET) = 1; }
    {
        RowSpan rows_(t, end_row - start_row + 1);
        t.put(start_row, col + X_INPUT_OFFSET, x);
        t.put(start_row, col + Y_INPUT_OFFSET, y);
    }
    // the one-hot step selector (fp.rs:216-222 writes all twelve bits of `selector` on every row; the eleven zeros land on zeros)
    for (size_t row = start_row; row <= end_row && row < start_row + 12; row++) t.at(row, col + SELECTOR_OFFSET + (row - start_row)) = 1;
    (void)selector;
    L24 prev;
    prev.fill(0);
    for (size_t i = 0; i < 12; i++) {
        uint32_t xy[13], carries[12];
        multiply_by_slice(x, y[i], xy, carries);
        t.put(start_row + i, col + XY_OFFSET, xy, 13);
        t.put(start_row + i, col + XY_CARRIES_OFFSET, carries, 12);
        L24 shifted;
        shifted.fill(0);
        for (size_t j = 0; j < 13; j++) shifted[j + i] = xy[j];
        t.put(start_row + i, col + SHIFTED_XY_OFFSET, shifted);
        L24 sum, sc;
        add_u32_slices(shifted, prev, sum, sc);
        t.put(start_row + i, col + SUM_OFFSET, sum);
        t.put(start_row + i, col + SUM_CARRIES_OFFSET, sc);
        prev = sum;
    }
}
L12 fill_reduction_trace(Trace& t, const L24& x, size_t start_row, size_t end_row, size_t col) {  // fp.rs:386-428
    L12 div, rem;
    div_rem_modulus(x, div, rem);
    fill_multiplication_trace_no_mod_reduction(t, div, MODULUS, start_row, end_row, col + REDUCE_MULTIPLICATION_OFFSET);
    { RowSpan rows_(t, end_row - start_row + 1); t.put(start_row, col + REDUCE_X_OFFSET, x); }
    L24 div_x_mod = mul_wide(div, MODULUS);
    { RowSpan rows_(t, end_row - start_row + 1); t.put(start_row, col + REDUCED_OFFSET, rem); }
    fill_addition_trace(t, div_x_mod, widen(rem), start_row + 11, col + REDUCTION_ADDITION_OFFSET);
    return rem;
}

// ------------------------------------------------------------------ constraints
// fp.rs:443-574
void add_multiplication_constraints(CS& cs, size_t sc, const Expr& bs) {
    const Expr msel = cs.L(sc + MULTIPLICATION_SELECTOR_OFFSET);
    for (size_t i = 0; i < 12; i++) {
        cs.ct(bs * msel * (cs.L(sc + X_INPUT_OFFSET + i) - cs.N(sc + X_INPUT_OFFSET + i)));
        cs.ct(bs * msel * (cs.L(sc + Y_INPUT_OFFSET + i) - cs.N(sc + Y_INPUT_OFFSET + i)));
    }
    for (size_t i = 0; i < 12; i++) {
        const Expr sel = cs.L(sc + SELECTOR_OFFSET + i);
        for (size_t j = 0; j < 12; j++) {
            Expr prod = cs.L(sc + X_INPUT_OFFSET + j) * cs.L(sc + Y_INPUT_OFFSET + i);
            if (j == 0)
                cs.ct(bs * sel * (prod - cs.L(sc + XY_OFFSET + j) - (cs.L(sc + XY_CARRIES_OFFSET + j) * CS::K(TWO32))));
            else
                cs.ct(bs * sel * (prod + cs.L(sc + XY_CARRIES_OFFSET + j - 1) - cs.L(sc + XY_OFFSET + j) -
                                  (cs.L(sc + XY_CARRIES_OFFSET + j) * CS::K(TWO32))));
        }
    }
    cs.ct(bs * msel * (cs.L(sc + XY_OFFSET + 12) - cs.L(sc + XY_CARRIES_OFFSET + 11)));
    for (size_t i = 0; i < 12; i++) {
        const Expr sel = cs.L(sc + SELECTOR_OFFSET + i);
        for (size_t j = 0; j < 13; j++) cs.ct(bs * sel * (cs.L(sc + SHIFTED_XY_OFFSET + j + i) - cs.L(sc + XY_OFFSET + j)));
    }
    const Expr first = cs.L(sc + MULTIPLICATION_FIRST_ROW_OFFSET);
    for (size_t j = 0; j < 24; j++) {
        cs.c(bs * first * (cs.L(sc + SUM_OFFSET + j) - cs.L(sc + SHIFTED_XY_OFFSET + j)));
        cs.c(bs * first * cs.L(sc + SUM_CARRIES_OFFSET + j));
    }
    cs.ct(bs * msel * (cs.N(sc + SUM_OFFSET) + (cs.N(sc + SUM_CARRIES_OFFSET) * CS::K(TWO32)) - cs.N(sc + SHIFTED_XY_OFFSET) -
                       cs.L(sc + SUM_OFFSET)));
    for (size_t j = 1; j < 24; j++)
        cs.ct(bs * msel * (cs.N(sc + SUM_OFFSET + j) + (cs.N(sc + SUM_CARRIES_OFFSET + j) * CS::K(TWO32)) - cs.N(sc + SHIFTED_XY_OFFSET + j) -
                           cs.L(sc + SUM_OFFSET + j) - cs.N(sc + SUM_CARRIES_OFFSET + j - 1)));
}

// sum[j] + carry[j] * 2^32 - x[j] - y[j] - carry[j-1]   (shared shape of the limb-wise additions)
static Expr add_limb_body(CS& cs, size_t sum, size_t carry, size_t x, size_t y, size_t j) {
    Expr e = cs.L(sum + j) + (cs.L(carry + j) * CS::K(TWO32)) - cs.L(x + j) - cs.L(y + j);
    if (j > 0) e = e - cs.L(carry + j - 1);
    return e;
}
// diff[j] + y[j] (+ borrow[j-1]) - borrow[j] * 2^32 - x[j]
static Expr sub_limb_body(CS& cs, size_t diff, size_t borrow, size_t x, size_t y, size_t j) {
    Expr e = cs.L(diff + j) + cs.L(y + j);
    if (j > 0) e = e + cs.L(borrow + j - 1);
    return e - (cs.L(borrow + j) * CS::K(TWO32)) - cs.L(x + j);
}

void add_addition_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp.rs:696-737 (transition constraints)
    const Expr chk = cs.L(sc + ADDITION_CHECK_OFFSET);
    for (size_t j = 0; j < 24; j++)
        cs.ct(bs * chk * add_limb_body(cs, sc + ADDITION_SUM_OFFSET, sc + ADDITION_CARRY_OFFSET, sc + ADDITION_X_OFFSET, sc + ADDITION_Y_OFFSET, j));
}
void add_addition_fp_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp.rs:784-825 (plain constraints)
    const Expr chk = cs.L(sc + FP_ADDITION_CHECK_OFFSET);
    for (size_t j = 0; j < 12; j++)
        cs.c(bs * chk * add_limb_body(cs, sc + FP_ADDITION_SUM_OFFSET, sc + FP_ADDITION_CARRY_OFFSET, sc + FP_ADDITION_X_OFFSET, sc + FP_ADDITION_Y_OFFSET, j));
}
void add_subtraction_fp_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp.rs:874-915
    const Expr chk = cs.L(sc + FP_SUBTRACTION_CHECK_OFFSET);
    for (size_t j = 0; j < 12; j++)
        cs.c(bs * chk * sub_limb_body(cs, sc + FP_SUBTRACTION_DIFF_OFFSET, sc + FP_SUBTRACTION_BORROW_OFFSET, sc + FP_SUBTRACTION_X_OFFSET, sc + FP_SUBTRACTION_Y_OFFSET, j));
}
void add_negate_fp_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp.rs:967-994
    add_addition_fp_constraints(cs, sc, bs);
    cs.link_const(false, bs * cs.L(sc + FP_ADDITION_CHECK_OFFSET), sc + FP_ADDITION_SUM_OFFSET, MODULUS.data(), 12);
}
void add_fp_single_multiply_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp.rs:1024-1065
    const Expr chk = cs.L(sc + FP_MULTIPLY_SINGLE_CHECK_OFFSET);
    for (size_t j = 0; j < 12; j++) {
        Expr e = cs.L(sc + FP_MULTIPLY_SINGLE_SUM_OFFSET + j) + (cs.L(sc + FP_MULTIPLY_SINGLE_CARRY_OFFSET + j) * CS::K(TWO32)) -
                 cs.L(sc + FP_MULTIPLY_SINGLE_X_OFFSET + j) * cs.L(sc + FP_MULTIPLY_SINGLE_Y_OFFSET);
        if (j > 0) e = e - cs.L(sc + FP_MULTIPLY_SINGLE_CARRY_OFFSET + j - 1);
        cs.c(bs * chk * e);
    }
}
void add_fp_reduce_single_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp.rs:1114-1180
    const size_t mul = sc + FP_SINGLE_REDUCE_MULTIPLICATION_OFFSET, add = sc + FP_SINGLE_REDUCTION_ADDITION_OFFSET;
    cs.link_const(true, bs * cs.L(mul + FP_MULTIPLY_SINGLE_CHECK_OFFSET), mul + FP_MULTIPLY_SINGLE_X_OFFSET, MODULUS.data(), 12);
    add_fp_single_multiply_constraints(cs, mul, bs);
    const Expr achk = cs.L(add + FP_ADDITION_CHECK_OFFSET);
    cs.link(true, bs * achk, mul + FP_MULTIPLY_SINGLE_SUM_OFFSET, add + FP_ADDITION_X_OFFSET, 12);
    add_addition_fp_constraints(cs, add, bs);
    cs.link(true, bs * achk, sc + FP_SINGLE_REDUCED_OFFSET, add + FP_ADDITION_Y_OFFSET, 12);
    cs.link(true, bs * achk, sc + FP_SINGLE_REDUCE_X_OFFSET, add + FP_ADDITION_SUM_OFFSET, 12);
}
void add_subtraction_constraints(CS& cs, size_t sc, const Expr& bs) {  // fp.rs:1239-1280 (transition constraints)
    const Expr chk = cs.L(sc + SUBTRACTION_CHECK_OFFSET);
    for (size_t j = 0; j < 24; j++)
        cs.ct(bs * chk * sub_limb_body(cs, sc + SUBTRACTION_DIFF_OFFSET, sc + SUBTRACTION_BORROW_OFFSET, sc + SUBTRACTION_X_OFFSET, sc + SUBTRACTION_Y_OFFSET, j));
}
// fp.rs:1326-1378.  The checked value sits in the 12 columns BEFORE the block (start_col - 12 + i); the bit
// recomposition and bit-30 constraints are emitted inside the limb loop, i.e. 12 times each (App. B.4 items 1-2).
void add_range_check_constraints(CS& cs, size_t sc, const Expr& bs) {
    const Expr sel = cs.L(sc + RANGE_CHECK_SELECTOR_OFFSET);
    const L12& y = range_check_offset();
    const size_t bit_col = sc + RANGE_CHECK_BIT_DECOMP_OFFSET;
    for (size_t i = 0; i < 12; i++) {
        Expr e = cs.L(sc + RANGE_CHECK_SUM_OFFSET + i) + (cs.L(sc + RANGE_CHECK_SUM_CARRY_OFFSET + i) * CS::K(TWO32)) - CS::K(y[i]) - cs.L(sc - 12 + i);
        if (i > 0) e = e - cs.L(sc + RANGE_CHECK_SUM_CARRY_OFFSET + i - 1);
        cs.c(bs * sel * e);
        Expr rec = CS::K(0);
        for (size_t k = 0; k < 32; k++) rec = rec + cs.L(bit_col + k) * CS::K(1ULL << k);
        cs.c(bs * sel * (rec - cs.L(sc + RANGE_CHECK_SUM_OFFSET + 11)));
        cs.c(bs * sel * cs.L(bit_col + 30));
    }
}
void add_reduce_constraints(CS& cs, size_t sc, size_t selector_col, const Expr& bs) {  // fp.rs:1447-1553
    const size_t mul = sc + REDUCE_MULTIPLICATION_OFFSET, add = sc + REDUCTION_ADDITION_OFFSET;
    const Expr sel = cs.L(selector_col);
    cs.link_const(true, bs * sel, mul + Y_INPUT_OFFSET, MODULUS.data(), 12);
    add_multiplication_constraints(cs, mul, bs);
    cs.keep(true, bs * sel, sc + REDUCE_X_OFFSET, 24);
    cs.keep(true, bs * sel, sc + REDUCED_OFFSET, 12);
    const Expr achk = cs.L(add + ADDITION_CHECK_OFFSET);
    cs.link(true, bs * achk, mul + SUM_OFFSET, add + ADDITION_X_OFFSET, 24);
    add_addition_constraints(cs, add, bs);
    for (size_t i = 0; i < 24; i++) {
        if (i < 12) cs.ct(bs * achk * (cs.L(sc + REDUCED_OFFSET + i) - cs.L(add + ADDITION_Y_OFFSET + i)));
        else cs.ct(bs * achk * cs.L(add + ADDITION_Y_OFFSET + i));
    }
    cs.link(true, bs * achk, sc + REDUCE_X_OFFSET, add + ADDITION_SUM_OFFSET, 24);
}

}  // namespace starkhip
