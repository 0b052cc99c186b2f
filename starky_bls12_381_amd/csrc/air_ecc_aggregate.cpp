// ECCAggStark: aggregation of 512 G1 public keys under a participation bitmap (3339 columns x 8192 rows,
// constraint degree 4) -- the fifth STARK of the reference's pipeline.
// Restates /root/reference/src/g1.rs (column map :7-23, fill_trace_g1_addition :26-255,
// add_g1_addition_constraints :257-604) and /root/reference/src/ecc_aggregate.rs (column map :7-21,
// generate_trace :39-84, eval_packed_generic :96-269); public inputs as built by ec_aggregate_main,
// src/aggregate_proof.rs:181-221 (points, bits, aggregate).  The *_ext_circuit halves are out of scope.
//
// One affine addition (x3, y3) = (x1, y1) + (x2, y2) occupies 12 rows and is checked without a division:
//   (x1 + x2 + x3) * (x2 - x1)^2 == (y2 - y1)^2        and        (y1 + y3) * (x2 - x1) == (y2 - y1) * (x1 - x3)   (mod p)
// with differences formed as (a + p) - b on 12 x u32 limbs (never negative, not reduced).
#include <string.h>

#include <array>
#include <vector>

#include "airs.h"
#include "gadgets.h"

namespace starkhip {
using namespace lay;

namespace lay_g1 {  // src/g1.rs:7-23
constexpr size_t MULRR = FP_MULTIPLICATION_TOTAL_COLUMNS + REDUCTION_TOTAL + RANGE_CHECK_TOTAL;
constexpr size_t ADDSUB = FP_ADDITION_TOTAL + FP_SUBTRACTION_TOTAL;
constexpr size_t X1 = 0, Y1 = 12, X2 = 24, Y2 = 36, X3 = 48, Y3 = 60;
constexpr size_t X2_X1_DIFF = 72;
constexpr size_t Y2_Y1_DIFF = X2_X1_DIFF + ADDSUB;
constexpr size_t X2_X1_SQ = Y2_Y1_DIFF + ADDSUB;
constexpr size_t Y2_Y1_SQ = X2_X1_SQ + MULRR;
constexpr size_t X1_X2_X3_SUM = Y2_Y1_SQ + MULRR;
constexpr size_t X1_X2_X3_X2_X1_SQ = X1_X2_X3_SUM + 2 * FP_ADDITION_TOTAL;
constexpr size_t Y1_Y3 = X1_X2_X3_X2_X1_SQ + MULRR;
constexpr size_t X1_X3 = Y1_Y3 + FP_ADDITION_TOTAL;
constexpr size_t Y1_Y3_X2_X1 = X1_X3 + ADDSUB;
constexpr size_t Y2_Y1_X1_X3 = Y1_Y3_X2_X1 + MULRR;
constexpr size_t TOT_COL = Y2_Y1_X1_X3 + MULRR;
}  // namespace lay_g1

namespace lay_eccagg {  // src/ecc_aggregate.rs:7-21
constexpr size_t NUM_POINTS = 512;
constexpr size_t ROW_NUM = 0;
constexpr size_t PIS_IDX = ROW_NUM + 12;
constexpr size_t A_IS_INF = PIS_IDX + NUM_POINTS;
constexpr size_t B_IS_INF = A_IS_INF + 1;
constexpr size_t OP = B_IS_INF + 1;
constexpr size_t COLUMNS = OP + lay_g1::TOT_COL;
constexpr size_t POINTS = 0;
constexpr size_t BITS = POINTS + 24 * NUM_POINTS;
constexpr size_t RES = BITS + NUM_POINTS;
constexpr size_t PUBLIC_INPUTS = RES + 24;
static_assert(COLUMNS == 3339, "README.md:40 of the reference: 3339 columns");
}  // namespace lay_eccagg

// ---------------------------------------------------------------- big-limb helpers for the unreduced intermediates
static L12 add12(const L12& a, const L12& b) {  // a + b, must fit 384 bits
    L12 r;
    uint64_t c = 0;
    for (int i = 0; i < 12; i++) {
        uint64_t s = (uint64_t)a[i] + b[i] + c;
        r[i] = (uint32_t)s;
        c = s >> 32;
    }
    return r;
}
static L12 sub12(const L12& a, const L12& b) {  // a - b, a >= b
    L12 r;
    int64_t br = 0;
    for (int i = 0; i < 12; i++) {
        int64_t d = (int64_t)a[i] - b[i] - br;
        br = d < 0;
        r[i] = (uint32_t)(d + (br ? ((int64_t)1 << 32) : 0));
    }
    return r;
}
static L24 mul12(const L12& a, const L12& b) {
    L24 r{};
    for (int i = 0; i < 12; i++) {
        uint64_t c = 0;
        for (int j = 0; j < 12; j++) {
            uint64_t t = (uint64_t)a[i] * b[j] + r[i + j] + c;
            r[i + j] = (uint32_t)t;
            c = t >> 32;
        }
        r[i + 12] = (uint32_t)c;
    }
    return r;
}

// src/g1.rs:26-255.  Returns the sum.
static void fill_trace_g1_addition(Trace& t, const Fp pt1[2], const Fp pt2[2], size_t start_row, size_t col, Fp out[2]) {
    using namespace lay_g1;
    const Fp dy = pt2[1] - pt1[1], dx = pt2[0] - pt1[0];
    const Fp lambda = dy / dx;
    const Fp x3_fp = lambda * lambda - pt2[0] - pt1[0];
    const Fp y3_fp = lambda * (pt1[0] - x3_fp) - pt1[1];
    const size_t end_row = start_row + 11;
    const L12 &x1 = pt1[0].l, &y1 = pt1[1].l, &x2 = pt2[0].l, &y2 = pt2[1].l, &x3 = x3_fp.l, &y3 = y3_fp.l;
    const L12& p = modulus_limbs();
    {
        RowSpan rows_(t, end_row - start_row + 1);
        t.put(start_row, col + X1, x1);
        t.put(start_row, col + Y1, y1);
        t.put(start_row, col + X2, x2);
        t.put(start_row, col + Y2, y2);
        t.put(start_row, col + X3, x3);
        t.put(start_row, col + Y3, y3);
    }
    auto mul_block = [&](const L12& a, const L12& b, size_t c) {  // multiplication + reduction + range check of the remainder
        fill_multiplication_trace_no_mod_reduction(t, a, b, start_row, end_row, c);
        const L12 rem = fill_reduction_trace(t, mul12(a, b), start_row, end_row, c + FP_MULTIPLICATION_TOTAL_COLUMNS);
        fill_range_check_trace(t, rem, end_row, c + FP_MULTIPLICATION_TOTAL_COLUMNS + REDUCTION_TOTAL);
        return rem;
    };
    const L12 x2_mod = add12(x2, p), y2_mod = add12(y2, p), x1_mod = add12(x1, p);
    const L12 x2_x1 = sub12(x2_mod, x1), y2_y1 = sub12(y2_mod, y1);
    {
        RowSpan rows_(t, end_row - start_row + 1);
        fill_trace_addition_fp(t, x2, p, start_row, col + X2_X1_DIFF);
        fill_trace_subtraction_fp(t, x2_mod, x1, start_row, col + X2_X1_DIFF + FP_ADDITION_TOTAL);
        fill_trace_addition_fp(t, y2, p, start_row, col + Y2_Y1_DIFF);
        fill_trace_subtraction_fp(t, y2_mod, y1, start_row, col + Y2_Y1_DIFF + FP_ADDITION_TOTAL);
    }
    const L12 x2_x1_sq = mul_block(x2_x1, x2_x1, col + X2_X1_SQ);
    const L12 y2_y1_sq = mul_block(y2_y1, y2_y1, col + Y2_Y1_SQ);
    const L12 x1_x2 = add12(x1, x2), x1_x2_x3 = add12(x1_x2, x3);
    {
        RowSpan rows_(t, end_row - start_row + 1);
        fill_trace_addition_fp(t, x1, x2, start_row, col + X1_X2_X3_SUM);
        fill_trace_addition_fp(t, x1_x2, x3, start_row, col + X1_X2_X3_SUM + FP_ADDITION_TOTAL);
    }
    const L12 lhs1 = mul_block(x1_x2_x3, x2_x1_sq, col + X1_X2_X3_X2_X1_SQ);
    if (lhs1 != y2_y1_sq) throw std::runtime_error("g1 addition: slope identity does not hold");
    const L12 y1_y3 = add12(y1, y3), x1_x3 = sub12(x1_mod, x3);
    {
        RowSpan rows_(t, end_row - start_row + 1);
        fill_trace_addition_fp(t, y1, y3, start_row, col + Y1_Y3);
        fill_trace_addition_fp(t, x1, p, start_row, col + X1_X3);
        fill_trace_subtraction_fp(t, x1_mod, x3, start_row, col + X1_X3 + FP_ADDITION_TOTAL);
    }
    const L12 lhs2 = mul_block(y1_y3, x2_x1, col + Y1_Y3_X2_X1);
    const L12 rhs2 = mul_block(y2_y1, x1_x3, col + Y2_Y1_X1_X3);
    if (lhs2 != rhs2) throw std::runtime_error("g1 addition: y identity does not hold");
    out[0] = x3_fp;
    out[1] = y3_fp;
}

// src/g1.rs:257-604 (bit_selector = None)
static void add_g1_addition_constraints(CS& cs, size_t sc) {
    using namespace lay_g1;
    const Expr bs = CS::one();
    const L12& p = modulus_limbs();
    const size_t RED = FP_MULTIPLICATION_TOTAL_COLUMNS, RC = FP_MULTIPLICATION_TOTAL_COLUMNS + REDUCTION_TOTAL;
    {
        const Expr g = bs * cs.L(sc + X2_X1_SQ + MULTIPLICATION_SELECTOR_OFFSET);
        for (size_t i = 0; i < 12; i++)
            for (size_t c : {X1, Y1, X2, Y2, X3, Y3}) cs.ct(g * (cs.L(sc + c + i) - cs.N(sc + c + i)));
    }
    // (a + p) - b: an addition whose y input is the modulus, then a subtraction fed by its sum
    auto diff_block = [&](size_t blk, size_t a_col, size_t b_col) {
        const Expr ga = bs * cs.L(sc + blk + FP_ADDITION_CHECK_OFFSET);
        for (size_t i = 0; i < 12; i++) {
            cs.c(ga * (cs.L(sc + blk + FP_ADDITION_X_OFFSET + i) - cs.L(sc + a_col + i)));
            cs.c(ga * (cs.L(sc + blk + FP_ADDITION_Y_OFFSET + i) - CS::K(p[i])));
        }
        add_addition_fp_constraints(cs, sc + blk, bs);
        const size_t sub = blk + FP_ADDITION_TOTAL;
        const Expr gs = bs * cs.L(sc + sub + FP_SUBTRACTION_CHECK_OFFSET);
        for (size_t i = 0; i < 12; i++) {
            cs.c(gs * (cs.L(sc + sub + FP_SUBTRACTION_X_OFFSET + i) - cs.L(sc + blk + FP_ADDITION_SUM_OFFSET + i)));
            cs.c(gs * (cs.L(sc + sub + FP_SUBTRACTION_Y_OFFSET + i) - cs.L(sc + b_col + i)));
        }
        add_subtraction_fp_constraints(cs, sc + sub, bs);
    };
    // plain addition block with both inputs wired
    auto add_block = [&](size_t blk, size_t a_col, size_t b_col) {
        const Expr ga = bs * cs.L(sc + blk + FP_ADDITION_CHECK_OFFSET);
        for (size_t i = 0; i < 12; i++) {
            cs.c(ga * (cs.L(sc + blk + FP_ADDITION_X_OFFSET + i) - cs.L(sc + a_col + i)));
            cs.c(ga * (cs.L(sc + blk + FP_ADDITION_Y_OFFSET + i) - cs.L(sc + b_col + i)));
        }
        add_addition_fp_constraints(cs, sc + blk, bs);
    };
    // multiplication + reduction + range check with both inputs wired
    auto mul_block = [&](size_t blk, size_t a_col, size_t b_col) {
        const Expr gm = bs * cs.L(sc + blk + MULTIPLICATION_SELECTOR_OFFSET);
        for (size_t i = 0; i < 12; i++) {
            cs.c(gm * (cs.L(sc + blk + X_INPUT_OFFSET + i) - cs.L(sc + a_col + i)));
            cs.c(gm * (cs.L(sc + blk + Y_INPUT_OFFSET + i) - cs.L(sc + b_col + i)));
        }
        add_multiplication_constraints(cs, sc + blk, bs);
        const Expr gr = bs * cs.L(sc + blk + RC + RANGE_CHECK_SELECTOR_OFFSET);
        for (size_t i = 0; i < 24; i++) cs.c(gr * (cs.L(sc + blk + SUM_OFFSET + i) - cs.L(sc + blk + RED + REDUCE_X_OFFSET + i)));
        add_reduce_constraints(cs, sc + blk + RED, sc + blk + MULTIPLICATION_SELECTOR_OFFSET, bs);
        add_range_check_constraints(cs, sc + blk + RC, bs);
    };
    auto equal_reduced = [&](size_t blk_a, size_t blk_b) {
        const Expr g = bs * cs.L(sc + blk_a + MULTIPLICATION_SELECTOR_OFFSET);
        for (size_t i = 0; i < 12; i++) cs.c(g * (cs.L(sc + blk_a + RED + REDUCED_OFFSET + i) - cs.L(sc + blk_b + RED + REDUCED_OFFSET + i)));
    };
    const size_t X2_X1 = X2_X1_DIFF + FP_ADDITION_TOTAL + FP_SUBTRACTION_DIFF_OFFSET;
    const size_t Y2_Y1 = Y2_Y1_DIFF + FP_ADDITION_TOTAL + FP_SUBTRACTION_DIFF_OFFSET;
    diff_block(X2_X1_DIFF, X2, X1);
    diff_block(Y2_Y1_DIFF, Y2, Y1);
    mul_block(X2_X1_SQ, X2_X1, X2_X1);
    mul_block(Y2_Y1_SQ, Y2_Y1, Y2_Y1);
    add_block(X1_X2_X3_SUM, X1, X2);
    add_block(X1_X2_X3_SUM + FP_ADDITION_TOTAL, X1_X2_X3_SUM + FP_ADDITION_SUM_OFFSET, X3);
    mul_block(X1_X2_X3_X2_X1_SQ, X1_X2_X3_SUM + FP_ADDITION_TOTAL + FP_ADDITION_SUM_OFFSET, X2_X1_SQ + RED + REDUCED_OFFSET);
    equal_reduced(X1_X2_X3_X2_X1_SQ, Y2_Y1_SQ);
    add_block(Y1_Y3, Y1, Y3);
    diff_block(X1_X3, X1, X3);
    mul_block(Y1_Y3_X2_X1, Y1_Y3 + FP_ADDITION_SUM_OFFSET, X2_X1);
    mul_block(Y2_Y1_X1_X3, Y2_Y1, X1_X3 + FP_ADDITION_TOTAL + FP_SUBTRACTION_DIFF_OFFSET);
    equal_reduced(Y2_Y1_X1_X3, Y1_Y3_X2_X1);
}

AirProgram build_air_ecc_aggregate() {
    using namespace lay_eccagg;
    using namespace lay_g1;
    AirBuilder b(COLUMNS, PUBLIC_INPUTS, 4);
    CS cs(b);
    auto L = [&](size_t c) { return cs.L(c); };
    auto N = [&](size_t c) { return cs.N(c); };
    const Expr one = CS::one();
    // src/ecc_aggregate.rs:104-117: the row counter is a one-hot that rotates with period 12
    for (size_t i = 0; i < 12; i++) cs.cf(i == 0 ? L(ROW_NUM) - one : L(ROW_NUM + i));
    for (size_t i = 0; i < 12; i++) cs.ct(L(ROW_NUM + i) - N(ROW_NUM + (i + 1) % 12));
    // :119-141 which public key an addition consumes: a one-hot that advances by one per 12-row block
    for (size_t i = 0; i < NUM_POINTS; i++) cs.cf(i < 2 ? L(PIS_IDX + i) - one : L(PIS_IDX + i));
    const size_t LASTP = PIS_IDX + NUM_POINTS - 1;
    for (size_t i = 1; i + 1 < NUM_POINTS; i++) cs.ct((one - L(LASTP)) * N(ROW_NUM) * (L(PIS_IDX + i) - N(PIS_IDX + i + 1)));
    for (size_t i = 0; i < NUM_POINTS; i++) cs.ct(L(LASTP) * N(ROW_NUM) * N(PIS_IDX + i));
    // :143-159 first addition takes points 0 and 1 and their bits
    for (size_t i = 0; i < 12; i++) {
        cs.cf(L(OP + X1 + i) - b.PI(POINTS + i));
        cs.cf(L(OP + Y1 + i) - b.PI(POINTS + i + 12));
        cs.cf(L(OP + X2 + i) - b.PI(POINTS + 24 + i));
        cs.cf(L(OP + Y2 + i) - b.PI(POINTS + 24 + i + 12));
    }
    cs.cf(one - L(A_IS_INF) - b.PI(BITS));
    cs.cf(one - L(B_IS_INF) - b.PI(BITS + 1));
    // :161-181 every later block takes point idx as its second operand
    for (size_t idx = 2; idx < NUM_POINTS; idx++) {
        const Expr g = N(ROW_NUM) * N(PIS_IDX + idx);
        for (size_t i = 0; i < 12; i++) {
            cs.ct(g * (N(OP + X2 + i) - b.PI(POINTS + 24 * idx + i)));
            cs.ct(g * (N(OP + Y2 + i) - b.PI(POINTS + 24 * idx + i + 12)));
        }
        cs.ct(g * (one - N(B_IS_INF) - b.PI(BITS + idx)));
    }
    // :183-208 operands and result are constant inside a block
    for (size_t i = 0; i < 12; i++)
        for (size_t c : {X1, Y1, X2, Y2, X3, Y3}) cs.ct((one - N(ROW_NUM)) * (L(OP + c + i) - N(OP + c + i)));
    // :210-221 infinity flags
    cs.c(L(A_IS_INF) * (one - L(A_IS_INF)));
    cs.c(L(B_IS_INF) * (one - L(B_IS_INF)));
    cs.c(L(A_IS_INF) * L(B_IS_INF));
    cs.ct((one - N(ROW_NUM)) * (L(A_IS_INF) - N(A_IS_INF)));
    cs.ct((one - N(ROW_NUM)) * (L(B_IS_INF) - N(B_IS_INF)));
    // the running sum: operand 2 if operand 1 is infinity, operand 1 if operand 2 is, else the addition's result
    auto running = [&](size_t xcol1, size_t xcol2, size_t xcol3, size_t i) {
        return L(A_IS_INF) * L(OP + xcol2 + i) + L(B_IS_INF) * L(OP + xcol1 + i) + (one - L(A_IS_INF) - L(B_IS_INF)) * L(OP + xcol3 + i);
    };
    // :223-244 it becomes the next block's first operand
    for (size_t i = 0; i < 12; i++) {
        cs.ct(N(ROW_NUM) * (one - L(LASTP)) * (running(X1, X2, X3, i) - N(OP + X1 + i)));
        cs.ct(N(ROW_NUM) * (one - L(LASTP)) * (running(Y1, Y2, Y3, i) - N(OP + Y1 + i)));
    }
    add_g1_addition_constraints(cs, OP);  // :246
    // :248-269 after the last block it is the public aggregate
    for (size_t i = 0; i < 12; i++) {
        cs.ct(N(ROW_NUM) * L(LASTP) * (running(X1, X2, X3, i) - b.PI(RES + i)));
        cs.ct(N(ROW_NUM) * L(LASTP) * (running(Y1, Y2, Y3, i) - b.PI(RES + i + 12)));
    }
    return b.finish();
}

// src/ecc_aggregate.rs:39-84 without the trace: the aggregate of the points whose bit is set
static void g1_aggregate(const Fp (*pts)[2], const uint8_t* bits, size_t n, Trace* t, Fp res[2]) {
    using namespace lay_eccagg;
    Fp acc[2], sum[2];
    size_t row = 0;
    if (t) {
        fill_trace_g1_addition(*t, pts[0], pts[1], row, OP, sum);
    } else {
        const Fp lambda = (pts[1][1] - pts[0][1]) / (pts[1][0] - pts[0][0]);
        sum[0] = lambda * lambda - pts[1][0] - pts[0][0];
        sum[1] = lambda * (pts[0][0] - sum[0]) - pts[0][1];
    }
    acc[0] = sum[0];
    acc[1] = sum[1];
    if (!bits[0]) {
        acc[0] = pts[1][0];
        acc[1] = pts[1][1];
    } else if (!bits[1]) {
        acc[0] = pts[0][0];
        acc[1] = pts[0][1];
    }
    if (t)
        for (size_t r = row; r < row + 12; r++) {
            t->at(r, A_IS_INF) = !bits[0];
            t->at(r, B_IS_INF) = !bits[1];
        }
    for (size_t i = 2; i < n; i++) {
        row += 12;
        if (t) {
            fill_trace_g1_addition(*t, acc, pts[i], row, OP, sum);
            for (size_t r = row; r < row + 12; r++) {
                t->at(r, A_IS_INF) = 0;
                t->at(r, B_IS_INF) = !bits[i];
            }
        } else if (bits[i]) {
            const Fp lambda = (pts[i][1] - acc[1]) / (pts[i][0] - acc[0]);
            sum[0] = lambda * lambda - pts[i][0] - acc[0];
            sum[1] = lambda * (acc[0] - sum[0]) - acc[1];
        }
        if (bits[i]) {
            acc[0] = sum[0];
            acc[1] = sum[1];
        }
    }
    res[0] = acc[0];
    res[1] = acc[1];
}

}  // namespace starkhip

using namespace starkhip;

static void load_points(const uint32_t* points, std::vector<std::array<Fp, 2>>& out) {
    out.resize(lay_eccagg::NUM_POINTS);
    for (size_t i = 0; i < lay_eccagg::NUM_POINTS; i++) {
        for (int k = 0; k < 12; k++) {
            out[i][0].l[k] = points[24 * i + k];
            out[i][1].l[k] = points[24 * i + 12 + k];
        }
    }
}

extern "C" int starkhip_native_g1_aggregate(const uint32_t* points, const uint8_t* bits, uint32_t out[24]) {
    try {
        std::vector<std::array<Fp, 2>> pts;
        load_points(points, pts);
        Fp res[2];
        g1_aggregate((const Fp(*)[2])pts.data(), bits, lay_eccagg::NUM_POINTS, nullptr, res);
        for (int k = 0; k < 12; k++) {
            out[k] = res[0].l[k];
            out[12 + k] = res[1].l[k];
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "starkhip_native_g1_aggregate: %s\n", e.what());
        return STARKHIP_ERR_BAD_SHAPE;
    }
    return STARKHIP_OK;
}

extern "C" int starkhip_trace_ecc_aggregate(const uint32_t* points, const uint8_t* bits, uint64_t* trace, size_t n_rows, uint64_t* public_inputs) {
    using namespace lay_eccagg;
    if (n_rows <= (NUM_POINTS - 1) * 12 || (n_rows & (n_rows - 1))) return STARKHIP_ERR_BAD_SHAPE;  // "stark doesn't have enough rows"
    try {
        std::vector<std::array<Fp, 2>> pts;
        load_points(points, pts);
        Trace t = open_trace(trace, n_rows, COLUMNS);
        for (size_t i = 0; i < n_rows; i++) t.at(i, ROW_NUM + i % 12) = 1;
        size_t row = 0;
        for (size_t i = 0; i < NUM_POINTS; i++) {
            if (i >= 2) row += 12;
            for (size_t r = row; r < row + 12; r++) t.at(r, PIS_IDX + i) = 1;
        }
        Fp res[2];
        g1_aggregate((const Fp(*)[2])pts.data(), bits, NUM_POINTS, &t, res);
        for (size_t i = 0; i < 24 * NUM_POINTS; i++) public_inputs[POINTS + i] = points[i];
        for (size_t i = 0; i < NUM_POINTS; i++) public_inputs[BITS + i] = bits[i] ? 1 : 0;
        for (size_t i = 0; i < 12; i++) {
            public_inputs[RES + i] = res[0].l[i];
            public_inputs[RES + 12 + i] = res[1].l[i];
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "starkhip_trace_ecc_aggregate: %s\n", e.what());
        return STARKHIP_ERR_BAD_SHAPE;
    }
    return STARKHIP_OK;
}
