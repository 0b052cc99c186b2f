// PairingPrecompStark: the 68 line-function coefficient triples of a G2 point (29376 columns x 1024 rows, degree 4).
// Restates /root/reference/src/calc_pairing_precomp.rs: layout (:49-131), generate_trace (:150-348),
// eval_packed_generic (:376-2123), constraint_degree (:3320-3322); public inputs as built by
// calc_pairing_precomp, src/aggregate_proof.rs:36-55.  The bit-0 ("doubling") and bit-1 ("addition") steps are two
// layouts over the SAME column window, selected by BIT1_SELECTOR.
#include <stdio.h>

#include "airs.h"
#include "gadgets.h"
#include "wiring.h"

namespace starkhip {
using namespace lay;
namespace PC = lay_precomp;
using namespace wire;
using bls::Fp;
using bls::Fp2;

namespace {
inline Loc2 mulb_out(size_t t) { return {t + MULTIPLY_B_Z0_REDUCE_OFFSET + REDUCED_OFFSET, t + MULTIPLY_B_Z1_REDUCE_OFFSET + REDUCED_OFFSET}; }
inline Loc2 neg_y(size_t t) { return {t + FP2_ADDITION_0_OFFSET + FP_ADDITION_Y_OFFSET, t + FP2_ADDITION_1_OFFSET + FP_ADDITION_Y_OFFSET}; }

// fp2 x fp block fed from an Fp2 location and a constant Fp: per i < 12: X[i] - v.c0, X[12 + i] - v.c1, Y[i] - k[i]
// (`small`: the reference writes the constant as "Y[0] - c, Y[i>0] == 0", same values)
void fp2fp_const_in(CS& cs, const Expr& bs, size_t blk, Loc2 v, const bls::L12& k) {
    const Expr g = bs * cs.L(blk + FP2_FP_MUL_SELECTOR_OFFSET);
    for (size_t i = 0; i < 12; i++) {
        cs.c(g * (cs.L(blk + FP2_FP_X_INPUT_OFFSET + i) - cs.L(v.c0 + i)));
        cs.c(g * (cs.L(blk + FP2_FP_X_INPUT_OFFSET + 12 + i) - cs.L(v.c1 + i)));
        cs.c(g * (cs.L(blk + FP2_FP_Y_INPUT_OFFSET + i) - CS::K(k[i])));
    }
    add_fp2_fp_mul_constraints(cs, blk, bs);
}
// fp2-mul block: X from a 24-limb run or an Fp2 location, likewise Y; per i < 24: X link then Y link
void mul_in_mixed(CS& cs, const Expr& bs, size_t t, bool x_is_loc, Loc2 xl, size_t xcol, bool y_is_loc, Loc2 yl, size_t ycol) {
    const Expr g = bs * cs.L(t + FP2_FP2_SELECTOR_OFFSET);
    for (size_t i = 0; i < 24; i++) {
        const size_t xs = x_is_loc ? (i < 12 ? xl.c0 + i : xl.c1 + i - 12) : xcol + i;
        const size_t ys = y_is_loc ? (i < 12 ? yl.c0 + i : yl.c1 + i - 12) : ycol + i;
        cs.c(g * (cs.L(t + FP2_FP2_X_INPUT_OFFSET + i) - cs.L(xs)));
        cs.c(g * (cs.L(t + FP2_FP2_Y_INPUT_OFFSET + i) - cs.L(ys)));
    }
}
bls::L12 small_fp(uint32_t v) { return Fp::from_u32(v).l; }
}  // namespace

AirProgram build_air_pairing_precomp() {
    AirBuilder b(PC::COLUMNS, PC::PUBLIC_INPUTS, 4);
    CS cs(b);
    const size_t ZZ = PC::Z_MULT_Z_INV_OFFSET, XZ = PC::X_MULT_Z_INV_OFFSET, YZ = PC::Y_MULT_Z_INV_OFFSET;
    const size_t QX = PC::QX_OFFSET, QY = PC::QY_OFFSET, QZ = PC::QZ_OFFSET, RX = PC::RX_OFFSET, RY = PC::RY_OFFSET, RZ = PC::RZ_OFFSET;
    const Expr one = CS::one();

    // ---- z * z^-1 == 1, inputs tied to the public inputs on the first row (:388-519)
    for (size_t i = 0; i < 12; i++) {
        cs.cf(cs.L(ZZ + Z1_REDUCE_OFFSET + REDUCED_OFFSET + i) - CS::K(i == 0 ? 1 : 0));
        cs.cf(cs.L(ZZ + Z2_REDUCE_OFFSET + REDUCED_OFFSET + i));
    }
    for (size_t i = 0; i < 12; i++) {
        cs.cf(cs.L(ZZ + FP2_FP2_X_INPUT_OFFSET + i) - b.PI(PC::Z0_PUBLIC_INPUTS_OFFSET + i));
        cs.cf(cs.L(ZZ + FP2_FP2_X_INPUT_OFFSET + 12 + i) - b.PI(PC::Z1_PUBLIC_INPUTS_OFFSET + i));
    }
    add_fp2_mul_constraints(cs, ZZ, one);
    auto by_z_inv = [&](size_t blk, size_t pi0, size_t pi1) {
        for (size_t i = 0; i < 12; i++) {
            cs.cf(cs.L(blk + FP2_FP2_X_INPUT_OFFSET + i) - b.PI(pi0 + i));
            cs.cf(cs.L(blk + FP2_FP2_X_INPUT_OFFSET + 12 + i) - b.PI(pi1 + i));
            cs.cf(cs.L(blk + FP2_FP2_Y_INPUT_OFFSET + i) - cs.L(ZZ + X_0_Y_0_MULTIPLICATION_OFFSET + Y_INPUT_OFFSET + i));
            cs.cf(cs.L(blk + FP2_FP2_Y_INPUT_OFFSET + 12 + i) - cs.L(ZZ + X_0_Y_1_MULTIPLICATION_OFFSET + Y_INPUT_OFFSET + i));
        }
        add_fp2_mul_constraints(cs, blk, one);
    };
    by_z_inv(XZ, PC::X0_PUBLIC_INPUTS_OFFSET, PC::X1_PUBLIC_INPUTS_OFFSET);
    by_z_inv(YZ, PC::Y0_PUBLIC_INPUTS_OFFSET, PC::Y1_PUBLIC_INPUTS_OFFSET);
    for (size_t i = 0; i < 12; i++) {
        cs.cf(cs.L(XZ + Z1_REDUCE_OFFSET + REDUCED_OFFSET + i) - cs.L(QX + i));
        cs.cf(cs.L(XZ + Z2_REDUCE_OFFSET + REDUCED_OFFSET + i) - cs.L(QX + 12 + i));
        cs.cf(cs.L(YZ + Z1_REDUCE_OFFSET + REDUCED_OFFSET + i) - cs.L(QY + i));
        cs.cf(cs.L(YZ + Z2_REDUCE_OFFSET + REDUCED_OFFSET + i) - cs.L(QY + 12 + i));
        if (i == 0) cs.cf(cs.L(QZ + i) - one);
        else cs.cf(cs.L(QZ + i));
        cs.cf(cs.L(QZ + 12 + i));
    }
    for (size_t i = 0; i < 24; i++) {
        cs.ct(cs.L(QX + i) - cs.N(QX + i));
        cs.ct(cs.L(QY + i) - cs.N(QY + i));
        cs.ct(cs.L(QZ + i) - cs.N(QZ + i));
    }
    // ---- R registers (:527-605)
    const Expr bit1 = cs.L(PC::BIT1_SELECTOR_OFFSET);
    const Expr bit0 = one - bit1;
    const Expr first_loop = cs.L(PC::FIRST_LOOP_SELECTOR_OFFSET), first_row = cs.L(PC::FIRST_ROW_SELECTOR_OFFSET);
    const Expr nfl = cs.N(PC::FIRST_LOOP_SELECTOR_OFFSET), nfr = cs.N(PC::FIRST_ROW_SELECTOR_OFFSET);
    const size_t NRX = PC::NEW_RX_OFFSET, NRY = PC::NEW_RY_OFFSET, NRZ = PC::NEW_RZ_OFFSET;
    const size_t B1RX = PC::BIT1_RX_CALC_OFFSET, B1RY = PC::BIT1_RY_CALC_OFFSET, B1RZ = PC::BIT1_RZ_CALC_OFFSET;
    for (size_t i = 0; i < 24; i++) {
        cs.c(first_loop * first_row * (cs.L(RX + i) - cs.L(QX + i)));
        cs.c(first_loop * first_row * (cs.L(RY + i) - cs.L(QY + i)));
        cs.c(first_loop * first_row * (cs.L(RZ + i) - cs.L(QZ + i)));
        const size_t h = i < 12 ? 0 : 1, ii = i % 12;
        const Loc2 l0x = fp2fp_out(NRX), l0y = subred_out(NRY), l0z = mul_out(NRZ), l1x = mul_out(B1RX), l1y = subred_out(B1RY), l1z = mul_out(B1RZ);
        auto pick = [&](Loc2 l) { return (h ? l.c1 : l.c0) + ii; };
        cs.c(bit0 * (one - nfl) * nfr * (cs.N(RX + i) - cs.L(pick(l0x))));
        cs.c(bit0 * (one - nfl) * nfr * (cs.N(RY + i) - cs.L(pick(l0y))));
        cs.c(bit0 * (one - nfl) * nfr * (cs.N(RZ + i) - cs.L(pick(l0z))));
        cs.c(bit1 * (one - nfl) * nfr * (cs.N(RX + i) - cs.L(pick(l1x))));
        cs.c(bit1 * (one - nfl) * nfr * (cs.N(RY + i) - cs.L(pick(l1y))));
        cs.c(bit1 * (one - nfl) * nfr * (cs.N(RZ + i) - cs.L(pick(l1z))));
        cs.ct((one - nfr) * (cs.L(RX + i) - cs.N(RX + i)));
        cs.ct((one - nfr) * (cs.L(RY + i) - cs.N(RY + i)));
        cs.ct((one - nfr) * (cs.L(RZ + i) - cs.N(RZ + i)));
    }
    // ---- ell coefficients against the public inputs (:607-651)
    const size_t T0 = PC::T0_CALC_OFFSET, T1 = PC::T1_CALC_OFFSET, X0 = PC::X0_CALC_OFFSET, T2 = PC::T2_CALC_OFFSET, T3 = PC::T3_CALC_OFFSET;
    const size_t X1 = PC::X1_CALC_OFFSET, T4 = PC::T4_CALC_OFFSET, X2 = PC::X2_CALC_OFFSET, X3 = PC::X3_CALC_OFFSET, X4 = PC::X4_CALC_OFFSET;
    const size_t X5 = PC::X5_CALC_OFFSET, X6 = PC::X6_CALC_OFFSET, X7 = PC::X7_CALC_OFFSET, X8 = PC::X8_CALC_OFFSET, X9 = PC::X9_CALC_OFFSET;
    const size_t X10 = PC::X10_CALC_OFFSET, X11 = PC::X11_CALC_OFFSET, X12 = PC::X12_CALC_OFFSET, X13 = PC::X13_CALC_OFFSET;
    const size_t BT[19] = {PC::BIT1_T0_CALC_OFFSET,  PC::BIT1_T1_CALC_OFFSET,  PC::BIT1_T2_CALC_OFFSET,  PC::BIT1_T3_CALC_OFFSET,  PC::BIT1_T4_CALC_OFFSET,
                           PC::BIT1_T5_CALC_OFFSET,  PC::BIT1_T6_CALC_OFFSET,  PC::BIT1_T7_CALC_OFFSET,  PC::BIT1_T8_CALC_OFFSET,  PC::BIT1_T9_CALC_OFFSET,
                           PC::BIT1_T10_CALC_OFFSET, PC::BIT1_T11_CALC_OFFSET, PC::BIT1_T12_CALC_OFFSET, PC::BIT1_T13_CALC_OFFSET, PC::BIT1_T14_CALC_OFFSET,
                           PC::BIT1_T15_CALC_OFFSET, PC::BIT1_T16_CALC_OFFSET, PC::BIT1_T17_CALC_OFFSET, PC::BIT1_T18_CALC_OFFSET};
    {
        const Loc2 a0 = subred_out(X2), a1 = fp2fp_out(X4), a2 = neg_y(X5);
        const Loc2 c0 = subred_out(BT[6]), c1 = neg_y(BT[7]), c2 = subred_out(BT[3]);
        const size_t src0[6] = {a0.c0, a0.c1, a1.c0, a1.c1, a2.c0, a2.c1}, src1[6] = {c0.c0, c0.c1, c1.c0, c1.c1, c2.c0, c2.c1};
        for (size_t idx = 0; idx < 68; idx++) {
            const Expr sel = cs.L(PC::ELL_COEFFS_IDX_OFFSET + idx);
            for (size_t i = 0; i < 12; i++) {
                for (size_t h = 0; h < 6; h++) cs.c(bit0 * sel * (cs.L(src0[h] + i) - b.PI(PC::ELL_COEFFS_PUBLIC_INPUTS_OFFSET + idx * 72 + i + 12 * h)));
                for (size_t h = 0; h < 6; h++) cs.c(bit1 * sel * (cs.L(src1[h] + i) - b.PI(PC::ELL_COEFFS_PUBLIC_INPUTS_OFFSET + idx * 72 + i + 12 * h)));
            }
        }
    }
    // ---- bit-0 step (:653-1395), values as in native calc_precomp_stuff_loop0 (src/native.rs:293-326)
    const bls::L12 k_half = bls::mod_inverse_of_two().l;
    mul_in24(cs, bit0, T0, RY, RY, true);
    add_fp2_mul_constraints(cs, T0, bit0);
    mul_in24(cs, bit0, T1, RZ, RZ, true);
    add_fp2_mul_constraints(cs, T1, bit0);
    fp2fp_const_in(cs, bit0, X0, mul_out(T1), small_fp(3));
    {
        const Loc2 v = fp2fp_out(X0);
        cs.links(false, bit0, 12, {{T2 + MULTIPLY_B_SELECTOR_OFFSET, v.c0, T2 + MULTIPLY_B_X_OFFSET}, {T2 + MULTIPLY_B_SELECTOR_OFFSET, v.c1, T2 + MULTIPLY_B_X_OFFSET + 12}});
    }
    add_multiply_by_b_constraints(cs, T2, bit0);
    fp2fp_const_in(cs, bit0, T3, mulb_out(T2), small_fp(3));
    mul_in24(cs, bit0, X1, RY, RZ, true);
    add_fp2_mul_constraints(cs, X1, bit0);
    fp2fp_const_in(cs, bit0, T4, mul_out(X1), small_fp(2));
    sub_in(cs, bit0, X2, mulb_out(T2), mul_out(T0));
    add_subtraction_with_reduction_constraints(cs, X2, bit0);
    mul_in24(cs, bit0, X3, RX, RX, true);
    add_fp2_mul_constraints(cs, X3, bit0);
    fp2fp_const_in(cs, bit0, X4, mul_out(X3), small_fp(3));
    {
        const Loc2 v = fp2fp_out(T4);
        const size_t a0 = X5 + FP2_ADDITION_0_OFFSET, a1 = X5 + FP2_ADDITION_1_OFFSET;
        cs.links(false, bit0, 12, {{a0 + FP_ADDITION_CHECK_OFFSET, v.c0, a0 + FP_ADDITION_X_OFFSET}, {a1 + FP_ADDITION_CHECK_OFFSET, v.c1, a1 + FP_ADDITION_X_OFFSET}});
    }
    add_negate_fp2_constraints(cs, X5, bit0);
    sub_in(cs, bit0, X6, mul_out(T0), fp2fp_out(T3));
    add_subtraction_with_reduction_constraints(cs, X6, bit0);
    mul_in24(cs, bit0, X7, RX, RY, true);
    add_fp2_mul_constraints(cs, X7, bit0);
    mul_in(cs, bit0, X8, subred_out(X6), mul_out(X7), true);
    add_fp2_mul_constraints(cs, X8, bit0);
    add_in(cs, bit0, X9, mul_out(T0), fp2fp_out(T3));
    add_addition_with_reduction_constraints(cs, X9, bit0);
    fp2fp_const_in(cs, bit0, X10, addred_out(X9), k_half);
    mul_in(cs, bit0, X11, fp2fp_out(X10), fp2fp_out(X10), true);
    add_fp2_mul_constraints(cs, X11, bit0);
    mul_in(cs, bit0, X12, mulb_out(T2), mulb_out(T2), true);
    add_fp2_mul_constraints(cs, X12, bit0);
    fp2fp_const_in(cs, bit0, X13, mul_out(X12), small_fp(3));
    fp2fp_const_in(cs, bit0, NRX, mul_out(X8), k_half);
    sub_in(cs, bit0, NRY, mul_out(X11), fp2fp_out(X13));
    add_subtraction_with_reduction_constraints(cs, NRY, bit0);
    mul_in(cs, bit0, NRZ, mul_out(T0), fp2fp_out(T4), true);
    add_fp2_mul_constraints(cs, NRZ, bit0);
    // ---- bit-1 step (:1397-2121), values as in native calc_precomp_stuff_loop1 (src/native.rs:328-366)
    const Loc2 none = {0, 0};
    mul_in24(cs, bit1, BT[0], QY, RZ, true);
    add_fp2_mul_constraints(cs, BT[0], bit1);
    sub_in(cs, bit1, BT[1], raw(RY), mul_out(BT[0]));
    add_subtraction_with_reduction_constraints(cs, BT[1], bit1);
    mul_in24(cs, bit1, BT[2], QX, RZ, true);
    add_fp2_mul_constraints(cs, BT[2], bit1);
    sub_in(cs, bit1, BT[3], raw(RX), mul_out(BT[2]));
    add_subtraction_with_reduction_constraints(cs, BT[3], bit1);
    mul_in_mixed(cs, bit1, BT[4], true, subred_out(BT[1]), 0, false, none, QX);
    add_fp2_mul_constraints(cs, BT[4], bit1);
    mul_in_mixed(cs, bit1, BT[5], true, subred_out(BT[3]), 0, false, none, QY);
    add_fp2_mul_constraints(cs, BT[5], bit1);
    sub_in(cs, bit1, BT[6], mul_out(BT[4]), mul_out(BT[5]));
    add_subtraction_with_reduction_constraints(cs, BT[6], bit1);
    {
        const Loc2 v = subred_out(BT[1]);
        const size_t a0 = BT[7] + FP2_ADDITION_0_OFFSET, a1 = BT[7] + FP2_ADDITION_1_OFFSET;
        cs.links(false, bit1, 12, {{a0 + FP_ADDITION_CHECK_OFFSET, v.c0, a0 + FP_ADDITION_X_OFFSET}, {a1 + FP_ADDITION_CHECK_OFFSET, v.c1, a1 + FP_ADDITION_X_OFFSET}});
    }
    add_negate_fp2_constraints(cs, BT[7], bit1);
    mul_in(cs, bit1, BT[8], subred_out(BT[3]), subred_out(BT[3]), true);
    add_fp2_mul_constraints(cs, BT[8], bit1);
    mul_in(cs, bit1, BT[9], mul_out(BT[8]), subred_out(BT[3]), true);
    add_fp2_mul_constraints(cs, BT[9], bit1);
    mul_in(cs, bit1, BT[10], mul_out(BT[8]), raw(RX), true);
    add_fp2_mul_constraints(cs, BT[10], bit1);
    mul_in(cs, bit1, BT[11], subred_out(BT[1]), subred_out(BT[1]), true);
    add_fp2_mul_constraints(cs, BT[11], bit1);
    mul_in(cs, bit1, BT[12], mul_out(BT[11]), raw(RZ), true);
    add_fp2_mul_constraints(cs, BT[12], bit1);
    fp2fp_const_in(cs, bit1, BT[13], mul_out(BT[10]), small_fp(2));
    sub_in(cs, bit1, BT[14], mul_out(BT[9]), fp2fp_out(BT[13]));
    add_subtraction_with_reduction_constraints(cs, BT[14], bit1);
    add_in(cs, bit1, BT[15], subred_out(BT[14]), mul_out(BT[12]));
    add_addition_with_reduction_constraints(cs, BT[15], bit1);
    sub_in(cs, bit1, BT[16], mul_out(BT[10]), addred_out(BT[15]));
    add_subtraction_with_reduction_constraints(cs, BT[16], bit1);
    mul_in(cs, bit1, BT[17], subred_out(BT[16]), subred_out(BT[1]), true);
    add_fp2_mul_constraints(cs, BT[17], bit1);
    mul_in(cs, bit1, BT[18], mul_out(BT[9]), raw(RY), true);
    add_fp2_mul_constraints(cs, BT[18], bit1);
    mul_in(cs, bit1, B1RX, subred_out(BT[3]), addred_out(BT[15]), true);
    add_fp2_mul_constraints(cs, B1RX, bit1);
    sub_in(cs, bit1, B1RY, mul_out(BT[17]), mul_out(BT[18]));
    add_subtraction_with_reduction_constraints(cs, B1RY, bit1);
    mul_in_mixed(cs, bit1, B1RZ, false, none, RZ, true, mul_out(BT[9]), 0);
    add_fp2_mul_constraints(cs, B1RZ, bit1);
    return b.finish();
}

}  // namespace starkhip

using namespace starkhip;

static bls::Fp fp_of(const uint32_t* l) { bls::Fp r; for (int i = 0; i < 12; i++) r.l[i] = l[i]; return r; }
static bls::Fp2 fp2_of(const uint32_t* l) { return bls::Fp2(fp_of(l), fp_of(l + 12)); }

// PairingPrecompStark::generate_trace (:150-348) + public inputs (src/aggregate_proof.rs:36-55)
extern "C" int starkhip_trace_pairing_precomp(const uint32_t qx_[24], const uint32_t qy_[24], const uint32_t qz_[24], uint64_t* trace, size_t n_rows,
                                              uint64_t* public_inputs) {
    if (n_rows < 16 || (n_rows & (n_rows - 1))) return STARKHIP_ERR_BAD_SHAPE;
    try {
        const Fp2 x = fp2_of(qx_), y = fp2_of(qy_), z = fp2_of(qz_);
        Trace t = open_trace(trace, n_rows, PC::COLUMNS);
        const Fp2 z_inv = z.invert();
        const Fp2 qx = x * z.invert(), qy = y * z.invert(), qz = Fp2::one();  // calc_qs (native.rs:283-291)
        const size_t num_coeffs = 68;
        const Fp three = Fp::from_u32(3), two = Fp::from_u32(2), k = bls::mod_inverse_of_two();
        // The running point (rx, ry, rz) and the bit state at the start of every 12-row block come from a native pass first;
        // after that a block's rows depend only on them, so ranges of blocks -- and the three multiplications that span all
        // rows -- are tasks for fill_tasks (trace_tasks.cpp).
        struct Block {
            Fp2 rx, ry, rz;
            bool bit1;
        };
        const size_t n_blocks = n_rows / 12 + 1;  // the last one is cut off by the end of the trace: header rows only
        std::vector<Block> at(n_blocks);
        {
            Fp2 rx = qx, ry = qy, rz = qz;
            int bit_pos = 62;
            bool bit1 = false;
            for (size_t n = 0; n < n_blocks; n++) {
                at[n] = {rx, ry, rz, bit1};
                if ((n + 1) * 12 > n_rows) break;
                if (!bit1) {
                    const std::vector<Fp2> v = bls::calc_precomp_stuff_loop0(rx, ry, rz);
                    rx = v[0]; ry = v[1]; rz = v[2];
                    bit1 = (bls::BLS_X >> bit_pos) & 1;
                    bit_pos = bit1 ? bit_pos : (bit_pos > 0 ? bit_pos - 1 : 0);
                } else {
                    const std::vector<Fp2> w = bls::calc_precomp_stuff_loop1(rx, ry, rz, qx, qy);
                    rx = w[0]; ry = w[1]; rz = w[2];
                    bit1 = false;
                    bit_pos = bit_pos > 0 ? bit_pos - 1 : 0;
                }
            }
        }
        const size_t per_task = t.log && trace_threads() > 1 ? 6 : n_blocks;  // blocks per task
        const size_t n_block_tasks = (n_blocks + per_task - 1) / per_task;
        fill_tasks(t, n_block_tasks + 1, [&](Trace& t, size_t task) {
            if (task == n_block_tasks) {
                // the three global multiplications span ALL rows as one "12-row" gadget call (App. B.4 item 13)
                generate_trace_fp2_mul(t, z, z_inv, 0, n_rows - 1, PC::Z_MULT_Z_INV_OFFSET);
                generate_trace_fp2_mul(t, x, z_inv, 0, n_rows - 1, PC::X_MULT_Z_INV_OFFSET);
                generate_trace_fp2_mul(t, y, z_inv, 0, n_rows - 1, PC::Y_MULT_Z_INV_OFFSET);
                for (size_t row = 0; row < n_rows; row++) {
                    t.put(row, PC::QX_OFFSET, qx);
                    t.put(row, PC::QY_OFFSET, qy);
                    t.put(row, PC::QZ_OFFSET, qz);
                }
                return;
            }
            for (size_t n = task * per_task; n < std::min(n_blocks, (task + 1) * per_task); n++) {
                const Fp2 &rx = at[n].rx, &ry = at[n].ry, &rz = at[n].rz;
                const bool bit1 = at[n].bit1;
                    const size_t start_row = n * 12, end_row = (n + 1) * 12;
                    for (size_t row = start_row; row < std::min(end_row, n_rows); row++) {
                        if (n == 0) t.at(row, PC::FIRST_LOOP_SELECTOR_OFFSET) = 1;
                        t.put(row, PC::RX_OFFSET, rx);
                        t.put(row, PC::RY_OFFSET, ry);
                        t.put(row, PC::RZ_OFFSET, rz);
                        if (bit1) t.at(row, PC::BIT1_SELECTOR_OFFSET) = 1;
                        if (n < num_coeffs) t.at(row, PC::ELL_COEFFS_IDX_OFFSET + n) = 1;
                    }
                    t.at(start_row, PC::FIRST_ROW_SELECTOR_OFFSET) = 1;
                    if (end_row > n_rows) break;  // (only the last block)
                    const size_t r0 = start_row, r1 = end_row - 1;
                    auto rows_sub = [&](const Fp2& a, const Fp2& bb, size_t col) { for (size_t r = r0; r <= r1; r++) fill_trace_subtraction_with_reduction(t, a, bb, r, col); };
                    auto rows_add = [&](const Fp2& a, const Fp2& bb, size_t col) { for (size_t r = r0; r <= r1; r++) fill_trace_addition_with_reduction(t, a, bb, r, col); };
                    auto rows_neg = [&](const Fp2& a, size_t col) { for (size_t r = r0; r <= r1; r++) fill_trace_negate_fp2(t, a, r, col); };
                    if (!bit1) {
                        // v = [new_rx, new_ry, new_rz, t0, t1, x0, t2, t3, x1, t4, x3, x2, x4, x5, x6, x7, x8, x9, x10, x11, x12, x13]
                        const std::vector<Fp2> v = bls::calc_precomp_stuff_loop0(rx, ry, rz);
                        generate_trace_fp2_mul(t, ry, ry, r0, r1, PC::T0_CALC_OFFSET);
                        generate_trace_fp2_mul(t, rz, rz, r0, r1, PC::T1_CALC_OFFSET);
                        fill_trace_fp2_fp_mul(t, v[4], three, r0, r1, PC::X0_CALC_OFFSET);
                        fill_multiply_by_b_trace(t, v[5], r0, r1, PC::T2_CALC_OFFSET);
                        fill_trace_fp2_fp_mul(t, v[6], three, r0, r1, PC::T3_CALC_OFFSET);
                        generate_trace_fp2_mul(t, ry, rz, r0, r1, PC::X1_CALC_OFFSET);
                        fill_trace_fp2_fp_mul(t, v[8], two, r0, r1, PC::T4_CALC_OFFSET);
                        rows_sub(v[6], v[3], PC::X2_CALC_OFFSET);
                        generate_trace_fp2_mul(t, rx, rx, r0, r1, PC::X3_CALC_OFFSET);
                        fill_trace_fp2_fp_mul(t, v[10], three, r0, r1, PC::X4_CALC_OFFSET);
                        rows_neg(v[9], PC::X5_CALC_OFFSET);
                        rows_sub(v[3], v[7], PC::X6_CALC_OFFSET);
                        generate_trace_fp2_mul(t, rx, ry, r0, r1, PC::X7_CALC_OFFSET);
                        generate_trace_fp2_mul(t, v[14], v[15], r0, r1, PC::X8_CALC_OFFSET);
                        rows_add(v[3], v[7], PC::X9_CALC_OFFSET);
                        fill_trace_fp2_fp_mul(t, v[17], k, r0, r1, PC::X10_CALC_OFFSET);
                        generate_trace_fp2_mul(t, v[18], v[18], r0, r1, PC::X11_CALC_OFFSET);
                        generate_trace_fp2_mul(t, v[6], v[6], r0, r1, PC::X12_CALC_OFFSET);
                        fill_trace_fp2_fp_mul(t, v[20], three, r0, r1, PC::X13_CALC_OFFSET);
                        fill_trace_fp2_fp_mul(t, v[16], k, r0, r1, PC::NEW_RX_OFFSET);
                        rows_sub(v[19], v[21], PC::NEW_RY_OFFSET);
                        generate_trace_fp2_mul(t, v[3], v[9], r0, r1, PC::NEW_RZ_OFFSET);
                    } else {
                        // w = [new_rx, new_ry, new_rz, t0, t1, ..., t18]
                        const std::vector<Fp2> w = bls::calc_precomp_stuff_loop1(rx, ry, rz, qx, qy);
                        generate_trace_fp2_mul(t, qy, rz, r0, r1, PC::BIT1_T0_CALC_OFFSET);
                        rows_sub(ry, w[3], PC::BIT1_T1_CALC_OFFSET);
                        generate_trace_fp2_mul(t, qx, rz, r0, r1, PC::BIT1_T2_CALC_OFFSET);
                        rows_sub(rx, w[5], PC::BIT1_T3_CALC_OFFSET);
                        generate_trace_fp2_mul(t, w[4], qx, r0, r1, PC::BIT1_T4_CALC_OFFSET);
                        generate_trace_fp2_mul(t, w[6], qy, r0, r1, PC::BIT1_T5_CALC_OFFSET);
                        rows_sub(w[7], w[8], PC::BIT1_T6_CALC_OFFSET);
                        rows_neg(w[4], PC::BIT1_T7_CALC_OFFSET);
                        generate_trace_fp2_mul(t, w[6], w[6], r0, r1, PC::BIT1_T8_CALC_OFFSET);
                        generate_trace_fp2_mul(t, w[11], w[6], r0, r1, PC::BIT1_T9_CALC_OFFSET);
                        generate_trace_fp2_mul(t, w[11], rx, r0, r1, PC::BIT1_T10_CALC_OFFSET);
                        generate_trace_fp2_mul(t, w[4], w[4], r0, r1, PC::BIT1_T11_CALC_OFFSET);
                        generate_trace_fp2_mul(t, w[14], rz, r0, r1, PC::BIT1_T12_CALC_OFFSET);
                        fill_trace_fp2_fp_mul(t, w[13], two, r0, r1, PC::BIT1_T13_CALC_OFFSET);
                        rows_sub(w[12], w[16], PC::BIT1_T14_CALC_OFFSET);
                        rows_add(w[17], w[15], PC::BIT1_T15_CALC_OFFSET);
                        rows_sub(w[13], w[18], PC::BIT1_T16_CALC_OFFSET);
                        generate_trace_fp2_mul(t, w[19], w[4], r0, r1, PC::BIT1_T17_CALC_OFFSET);
                        generate_trace_fp2_mul(t, w[12], ry, r0, r1, PC::BIT1_T18_CALC_OFFSET);
                        generate_trace_fp2_mul(t, w[6], w[18], r0, r1, PC::BIT1_RX_CALC_OFFSET);
                        rows_sub(w[20], w[21], PC::BIT1_RY_CALC_OFFSET);
                        generate_trace_fp2_mul(t, rz, w[12], r0, r1, PC::BIT1_RZ_CALC_OFFSET);
                    }
        
            }
        });
        // public inputs: x, y, z (72) then 68 x 72 ell coefficients of the native precompute
        size_t p = 0;
        for (const Fp2* v : {&x, &y, &z})
            for (int h = 0; h < 2; h++)
                for (int kk = 0; kk < 12; kk++) public_inputs[p++] = v->c[h].l[kk];
        const std::vector<bls::EllCoeff> ell = bls::calc_pairing_precomp(x, y, z);
        for (const auto& c : ell)
            for (int a = 0; a < 3; a++)
                for (int h = 0; h < 2; h++)
                    for (int kk = 0; kk < 12; kk++) public_inputs[p++] = c[a].c[h].l[kk];
        if (p != PC::PUBLIC_INPUTS) return STARKHIP_ERR_BAD_SHAPE;
    } catch (const std::exception& e) {
        fprintf(stderr, "starkhip_trace_pairing_precomp: %s\n", e.what());
        return STARKHIP_ERR_BAD_SHAPE;
    }
    return STARKHIP_OK;
}
