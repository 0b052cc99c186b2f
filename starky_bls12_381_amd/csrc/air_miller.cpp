// MillerLoopStark: the Miller loop of the pairing (97330 columns x 1024 rows, degree 3).
// Restates /root/reference/src/miller_loop.rs: layout (:48-63), fill_trace_miller_loop (:87-146),
// add_miller_loop_constraints (:191-411), eval_packed_generic (:644-677), constraint_degree (:713-715);
// public inputs as built by miller_loop_main, src/aggregate_proof.rs:78-101.
#include <stdio.h>

#include "airs.h"
#include "gadgets.h"
#include "wiring.h"

namespace starkhip {
using namespace lay;
namespace M = lay_miller;
using namespace wire;
using bls::Fp;
using bls::Fp12;
using bls::Fp2;
using bls::Fp6;

AirProgram build_air_miller_loop() {
    AirBuilder b(M::COLUMNS, M::PUBLIC_INPUTS, 3);
    CS cs(b);
    // ---- eval_packed_generic (:655-676)
    for (size_t i = 0; i < 12; i++) {
        cs.c(cs.L(M::PX_OFFSET + i) - b.PI(M::PIS_PX_OFFSET + i));
        cs.c(cs.L(M::PY_OFFSET + i) - b.PI(M::PIS_PY_OFFSET + i));
    }
    for (size_t i = 0; i < 68; i++)
        for (size_t j = 0; j < 72; j++) cs.c(cs.L(M::ELL_COEFFS_INDEX_OFFEST + i) * (cs.L(M::ELL_COEFFS_OFFSET + j) - b.PI(M::PIS_ELL_COEFFS_OFFSET + i * 72 + j)));
    for (size_t i = 0; i < 144; i++) cs.c(cs.L(M::MILLER_LOOP_RES_OFFSET + i) - b.PI(M::PIS_RES_OFFSET + i));

    // ---- add_miller_loop_constraints(start_col = 0, bit_selector = None) (:191-411)
    const Expr bs = CS::one();
    const size_t F12 = M::F12_OFFSET, o1 = M::O1_CALC_OFFSET, o4 = M::O4_CALC_OFFSET, m014 = M::F12_MUL_BY_014_OFFSET, sq = M::F12_SQ_CALC_OFFSET;
    const size_t res = M::MILLER_LOOP_RES_OFFSET, conj = M::RES_CONJUGATE_OFFSET, ell = M::ELL_COEFFS_OFFSET;
    for (size_t i = 0; i < 12; i++) {
        cs.ct(cs.L(M::PX_OFFSET + i) - cs.N(M::PX_OFFSET + i));
        cs.ct(cs.L(M::PY_OFFSET + i) - cs.N(M::PY_OFFSET + i));
    }
    for (size_t i = 0; i < 144; i++) {
        if (i == 0) cs.c(cs.L(M::FIRST_BIT_SELECTOR_OFFSET) * (cs.L(F12 + i) - CS::one()));
        else cs.c(cs.L(M::FIRST_BIT_SELECTOR_OFFSET) * cs.L(F12 + i));
    }
    const Expr nfirst = cs.N(M::FIRST_ROW_SELECTOR_OFFSET), nbit1 = cs.N(M::BIT1_SELECTOR_OFFSET);
    for (size_t i = 0; i < 12; i++)
        for (size_t j = 0; j < 6; j++) {
            cs.c(bs * nfirst * nbit1 * (cs.N(F12 + j * 12 + i) - cs.L(addred6_out(m014 + MULTIPLY_BY_014_X_CALC_OFFSET, j) + i)));
            cs.c(bs * nfirst * nbit1 * (cs.N(F12 + j * 12 + i + 72) - cs.L(subred6_out(m014 + MULTIPLY_BY_014_Y_CALC_OFFSET, j) + i)));
            cs.c(bs * nfirst * (CS::one() - nbit1) * (cs.N(F12 + j * 12 + i) - cs.L(addred6_out(sq + FP12_MUL_X_CALC_OFFSET, j) + i)));
            cs.c(bs * nfirst * (CS::one() - nbit1) * (cs.N(F12 + j * 12 + i + 72) - cs.L(subred6_out(sq + FP12_MUL_Y_CALC_OFFSET, j) + i)));
        }
    auto fp2fp_inputs = [&](size_t blk, size_t xcol, size_t ycol) {
        const Expr g = bs * cs.L(blk + FP2_FP_MUL_SELECTOR_OFFSET);
        for (size_t i = 0; i < 24; i++) {
            cs.c(g * (cs.L(blk + FP2_FP_X_INPUT_OFFSET + i) - cs.L(xcol + i)));
            if (i < 12) cs.c(g * (cs.L(blk + FP2_FP_Y_INPUT_OFFSET + i) - cs.L(ycol + i)));
        }
        add_fp2_fp_mul_constraints(cs, blk, bs);
    };
    fp2fp_inputs(o1, ell + 24, M::PX_OFFSET);
    fp2fp_inputs(o4, ell + 48, M::PY_OFFSET);
    {
        const Expr g = bs * cs.L(m014 + MULTIPLY_BY_014_SELECTOR_OFFSET);
        for (size_t i = 0; i < 12; i++) {
            for (size_t j = 0; j < 12; j++) cs.c(g * (cs.L(m014 + MULTIPLY_BY_014_INPUT_OFFSET + j * 12 + i) - cs.L(F12 + j * 12 + i)));
            for (size_t j = 0; j < 2; j++) {
                const size_t z = (j == 0 ? X0_Y_REDUCE_OFFSET : X1_Y_REDUCE_OFFSET) + REDUCED_OFFSET;
                cs.c(g * (cs.L(m014 + MULTIPLY_BY_014_O0_OFFSET + j * 12 + i) - cs.L(ell + j * 12 + i)));
                cs.c(g * (cs.L(m014 + MULTIPLY_BY_014_O1_OFFSET + j * 12 + i) - cs.L(o1 + z + i)));
                cs.c(g * (cs.L(m014 + MULTIPLY_BY_014_O4_OFFSET + j * 12 + i) - cs.L(o4 + z + i)));
            }
        }
    }
    add_multiply_by_014_constraints(cs, m014, bs);
    {
        const Expr g = bs * cs.L(sq + FP12_MUL_SELECTOR_OFFSET);
        for (size_t i = 0; i < 12; i++) {
            for (size_t j = 0; j < 6; j++) {
                cs.c(g * (cs.L(addred6_out(m014 + MULTIPLY_BY_014_X_CALC_OFFSET, j) + i) - cs.L(sq + FP12_MUL_X_INPUT_OFFSET + j * 12 + i)));
                cs.c(g * (cs.L(subred6_out(m014 + MULTIPLY_BY_014_Y_CALC_OFFSET, j) + i) - cs.L(sq + FP12_MUL_X_INPUT_OFFSET + j * 12 + i + 72)));
            }
            // the reference writes `bit_selector_val * X - Y` without parentheses and without the block selector (App. B.4 item 5)
            for (size_t j = 0; j < 12; j++) cs.c(bs * cs.L(sq + FP12_MUL_X_INPUT_OFFSET + j * 12 + i) - cs.L(sq + FP12_MUL_Y_INPUT_OFFSET + j * 12 + i));
        }
    }
    add_fp12_multiplication_constraints(cs, sq, bs);
    for (size_t i = 0; i < 12; i++)
        for (size_t jk = 0; jk < 6; jk++) {
            const size_t a = add6_block(conj, jk);
            cs.c(bs * cs.L(a + FP_ADDITION_CHECK_OFFSET) * (cs.L(a + FP_ADDITION_X_OFFSET + i) - cs.L(res + 72 + jk * 12 + i)));
        }
    add_negate_fp6_constraints(cs, conj, bs);
    {
        const Expr last = cs.L(M::LAST_BIT_SELECTOR_OFFSET);
        const size_t mx = m014 + MULTIPLY_BY_014_X_CALC_OFFSET, my = m014 + MULTIPLY_BY_014_Y_CALC_OFFSET;
        for (size_t i = 0; i < 12; i++)
            for (size_t jk = 0; jk < 6; jk++) {
                const size_t ax = add6_block(mx, jk), sy = sub6_block(my, jk), ac = add6_block(conj, jk);
                cs.c(bs * last * cs.L(ax + FP_ADDITION_CHECK_OFFSET) * (cs.L(addred6_out(mx, jk) + i) - cs.L(res + jk * 12 + i)));
                cs.c(bs * last * cs.L(sy + FP_SUBTRACTION_CHECK_OFFSET) * (cs.L(subred6_out(my, jk) + i) - cs.L(ac + FP_ADDITION_Y_OFFSET + i)));
            }
    }
    return b.finish();
}

}  // namespace starkhip

using namespace starkhip;

static bls::Fp fp_of(const uint32_t* l) { bls::Fp r; for (int i = 0; i < 12; i++) r.l[i] = l[i]; return r; }
static bls::Fp2 fp2_of(const uint32_t* l) { return bls::Fp2(fp_of(l), fp_of(l + 12)); }

// MillerLoopStark::generate_trace (:157-160) -> fill_trace_miller_loop(0, n-1, 0) (:87-146), public inputs per
// miller_loop_main (src/aggregate_proof.rs:78-101): px, py, 68 x 72 ell coefficients of the native precompute, result.
extern "C" int starkhip_trace_miller_loop(const uint32_t px[12], const uint32_t py[12], const uint32_t qx[24], const uint32_t qy[24],
                                          const uint32_t qz[24], uint64_t* trace, size_t n_rows, uint64_t* public_inputs) {
    if (n_rows < 2 || (n_rows & (n_rows - 1))) return STARKHIP_ERR_BAD_SHAPE;
    try {
        const bls::Fp x = fp_of(px), y = fp_of(py);
        const bls::Fp2 QX = fp2_of(qx), QY = fp2_of(qy), QZ = fp2_of(qz);
        std::vector<bls::EllCoeff> ell = bls::calc_pairing_precomp(QX, QY, QZ);
        Fp12 native_res = bls::miller_loop(x, y, QX, QY, QZ);
        Trace t = open_trace(trace, n_rows, M::COLUMNS);
        // The running value f12 at the start of every 12-row block comes from a native pass first (68 sparse products and
        // squarings); after that a block's rows depend only on (f12, bit state, ell[j]), so ranges of blocks -- and the columns
        // that span all rows -- are tasks for fill_tasks (trace_tasks.cpp).
        struct Block {
            Fp12 f12;
            int i;
            bool bitone;
        };
        const size_t blocks = std::min(n_rows / 12, ell.size());
        std::vector<Block> at(blocks);
        Fp12 f12 = Fp12::one();
        {
            int i = 62;  // bits() - 2
            bool bitone = false;
            for (size_t j = 0; j < blocks; j++) {
                at[j] = {f12, i, bitone};
                const bls::EllCoeff& e = ell[j];
                f12 = f12.multiply_by_014(e[0], e[1] * x, e[2] * y);
                if (((bls::BLS_X >> i) & 1) && !bitone) {
                    bitone = true;
                } else if (j + 1 < ell.size()) {
                    f12 = f12 * f12;
                    i -= 1;
                    bitone = false;
                }
            }
        }
        f12 = f12.conjugate();
        const size_t per_task = t.log && trace_threads() > 1 ? 4 : blocks ? blocks : 1;  // blocks per task
        const size_t n_block_tasks = (blocks + per_task - 1) / per_task;
        fill_tasks(t, n_block_tasks + 2, [&](Trace& part, size_t k) {
            if (k == n_block_tasks) {
                for (size_t row = 0; row < n_rows; row++) {
                    part.put(row, M::PX_OFFSET, x.l);
                    part.put(row, M::PY_OFFSET, y.l);
                }
                for (size_t row = 0; row < n_rows; row++) part.put(row, M::MILLER_LOOP_RES_OFFSET, f12);
                return;
            }
            if (k == n_block_tasks + 1) {
                for (size_t row = 0; row < n_rows; row++) fill_trace_negate_fp6(part, f12.c6(1), row, M::RES_CONJUGATE_OFFSET);
                return;
            }
            for (size_t j = k * per_task; j < std::min(blocks, (k + 1) * per_task); j++) {
                const size_t s_row = j * 12, e_row = (j + 1) * 12 - 1;
                const Block& b = at[j];
                {
                    RowSpan rows_(part, e_row - s_row + 1);
                    if (j == 0) part.at(s_row, M::FIRST_BIT_SELECTOR_OFFSET) = 1;
                    if (b.i == 0) part.at(s_row, M::LAST_BIT_SELECTOR_OFFSET) = 1;
                    if (b.bitone) part.at(s_row, M::BIT1_SELECTOR_OFFSET) = 1;
                    part.at(s_row, M::ELL_COEFFS_INDEX_OFFEST + j) = 1;
                    for (size_t c = 0; c < 3; c++) part.put(s_row, M::ELL_COEFFS_OFFSET + c * 24, ell[j][c]);
                    part.put(s_row, M::F12_OFFSET, b.f12);
                }
                if (j != 0) part.at(s_row, M::FIRST_ROW_SELECTOR_OFFSET) = 1;
                const bls::EllCoeff& e = ell[j];
                fill_trace_fp2_fp_mul(part, e[1], x, s_row, e_row, M::O1_CALC_OFFSET);
                const Fp2 o1 = e[1] * x;
                fill_trace_fp2_fp_mul(part, e[2], y, s_row, e_row, M::O4_CALC_OFFSET);
                const Fp2 o4 = e[2] * y;
                fill_trace_multiply_by_014(part, b.f12, e[0], o1, o4, s_row, e_row, M::F12_MUL_BY_014_OFFSET);
                const Fp12 g = b.f12.multiply_by_014(e[0], o1, o4);
                fill_trace_fp12_multiplication(part, g, g, s_row, e_row, M::F12_SQ_CALC_OFFSET);
            }
        });
        // public inputs
        size_t p = 0;
        for (int k = 0; k < 12; k++) public_inputs[p++] = x.l[k];
        for (int k = 0; k < 12; k++) public_inputs[p++] = y.l[k];
        for (const auto& c : ell)
            for (int a = 0; a < 3; a++)
                for (int h = 0; h < 2; h++)
                    for (int k = 0; k < 12; k++) public_inputs[p++] = c[a].c[h].l[k];
        for (int a = 0; a < 12; a++)
            for (int k = 0; k < 12; k++) public_inputs[p++] = native_res.c[a].l[k];
        if (p != M::PUBLIC_INPUTS) return STARKHIP_ERR_BAD_SHAPE;
    } catch (const std::exception& e) {
        fprintf(stderr, "starkhip_trace_miller_loop: %s\n", e.what());
        return STARKHIP_ERR_BAD_SHAPE;
    }
    return STARKHIP_OK;
}
