// FinalExponentiateStark: the final exponentiation of the pairing (73527 columns x 8192 rows, degree 5).
// Restates /root/reference/src/final_exponentiate.rs: layout (:37-77), row schedule (:80-119), op wrappers
// (:137-228), generate_trace (:240-279), op link constraints (:283-827), eval_packed_generic (:907-1136),
// constraint_degree (:1362-1364); public inputs as built by final_exponentiate_main, src/aggregate_proof.rs:158-165.
#include <stdio.h>

#include <vector>

#include "airs.h"
#include "gadgets.h"
#include "wiring.h"

namespace starkhip {
using namespace lay;
using namespace lay_finalexp;
using bls::Fp12;

namespace {

enum OpKind { OP_FROB, OP_MUL, OP_DIV, OP_CEXP, OP_CONJ, OP_CSQ };
struct Op {
    OpKind kind;
    size_t row;       // first row of the op
    int a, b;         // operand slots: -1 = INPUT, k = T_k
    int out;          // result slot T_out
    size_t pow;       // frobenius power
};
// The 32 steps T0..T31 of native final_exponentiate (src/native.rs:1311-1345) in trace order.
// OP_DIV is step T1: the trace holds the multiplication  T1 * INPUT = T0  (App. B.4 item 9).
const Op OPS[32] = {
    {OP_FROB, T0_ROW, -1, 0, 0, 6},   {OP_DIV, T1_ROW, 0, -1, 1, 0},    {OP_FROB, T2_ROW, 1, 0, 2, 2},    {OP_MUL, T3_ROW, 2, 1, 3, 0},
    {OP_CEXP, T4_ROW, 3, 0, 4, 0},    {OP_CONJ, T5_ROW, 4, 0, 5, 0},    {OP_CSQ, T6_ROW, 3, 0, 6, 0},     {OP_CONJ, T7_ROW, 6, 0, 7, 0},
    {OP_MUL, T8_ROW, 7, 5, 8, 0},     {OP_CEXP, T9_ROW, 8, 0, 9, 0},    {OP_CONJ, T10_ROW, 9, 0, 10, 0},  {OP_CEXP, T11_ROW, 10, 0, 11, 0},
    {OP_CONJ, T12_ROW, 11, 0, 12, 0}, {OP_CEXP, T13_ROW, 12, 0, 13, 0}, {OP_CONJ, T14_ROW, 13, 0, 14, 0}, {OP_CSQ, T15_ROW, 5, 0, 15, 0},
    {OP_MUL, T16_ROW, 14, 15, 16, 0}, {OP_CEXP, T17_ROW, 16, 0, 17, 0}, {OP_CONJ, T18_ROW, 17, 0, 18, 0}, {OP_MUL, T19_ROW, 5, 12, 19, 0},
    {OP_FROB, T20_ROW, 19, 0, 20, 2}, {OP_MUL, T21_ROW, 10, 3, 21, 0},  {OP_FROB, T22_ROW, 21, 0, 22, 3}, {OP_CONJ, T23_ROW, 3, 0, 23, 0},
    {OP_MUL, T24_ROW, 16, 23, 24, 0}, {OP_FROB, T25_ROW, 24, 0, 25, 1}, {OP_CONJ, T26_ROW, 8, 0, 26, 0},  {OP_MUL, T27_ROW, 18, 26, 27, 0},
    {OP_MUL, T28_ROW, 27, 3, 28, 0},  {OP_MUL, T29_ROW, 20, 22, 29, 0}, {OP_MUL, T30_ROW, 29, 25, 30, 0}, {OP_MUL, T31_ROW, 30, 28, 31, 0},
};
const size_t T_OFF[32] = {FINAL_EXP_T0_OFFSET,  FINAL_EXP_T1_OFFSET,  FINAL_EXP_T2_OFFSET,  FINAL_EXP_T3_OFFSET,  FINAL_EXP_T4_OFFSET,  FINAL_EXP_T5_OFFSET,
                          FINAL_EXP_T6_OFFSET,  FINAL_EXP_T7_OFFSET,  FINAL_EXP_T8_OFFSET,  FINAL_EXP_T9_OFFSET,  FINAL_EXP_T10_OFFSET, FINAL_EXP_T11_OFFSET,
                          FINAL_EXP_T12_OFFSET, FINAL_EXP_T13_OFFSET, FINAL_EXP_T14_OFFSET, FINAL_EXP_T15_OFFSET, FINAL_EXP_T16_OFFSET, FINAL_EXP_T17_OFFSET,
                          FINAL_EXP_T18_OFFSET, FINAL_EXP_T19_OFFSET, FINAL_EXP_T20_OFFSET, FINAL_EXP_T21_OFFSET, FINAL_EXP_T22_OFFSET, FINAL_EXP_T23_OFFSET,
                          FINAL_EXP_T24_OFFSET, FINAL_EXP_T25_OFFSET, FINAL_EXP_T26_OFFSET, FINAL_EXP_T27_OFFSET, FINAL_EXP_T28_OFFSET, FINAL_EXP_T29_OFFSET,
                          FINAL_EXP_T30_OFFSET, FINAL_EXP_T31_OFFSET};
inline size_t slot_col(int s) { return s < 0 ? FINAL_EXP_INPUT_OFFSET : T_OFF[s]; }
const size_t OPW = FINAL_EXP_OP_OFFSET;
const size_t N_ROWS = 8192;
const size_t OP_SELECTORS[5] = {FINAL_EXP_FORBENIUS_MAP_SELECTOR, FINAL_EXP_CYCLOTOMIC_EXP_SELECTOR, FINAL_EXP_MUL_SELECTOR,
                                FINAL_EXP_CYCLOTOMIC_SQ_SELECTOR, FINAL_EXP_CONJUGATE_SELECTOR};

size_t op_rows(OpKind k) {
    switch (k) {
        case OP_FROB: return FP12_FORBENIUS_MAP_ROWS;
        case OP_MUL: case OP_DIV: return FP12_MUL_ROWS;
        case OP_CEXP: return CYCLOTOMIC_EXP_ROWS;
        case OP_CSQ: return CYCLOTOMIC_SQ_ROWS;
        default: return CONJUGATE_ROWS;
    }
}
size_t op_selector(OpKind k) {
    switch (k) {
        case OP_FROB: return FINAL_EXP_FORBENIUS_MAP_SELECTOR;
        case OP_MUL: case OP_DIV: return FINAL_EXP_MUL_SELECTOR;
        case OP_CEXP: return FINAL_EXP_CYCLOTOMIC_EXP_SELECTOR;
        case OP_CSQ: return FINAL_EXP_CYCLOTOMIC_SQ_SELECTOR;
        default: return FINAL_EXP_CONJUGATE_SELECTOR;
    }
}
// i-th Fp of the frobenius gadget's result inside the op window (final_exponentiate.rs:322-350)
size_t frob_out(size_t i) {
    const size_t r0 = FP12_FORBENIUS_MAP_R0_CALC_OFFSET;
    switch (i) {
        case 0: return r0 + FP6_FORBENIUS_MAP_X_CALC_OFFSET + FP2_FORBENIUS_MAP_INPUT_OFFSET;
        case 1: return r0 + FP6_FORBENIUS_MAP_X_CALC_OFFSET + FP2_FORBENIUS_MAP_T0_CALC_OFFSET + FP_MULTIPLICATION_TOTAL_COLUMNS + REDUCED_OFFSET;
        case 2: return r0 + FP6_FORBENIUS_MAP_Y_CALC_OFFSET + Z1_REDUCE_OFFSET + REDUCED_OFFSET;
        case 3: return r0 + FP6_FORBENIUS_MAP_Y_CALC_OFFSET + Z2_REDUCE_OFFSET + REDUCED_OFFSET;
        case 4: return r0 + FP6_FORBENIUS_MAP_Z_CALC_OFFSET + Z1_REDUCE_OFFSET + REDUCED_OFFSET;
        case 5: return r0 + FP6_FORBENIUS_MAP_Z_CALC_OFFSET + Z2_REDUCE_OFFSET + REDUCED_OFFSET;
        default: {
            const size_t blk[3] = {FP12_FORBENIUS_MAP_C0_CALC_OFFSET, FP12_FORBENIUS_MAP_C1_CALC_OFFSET, FP12_FORBENIUS_MAP_C2_CALC_OFFSET};
            return blk[(i - 6) / 2] + ((i - 6) % 2 ? Z2_REDUCE_OFFSET : Z1_REDUCE_OFFSET) + REDUCED_OFFSET;
        }
    }
}

// every row of an op carries exactly that op's selector pattern (:299-320 and siblings)
void op_selector_pattern(CS& cs, const Op& op) {
    const size_t mine = op_selector(op.kind);
    for (size_t i = op.row; i < op.row + op_rows(op.kind); i++) {
        const Expr rs = cs.L(FINAL_EXP_ROW_SELECTORS + i);
        for (size_t s : OP_SELECTORS) {
            if (s == mine) cs.c(rs * (cs.L(s) - CS::one()));
            else cs.c(rs * cs.L(s));
        }
    }
}

void op_links(CS& cs, const Op& op) {
    using namespace wire;
    op_selector_pattern(cs, op);
    const Expr rs = cs.L(FINAL_EXP_ROW_SELECTORS + op.row);
    const size_t a = slot_col(op.a), b = slot_col(op.b), out = slot_col(op.out);
    switch (op.kind) {
        case OP_FROB:  // :283-366
            cs.link(false, rs, a, OPW + FP12_FORBENIUS_MAP_INPUT_OFFSET, 144);
            cs.c(rs * (cs.L(OPW + FP12_FORBENIUS_MAP_POW_OFFSET) - CS::K(op.pow)));
            for (size_t i = 0; i < 12; i++)
                for (size_t j = 0; j < 12; j++) cs.c(rs * (cs.L(OPW + frob_out(j) + i) - cs.L(out + j * 12 + i)));
            break;
        case OP_MUL:
        case OP_DIV: {  // :443-510; for T1 the reference passes (x = T1, y = INPUT, res = T0)
            const size_t x = op.kind == OP_DIV ? out : a, y = b, res = op.kind == OP_DIV ? a : out;
            for (size_t i = 0; i < 144; i++) {
                cs.c(rs * (cs.L(x + i) - cs.L(OPW + FP12_MUL_X_INPUT_OFFSET + i)));
                cs.c(rs * (cs.L(y + i) - cs.L(OPW + FP12_MUL_Y_INPUT_OFFSET + i)));
            }
            for (size_t i = 0; i < 12; i++)
                for (size_t j = 0; j < 6; j++) {
                    cs.c(rs * (cs.L(res + j * 12 + i) - cs.L(addred6_out(OPW + FP12_MUL_X_CALC_OFFSET, j) + i)));
                    cs.c(rs * (cs.L(res + 72 + j * 12 + i) - cs.L(subred6_out(OPW + FP12_MUL_Y_CALC_OFFSET, j) + i)));
                }
            break;
        }
        case OP_CEXP: {  // :569-621
            cs.link(false, rs, a, OPW + INPUT_OFFSET, 144);
            const Expr last = cs.L(FINAL_EXP_ROW_SELECTORS + op.row + CYCLOTOMIC_EXP_ROWS - 1) * cs.L(OPW + RES_ROW_SELECTOR_OFFSET);
            cs.link(false, last, out, OPW + Z_OFFSET, 144);
            break;
        }
        case OP_CONJ:  // :666-715
            cs.link(false, rs, a, OPW + FP12_CONJUGATE_INPUT_OFFSET, 144);
            cs.link(false, rs, out, OPW + FP12_CONJUGATE_OUTPUT_OFFSET, 144);
            break;
        case OP_CSQ: {  // :758-827
            static const size_t C[6] = {CYCLOTOMIC_SQ_C0_CALC_OFFSET, CYCLOTOMIC_SQ_C1_CALC_OFFSET, CYCLOTOMIC_SQ_C2_CALC_OFFSET,
                                        CYCLOTOMIC_SQ_C3_CALC_OFFSET, CYCLOTOMIC_SQ_C4_CALC_OFFSET, CYCLOTOMIC_SQ_C5_CALC_OFFSET};
            cs.link(false, rs, a, OPW + CYCLOTOMIC_SQ_INPUT_OFFSET, 144);
            for (size_t i = 0; i < 12; i++)
                for (size_t j = 0; j < 6; j++)
                    for (size_t k = 0; k < 2; k++)
                        cs.c(rs * (cs.L(OPW + C[j] + FP2_ADDITION_TOTAL + RR * k + FP_SINGLE_REDUCED_OFFSET + i) - cs.L(out + j * 24 + k * 12 + i)));
            break;
        }
    }
}

}  // namespace

AirProgram build_air_final_exp() {
    AirBuilder b(COLUMNS, PUBLIC_INPUTS, 5);
    CS cs(b);
    for (size_t i = 0; i < 144; i++) {  // :919-927
        cs.c(cs.L(FINAL_EXP_INPUT_OFFSET + i) - b.PI(PIS_INPUT_OFFSET + i));
        cs.c(cs.L(FINAL_EXP_T31_OFFSET + i) - b.PI(PIS_OUTPUT_OFFSET + i));
    }
    // one-hot row selector shift register (:931-956)
    for (size_t i = 0; i < N_ROWS; i++) cs.cf(cs.L(FINAL_EXP_ROW_SELECTORS + i) - CS::K(i == 0 ? 1 : 0));
    for (size_t i = 0; i + 1 < N_ROWS; i++) cs.ct(cs.L(FINAL_EXP_ROW_SELECTORS + i) - cs.N(FINAL_EXP_ROW_SELECTORS + i + 1));
    for (size_t i = 0; i < N_ROWS; i++) cs.cl(cs.L(FINAL_EXP_ROW_SELECTORS + i) - CS::K(i + 1 == N_ROWS ? 1 : 0));
    // the input and every T_j block are constant over the rows (:958-1033)
    for (size_t i = 0; i < 144; i++) {
        cs.ct(cs.L(FINAL_EXP_INPUT_OFFSET + i) - cs.N(FINAL_EXP_INPUT_OFFSET + i));
        for (size_t j = 0; j < 32; j++) cs.ct(cs.L(T_OFF[j] + i) - cs.N(T_OFF[j] + i));
    }
    for (const Op& op : OPS) op_links(cs, op);  // :1036-1129
    // the five gadget trees share the op window, each gated by its op selector (:1131-1135)
    add_fp12_forbenius_map_constraints(cs, OPW, cs.L(FINAL_EXP_FORBENIUS_MAP_SELECTOR));
    add_fp12_multiplication_constraints(cs, OPW, cs.L(FINAL_EXP_MUL_SELECTOR));
    add_cyclotomic_exp_constraints(cs, OPW, cs.L(FINAL_EXP_CYCLOTOMIC_EXP_SELECTOR));
    add_fp12_conjugate_constraints(cs, OPW, cs.L(FINAL_EXP_CONJUGATE_SELECTOR));
    add_cyclotomic_sq_constraints(cs, OPW, cs.L(FINAL_EXP_CYCLOTOMIC_SQ_SELECTOR));
    return b.finish();
}

}  // namespace starkhip

using namespace starkhip;

// FinalExponentiateStark::generate_trace (:240-279) + public inputs (src/aggregate_proof.rs:158-165).
// The 32 results come from the native chain first (12 ms); after that every op's rows depend only on its operands, so the ops
// -- the five cyclotomic exponentiations, 88 % of the time, cut into step ranges -- are tasks for fill_tasks (trace_tasks.cpp).
namespace {
struct FillTask {
    int op;          // index into OPS, or -1: the columns that span all rows (row selectors, input, the T_j blocks)
    size_t j0, j1;   // step range of a cyclotomic exponentiation
};

void fill_task(Trace& t, const FillTask& task, const Fp12& X, const Fp12 (&T)[32], size_t n_rows) {
    auto val = [&](int s) -> const Fp12& { return s < 0 ? X : T[s]; };
    if (task.op < 0) {
        for (size_t row = 0; row < n_rows; row++) t.at(row, FINAL_EXP_ROW_SELECTORS + row) = 1;
        t.put_rows(0, n_rows, FINAL_EXP_INPUT_OFFSET, X);  // constant over the rows (final_exponentiate.rs:958-1033)
        for (const Op& op : OPS) t.put_rows(0, n_rows, T_OFF[op.out], T[op.out]);
        return;
    }
    const Op& op = OPS[task.op];
    const size_t r0 = op.row, r1 = op.row + op_rows(op.kind) - 1;
    const Fp12& a = val(op.a);
    if (task.j0 == 0)
        { RowSpan rows_(t, r1 - r0 + 1); t.at(r0, op_selector(op.kind)) = 1; }
    switch (op.kind) {
        case OP_FROB: fill_trace_fp12_forbenius_map(t, a, op.pow, r0, r1, OPW); break;
        case OP_MUL: fill_trace_fp12_multiplication(t, a, val(op.b), r0, r1, OPW); break;
        case OP_DIV: fill_trace_fp12_multiplication(t, T[op.out], val(op.b), r0, r1, OPW); break;  // res * y == x
        case OP_CEXP: fill_trace_cyclotomic_exp_steps(t, a, r0, r1, OPW, task.j0, task.j1); break;
        case OP_CONJ: fill_trace_fp12_conjugate(t, a, r0, OPW); break;
        case OP_CSQ: fill_trace_cyclotomic_sq(t, a, r0, r1, OPW); break;
    }
}
}  // namespace

extern "C" int starkhip_trace_final_exp(const uint32_t x[144], uint64_t* trace, size_t n_rows, uint64_t* public_inputs) {
    if (n_rows != N_ROWS) return STARKHIP_ERR_BAD_SHAPE;  // the layout holds exactly 8192 row-selector columns
    try {
        const Fp12 X = Fp12::from_limbs(x);
        Trace t = open_trace(trace, n_rows, COLUMNS);
        Fp12 T[32];
        for (const Op& op : OPS) {  // src/native.rs:1311-1345
            const Fp12& a = op.a < 0 ? X : T[op.a];
            const Fp12& b = op.b < 0 ? X : T[op.b];
            switch (op.kind) {
                case OP_FROB: T[op.out] = a.forbenius_map(op.pow); break;
                case OP_MUL: T[op.out] = a * b; break;
                case OP_DIV: T[op.out] = a / b; break;
                case OP_CEXP: T[op.out] = a.cyclotomic_exponent(); break;
                case OP_CONJ: T[op.out] = a.conjugate(); break;
                case OP_CSQ: T[op.out] = a.cyclotomic_square(); break;
            }
        }
        std::vector<FillTask> tasks;
        const size_t cuts = t.log && trace_threads() > 1 ? 5 : 1;  // parts per cyclotomic exponentiation: 53 tasks of comparable weight
        tasks.push_back({-1, 0, 70});
        for (int k = 0; k < 32; k++) {
            if (OPS[k].kind == OP_CEXP)
                for (size_t c = 0; c < cuts; c++) tasks.push_back({k, 70 * c / cuts, 70 * (c + 1) / cuts});
            else
                tasks.push_back({k, 0, 70});
        }
        fill_tasks(t, tasks.size(), [&](Trace& part, size_t k) { fill_task(part, tasks[k], X, T, n_rows); });
        uint32_t out[144];
        T[31].to_limbs(out);
        for (int i = 0; i < 144; i++) {
            public_inputs[i] = x[i];
            public_inputs[144 + i] = out[i];
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "starkhip_trace_final_exp: %s\n", e.what());
        return STARKHIP_ERR_BAD_SHAPE;
    }
    return STARKHIP_OK;
}
