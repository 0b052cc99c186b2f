// Gadget library shared by the four AIRs: trace fillers (the witness side) and constraint emitters
// (the AIR side) for Fp / Fp2 / Fp6 / Fp12 arithmetic on 12 x u32 limbs, one limb per trace cell.
// Restates the `fill_*` and packed `add_*_constraints` halves of /root/reference/src/{fp,fp2,fp6,fp12}.rs;
// the `*_ext_circuit` halves (recursive verifier) are out of scope.  Constraint ORDER follows the
// reference exactly because it fixes the power of alpha each constraint receives.
#pragma once
#include <stdint.h>

#include <functional>

#include "air_ir.h"
#include "air_layout.h"
#include "native.h"
#include "trace_log.h"

namespace starkhip {

using bls::Fp;
using bls::Fp12;
using bls::Fp2;
using bls::Fp6;
using bls::L12;
using bls::L24;

// Trace sink of the fillers: either a row-major matrix [rows][cols] of canonical Goldilocks cells (d != nullptr), or a
// recorder of the same writes as runs (trace_log.h; log != nullptr) that the device expands.
struct Trace {
    uint64_t* d;
    size_t rows, cols;
    TraceLog* log;
    // Row span (RowSpan below): every write at row r stands for the same write on rows r .. r + span - 1.  The reference fills a
    // gadget that is constant over a 12-row block with `for row in start..=end { fill_*(row) }` (e.g. src/fp12.rs:96, src/fp2.rs:146-154):
    // twelve times the same natives and the same cells.  The fillers here run the body ONCE under a span of twelve -- one record
    // with a run of twelve rows when recording (what twelve consecutive put()s leave), twelve rows written when filling a matrix.
    size_t span = 1;
    Trace(uint64_t* dense, size_t r, size_t c, TraceLog* l = nullptr) : d(dense), rows(r), cols(c), log(l) {}
    struct Cell {  // `t.at(row, col) = v`
        Trace& t;
        size_t row, col;
        void operator=(uint64_t v) {
            for (size_t r = row; r < row + t.span; r++) {
                if (t.log) {
                    if (v != 0 && t.span > 1) {  // a run of the same non-zero value
                        if (v >> 32) throw std::runtime_error("trace_log: cell value does not fit 32 bits");
                        const uint32_t w = (uint32_t)v;
                        t.log->put_rows(row, t.span, col, &w, 1);
                        return;
                    }
                    t.log->set(r, col, v);
                } else {
                    t.d[r * t.cols + col] = v;
                }
            }
        }
    };
    Cell at(size_t row, size_t col) { return Cell{*this, row, col}; }
    void put(size_t row, size_t col, const uint32_t* v, size_t n) { put_rows(row, span, col, v, n); }
    // the same vector on n_rows consecutive rows (one record when recording)
    void put_rows(size_t row, size_t n_rows, size_t col, const uint32_t* v, size_t n) {
        if (log) {
            log->put_rows(row, n_rows, col, v, n);
            return;
        }
        for (size_t r = row; r < row + n_rows; r++) {
            uint64_t* p = d + r * cols + col;
            for (size_t i = 0; i < n; i++) p[i] = v[i];
        }
    }
    void put_rows(size_t row, size_t n_rows, size_t col, const Fp12& v) { for (int i = 0; i < 12; i++) put_rows(row, n_rows, col + 12 * i, v.c[i].l.data(), 12); }
    void put(size_t row, size_t col, const L12& v) { put(row, col, v.data(), 12); }
    void put(size_t row, size_t col, const L24& v) { put(row, col, v.data(), 24); }
    void put(size_t row, size_t col, const Fp2& v) { put(row, col, v.c[0].l); put(row, col + 12, v.c[1].l); }
    void put(size_t row, size_t col, const Fp6& v) { for (int i = 0; i < 6; i++) put(row, col + 12 * i, v.c[i].l); }
    void put(size_t row, size_t col, const Fp12& v) { for (int i = 0; i < 12; i++) put(row, col + 12 * i, v.c[i].l); }
};

struct RowSpan {  // `{ RowSpan rows(t, n); fill_x(t, ..., first_row, col); }` == `for row in first_row .. first_row + n { fill_x(t, ..., row, col) }`
    Trace& t;
    size_t old;
    RowSpan(Trace& tr, size_t n) : t(tr), old(tr.span) {
        if (old != 1) throw std::runtime_error("trace: nested row spans");
        t.span = n;
    }
    ~RowSpan() { t.span = old; }
};

// What a generator calls first: a zeroed dense matrix over the caller's buffer, or -- when the calling thread is armed by
// starkhip_trace_log_begin -- a recorder (the buffer argument is then ignored and may be null).
TraceLog*& armed_trace_log();  // thread-local, capi.cpp
int trace_threads();           // host threads one generator call may use when recording (starkhip_trace_set_threads), capi.cpp
inline Trace open_trace(uint64_t* dense, size_t rows, size_t cols) {
    if (TraceLog* log = armed_trace_log()) {
        if (!log->offsets.empty() || log->rows) throw std::runtime_error("trace_log: one generator call per log");
        log->reset(rows, cols);
        return Trace(nullptr, rows, cols, log);
    }
    if (!dense) throw std::runtime_error("trace: null buffer");
    memset(dense, 0, rows * cols * sizeof(uint64_t));
    return Trace(dense, rows, cols);
}

// Run fill(trace, k) for k < n_tasks; the tasks must write disjoint cells.  Serially into `t`, or -- when `t` records and
// trace_threads() > 1 -- on a pool, every task into a log of its own that `t`'s log takes over in task order (trace_tasks.cpp).
void fill_tasks(Trace& t, size_t n_tasks, const std::function<void(Trace&, size_t)>& fill);

// Constraint sink: thin sugar over AirBuilder for the "gate * (a - b)" families that make up most constraints.
struct CS {
    AirBuilder& b;
    explicit CS(AirBuilder& bb) : b(bb) {}
    Expr L(size_t c) const { return b.L((uint32_t)c); }
    Expr N(size_t c) const { return b.N((uint32_t)c); }
    static Expr K(uint64_t v) { return AirBuilder::C(v); }
    static Expr one() { return AirBuilder::one(); }
    void c(const Expr& e) { b.constraint(e); }
    void ct(const Expr& e) { b.transition(e); }
    void cf(const Expr& e) { b.first_row(e); }
    void cl(const Expr& e) { b.last_row(e); }
    void emit(bool transition, const Expr& e) { transition ? b.transition(e) : b.constraint(e); }
    // gate * (local[a + i] - local[bcol + i]), i < n
    void link(bool transition, const Expr& gate, size_t a, size_t bcol, size_t n) {
        for (size_t i = 0; i < n; i++) emit(transition, gate * (L(a + i) - L(bcol + i)));
    }
    // gate * (local[a + i] - next[a + i])
    void keep(bool transition, const Expr& gate, size_t a, size_t n) {
        for (size_t i = 0; i < n; i++) emit(transition, gate * (L(a + i) - N(a + i)));
    }
    // interleaved copy constraints: for each i < n, for each spec in order:  bs * local[gate] * (local[a + i] - local[b + i])
    struct LinkSpec {
        size_t gate, a, b;
    };
    void links(bool transition, const Expr& bs, size_t n, std::initializer_list<LinkSpec> specs) {
        for (size_t i = 0; i < n; i++)
            for (const LinkSpec& s : specs) emit(transition, bs * L(s.gate) * (L(s.a + i) - L(s.b + i)));
    }
    // gate * (local[a + i] - k[i])
    void link_const(bool transition, const Expr& gate, size_t a, const uint32_t* k, size_t n) {
        for (size_t i = 0; i < n; i++) emit(transition, gate * (L(a + i) - K(k[i])));
    }
};

// limb constants used inside constraints
const L12& modulus_limbs();        // p
const L24& modulus_sq_limbs();     // p^2           (src/fp2.rs:803-811)
const L12& range_check_offset();   // 2^382 - p     (src/fp.rs:1343-1344)

// ---- Fp (src/fp.rs)
void fill_addition_trace(Trace& t, const L24& x, const L24& y, size_t row, size_t col);
void fill_trace_addition_fp(Trace& t, const L12& x, const L12& y, size_t row, size_t col);
void fill_trace_negate_fp(Trace& t, const L12& x, size_t row, size_t col);
void fill_subtraction_trace(Trace& t, const L24& x, const L24& y, size_t row, size_t col);
void fill_trace_subtraction_fp(Trace& t, const L12& x, const L12& y, size_t row, size_t col);
void fill_trace_multiply_single_fp(Trace& t, const L12& x, uint32_t y, size_t row, size_t col);
L12 fill_trace_reduce_single(Trace& t, const L12& x, size_t row, size_t col);
void fill_range_check_trace(Trace& t, const L12& x, size_t row, size_t col);
void fill_multiplication_trace_no_mod_reduction(Trace& t, const L12& x, const L12& y, size_t start_row, size_t end_row, size_t col);
L12 fill_reduction_trace(Trace& t, const L24& x, size_t start_row, size_t end_row, size_t col);

void add_multiplication_constraints(CS& cs, size_t col, const Expr& bs);
void add_addition_constraints(CS& cs, size_t col, const Expr& bs);
void add_addition_fp_constraints(CS& cs, size_t col, const Expr& bs);
void add_subtraction_fp_constraints(CS& cs, size_t col, const Expr& bs);
void add_negate_fp_constraints(CS& cs, size_t col, const Expr& bs);
void add_fp_single_multiply_constraints(CS& cs, size_t col, const Expr& bs);
void add_fp_reduce_single_constraints(CS& cs, size_t col, const Expr& bs);
void add_subtraction_constraints(CS& cs, size_t col, const Expr& bs);
void add_range_check_constraints(CS& cs, size_t col, const Expr& bs);
void add_reduce_constraints(CS& cs, size_t col, size_t selector_col, const Expr& bs);

// ---- Fp2 (src/fp2.rs)
void fill_trace_addition_fp2(Trace& t, const Fp2& x, const Fp2& y, size_t row, size_t col);
void fill_trace_subtraction_fp2(Trace& t, const Fp2& x, const Fp2& y, size_t row, size_t col);
void fill_trace_negate_fp2(Trace& t, const Fp2& x, size_t row, size_t col);
void generate_trace_fp2_mul(Trace& t, const Fp2& x, const Fp2& y, size_t start_row, size_t end_row, size_t col);
void fill_trace_fp2_fp_mul(Trace& t, const Fp2& x, const Fp& y, size_t start_row, size_t end_row, size_t col);
void fill_trace_subtraction_with_reduction(Trace& t, const Fp2& x, const Fp2& y, size_t row, size_t col);
void fill_multiply_by_b_trace(Trace& t, const Fp2& x, size_t start_row, size_t end_row, size_t col);
void fill_trace_addition_with_reduction(Trace& t, const Fp2& x, const Fp2& y, size_t row, size_t col);
void fill_trace_non_residue_multiplication(Trace& t, const Fp2& x, size_t row, size_t col);
void fill_trace_fp4_sq(Trace& t, const Fp2& x, const Fp2& y, size_t start_row, size_t end_row, size_t col);
void fill_trace_fp2_forbenius_map(Trace& t, const Fp2& x, size_t pow, size_t start_row, size_t end_row, size_t col);

void add_addition_fp2_constraints(CS& cs, size_t col, const Expr& bs);
void add_subtraction_fp2_constraints(CS& cs, size_t col, const Expr& bs);
void add_fp2_single_multiply_constraints(CS& cs, size_t col, const Expr& bs);
void add_negate_fp2_constraints(CS& cs, size_t col, const Expr& bs);
void add_fp2_mul_constraints(CS& cs, size_t col, const Expr& bs);
void add_fp2_fp_mul_constraints(CS& cs, size_t col, const Expr& bs);
void add_multiply_by_b_constraints(CS& cs, size_t col, const Expr& bs);
void add_subtraction_with_reduction_constraints(CS& cs, size_t col, const Expr& bs);
void add_addition_with_reduction_constraints(CS& cs, size_t col, const Expr& bs);
void add_non_residue_multiplication_constraints(CS& cs, size_t col, const Expr& bs);
void add_fp4_sq_constraints(CS& cs, size_t col, const Expr& bs);
void add_fp2_forbenius_map_constraints(CS& cs, size_t col, const Expr& bs);

// ---- Fp6 (src/fp6.rs)
void fill_trace_addition_fp6(Trace& t, const Fp6& x, const Fp6& y, size_t row, size_t col);
void fill_trace_addition_with_reduction_fp6(Trace& t, const Fp6& x, const Fp6& y, size_t row, size_t col);
void fill_trace_subtraction_fp6(Trace& t, const Fp6& x, const Fp6& y, size_t row, size_t col);
void fill_trace_negate_fp6(Trace& t, const Fp6& x, size_t row, size_t col);
void fill_trace_subtraction_with_reduction_fp6(Trace& t, const Fp6& x, const Fp6& y, size_t row, size_t col);
void fill_trace_non_residue_multiplication_fp6(Trace& t, const Fp6& x, size_t row, size_t col);
void fill_trace_fp6_multiplication(Trace& t, const Fp6& x, const Fp6& y, size_t start_row, size_t end_row, size_t col);
void fill_trace_multiply_by_1(Trace& t, const Fp6& x, const Fp2& b1, size_t start_row, size_t end_row, size_t col);
void fill_trace_multiply_by_01(Trace& t, const Fp6& x, const Fp2& b0, const Fp2& b1, size_t start_row, size_t end_row, size_t col);
void fill_trace_fp6_forbenius_map(Trace& t, const Fp6& x, size_t pow, size_t start_row, size_t end_row, size_t col);

void add_addition_fp6_constraints(CS& cs, size_t col, const Expr& bs);
void add_addition_with_reduction_constraints_fp6(CS& cs, size_t col, const Expr& bs);
void add_subtraction_fp6_constraints(CS& cs, size_t col, const Expr& bs);
void add_negate_fp6_constraints(CS& cs, size_t col, const Expr& bs);
void add_subtraction_with_reduction_constraints_fp6(CS& cs, size_t col, const Expr& bs);
void add_non_residue_multiplication_fp6_constraints(CS& cs, size_t col, const Expr& bs);
void add_fp6_multiplication_constraints(CS& cs, size_t col, const Expr& bs);
void add_multiply_by_1_constraints(CS& cs, size_t col, const Expr& bs);
void add_multiply_by_01_constraints(CS& cs, size_t col, const Expr& bs);
void add_fp6_forbenius_map_constraints(CS& cs, size_t col, const Expr& bs);

// ---- Fp12 (src/fp12.rs)
void fill_trace_multiply_by_014(Trace& t, const Fp12& x, const Fp2& o0, const Fp2& o1, const Fp2& o4, size_t start_row, size_t end_row, size_t col);
void fill_trace_fp12_multiplication(Trace& t, const Fp12& x, const Fp12& y, size_t start_row, size_t end_row, size_t col);
void fill_trace_cyclotomic_sq(Trace& t, const Fp12& x, size_t start_row, size_t end_row, size_t col);
void fill_trace_cyclotomic_exp(Trace& t, const Fp12& x, size_t start_row, size_t end_row, size_t col);
void fill_trace_cyclotomic_exp_steps(Trace& t, const Fp12& x, size_t start_row, size_t end_row, size_t col, size_t j0, size_t j1);
void fill_trace_fp12_forbenius_map(Trace& t, const Fp12& x, size_t pow, size_t start_row, size_t end_row, size_t col);
void fill_trace_fp12_conjugate(Trace& t, const Fp12& x, size_t row, size_t col);

void add_multiply_by_014_constraints(CS& cs, size_t col, const Expr& bs);
void add_fp12_multiplication_constraints(CS& cs, size_t col, const Expr& bs);
void add_cyclotomic_sq_constraints(CS& cs, size_t col, const Expr& bs);
void add_cyclotomic_exp_constraints(CS& cs, size_t col, const Expr& bs);
void add_fp12_forbenius_map_constraints(CS& cs, size_t col, const Expr& bs);
void add_fp12_conjugate_constraints(CS& cs, size_t col, const Expr& bs);

}  // namespace starkhip
