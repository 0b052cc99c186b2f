// Flat constraint program ("AIR IR") + the symbolic builder that produces it.
//
// The reference evaluates an AIR by calling S::eval_packed_generic once per LDE
// point (e.g. /root/reference/src/fp12_mul.rs:58-99) with a ConstraintConsumer that
// folds constraint k into acc = acc*alpha + c_k.  A GPU cannot
// call Rust generics, so here each AIR is *described once* as data: the gadget
// functions (air_fp*.cpp) run the same control flow as the reference's
// add_*_constraints functions but on symbolic expressions, and every
// yield_constr.constraint*/ call becomes one record of this program, in the same
// order (the order fixes the power of alpha each constraint gets).
//
// Every constraint of the four AIRs has the shape  mask(kind) * gate_1..gate_g * body
//  where gate_i is a trace cell c or (1 - c) and body is a short
// sum of  coef * cell * cell...  terms.  Consecutive constraints with identical
// (kind, gates) form a GROUP so an evaluator can Horner-fold the bodies and multiply
// by the gate product once:
//     acc_j <- acc_j * alpha_j^m + mask * G * (sum_k body_k alpha_j^(m-1-k))
// which equals the reference's per-constraint fold exactly (field arithmetic is exact).
//
// Code stream (uint32 words):
//   GROUP word : [3:0]=1  [5:4]=kind  [15:8]=n_gates  [31:16]=m (1..AIR_MAX_GROUP)
//   n_gates x cellref
//   m constraints, each = 1+ TERM records; TERM word:
//       [1:0]=nf (cell factors)  [4:2]=ck  [5]=last term of this constraint  [31:6]=idx
//     ck: 0 => +1, 1 => -1, 2 => consts[idx], 3 => +public_inputs[idx], 4 => -public_inputs[idx]
//     followed by nf x cellref
//   END word   : 0
// cellref: [23:0]=column  [30]=next row  [31]=complement (gates only: value 1 - cell)
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "gl.h"

namespace starkhip {

enum : uint32_t { KIND_PLAIN = 0, KIND_TRANSITION = 1, KIND_FIRST = 2, KIND_LAST = 3 };
enum : uint32_t { CK_PLUS = 0, CK_MINUS = 1, CK_CONST = 2, CK_PI = 3, CK_NEG_PI = 4 };
enum : uint32_t { REF_NEXT = 1u << 30, REF_COMPL = 1u << 31, REF_COL_MASK = 0xFFFFFFu };
static const uint32_t AIR_MAX_GROUP = 255;
static const uint64_t AIR_MAGIC = 0x3152495F52494153ULL;  // "SAIR_IR1"

struct AirProgram {
    uint32_t n_cols = 0, n_pis = 0, degree = 0, n_constraints = 0;
    std::vector<uint64_t> consts;
    std::vector<uint32_t> code;
    // per group: word offset of the GROUP word, and number of constraints before it
    std::vector<uint32_t> group_off, group_k0;

    // blob layout (u64 words): magic, n_cols, n_pis, degree, n_constraints, n_consts,
    // n_code_words, n_groups, consts[], code[] (two u32 per u64, zero padded),
    // group_off[] / group_k0[] packed as (off | k0<<32)
    std::vector<uint64_t> serialize() const {
        std::vector<uint64_t> b;
        b.push_back(AIR_MAGIC);
        b.push_back(n_cols);
        b.push_back(n_pis);
        b.push_back(degree);
        b.push_back(n_constraints);
        b.push_back(consts.size());
        b.push_back(code.size());
        b.push_back(group_off.size());
        b.insert(b.end(), consts.begin(), consts.end());
        for (size_t i = 0; i < code.size(); i += 2) {
            uint64_t w = code[i];
            if (i + 1 < code.size()) w |= (uint64_t)code[i + 1] << 32;
            b.push_back(w);
        }
        for (size_t i = 0; i < group_off.size(); i++)
            b.push_back((uint64_t)group_off[i] | ((uint64_t)group_k0[i] << 32));
        return b;
    }
};

// ------------------------------------------------------------------ symbolic layer
struct Mono {
    gl_t coef;                  // canonical, non-zero
    int pi;                     // -1 or public-input index (coefficient multiplies PI[pi])
    std::vector<uint32_t> f;    // sorted cell refs (col | REF_NEXT)
    bool same_vars(const Mono& o) const { return pi == o.pi && f == o.f; }
};

struct Poly {
    std::vector<Mono> m;
    void add_mono(const Mono& x) {
        if (x.coef == 0) return;
        for (size_t i = 0; i < m.size(); i++)
            if (m[i].same_vars(x)) {
                m[i].coef = gl_add(m[i].coef, x.coef);
                if (m[i].coef == 0) m.erase(m.begin() + i);
                return;
            }
        m.push_back(x);
    }
    bool is_const(gl_t c) const { return m.size() == 1 && m[0].pi < 0 && m[0].f.empty() && m[0].coef == c; }
    bool is_single_cell() const { return m.size() == 1 && m[0].pi < 0 && m[0].f.size() == 1 && m[0].coef == 1; }
    // 1 - cell ?
    bool is_complement(uint32_t* cell) const {
        if (m.size() != 2) return false;
        const Mono *one = nullptr, *neg = nullptr;
        for (auto& x : m) {
            if (x.pi < 0 && x.f.empty() && x.coef == 1) one = &x;
            if (x.pi < 0 && x.f.size() == 1 && x.coef == GL_P - 1) neg = &x;
        }
        if (!one || !neg) return false;
        *cell = neg->f[0];
        return true;
    }
};

inline Poly poly_mul(const Poly& a, const Poly& b) {
    Poly r;
    for (auto& x : a.m)
        for (auto& y : b.m) {
            Mono z;
            z.coef = gl_mul(x.coef, y.coef);
            if (x.pi >= 0 && y.pi >= 0) throw std::runtime_error("air_ir: product of two public inputs");
            z.pi = x.pi >= 0 ? x.pi : y.pi;
            z.f = x.f;
            z.f.insert(z.f.end(), y.f.begin(), y.f.end());
            std::sort(z.f.begin(), z.f.end());
            r.add_mono(z);
        }
    return r;
}

// value = prod(gates) * body
struct Expr {
    std::vector<uint32_t> gates;  // cellref, may carry REF_COMPL
    Poly body;

    Expr() {}
    static Expr constant(gl_t c) {
        Expr e;
        Mono m;
        m.coef = gl_from_u64(c);
        m.pi = -1;
        e.body.add_mono(m);
        return e;
    }
    static Expr cell(uint32_t ref) {
        Expr e;
        Mono m;
        m.coef = 1;
        m.pi = -1;
        m.f.push_back(ref);
        e.body.m.push_back(m);
        return e;
    }
    static Expr pub(int i) {
        Expr e;
        Mono m;
        m.coef = 1;
        m.pi = i;
        e.body.m.push_back(m);
        return e;
    }
    // fold gates into the body (expanded polynomial)
    Poly expanded() const {
        Poly p = body;
        for (uint32_t g : gates) {
            Poly q;
            Mono c;
            c.coef = 1;
            c.pi = -1;
            c.f.push_back(g & ~REF_COMPL);
            if (g & REF_COMPL) {
                Mono one;
                one.coef = 1;
                one.pi = -1;
                q.m.push_back(one);
                c.coef = GL_P - 1;
            }
            q.m.push_back(c);
            p = poly_mul(p, q);
        }
        return p;
    }
};

inline Expr operator+(const Expr& a, const Expr& b) {
    Expr r;
    r.body = a.expanded();
    Poly pb = b.expanded();
    for (auto& x : pb.m) r.body.add_mono(x);
    return r;
}
inline Expr operator-(const Expr& a, const Expr& b) {
    Expr r;
    r.body = a.expanded();
    Poly pb = b.expanded();
    for (auto x : pb.m) {
        x.coef = gl_neg(x.coef);
        r.body.add_mono(x);
    }
    return r;
}
inline Expr operator*(const Expr& a, const Expr& b) {
    Expr r;
    r.gates = a.gates;
    r.gates.insert(r.gates.end(), b.gates.begin(), b.gates.end());
    uint32_t c;
    if (a.body.is_const(1)) {
        r.body = b.body;
    } else if (b.body.is_const(1)) {
        r.body = a.body;
    } else if (a.body.is_single_cell()) {
        r.gates.push_back(a.body.m[0].f[0]);
        r.body = b.body;
    } else if (b.body.is_single_cell()) {
        r.gates.push_back(b.body.m[0].f[0]);
        r.body = a.body;
    } else if (a.body.is_complement(&c)) {
        r.gates.push_back(c | REF_COMPL);
        r.body = b.body;
    } else if (b.body.is_complement(&c)) {
        r.gates.push_back(c | REF_COMPL);
        r.body = a.body;
    } else {
        r.body = poly_mul(a.body, b.body);
    }
    return r;
}
inline Expr operator*(const Expr& a, uint64_t k) { return a * Expr::constant(k); }
inline Expr operator-(const Expr& a, uint64_t k) { return a - Expr::constant(k); }
inline Expr operator+(const Expr& a, uint64_t k) { return a + Expr::constant(k); }

// ------------------------------------------------------------------ builder
class AirBuilder {
  public:
    AirBuilder(uint32_t n_cols, uint32_t n_pis, uint32_t degree) {
        prog_.n_cols = n_cols;
        prog_.n_pis = n_pis;
        prog_.degree = degree;
    }

    Expr L(uint32_t col) const { check_col(col); return Expr::cell(col); }
    Expr N(uint32_t col) const { check_col(col); return Expr::cell(col | REF_NEXT); }
    Expr PI(uint32_t i) const {
        if (i >= prog_.n_pis) throw std::runtime_error("air_ir: public input index out of range");
        return Expr::pub((int)i);
    }
    static Expr C(uint64_t c) { return Expr::constant(c); }
    static Expr one() { return Expr::constant(1); }

    void constraint(const Expr& e) { emit(KIND_PLAIN, e); }
    void transition(const Expr& e) { emit(KIND_TRANSITION, e); }
    void first_row(const Expr& e) { emit(KIND_FIRST, e); }
    void last_row(const Expr& e) { emit(KIND_LAST, e); }

    uint32_t count() const { return prog_.n_constraints; }

    AirProgram finish() {
        flush_group();
        prog_.code.push_back(0);
        return prog_;
    }

  private:
    struct Pending {
        std::vector<uint32_t> words;  // encoded terms of one constraint
    };
    AirProgram prog_;
    std::map<uint64_t, uint32_t> const_idx_;
    bool open_ = false;
    uint32_t cur_kind_ = 0;
    std::vector<uint32_t> cur_gates_;
    std::vector<Pending> cur_;

    void check_col(uint32_t col) const {
        if (col >= prog_.n_cols) throw std::runtime_error("air_ir: column " + std::to_string(col) + " out of range");
    }

    uint32_t const_index(gl_t c) {
        auto it = const_idx_.find(c);
        if (it != const_idx_.end()) return it->second;
        uint32_t i = (uint32_t)prog_.consts.size();
        prog_.consts.push_back(c);
        const_idx_[c] = i;
        return i;
    }

    void emit(uint32_t kind, const Expr& e) {
        std::vector<uint32_t> gates = e.gates;
        std::sort(gates.begin(), gates.end());
        Poly body = e.body;
        if (body.m.empty()) {
            // identically-zero constraint: keep its index (alpha power) with an explicit 0 term
            Mono z;
            z.coef = 0;
            z.pi = -1;
            body.m.push_back(z);
        }
        size_t maxf = 0;
        for (auto& m : body.m) maxf = std::max(maxf, m.f.size());
        size_t deg = gates.size() + maxf + ((kind == KIND_FIRST || kind == KIND_LAST) ? 1 : 0);
        if (deg > prog_.degree)
            throw std::runtime_error("air_ir: constraint " + std::to_string(prog_.n_constraints) + " has degree " +
                                     std::to_string(deg) + " > " + std::to_string(prog_.degree));
        if (!open_ || kind != cur_kind_ || gates != cur_gates_ || cur_.size() >= AIR_MAX_GROUP) {
            flush_group();
            open_ = true;
            cur_kind_ = kind;
            cur_gates_ = gates;
        }
        Pending p;
        for (size_t t = 0; t < body.m.size(); t++) {
            const Mono& m = body.m[t];
            if (m.f.size() > 3) throw std::runtime_error("air_ir: term with more than 3 cell factors");
            uint32_t ck, idx = 0;
            if (m.pi >= 0) {
                if (m.coef == 1) ck = CK_PI;
                else if (m.coef == GL_P - 1) ck = CK_NEG_PI;
                else throw std::runtime_error("air_ir: scaled public input");
                idx = (uint32_t)m.pi;
            } else if (m.coef == 1) {
                ck = CK_PLUS;
            } else if (m.coef == GL_P - 1) {
                ck = CK_MINUS;
            } else {
                ck = CK_CONST;
                idx = const_index(m.coef);
            }
            uint32_t w = (uint32_t)m.f.size() | (ck << 2) | ((t + 1 == body.m.size()) ? 1u << 5 : 0) | (idx << 6);
            p.words.push_back(w);
            for (uint32_t r : m.f) p.words.push_back(r);
        }
        cur_.push_back(std::move(p));
        prog_.n_constraints++;
    }

    void flush_group() {
        if (!open_) return;
        prog_.group_off.push_back((uint32_t)prog_.code.size());
        prog_.group_k0.push_back(prog_.n_constraints - (uint32_t)cur_.size());
        prog_.code.push_back(1u | (cur_kind_ << 4) | ((uint32_t)cur_gates_.size() << 8) | ((uint32_t)cur_.size() << 16));
        for (uint32_t g : cur_gates_) prog_.code.push_back(g);
        for (auto& p : cur_) prog_.code.insert(prog_.code.end(), p.words.begin(), p.words.end());
        cur_.clear();
        open_ = false;
    }
};

}  // namespace starkhip
