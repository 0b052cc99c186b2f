// Host evaluator of the flat constraint program over an arbitrary field (used by the verifier
// over the quadratic extension; the device kernel in kernels_quotient.hip is the base-field twin).
// Uses the grouped Horner fold described in air_ir.h.
#pragma once
#include <vector>

#include "air_ir.h"

namespace starkhip {

// Field policy for gl2_t
struct ExtOps {
    typedef gl2_t T;
    static T zero() { return gl2_zero(); }
    static T one() { return gl2_one(); }
    static T from_base(gl_t x) { return gl2_from_base(x); }
    static T add(T a, T b) { return gl2_add(a, b); }
    static T sub(T a, T b) { return gl2_sub(a, b); }
    static T mul(T a, T b) { return gl2_mul(a, b); }
    static T mul_base(T a, gl_t b) { return gl2_mul_base(a, b); }
};

// Field policy for gl_t (base field)
struct BaseOps {
    typedef gl_t T;
    static T zero() { return 0; }
    static T one() { return 1; }
    static T from_base(gl_t x) { return x; }
    static T add(T a, T b) { return gl_add(a, b); }
    static T sub(T a, T b) { return gl_sub(a, b); }
    static T mul(T a, T b) { return gl_mul(a, b); }
    static T mul_base(T a, gl_t b) { return gl_mul(a, b); }
};

// acc[j] = sum_k mask(kind_k) * c_k * alpha_j^(K-1-k)
template <class O>
void air_eval_folded(const AirProgram& p, const typename O::T* local, const typename O::T* next, const gl_t* pis,
                     const typename O::T masks[4], const typename O::T* alphas, int n_alpha, typename O::T* acc) {
    typedef typename O::T T;
    const uint32_t* w = p.code.data();
    for (int j = 0; j < n_alpha; j++) acc[j] = O::zero();
    std::vector<T> t(n_alpha), apow(n_alpha);
    while ((*w & 15u) == 1u) {
        uint32_t kind = (*w >> 4) & 3u, ng = (*w >> 8) & 255u, m = *w >> 16;
        w++;
        T G = masks[kind];
        for (uint32_t g = 0; g < ng; g++, w++) {
            T v = ((*w & REF_NEXT) ? next : local)[*w & REF_COL_MASK];
            if (*w & REF_COMPL) v = O::sub(O::one(), v);
            G = O::mul(G, v);
        }
        for (int j = 0; j < n_alpha; j++) {
            t[j] = O::zero();
            apow[j] = O::one();
        }
        for (uint32_t c = 0; c < m; c++) {
            T body = O::zero();
            for (;;) {
                uint32_t tw = *w++;
                uint32_t nf = tw & 3u, ck = (tw >> 2) & 7u, idx = tw >> 6;
                T v = O::one();
                for (uint32_t f = 0; f < nf; f++, w++) v = O::mul(v, ((*w & REF_NEXT) ? next : local)[*w & REF_COL_MASK]);
                switch (ck) {
                    case CK_PLUS: body = O::add(body, v); break;
                    case CK_MINUS: body = O::sub(body, v); break;
                    case CK_CONST: body = O::add(body, O::mul_base(v, p.consts[idx])); break;
                    case CK_PI: body = O::add(body, O::mul_base(v, pis[idx])); break;
                    default: body = O::sub(body, O::mul_base(v, pis[idx])); break;
                }
                if (tw & 32u) break;
            }
            for (int j = 0; j < n_alpha; j++) {
                t[j] = O::add(O::mul(t[j], alphas[j]), body);
                apow[j] = O::mul(apow[j], alphas[j]);
            }
        }
        for (int j = 0; j < n_alpha; j++) acc[j] = O::add(O::mul(acc[j], apow[j]), O::mul(G, t[j]));
    }
}

}  // namespace starkhip
