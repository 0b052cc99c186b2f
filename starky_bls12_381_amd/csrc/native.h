// Software BLS12-381 tower on 12 x u32 limbs: the host-side arithmetic that produces witnesses and
// public inputs for the four AIRs.  Restates /root/reference/src/native.rs (and the helpers of
// src/big_arithmetic.rs) including its non-canonical corner: Neg returns p - x, so -0 == p
// (native.rs:436-443).  Host only.
#pragma once
#include <stdint.h>
#include <string.h>

#include <array>
#include <vector>

namespace starkhip {
namespace bls {

typedef std::array<uint32_t, 12> L12;
typedef std::array<uint32_t, 24> L24;

extern const L12 MODULUS;  // p, little-endian u32 limbs
// curve parameter |x| = 0xd201000000010000 (native.rs:20-22)
static const uint64_t BLS_X = 0xd201000000010000ULL;

// ---- limb helpers (native.rs:55-181, 234-281)
void multiply_by_slice(const L12& x, uint32_t y, uint32_t res[13], uint32_t carries[12]);
void add_u32_slices(const L24& x, const L24& y, L24& sum, L24& carries);
void add_u32_slices_12(const L12& x, const L12& y, L12& sum, L12& carries);
void sub_u32_slices(const L24& x, const L24& y, L24& diff, L24& borrows);
void sub_u32_slices_12(const L12& x, const L12& y, L12& diff, L12& borrows);
void mul_u32_slice_u32(const L12& x, uint32_t y, L12& res, L12& carries);
// x / p and x % p for a 24-limb x whose quotient fits 12 limbs (get_div_rem_modulus_from_biguint_12)
void div_rem_modulus(const L24& x, L12& div, L12& rem);
L24 mul_wide(const L12& x, const L12& y);  // full product (mul_fp_without_reduction)
L24 widen(const L12& x);

struct Fp {
    L12 l;
    Fp() { l.fill(0); }
    explicit Fp(const L12& v) : l(v) {}
    static Fp from_u32(uint32_t v) { Fp r; r.l[0] = v; return r; }
    static Fp one() { return from_u32(1); }
    static Fp zero() { return Fp(); }
    bool operator==(const Fp& o) const { return l == o.l; }
    Fp invert() const;
};
Fp operator+(const Fp& a, const Fp& b);  // add_fp: one conditional subtraction of p
Fp operator-(const Fp& a, const Fp& b);  // sub_fp: (p + a - b) mod p
Fp operator*(const Fp& a, const Fp& b);  // mul_fp
Fp operator-(const Fp& a);               // p - a (NOT reduced)
Fp operator/(const Fp& a, const Fp& b);

struct Fp2 {
    Fp c[2];
    Fp2() {}
    Fp2(const Fp& a, const Fp& b) { c[0] = a; c[1] = b; }
    static Fp2 one() { return Fp2(Fp::one(), Fp::zero()); }
    static Fp2 zero() { return Fp2(); }
    bool operator==(const Fp2& o) const { return c[0] == o.c[0] && c[1] == o.c[1]; }
    Fp2 multiply_by_b() const;
    Fp2 mul_by_nonresidue() const;
    Fp2 invert() const;
    Fp2 forbenius_map(size_t pow) const;
};
Fp2 operator+(const Fp2& a, const Fp2& b);
Fp2 operator-(const Fp2& a, const Fp2& b);
Fp2 operator*(const Fp2& a, const Fp2& b);
Fp2 operator*(const Fp2& a, const Fp& b);
Fp2 operator-(const Fp2& a);

struct Fp6 {
    Fp c[6];
    Fp2 c2(int i) const { return Fp2(c[2 * i], c[2 * i + 1]); }
    static Fp6 from_fp2(const Fp2& a, const Fp2& b, const Fp2& d) {
        Fp6 r;
        r.c[0] = a.c[0]; r.c[1] = a.c[1]; r.c[2] = b.c[0]; r.c[3] = b.c[1]; r.c[4] = d.c[0]; r.c[5] = d.c[1];
        return r;
    }
    Fp6 invert() const;
    Fp6 multiply_by_01(const Fp2& b0, const Fp2& b1) const;
    Fp6 multiply_by_1(const Fp2& b1) const;
    Fp6 forbenius_map(size_t pow) const;
};
Fp6 operator+(const Fp6& a, const Fp6& b);
Fp6 operator-(const Fp6& a, const Fp6& b);
Fp6 operator*(const Fp6& a, const Fp6& b);
Fp6 operator-(const Fp6& a);
Fp6 mul_by_nonresidue(const Fp6& x);

struct Fp12 {
    Fp c[12];
    static Fp12 one() { Fp12 r; r.c[0] = Fp::one(); return r; }
    Fp6 c6(int i) const { Fp6 r; for (int k = 0; k < 6; k++) r.c[k] = c[6 * i + k]; return r; }
    Fp2 c2(int i) const { return Fp2(c[2 * i], c[2 * i + 1]); }
    static Fp12 from_fp6(const Fp6& a, const Fp6& b) { Fp12 r; for (int k = 0; k < 6; k++) { r.c[k] = a.c[k]; r.c[6 + k] = b.c[k]; } return r; }
    bool operator==(const Fp12& o) const { for (int k = 0; k < 12; k++) if (!(c[k] == o.c[k])) return false; return true; }
    Fp12 invert() const;
    Fp12 forbenius_map(size_t pow) const;
    Fp12 multiply_by_014(const Fp2& o0, const Fp2& o1, const Fp2& o4) const;
    Fp12 conjugate() const;
    Fp12 cyclotomic_square() const;
    Fp12 cyclotomic_exponent() const;
    Fp12 final_exponentiate() const;
    void to_limbs(uint32_t out[144]) const { for (int k = 0; k < 12; k++) memcpy(out + 12 * k, c[k].l.data(), 48); }
    static Fp12 from_limbs(const uint32_t in[144]) { Fp12 r; for (int k = 0; k < 12; k++) memcpy(r.c[k].l.data(), in + 12 * k, 48); return r; }
};
Fp12 operator+(const Fp12& a, const Fp12& b);
Fp12 operator*(const Fp12& a, const Fp12& b);
Fp12 operator/(const Fp12& a, const Fp12& b);

void fp4_square(const Fp2& a, const Fp2& b, Fp2& out0, Fp2& out1);

// Frobenius coefficient tables (native.rs:1052-1056, 1069-1125, 1148-1199)
extern const Fp FP2_FROBENIUS_COEFF[2];
const Fp2* fp6_frobenius_coeff_1();   // [6]
const Fp2* fp6_frobenius_coeff_2();   // [6]
const Fp2* fp12_frobenius_coeff();    // [12]

typedef std::array<Fp2, 3> EllCoeff;
std::vector<EllCoeff> calc_pairing_precomp(const Fp2& x, const Fp2& y, const Fp2& z);
Fp12 miller_loop(const Fp& g1x, const Fp& g1y, const Fp2& g2x, const Fp2& g2y, const Fp2& g2z);
std::vector<Fp2> calc_precomp_stuff_loop0(const Fp2& rx, const Fp2& ry, const Fp2& rz);
std::vector<Fp2> calc_precomp_stuff_loop1(const Fp2& rx, const Fp2& ry, const Fp2& rz, const Fp2& qx, const Fp2& qy);
Fp mod_inverse_of_two();

Fp fp_from_decimal(const char* s);

}  // namespace bls
}  // namespace starkhip
