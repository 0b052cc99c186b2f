// Software BLS12-381 tower (see native.h).  Each function names the reference function it follows.
#include "native.h"

#include <stdexcept>

namespace starkhip {
namespace bls {

// p = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab  (native.rs:12-14)
const L12 MODULUS = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                     0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};

// ------------------------------------------------------------------ limb helpers
void multiply_by_slice(const L12& x, uint32_t y, uint32_t res[13], uint32_t carries[12]) {  // native.rs:55-69
    uint32_t prev = 0;
    for (int i = 0; i < 12; i++) {
        uint64_t t = (uint64_t)x[i] * y + prev;
        res[i] = (uint32_t)t;
        prev = (uint32_t)(t >> 32);
        carries[i] = prev;
    }
    res[12] = prev;
}
void add_u32_slices(const L24& x, const L24& y, L24& sum, L24& carries) {  // native.rs:71-84
    uint32_t prev = 0;
    for (int i = 0; i < 24; i++) {
        uint64_t s = (uint64_t)x[i] + y[i] + prev;
        sum[i] = (uint32_t)s;
        prev = (uint32_t)(s >> 32);
        carries[i] = prev;
    }
}
void add_u32_slices_12(const L12& x, const L12& y, L12& sum, L12& carries) {  // native.rs:86-99
    uint32_t prev = 0;
    for (int i = 0; i < 12; i++) {
        uint64_t s = (uint64_t)x[i] + y[i] + prev;
        sum[i] = (uint32_t)s;
        prev = (uint32_t)(s >> 32);
        carries[i] = prev;
    }
}
// native.rs:102-118: note `y[i] + prev_borrow` is evaluated in u32 (wraps when y[i] == 0xffffffff and a borrow is pending;
// a release build of the reference wraps silently) -- keep u32 semantics.
void sub_u32_slices(const L24& x, const L24& y, L24& diff, L24& borrows) {
    uint32_t prev = 0;
    for (int i = 0; i < 24; i++) {
        uint32_t yb = y[i] + prev;
        if (x[i] >= yb) {
            diff[i] = x[i] - y[i] - prev;
            borrows[i] = 0;
            prev = 0;
        } else {
            diff[i] = (uint32_t)((1ULL << 32) + x[i] - y[i] - prev);
            borrows[i] = 1;
            prev = 1;
        }
    }
}
void sub_u32_slices_12(const L12& x, const L12& y, L12& diff, L12& borrows) {  // native.rs:121-138
    uint32_t prev = 0;
    for (int i = 0; i < 12; i++) {
        uint32_t yb = y[i] + prev;
        if (x[i] >= yb) {
            diff[i] = x[i] - y[i] - prev;
            borrows[i] = 0;
            prev = 0;
        } else {
            diff[i] = (uint32_t)((1ULL << 32) + x[i] - y[i] - prev);
            borrows[i] = 1;
            prev = 1;
        }
    }
    if (borrows[11] != 0) throw std::runtime_error("sub_u32_slices_12: x < y");
}
void mul_u32_slice_u32(const L12& x, uint32_t y, L12& res, L12& carries) {  // native.rs:140-152
    uint32_t prev = 0;
    for (int i = 0; i < 12; i++) {
        uint64_t t = (uint64_t)x[i] * y + prev;
        res[i] = (uint32_t)t;
        carries[i] = (uint32_t)(t >> 32);
        prev = carries[i];
    }
    if (prev != 0) throw std::runtime_error("mul_u32_slice_u32: overflow");
}
L24 widen(const L12& x) {
    L24 r;
    r.fill(0);
    for (int i = 0; i < 12; i++) r[i] = x[i];
    return r;
}
L24 mul_wide(const L12& x, const L12& y) {  // mul_fp_without_reduction, native.rs:495-500
    uint64_t acc[25] = {0};
    L24 r;
    uint32_t t[24] = {0};
    for (int i = 0; i < 12; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < 12; j++) {
            uint64_t cur = (uint64_t)t[i + j] + (uint64_t)x[j] * y[i] + carry;
            t[i + j] = (uint32_t)cur;
            carry = cur >> 32;
        }
        t[i + 12] = (uint32_t)carry;
    }
    (void)acc;
    for (int i = 0; i < 24; i++) r[i] = t[i];
    return r;
}

// floor(x / p) and x mod p for the fixed 12-limb divisor p; x / p must fit 12 limbs (get_div_rem_modulus_from_biguint_12,
// native.rs:277-281 -- BigUint division there; any exact long division gives the same two numbers).  Knuth's algorithm D on 64-bit
// limbs -- six divisor limbs, one hardware 128 / 64 division per quotient limb -- and only as many quotient limbs as x has above p's
// length: the reduce gadget's operands (twelve or thirteen 32-bit limbs: quotient below 16) and the subtraction's p + a - b take one
// step, a product of two field elements seven.  Round 5 ran the 32-bit form over all thirteen positions whatever x was: a third of a
// FinalExp recording (tools/experiments/recording_cpu_probe.cpp).
void div_rem_modulus(const L24& x, L12& div, L12& rem) {
    constexpr int n = 6, s = 3;  // p's top 64-bit limb 0x1a0111ea397fe69a has three leading zeros
    static const uint64_t V[n] = {  // p << 3
        (0xb9feffffffffaaabull << s),
        (0x1eabfffeb153ffffull << s) | (0xb9feffffffffaaabull >> (64 - s)),
        (0x6730d2a0f6b0f624ull << s) | (0x1eabfffeb153ffffull >> (64 - s)),
        (0x64774b84f38512bfull << s) | (0x6730d2a0f6b0f624ull >> (64 - s)),
        (0x4b1ba7b6434bacd7ull << s) | (0x64774b84f38512bfull >> (64 - s)),
        (0x1a0111ea397fe69aull << s) | (0x4b1ba7b6434bacd7ull >> (64 - s))};
    uint64_t X[12];
    for (int i = 0; i < 12; i++) X[i] = (uint64_t)x[2 * i] | ((uint64_t)x[2 * i + 1] << 32);
    int m = 12;
    while (m > n && X[m - 1] == 0) m--;
    uint64_t u[13], q[7] = {0, 0, 0, 0, 0, 0, 0};
    u[m] = X[m - 1] >> (64 - s);
    for (int i = m - 1; i > 0; i--) u[i] = (X[i] << s) | (X[i - 1] >> (64 - s));
    u[0] = X[0] << s;
    typedef unsigned __int128 u128;
    for (int j = m - n; j >= 0; j--) {
        uint64_t qhat, rhat;
        bool check = true;
        if (u[j + n] >= V[n - 1]) {  // == V[n - 1] (the running remainder is below the divisor): the quotient limb is 2^64 - 1 or 2^64 - 2
            qhat = ~0ull;
            const u128 r = (u128)u[j + n - 1] + V[n - 1];
            rhat = (uint64_t)r;
            check = (r >> 64) == 0;
        } else {
            asm("divq %4" : "=a"(qhat), "=d"(rhat) : "a"(u[j + n - 1]), "d"(u[j + n]), "r"(V[n - 1]) : "cc");
        }
        while (check && (u128)qhat * V[n - 2] > (((u128)rhat << 64) | u[j + n - 2])) {
            qhat--;
            const u128 r = (u128)rhat + V[n - 1];
            rhat = (uint64_t)r;
            check = (r >> 64) == 0;
        }
        uint64_t carry = 0, borrow = 0;
        for (int i = 0; i < n; i++) {
            const u128 pr = (u128)qhat * V[i] + carry;
            carry = (uint64_t)(pr >> 64);
            const uint64_t lo = (uint64_t)pr, t = u[i + j] - lo, t2 = t - borrow;
            borrow = (u[i + j] < lo) | (t < borrow);
            u[i + j] = t2;
        }
        const uint64_t t = u[j + n] - carry, t2 = t - borrow;
        const bool negative = (u[j + n] < carry) | (t < borrow);
        u[j + n] = t2;
        if (negative) {  // qhat was one too large (probability ~ 2^-63): add the divisor back
            qhat--;
            uint64_t c = 0;
            for (int i = 0; i < n; i++) {
                const u128 a = (u128)u[i + j] + V[i] + c;
                u[i + j] = (uint64_t)a;
                c = (uint64_t)(a >> 64);
            }
            u[j + n] += c;
        }
        q[j] = qhat;
    }
    if (q[6] != 0) throw std::runtime_error("div_rem_modulus: quotient does not fit 12 limbs");
    for (int i = 0; i < 6; i++) {
        div[2 * i] = (uint32_t)q[i];
        div[2 * i + 1] = (uint32_t)(q[i] >> 32);
        const uint64_t r = (u[i] >> s) | (u[i + 1] << (64 - s));
        rem[2 * i] = (uint32_t)r;
        rem[2 * i + 1] = (uint32_t)(r >> 32);
    }
}

static bool less_than(const uint32_t* a, const uint32_t* b, int len) {  // big_arithmetic.rs big_less_than
    for (int i = len - 1; i >= 0; i--) {
        if (a[i] < b[i]) return true;
        if (b[i] < a[i]) return false;
    }
    return false;
}

// ------------------------------------------------------------------ Fp
Fp operator+(const Fp& a, const Fp& b) {  // add_fp, native.rs:459-474
    uint32_t s[13], m[13];
    uint32_t carry = 0;
    for (int i = 0; i < 12; i++) {
        uint64_t t = (uint64_t)a.l[i] + b.l[i] + carry;
        s[i] = (uint32_t)t;
        carry = (uint32_t)(t >> 32);
        m[i] = MODULUS[i];
    }
    s[12] = carry;
    m[12] = 0;
    Fp r;
    if (less_than(s, m, 13)) {
        for (int i = 0; i < 12; i++) r.l[i] = s[i];
    } else {  // big_sub once
        uint32_t c = 0;
        for (int i = 0; i < 13; i++) {
            uint64_t bc = (uint64_t)m[i] + c;
            uint32_t d;
            if ((uint64_t)s[i] >= bc) {
                d = s[i] - (uint32_t)bc;
                c = 0;
            } else {
                d = (uint32_t)((1ULL << 32) + s[i] - bc);
                c = 1;
            }
            if (i < 12) r.l[i] = d;
        }
    }
    return r;
}
Fp operator-(const Fp& a, const Fp& b) {  // sub_fp, native.rs:507-514: (p + a - b) mod p
    // p + a - b as a 13-limb value (b <= p + a is required; the reference would panic otherwise)
    uint32_t t[13];
    uint32_t carry = 0;
    for (int i = 0; i < 12; i++) {
        uint64_t s = (uint64_t)MODULUS[i] + a.l[i] + carry;
        t[i] = (uint32_t)s;
        carry = (uint32_t)(s >> 32);
    }
    t[12] = carry;
    int64_t borrow = 0;
    for (int i = 0; i < 13; i++) {
        int64_t d = (int64_t)t[i] - (i < 12 ? (int64_t)b.l[i] : 0) + borrow;
        t[i] = (uint32_t)d;
        borrow = d >> 32;
    }
    if (borrow) throw std::runtime_error("sub_fp: underflow");
    L24 w;
    w.fill(0);
    for (int i = 0; i < 13; i++) w[i] = t[i];
    L12 q;
    Fp r;
    div_rem_modulus(w, q, r.l);
    return r;
}
Fp operator*(const Fp& a, const Fp& b) {  // mul_fp, native.rs:486-493
    L24 w = mul_wide(a.l, b.l);
    L12 q;
    Fp r;
    div_rem_modulus(w, q, r.l);
    return r;
}
Fp operator-(const Fp& a) {  // Neg, native.rs:436-443: p - x without reduction
    Fp r;
    int64_t borrow = 0;
    for (int i = 0; i < 12; i++) {
        int64_t d = (int64_t)MODULUS[i] - (int64_t)a.l[i] + borrow;
        r.l[i] = (uint32_t)d;
        borrow = d >> 32;
    }
    if (borrow) throw std::runtime_error("neg_fp: operand above the modulus");
    return r;
}
Fp Fp::invert() const {  // mod_inverse (native.rs:183-223) returns the canonical inverse, 0 for 0; Fermat gives the same value
    // exponent p - 2
    L12 e = MODULUS;
    e[0] -= 2;
    Fp result = Fp::one(), base = *this;
    // reduce base first (the reference's egcd works on the integer value; x mod p has the same inverse)
    {
        L12 q;
        Fp t;
        div_rem_modulus(widen(base.l), q, t.l);
        base = t;
    }
    for (int i = 0; i < 384; i++) {
        if ((e[i / 32] >> (i % 32)) & 1) result = result * base;
        base = base * base;
    }
    return result;
}
Fp operator/(const Fp& a, const Fp& b) { return a * b.invert(); }  // native.rs:406-414

Fp fp_from_decimal(const char* s) {
    Fp r;
    for (; *s; s++) {
        uint64_t carry = (uint64_t)(*s - '0');
        for (int i = 0; i < 12; i++) {
            uint64_t t = (uint64_t)r.l[i] * 10 + carry;
            r.l[i] = (uint32_t)t;
            carry = t >> 32;
        }
    }
    return r;
}

Fp mod_inverse_of_two() { return Fp::from_u32(2).invert(); }

// ------------------------------------------------------------------ Fp2 (native.rs:523-710)
Fp2 operator+(const Fp2& a, const Fp2& b) { return Fp2(a.c[0] + b.c[0], a.c[1] + b.c[1]); }
Fp2 operator-(const Fp2& a, const Fp2& b) { return Fp2(a.c[0] - b.c[0], a.c[1] - b.c[1]); }
Fp2 operator*(const Fp2& a, const Fp2& b) {  // mul_fp2, native.rs:702-710
    Fp c0 = (a.c[0] * b.c[0]) - (a.c[1] * b.c[1]);
    Fp c1 = (a.c[0] * b.c[1]) + (a.c[1] * b.c[0]);
    return Fp2(c0, c1);
}
Fp2 operator*(const Fp2& a, const Fp& b) { return Fp2(a.c[0] * b, a.c[1] * b); }  // native.rs:663-684
Fp2 operator-(const Fp2& a) { return Fp2(-a.c[0], -a.c[1]); }
Fp2 Fp2::multiply_by_b() const {  // native.rs:539-543
    Fp t0 = c[0] * Fp::from_u32(4), t1 = c[1] * Fp::from_u32(4);
    return Fp2(t0 - t1, t0 + t1);
}
Fp2 Fp2::mul_by_nonresidue() const { return Fp2(c[0] - c[1], c[0] + c[1]); }  // native.rs:545-549
Fp2 Fp2::invert() const {  // native.rs:551-560
    Fp factor = ((c[0] * c[0]) + (c[1] * c[1])).invert();
    return Fp2(factor * c[0], factor * (-c[1]));
}

const Fp FP2_FROBENIUS_COEFF[2] = {
    Fp::from_u32(1),
    fp_from_decimal("4002409555221667393417789825735904156556882819939007885332058136124031650490837864442687629129015664037894272559786")};
Fp2 Fp2::forbenius_map(size_t pow) const { return Fp2(c[0], c[1] * FP2_FROBENIUS_COEFF[pow % 2]); }  // native.rs:1058-1064

// ------------------------------------------------------------------ Fp6 (native.rs:716-918)
Fp6 operator+(const Fp6& a, const Fp6& b) { Fp6 r; for (int i = 0; i < 6; i++) r.c[i] = a.c[i] + b.c[i]; return r; }
Fp6 operator-(const Fp6& a, const Fp6& b) { Fp6 r; for (int i = 0; i < 6; i++) r.c[i] = a.c[i] - b.c[i]; return r; }
Fp6 operator-(const Fp6& a) { Fp6 r; for (int i = 0; i < 6; i++) r.c[i] = -a.c[i]; return r; }
Fp6 operator*(const Fp6& x, const Fp6& y) {  // mul_fp6, native.rs:824-861
    Fp2 c0 = x.c2(0), c1 = x.c2(1), c2 = x.c2(2), r0 = y.c2(0), r1 = y.c2(1), r2 = y.c2(2);
    Fp2 t0 = c0 * r0, t1 = c1 * r1, t2 = c2 * r2;
    Fp2 t5 = (c1 + c2) * (r1 + r2);
    Fp2 t8 = ((t5 - t1) - t2).mul_by_nonresidue();
    Fp2 xx = t8 + t0;
    Fp2 t11 = (c0 + c1) * (r0 + r1);
    Fp2 t13 = (t11 - t0) - t1;
    Fp2 yy = t13 + t2.mul_by_nonresidue();
    Fp2 t17 = (c0 + c2) * (r0 + r2);
    Fp2 zz = ((t17 - t0) - t2) + t1;
    return Fp6::from_fp2(xx, yy, zz);
}
Fp6 mul_by_nonresidue(const Fp6& x) {  // native.rs:863-873
    Fp2 c0 = x.c2(2).mul_by_nonresidue();
    return Fp6::from_fp2(c0, x.c2(0), x.c2(1));
}
Fp6 Fp6::invert() const {  // native.rs:720-734
    Fp2 c0 = c2(0), c1 = c2(1), c2_ = c2(2);
    Fp2 t0 = (c0 * c0) - (c2_ * c1).mul_by_nonresidue();
    Fp2 t1 = (c2_ * c2_).mul_by_nonresidue() - (c0 * c1);
    Fp2 t2 = (c1 * c1) - (c0 * c2_);
    Fp2 t4 = (((c2_ * t1) + (c1 * t2)).mul_by_nonresidue() + (c0 * t0)).invert();
    return Fp6::from_fp2(t4 * t0, t4 * t1, t4 * t2);
}
Fp6 Fp6::multiply_by_01(const Fp2& b0, const Fp2& b1) const {  // native.rs:876-899
    Fp2 c0 = c2(0), c1 = c2(1), c2_ = c2(2);
    Fp2 t0 = c0 * b0, t1 = c1 * b1;
    Fp2 x = (c2_ * b1).mul_by_nonresidue() + t0;
    Fp2 y = (((b0 + b1) * (c0 + c1)) - t0) - t1;
    Fp2 z = (c2_ * b0) + t1;
    return Fp6::from_fp2(x, y, z);
}
Fp6 Fp6::multiply_by_1(const Fp2& b1) const {  // native.rs:901-917
    Fp2 c0 = c2(0), c1 = c2(1), c2_ = c2(2);
    return Fp6::from_fp2((c2_ * b1).mul_by_nonresidue(), c0 * b1, c1 * b1);
}

static Fp2 fp2_dec(const char* a, const char* b) { return Fp2(fp_from_decimal(a), fp_from_decimal(b)); }
#define D_A "4002409555221667392624310435006688643935503118305586438271171395842971157480381377015405980053539358417135540939436"
#define D_B "793479390729215512621379701633421447060886740281060493010456487427281649075476305620758731620350"
#define D_C "4002409555221667392624310435006688643935503118305586438271171395842971157480381377015405980053539358417135540939437"
#define D_PM1 "4002409555221667393417789825735904156556882819939007885332058136124031650490837864442687629129015664037894272559786"
#define D_B1 "793479390729215512621379701633421447060886740281060493010456487427281649075476305620758731620351"
const Fp2* fp6_frobenius_coeff_1() {  // native.rs:1069-1096
    static const Fp2 t[6] = {fp2_dec("1", "0"), fp2_dec("0", D_A), fp2_dec(D_B, "0"), fp2_dec("0", "1"), fp2_dec(D_B, "0"), fp2_dec("0", D_A)};
    return t;
}
const Fp2* fp6_frobenius_coeff_2() {  // native.rs:1098-1125
    static const Fp2 t[6] = {fp2_dec("1", "0"), fp2_dec(D_C, "0"), fp2_dec(D_A, "0"), fp2_dec(D_PM1, "0"), fp2_dec(D_B, "0"), fp2_dec("0", D_B1)};
    return t;
}
#define E_1A "3850754370037169011952147076051364057158807420970682438676050522613628423219637725072182697113062777891589506424760"
#define E_1B "151655185184498381465642749684540099398075398968325446656007613510403227271200139370504932015952886146304766135027"
#define E_3A "2973677408986561043442465346520108879172042883009249989176415018091420807192182638567116318576472649347015917690530"
#define E_3B "1028732146235106349975324479215795277384839936929757896155643118032610843298655225875571310552543014690878354869257"
#define E_5A "3125332594171059424908108096204648978570118281977575435832422631601824034463382777937621250592425535493320683825557"
#define E_5B "877076961050607968509681729531255177986764537961432449499635504522207616027455086505066378536590128544573588734230"
const Fp2* fp12_frobenius_coeff() {  // native.rs:1148-1199
    static const Fp2 t[12] = {fp2_dec("1", "0"),   fp2_dec(E_1A, E_1B), fp2_dec(D_B1, "0"), fp2_dec(E_3A, E_3B), fp2_dec(D_B, "0"), fp2_dec(E_5A, E_5B),
                              fp2_dec(D_PM1, "0"), fp2_dec(E_1B, E_1A), fp2_dec(D_A, "0"),  fp2_dec(E_3B, E_3A), fp2_dec(D_C, "0"), fp2_dec(E_5B, E_5A)};
    return t;
}
Fp6 Fp6::forbenius_map(size_t pow) const {  // native.rs:1126-1144
    return Fp6::from_fp2(c2(0).forbenius_map(pow), c2(1).forbenius_map(pow) * fp6_frobenius_coeff_1()[pow % 6],
                         c2(2).forbenius_map(pow) * fp6_frobenius_coeff_2()[pow % 6]);
}

// ------------------------------------------------------------------ Fp12 (native.rs:920-1345)
Fp12 operator+(const Fp12& a, const Fp12& b) { Fp12 r; for (int i = 0; i < 12; i++) r.c[i] = a.c[i] + b.c[i]; return r; }
Fp12 operator*(const Fp12& x, const Fp12& y) {  // mul_fp_12, native.rs:1009-1027
    Fp6 c0 = x.c6(0), c1 = x.c6(1), r0 = y.c6(0), r1 = y.c6(1);
    Fp6 t0 = c0 * r0, t1 = c1 * r1;
    Fp6 xx = t0 + mul_by_nonresidue(t1);
    Fp6 t5 = (c0 + c1) * (r0 + r1);
    Fp6 yy = (t5 - t0) - t1;
    return Fp12::from_fp6(xx, yy);
}
Fp12 Fp12::invert() const {  // native.rs:930-938
    Fp6 c0 = c6(0), c1 = c6(1);
    Fp6 t = ((c0 * c0) - mul_by_nonresidue(c1 * c1)).invert();
    return Fp12::from_fp6(c0 * t, -(c1 * t));
}
Fp12 operator/(const Fp12& a, const Fp12& b) { return a * b.invert(); }
Fp12 Fp12::forbenius_map(size_t pow) const {  // native.rs:1201-1221
    Fp6 r0 = c6(0).forbenius_map(pow);
    Fp6 t = c6(1).forbenius_map(pow);
    Fp2 coeff = fp12_frobenius_coeff()[pow % 12];
    return Fp12::from_fp6(r0, Fp6::from_fp2(t.c2(0) * coeff, t.c2(1) * coeff, t.c2(2) * coeff));
}
Fp12 Fp12::multiply_by_014(const Fp2& o0, const Fp2& o1, const Fp2& o4) const {  // native.rs:1225-1241
    Fp6 c0 = c6(0), c1 = c6(1);
    Fp6 t0 = c0.multiply_by_01(o0, o1);
    Fp6 t1 = c1.multiply_by_1(o4);
    Fp6 t2 = mul_by_nonresidue(t1);
    Fp6 x = t2 + t0;
    Fp6 t3 = c1 + c0;
    Fp2 t4 = o1 + o4;
    Fp6 t5 = t3.multiply_by_01(o0, t4);
    Fp6 y = (t5 - t0) - t1;
    return Fp12::from_fp6(x, y);
}
Fp12 Fp12::conjugate() const {  // native.rs:1243-1249 (uses the unreduced negation)
    Fp12 r = *this;
    for (int i = 6; i < 12; i++) r.c[i] = -r.c[i];
    return r;
}
void fp4_square(const Fp2& a, const Fp2& b, Fp2& out0, Fp2& out1) {  // native.rs:225-232
    Fp2 a2 = a * a, b2 = b * b;
    out0 = b2.mul_by_nonresidue() + a2;
    out1 = (((a + b) * (a + b)) - a2) - b2;
}
Fp12 Fp12::cyclotomic_square() const {  // native.rs:1251-1298
    Fp two = Fp::from_u32(2);
    Fp2 c0c0 = c2(0), c0c1 = c2(1), c0c2 = c2(2), c1c0 = c2(3), c1c1 = c2(4), c1c2 = c2(5);
    Fp2 t00, t01, t10, t11, t20, t21;
    fp4_square(c0c0, c1c1, t00, t01);
    fp4_square(c1c0, c0c2, t10, t11);
    fp4_square(c0c1, c1c2, t20, t21);
    Fp2 t3 = t21.mul_by_nonresidue();
    Fp2 r0 = ((t00 - c0c0) * two) + t00;
    Fp2 r1 = ((t10 - c0c1) * two) + t10;
    Fp2 r2 = ((t20 - c0c2) * two) + t20;
    Fp2 r3 = ((t3 + c1c0) * two) + t3;
    Fp2 r4 = ((t01 + c1c1) * two) + t01;
    Fp2 r5 = ((t11 + c1c2) * two) + t11;
    Fp12 r;
    const Fp2* rs[6] = {&r0, &r1, &r2, &r3, &r4, &r5};
    for (int i = 0; i < 6; i++) {
        r.c[2 * i] = rs[i]->c[0];
        r.c[2 * i + 1] = rs[i]->c[1];
    }
    return r;
}
Fp12 Fp12::cyclotomic_exponent() const {  // native.rs:1300-1309
    Fp12 z = Fp12::one();
    for (int i = 63; i >= 0; i--) {
        z = z.cyclotomic_square();
        if ((BLS_X >> i) & 1) z = z * *this;
    }
    return z;
}
Fp12 Fp12::final_exponentiate() const {  // native.rs:1311-1345
    const Fp12& self = *this;
    Fp12 t_0 = self.forbenius_map(6);
    Fp12 t_1 = t_0 / self;
    Fp12 t_2 = t_1.forbenius_map(2);
    Fp12 t_3 = t_2 * t_1;
    Fp12 t_4 = t_3.cyclotomic_exponent();
    Fp12 t_5 = t_4.conjugate();
    Fp12 t_6 = t_3.cyclotomic_square();
    Fp12 t_7 = t_6.conjugate();
    Fp12 t_8 = t_7 * t_5;
    Fp12 t_9 = t_8.cyclotomic_exponent();
    Fp12 t_10 = t_9.conjugate();
    Fp12 t_11 = t_10.cyclotomic_exponent();
    Fp12 t_12 = t_11.conjugate();
    Fp12 t_13 = t_12.cyclotomic_exponent();
    Fp12 t_14 = t_13.conjugate();
    Fp12 t_15 = t_5.cyclotomic_square();
    Fp12 t_16 = t_14 * t_15;
    Fp12 t_17 = t_16.cyclotomic_exponent();
    Fp12 t_18 = t_17.conjugate();
    Fp12 t_19 = t_5 * t_12;
    Fp12 t_20 = t_19.forbenius_map(2);
    Fp12 t_21 = t_10 * t_3;
    Fp12 t_22 = t_21.forbenius_map(3);
    Fp12 t_23 = t_3.conjugate();
    Fp12 t_24 = t_16 * t_23;
    Fp12 t_25 = t_24.forbenius_map(1);
    Fp12 t_26 = t_8.conjugate();
    Fp12 t_27 = t_18 * t_26;
    Fp12 t_28 = t_27 * t_3;
    Fp12 t_29 = t_20 * t_22;
    Fp12 t_30 = t_29 * t_25;
    return t_30 * t_28;
}

// ------------------------------------------------------------------ pairing precompute / Miller loop
std::vector<Fp2> calc_precomp_stuff_loop0(const Fp2& rx, const Fp2& ry, const Fp2& rz) {  // native.rs:293-326
    Fp three = Fp::from_u32(3), two = Fp::from_u32(2), k = mod_inverse_of_two();
    Fp2 t0 = ry * ry, t1 = rz * rz, x0 = t1 * three;
    Fp2 t2 = x0.multiply_by_b(), t3 = t2 * three, x1 = ry * rz, t4 = x1 * two;
    Fp2 x2 = t2 - t0, x3 = rx * rx, x4 = x3 * three, x5 = -t4;
    Fp2 x6 = t0 - t3, x7 = rx * ry, x8 = x6 * x7, x9 = t0 + t3, x10 = x9 * k, x11 = x10 * x10, x12 = t2 * t2, x13 = x12 * three;
    Fp2 new_rx = x8 * k, new_ry = x11 - x13, new_rz = t0 * t4;
    return {new_rx, new_ry, new_rz, t0, t1, x0, t2, t3, x1, t4, x3, x2, x4, x5, x6, x7, x8, x9, x10, x11, x12, x13};
}
std::vector<Fp2> calc_precomp_stuff_loop1(const Fp2& rx, const Fp2& ry, const Fp2& rz, const Fp2& qx, const Fp2& qy) {  // native.rs:328-366
    Fp2 t0 = qy * rz, t1 = ry - t0, t2 = qx * rz, t3 = rx - t2, t4 = t1 * qx, t5 = t3 * qy, t6 = t4 - t5, t7 = -t1;
    Fp2 t8 = t3 * t3, t9 = t8 * t3, t10 = t8 * rx, t11 = t1 * t1, t12 = t11 * rz, t13 = t10 * Fp::from_u32(2);
    Fp2 t14 = t9 - t13, t15 = t14 + t12, t16 = t10 - t15, t17 = t16 * t1, t18 = t9 * ry;
    Fp2 new_rx = t3 * t15, new_ry = t17 - t18, new_rz = rz * t9;
    return {new_rx, new_ry, new_rz, t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11, t12, t13, t14, t15, t16, t17, t18};
}
std::vector<EllCoeff> calc_pairing_precomp(const Fp2& x, const Fp2& y, const Fp2& z) {  // native.rs:1358-1437
    Fp2 qx = x * z.invert(), qy = y * z.invert();
    Fp2 rx = qx, ry = qy, rz = Fp2::one();
    std::vector<EllCoeff> ell;
    for (int i = 62; i >= 0; i--) {
        std::vector<Fp2> v = calc_precomp_stuff_loop0(rx, ry, rz);
        ell.push_back({v[11], v[12], v[13]});  // x2, x4, x5
        rx = v[0]; ry = v[1]; rz = v[2];
        if ((BLS_X >> i) & 1) {
            std::vector<Fp2> w = calc_precomp_stuff_loop1(rx, ry, rz, qx, qy);
            ell.push_back({w[9], w[10], w[6]});  // t6, t7, t3
            rx = w[0]; ry = w[1]; rz = w[2];
        }
    }
    return ell;
}
Fp12 miller_loop(const Fp& px, const Fp& py, const Fp2& g2x, const Fp2& g2y, const Fp2& g2z) {  // native.rs:1440-1468
    std::vector<EllCoeff> pre = calc_pairing_precomp(g2x, g2y, g2z);
    Fp12 f = Fp12::one();
    size_t j = 0;
    for (int i = 62; i >= 0; i--) {
        f = f.multiply_by_014(pre[j][0], pre[j][1] * px, pre[j][2] * py);
        if ((BLS_X >> i) & 1) {
            j++;
            f = f.multiply_by_014(pre[j][0], pre[j][1] * px, pre[j][2] * py);
        }
        if (i != 0) f = f * f;
        j++;
    }
    return f.conjugate();
}

}  // namespace bls
}  // namespace starkhip
