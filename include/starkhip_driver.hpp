// C++ host drivers above the C ABI of starkhip.h: the STARK half of the reference's
// /root/reference/src/aggregate_proof.rs, function for function --
//   calc_pairing_precomp :23-72, miller_loop_main :74-118, fp12_mul_main :120-151, final_exponentiate_main :153-179,
//   ec_aggregate_main :181-221, and the six-proof sequence of generate_aggregate_proof :286-370
// each = per-AIR StarkConfig, generate_trace + public inputs, prove, verify_stark_proof; a failed step throws (the
// reference unwraps).  Header-only; link with libstarkhip.so.  Points are u32 limb arrays as `get_u32_slice()` yields
// them: Fp = 12 limbs, Fp2 = 24 (c0 then c1), Fp12 = 144.
#pragma once
#include <stdint.h>
#include <string.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "starkhip.h"

namespace starkhip_driver {

struct Error : std::runtime_error {
    int code;
    Error(const char* what, int c) : std::runtime_error(std::string(what) + ": " + starkhip_error_string(c)), code(c) {}
};
inline void check(const char* what, int rc) {
    if (rc != STARKHIP_OK) throw Error(what, rc);
}

// -G1 generator: the fixed G1 operand of the second pairing (src/aggregate_proof.rs:336-337), little-endian u32 limbs
static const uint32_t NEG_G1_X[12] = {0xdb22c6bb, 0xfb3af00a, 0xf97a1aef, 0x6c55e83f, 0x171bac58, 0xa14e3a3f,
                                      0x9774b905, 0xc3688c4f, 0x4fa9ac0f, 0x2695638c, 0x3197d794, 0x17f1d3a7};
static const uint32_t NEG_G1_Y[12] = {0xb939c2ca, 0xad54dcd6, 0x0ecb751b, 0x4e6f38ba, 0xcaac4236, 0x6655b9d5,
                                      0x1db507c9, 0x67816aef, 0xcf2e21f2, 0xaa7d76c8, 0x55d545a8, 0x114d1d68};

struct Proof {
    starkhip_air_t air;
    starkhip_config_t config;
    std::vector<uint64_t> words;  // blob of starkhip.h
    size_t n_public_inputs = 0;
    const uint64_t* public_inputs() const { return words.data() + words.size() - n_public_inputs; }
};

class Prover {
  public:
    explicit Prover(int device = 0) { check("starkhip_init", starkhip_init(device, &ctx_)); }
    ~Prover() { starkhip_shutdown(ctx_); }
    Prover(const Prover&) = delete;
    Prover& operator=(const Prover&) = delete;

    // prove + verify_stark_proof (src/aggregate_proof.rs:59-67) from a recorded (compact) trace: the generator's runs are
    // expanded on the device, so the 4.8 GB of FinalExp rows never exist on the host
    Proof prove(starkhip_air_t air, const void* trace_log, const std::vector<uint64_t>& pis) {
        Proof p;
        p.air = air;
        check("starkhip_config_for_air", starkhip_config_for_air(air, &p.config));
        uint64_t* blob = nullptr;
        size_t words = 0;
        check("starkhip_prove_compact",
              starkhip_prove_compact(ctx_, air, &p.config, trace_log, pis.data(), pis.size(), STARKHIP_POW_SEARCH, &blob, &words));
        p.words.assign(blob, blob + words);
        starkhip_free(blob);
        p.n_public_inputs = pis.size();
        check("starkhip_verify", starkhip_verify(air, &p.config, p.words.data(), p.words.size()));
        return p;
    }

  private:
    void* ctx_ = nullptr;
};

namespace detail {
// generate_trace + public inputs of one AIR, recorded: `gen(trace, n_rows, pis)` is the one starkhip_trace_* call
struct Witness {
    void* log = nullptr;
    std::vector<uint64_t> pis;
    template <class Gen>
    Witness(starkhip_air_t air, const char* what, Gen gen) : pis(starkhip_air_public_inputs(air)) {
        check("starkhip_trace_log_begin", starkhip_trace_log_begin(&log));
        const int rc = gen((uint64_t*)nullptr, (size_t)starkhip_air_default_rows(air), pis.data());
        const int rc_end = starkhip_trace_log_end(log);
        if (rc != STARKHIP_OK || rc_end != STARKHIP_OK) {
            starkhip_trace_log_free(log);
            log = nullptr;
            throw Error(what, rc != STARKHIP_OK ? rc : rc_end);
        }
    }
    ~Witness() { starkhip_trace_log_free(log); }
    Witness(const Witness&) = delete;
    Witness& operator=(const Witness&) = delete;
};
}  // namespace detail

inline Proof calc_pairing_precomp(Prover& pv, const uint32_t x[24], const uint32_t y[24], const uint32_t z[24]) {
    detail::Witness w(STARKHIP_AIR_PAIRING_PRECOMP, "starkhip_trace_pairing_precomp",
                      [&](uint64_t* t, size_t n, uint64_t* pis) { return starkhip_trace_pairing_precomp(x, y, z, t, n, pis); });
    return pv.prove(STARKHIP_AIR_PAIRING_PRECOMP, w.log, w.pis);
}
inline Proof miller_loop_main(Prover& pv, const uint32_t x[12], const uint32_t y[12], const uint32_t q_x[24], const uint32_t q_y[24],
                              const uint32_t q_z[24]) {
    detail::Witness w(STARKHIP_AIR_MILLER_LOOP, "starkhip_trace_miller_loop",
                      [&](uint64_t* t, size_t n, uint64_t* pis) { return starkhip_trace_miller_loop(x, y, q_x, q_y, q_z, t, n, pis); });
    return pv.prove(STARKHIP_AIR_MILLER_LOOP, w.log, w.pis);
}
inline Proof fp12_mul_main(Prover& pv, const uint32_t x[144], const uint32_t y[144]) {
    detail::Witness w(STARKHIP_AIR_FP12_MUL, "starkhip_trace_fp12_mul",
                      [&](uint64_t* t, size_t n, uint64_t* pis) { return starkhip_trace_fp12_mul(x, y, t, n, pis); });
    return pv.prove(STARKHIP_AIR_FP12_MUL, w.log, w.pis);
}
inline Proof final_exponentiate_main(Prover& pv, const uint32_t x[144]) {
    detail::Witness w(STARKHIP_AIR_FINAL_EXP, "starkhip_trace_final_exp",
                      [&](uint64_t* t, size_t n, uint64_t* pis) { return starkhip_trace_final_exp(x, t, n, pis); });
    return pv.prove(STARKHIP_AIR_FINAL_EXP, w.log, w.pis);
}
// points: 512 x [x(12), y(12)]; bits: 512 bytes.  The aggregate is the last 24 public inputs of the proof.
inline Proof ec_aggregate_main(Prover& pv, const uint32_t* points, const uint8_t* bits) {
    detail::Witness w(STARKHIP_AIR_ECC_AGGREGATE, "starkhip_trace_ecc_aggregate",
                      [&](uint64_t* t, size_t n, uint64_t* pis) { return starkhip_trace_ecc_aggregate(points, bits, t, n, pis); });
    return pv.prove(STARKHIP_AIR_ECC_AGGREGATE, w.log, w.pis);
}

struct SignatureProofs {
    Proof pp1, ml1, pp2, ml2, fp12_mul, final_exp;
    bool valid = false;  // final_exponentiate(ml1 * ml2) == 1
    bool linked = false; // the public-input equalities the reference's recursive aggregation enforces
};

// pk = (x, y) of the aggregate public key; hm, sig = (x, y, z) of H(m) and of the signature (Fp2 limbs each).
inline SignatureProofs prove_signature(Prover& pv, const uint32_t pk_x[12], const uint32_t pk_y[12], const uint32_t hm[3][24],
                                       const uint32_t sig[3][24]) {
    SignatureProofs s;
    uint32_t ml1[144], ml2[144], prod[144];
    check("native_miller_loop", starkhip_native_miller_loop(pk_x, pk_y, hm[0], hm[1], hm[2], ml1));
    check("native_miller_loop", starkhip_native_miller_loop(NEG_G1_X, NEG_G1_Y, sig[0], sig[1], sig[2], ml2));
    check("native_fp12_mul", starkhip_native_fp12_mul(ml1, ml2, prod));
    s.pp1 = calc_pairing_precomp(pv, hm[0], hm[1], hm[2]);
    s.ml1 = miller_loop_main(pv, pk_x, pk_y, hm[0], hm[1], hm[2]);
    s.pp2 = calc_pairing_precomp(pv, sig[0], sig[1], sig[2]);
    s.ml2 = miller_loop_main(pv, NEG_G1_X, NEG_G1_Y, sig[0], sig[1], sig[2]);
    s.fp12_mul = fp12_mul_main(pv, ml1, ml2);
    s.final_exp = final_exponentiate_main(pv, prod);
    const uint64_t* fe = s.final_exp.public_inputs();
    s.valid = fe[144] == 1;
    for (int i = 1; i < 144; i++) s.valid = s.valid && fe[144 + i] == 0;
    auto same = [](const uint64_t* a, const uint64_t* b, size_t n) { return memcmp(a, b, n * sizeof(uint64_t)) == 0; };
    const size_t ELL = 68 * 3 * 24;  // ell_coeffs limbs
    s.linked = same(s.pp1.public_inputs() + 72, s.ml1.public_inputs() + 24, ELL) && same(s.pp2.public_inputs() + 72, s.ml2.public_inputs() + 24, ELL) &&
               same(s.ml1.public_inputs() + 24 + ELL, s.fp12_mul.public_inputs(), 144) &&
               same(s.ml2.public_inputs() + 24 + ELL, s.fp12_mul.public_inputs() + 144, 144) &&
               same(s.fp12_mul.public_inputs() + 288, s.final_exp.public_inputs(), 144);
    return s;
}

}  // namespace starkhip_driver
