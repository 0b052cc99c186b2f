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

#include <malloc.h>

#include <array>
#include <chrono>
#include <future>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "starkhip.h"

namespace starkhip_driver {

// For a DRIVER process that proves all the time: keep glibc from handing every large buffer (150 MB trace recordings, 50 MB
// proofs) back to the kernel, so that the next one does not start with page faults on fresh memory -- in a process with dozens
// of threads those serialise on the address-space lock.  Process-wide, so a driver's choice, not the library's.
inline void tune_host_allocator() {
#ifdef __GLIBC__
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 1 << 28);
#endif
}

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

// A proof blob as the library hands it over (starkhip.h): owned, released with starkhip_free -- never copied (a FinalExp proof
// is 52 MB; copying 48 of them into fresh host memory costs more than proving one).
class Words {
  public:
    Words() = default;
    Words(uint64_t* p, size_t n) : p_(p), n_(n) {}
    Words(Words&& o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    Words& operator=(Words&& o) noexcept {
        if (this != &o) {
            starkhip_free(p_);
            p_ = o.p_; n_ = o.n_;
            o.p_ = nullptr; o.n_ = 0;
        }
        return *this;
    }
    Words(const Words&) = delete;
    Words& operator=(const Words&) = delete;
    ~Words() { starkhip_free(p_); }
    const uint64_t* data() const { return p_; }
    size_t size() const { return n_; }
    uint64_t operator[](size_t i) const { return p_[i]; }

  private:
    uint64_t* p_ = nullptr;
    size_t n_ = 0;
};

struct Proof {
    starkhip_air_t air;
    starkhip_config_t config;
    Words words;  // blob of starkhip.h
    size_t n_public_inputs = 0;
    starkhip_ticket_info_t info{};  // pool proofs: phase times and the job's timeline (starkhip_pool_wait)
    const uint64_t* public_inputs() const { return words.data() + words.size() - n_public_inputs; }
    Proof() = default;
    Proof(Proof&&) = default;
    Proof& operator=(Proof&&) = default;
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
        p.words = Words(blob, words);
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
    bool valid = false;  // final_exponentiate(ml1 * ml2) == 1, read from what was PROVEN (the final_exp proof's output public inputs)
    bool linked = false; // the public-input equalities the reference's recursive aggregation enforces
    // the host natives' value of final_exponentiate(ml1 * ml2), where the driver computed it (pool drivers): the proven output
    // must equal it -- the same cross-check aggregate.signature_is_valid makes in the Python harness
    std::vector<uint32_t> native_final;
    bool native_agrees = true;
};

inline void finish_links(SignatureProofs& s);  // valid / linked from the six proofs' public inputs (below)

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
    finish_links(s);
    return s;
}

// ---- the same on the library's proof pool (starkhip_pool_*): every driver becomes submit + wait, so the six proofs of one
// signature -- or of a whole batch -- are in flight together; generate_trace runs on the pool's generator threads and the
// pool's scheduler merges the small AIRs' trace commitments (starkhip.h).
// Over SEVERAL devices (starkhip_multipool_*) it is the same object: the reference's caller is one process
// (src/aggregate_proof.rs:304-370, :402-414), so `Pool(devices, cfg)` gives that process a pool per device behind the same
// submit / wait -- jobs go to the pool with the least outstanding work, FinalExp-class jobs to the pool with the fewest of them
// (longest processing time first).  No process group and no collective: the proofs are independent.
class Pool {
  public:
    explicit Pool(const starkhip_pool_config_t& cfg) { check("starkhip_pool_create", starkhip_pool_create(&cfg, &pool_)); }
    explicit Pool(int device = 0) {
        starkhip_pool_config_t cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.device = device;
        check("starkhip_pool_create", starkhip_pool_create(&cfg, &pool_));
    }
    // one pool per entry of `devices` (an ordinal may repeat), each configured by `cfg`
    Pool(const std::vector<int>& devices, const starkhip_pool_config_t& cfg) : multi_(true) {
        check("starkhip_multipool_create", starkhip_multipool_create(devices.data(), devices.size(), &cfg, &pool_));
    }
    ~Pool() {
        if (multi_) starkhip_multipool_destroy(pool_);
        else starkhip_pool_destroy(pool_);
    }
    Pool(const Pool&) = delete;
    Pool& operator=(const Pool&) = delete;
    size_t devices() const { return multi_ ? starkhip_multipool_size(pool_) : 1; }
    // which pool a ticket's job went to (0 on a single pool)
    int slot_of(uint64_t ticket) const { return multi_ ? starkhip_multipool_ticket_slot(pool_, ticket) : 0; }

    // generate_trace + prove of one driver (src/aggregate_proof.rs:23-179) from its operands; returns the ticket
    uint64_t submit(starkhip_air_t air, const std::vector<uint32_t>& operands) {
        uint64_t t = 0;
        if (multi_)
            check("starkhip_multipool_submit_witness",
                  starkhip_multipool_submit_witness(pool_, -1, air, nullptr, operands.data(), operands.size(), STARKHIP_POW_SEARCH, &t));
        else
            check("starkhip_pool_submit_witness", starkhip_pool_submit_witness(pool_, air, nullptr, operands.data(), operands.size(), STARKHIP_POW_SEARCH, &t));
        return t;
    }
    // a whole batch of drivers at once, placed longest first over the devices (on one pool: in the given order)
    std::vector<uint64_t> submit_batch(const std::vector<starkhip_air_t>& airs, const std::vector<std::vector<uint32_t>>& operands) {
        std::vector<uint64_t> t(airs.size(), 0);
        if (!multi_) {
            for (size_t i = 0; i < airs.size(); i++) t[i] = submit(airs[i], operands[i]);
            return t;
        }
        std::vector<const uint32_t*> ptr;
        std::vector<size_t> len;
        for (const auto& o : operands) { ptr.push_back(o.data()); len.push_back(o.size()); }
        const int rc = starkhip_multipool_submit_witness_batch(pool_, airs.size(), airs.data(), ptr.data(), len.data(), STARKHIP_POW_SEARCH, t.data(), nullptr);
        if (rc != STARKHIP_OK) {  // not all-or-nothing: the jobs that were accepted run -- wait for them and drop their proofs before the error goes up
            for (uint64_t id : t) {
                if (!id) continue;
                uint64_t* blob = nullptr;
                size_t words = 0;
                starkhip_ticket_info_t info;
                if (starkhip_multipool_wait(pool_, id, &blob, &words, &info) == STARKHIP_OK) starkhip_free(blob);
            }
        }
        check("starkhip_multipool_submit_witness_batch", rc);
        return t;
    }
    // the finished proof of `ticket`; verify = the reference's verify_stark_proof(..).unwrap() right after prove
    Proof wait(starkhip_air_t air, uint64_t ticket, bool verify = true, starkhip_ticket_info_t* info = nullptr) {
        Proof p;
        p.air = air;
        check("starkhip_config_for_air", starkhip_config_for_air(air, &p.config));
        uint64_t* blob = nullptr;
        size_t words = 0;
        if (multi_) check("starkhip_multipool_wait", starkhip_multipool_wait(pool_, ticket, &blob, &words, &p.info));
        else check("starkhip_pool_wait", starkhip_pool_wait(pool_, ticket, &blob, &words, &p.info));
        if (info) *info = p.info;
        p.words = Words(blob, words);
        p.n_public_inputs = (size_t)starkhip_air_public_inputs(air);
        if (verify) check("starkhip_verify", starkhip_verify(air, &p.config, p.words.data(), p.words.size()));
        return p;
    }
    // launches of the commitment scheduler (summed over the devices' pools); slot >= 0: that device's pool alone
    starkhip_pool_stats_t stats(int slot = -1) {
        starkhip_pool_stats_t s;
        if (!multi_) {
            check("starkhip_pool_stats", starkhip_pool_stats(pool_, &s));
            return s;
        }
        memset(&s, 0, sizeof s);
        for (size_t k = 0; k < devices(); k++) {
            if (slot >= 0 && (size_t)slot != k) continue;
            starkhip_pool_stats_t one;
            check("starkhip_pool_stats", starkhip_pool_stats(starkhip_multipool_pool(pool_, k), &one));
            s.big_commit_launches += one.big_commit_launches;
            s.small_commit_launches += one.small_commit_launches;
            s.small_commit_requests += one.small_commit_requests;
            if (one.max_merged_commitments > s.max_merged_commitments) s.max_merged_commitments = one.max_merged_commitments;
        }
        return s;
    }

  private:
    void* pool_ = nullptr;
    bool multi_ = false;
};

namespace detail {
inline std::vector<uint32_t> pack(std::initializer_list<std::pair<const uint32_t*, size_t>> parts) {
    std::vector<uint32_t> v;
    for (const auto& p : parts) v.insert(v.end(), p.first, p.first + p.second);
    return v;
}
}  // namespace detail

// One signature's operands: aggregate public key (x, y), H(m) and signature as (x, y, z) Fp2 limbs.
struct SignatureOperands {
    uint32_t pk_x[12], pk_y[12], hm[3][24], sig[3][24];
};

// tickets of the six jobs of one signature, submitted in the reference's order; the two that need the native Miller-loop
// values (fp12_mul, final_exp: src/aggregate_proof.rs:352-363) are submitted by a helper thread as soon as those are known
struct SignatureTickets {
    uint64_t pp1 = 0, ml1 = 0, pp2 = 0, ml2 = 0;
    struct Tail {
        std::array<uint64_t, 2> tickets;      // fp12_mul, final_exp
        std::vector<uint32_t> native_final;   // native final_exponentiate(ml1 * ml2), computed after both jobs were submitted
    };
    std::future<Tail> tail;
};

inline SignatureTickets submit_signature(Pool& pool, const SignatureOperands& s) {
    SignatureTickets t;
    t.pp1 = pool.submit(STARKHIP_AIR_PAIRING_PRECOMP, detail::pack({{s.hm[0], 24}, {s.hm[1], 24}, {s.hm[2], 24}}));
    t.ml1 = pool.submit(STARKHIP_AIR_MILLER_LOOP, detail::pack({{s.pk_x, 12}, {s.pk_y, 12}, {s.hm[0], 24}, {s.hm[1], 24}, {s.hm[2], 24}}));
    t.pp2 = pool.submit(STARKHIP_AIR_PAIRING_PRECOMP, detail::pack({{s.sig[0], 24}, {s.sig[1], 24}, {s.sig[2], 24}}));
    t.ml2 = pool.submit(STARKHIP_AIR_MILLER_LOOP, detail::pack({{NEG_G1_X, 12}, {NEG_G1_Y, 12}, {s.sig[0], 24}, {s.sig[1], 24}, {s.sig[2], 24}}));
    t.tail = std::async(std::launch::async, [&pool, s]() {
        // the two native Miller loops are independent: one on a thread of its own
        std::vector<uint32_t> ml1(144), ml2(144), prod(144);
        auto other = std::async(std::launch::async, [&] { return starkhip_native_miller_loop(NEG_G1_X, NEG_G1_Y, s.sig[0], s.sig[1], s.sig[2], ml2.data()); });
        check("native_miller_loop", starkhip_native_miller_loop(s.pk_x, s.pk_y, s.hm[0], s.hm[1], s.hm[2], ml1.data()));
        check("native_miller_loop", other.get());
        check("native_fp12_mul", starkhip_native_fp12_mul(ml1.data(), ml2.data(), prod.data()));
        SignatureTickets::Tail out;
        out.tickets[1] = pool.submit(STARKHIP_AIR_FINAL_EXP, prod);  // the long pole first
        out.tickets[0] = pool.submit(STARKHIP_AIR_FP12_MUL, detail::pack({{ml1.data(), 144}, {ml2.data(), 144}}));
        out.native_final.resize(144);  // off the critical path: both jobs are already in the pool
        check("native_final_exponentiate", starkhip_native_final_exponentiate(prod.data(), out.native_final.data()));
        return out;
    });
    return t;
}

inline void finish_links(SignatureProofs& s) {
    const uint64_t* fe = s.final_exp.public_inputs();
    s.valid = fe[144] == 1;
    for (int i = 1; i < 144; i++) s.valid = s.valid && fe[144 + i] == 0;
    auto same = [](const uint64_t* a, const uint64_t* b, size_t n) { return memcmp(a, b, n * sizeof(uint64_t)) == 0; };
    const size_t ELL = 68 * 3 * 24;  // ell_coeffs limbs
    s.linked = same(s.pp1.public_inputs() + 72, s.ml1.public_inputs() + 24, ELL) && same(s.pp2.public_inputs() + 72, s.ml2.public_inputs() + 24, ELL) &&
               same(s.ml1.public_inputs() + 24 + ELL, s.fp12_mul.public_inputs(), 144) &&
               same(s.ml2.public_inputs() + 24 + ELL, s.fp12_mul.public_inputs() + 144, 144) &&
               same(s.fp12_mul.public_inputs() + 288, s.final_exp.public_inputs(), 144);
}

// the statement: the proofs are about THESE operands -- H(m) and the signature with Z = (1, 0), the key as ml1's G1 operand
// (the reference binds it through the ECCAggStark proof, src/aggregate_proof.rs:540-545), -G as ml2's
inline bool statement_holds(const SignatureProofs& p, const SignatureOperands& s) {
    auto is = [](const uint64_t* pis, const uint32_t* limbs, size_t n) {
        for (size_t i = 0; i < n; i++)
            if (pis[i] != limbs[i]) return false;
        return true;
    };
    static const uint32_t ONE[24] = {1};
    return is(p.pp1.public_inputs(), s.hm[0], 24) && is(p.pp1.public_inputs() + 24, s.hm[1], 24) && is(p.pp1.public_inputs() + 48, ONE, 24) &&
           is(p.pp2.public_inputs(), s.sig[0], 24) && is(p.pp2.public_inputs() + 24, s.sig[1], 24) && is(p.pp2.public_inputs() + 48, ONE, 24) &&
           is(p.ml1.public_inputs(), s.pk_x, 12) && is(p.ml1.public_inputs() + 12, s.pk_y, 12) && is(p.ml2.public_inputs(), NEG_G1_X, 12) &&
           is(p.ml2.public_inputs() + 12, NEG_G1_Y, 12);
}

inline SignatureProofs wait_signature(Pool& pool, SignatureTickets& t, bool verify = true) {
    SignatureProofs s;
    SignatureTickets::Tail tl = t.tail.get();
    const std::array<uint64_t, 2> tail = tl.tickets;
    s.pp1 = pool.wait(STARKHIP_AIR_PAIRING_PRECOMP, t.pp1, verify);
    s.ml1 = pool.wait(STARKHIP_AIR_MILLER_LOOP, t.ml1, verify);
    s.pp2 = pool.wait(STARKHIP_AIR_PAIRING_PRECOMP, t.pp2, verify);
    s.ml2 = pool.wait(STARKHIP_AIR_MILLER_LOOP, t.ml2, verify);
    s.fp12_mul = pool.wait(STARKHIP_AIR_FP12_MUL, tail[0], verify);
    s.final_exp = pool.wait(STARKHIP_AIR_FINAL_EXP, tail[1], verify);
    finish_links(s);
    s.native_final = std::move(tl.native_final);
    for (int i = 0; i < 144; i++) s.native_agrees = s.native_agrees && s.final_exp.public_inputs()[144 + i] == s.native_final[i];
    s.valid = s.valid && s.native_agrees;
    return s;
}

// the six proofs of one signature, all in flight at once
inline SignatureProofs prove_signature(Pool& pool, const SignatureOperands& s, bool verify = true) {
    SignatureTickets t = submit_signature(pool, s);
    return wait_signature(pool, t, verify);
}

// BASELINE configs[4]: a batch of signatures = 6 B proofs in flight on one pool
inline std::vector<SignatureProofs> prove_batch(Pool& pool, const std::vector<SignatureOperands>& sigs, bool verify = true) {
    std::vector<SignatureTickets> tickets;
    tickets.reserve(sigs.size());
    for (const SignatureOperands& s : sigs) tickets.push_back(submit_signature(pool, s));
    std::vector<SignatureProofs> out;
    out.reserve(sigs.size());
    for (SignatureTickets& t : tickets) out.push_back(wait_signature(pool, t, verify));
    return out;
}

}  // namespace starkhip_driver
