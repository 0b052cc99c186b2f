// BLS signature checks as STARK proofs through the C++ drivers of include/starkhip_driver.hpp -- compiled host code above the
// C ABI only, as a Rust caller of the reference's src/aggregate_proof.rs:304-370 would drive it.
//
//   signature_demo                      the six proofs of the reference's own vector (src/native.rs:1480-1498), one at a time on
//                                       one context (generate_trace + prove + verify each, as the reference does)
//   signature_demo --pool               the same six proofs in flight together on the library's proof pool
//   signature_demo --batch 8 [--operands tests/golden/signature_operands_8.bin] [--steps K] [--warmup W]
//                                       BASELINE configs[4]: B different signatures = 6 B proofs per step on the pool; prints
//                                       signatures/s with trace generation and natives INSIDE the timed region; afterwards every
//                                       proof of the last step is verified and checked against its statement
//   signature_demo --batch 8 --devices 0,1,2,3,4,5,6,7
//                                       the same from ONE process on several GPUs (starkhip_multipool_*: a pool per listed device --
//                                       an ordinal may repeat --, jobs placed longest first); --digests adds a digest of every proof
//                                       of the last step (the bytes do not depend on which device, pool or context made a proof)
// Exit code 0 = every proof verified, the public inputs chain, and final_exponentiate(ml1 * ml2) == 1 for every signature.
// Operand file: B records of 120 little-endian u32 limbs -- pk x, y (12 each), H(m) x, y (24 each), signature x, y (24 each);
// Z = (1, 0) is implied (tools/make_signature_operands.py derives them from the reference vector).  Build: make demo
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "starkhip_driver.hpp"

static const uint32_t PK_X[12] = {0xc07c5326, 0xe683161, 0x89f91998, 0xc3eb594f, 0x947c72f7, 0x434eb65c, 0x7d87bd95, 0x81cc34b6, 0x13491e37, 0x83be5cab, 0xe407bf46, 0x11065aa5};
static const uint32_t PK_Y[12] = {0x23aeff1d, 0xf23b9f28, 0x1943825c, 0xc78ee542, 0x474721f9, 0xc03b18e3, 0x48416215, 0x340e08c8, 0xf06ed4c4, 0x317638a6, 0x17818145, 0x16d944c6};
static const uint32_t HM[3][24] = {{0xab838fd2, 0x8612ddc1, 0x61ce5c0f, 0x350c5b24, 0x5799aad2, 0xb8848421, 0xe2098435, 0x56d9592e, 0x665b7143, 0xca5974b8, 0x8ada6391, 0xeb050f9, 0x1de2c0a7, 0x1531c6e6, 0xe1caed23, 0x8356420a, 0x7d33590c, 0xb5cdc285, 0xa6f29b2a, 0x4240281c, 0x33def041, 0x90ea1d03, 0x5b00fe7d, 0x113a929e},
                                   {0xa48d6ceb, 0x53eca773, 0xc59b1cac, 0xc42353a2, 0x5b8a553e, 0x18583abc, 0x909ef051, 0x1a633e3a, 0xb4e050ea, 0x8bdf1ffe, 0x3c8d7239, 0xa38bb4, 0x285b9c49, 0x52fe6b5c, 0x23927068, 0x698a5b2a, 0x3460269, 0xa64196b6, 0x4197ff9b, 0x806f1932, 0x9df92b6d, 0xa224c545, 0xcb8120, 0x783b595},
                                   {0x1, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0}};
static const uint32_t SIG[3][24] = {{0x7d9b6c52, 0xc8cab074, 0xf71a1b2d, 0xe67cd44f, 0xad8d126c, 0x1167ca07, 0xe9ce0399, 0x86d354ee, 0x3595b421, 0xcf0b7f57, 0xc77185b6, 0xbef22fc, 0xe4ce6846, 0xd30120d6, 0xe988d5a4, 0x1b3aad78, 0xe92076f, 0x457f1e4a, 0x6f93c2a6, 0xc4e0e722, 0x9d9f91d6, 0xf81e19f0, 0x8dc86269, 0xda59197},
                                    {0x47a7da5f, 0x951e4254, 0xbe6fbfc5, 0x48df24e8, 0x6d2b0652, 0xd8fd26d4, 0x42d22a2d, 0x33d38215, 0x3cedad7f, 0x7783a6f0, 0xb45a176f, 0x1099e699, 0x51e1e62d, 0xcc1047c0, 0xe57f924a, 0x47cb3632, 0xaadbc63, 0x7e1c66ab, 0xf075e068, 0x3083c45d, 0xe5d1379, 0xc3cb36f8, 0xcfb8185e, 0x4880abf},
                                    {0x1, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0, 0x0}};

using starkhip_driver::SignatureOperands;
using starkhip_driver::SignatureProofs;

static SignatureOperands reference_vector() {
    SignatureOperands s;
    memcpy(s.pk_x, PK_X, sizeof PK_X);
    memcpy(s.pk_y, PK_Y, sizeof PK_Y);
    memcpy(s.hm, HM, sizeof HM);
    memcpy(s.sig, SIG, sizeof SIG);
    return s;
}

static std::vector<SignatureOperands> load_operands(const char* path, size_t want) {
    std::vector<SignatureOperands> out;
    FILE* f = fopen(path, "rb");
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    uint32_t rec[120];
    while (out.size() < want && fread(rec, sizeof rec, 1, f) == 1) {
        SignatureOperands s;
        memset(&s, 0, sizeof s);
        memcpy(s.pk_x, rec, 48);
        memcpy(s.pk_y, rec + 12, 48);
        memcpy(s.hm[0], rec + 24, 96);
        memcpy(s.hm[1], rec + 48, 96);
        memcpy(s.sig[0], rec + 72, 96);
        memcpy(s.sig[1], rec + 96, 96);
        s.hm[2][0] = 1;
        s.sig[2][0] = 1;
        out.push_back(s);
    }
    fclose(f);
    if (out.size() < want) throw std::runtime_error(std::string(path) + ": fewer signatures than --batch asks for");
    return out;
}

static double seconds_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }

int main(int argc, char** argv) {
    size_t batch = 0, steps = 3, warmup = 1, pipeline = 1;
    bool use_pool = false, timeline = false, digests = false;
    std::vector<int> devices;
    const char* operands = nullptr;
    starkhip_pool_config_t cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.stream_priority = 1;
    cfg.warm_up = 1;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&]() -> const char* { return i + 1 < argc ? argv[++i] : "0"; };
        if (a == "--batch") batch = strtoul(val(), nullptr, 0), use_pool = true;
        else if (a == "--steps") steps = strtoul(val(), nullptr, 0);
        else if (a == "--pipeline") pipeline = std::max<size_t>(1, strtoul(val(), nullptr, 0));  // batches in flight: a steady stream of batches instead of one at a time
        else if (a == "--warmup") warmup = strtoul(val(), nullptr, 0);
        else if (a == "--operands") operands = val();
        else if (a == "--pool") use_pool = true;
        else if (a == "--timeline") timeline = true;  // per proof of the last step: submit, generation, proof (ms since the step began) on stderr
        else if (a == "--big") cfg.big_contexts = (unsigned)strtoul(val(), nullptr, 0);
        else if (a == "--small") cfg.small_contexts = (unsigned)strtoul(val(), nullptr, 0);
        else if (a == "--gen") cfg.generator_threads = (unsigned)strtoul(val(), nullptr, 0);
        else if (a == "--trace-threads") cfg.trace_threads = (unsigned)strtoul(val(), nullptr, 0);
        else if (a == "--policy") cfg.commit_policy = (unsigned)strtoul(val(), nullptr, 0);
        else if (a == "--priority") cfg.stream_priority = (unsigned)strtoul(val(), nullptr, 0);
        else if (a == "--warm") cfg.warm_up = (unsigned)strtoul(val(), nullptr, 0);
        else if (a == "--gather-ms") cfg.gather_ms = (float)atof(val());
        else if (a == "--device") cfg.device = atoi(val());
        else if (a == "--devices") {  // one pool per listed ordinal, in one process
            use_pool = true;
            for (const char* p = val(); *p;) {
                char* end = nullptr;
                devices.push_back((int)strtol(p, &end, 10));
                if (end == p) { fprintf(stderr, "signature_demo: --devices wants a comma-separated list of ordinals\n"); return 2; }
                p = *end == ',' ? end + 1 : end;
            }
        } else if (a == "--digests") digests = true;
        else {
            fprintf(stderr, "signature_demo: unknown option %s\n", a.c_str());
            return 2;
        }
    }
    try {
        if (!use_pool) {
            starkhip_driver::Prover prover(cfg.device);
            const auto t0 = std::chrono::steady_clock::now();
            const SignatureProofs s = starkhip_driver::prove_signature(prover, PK_X, PK_Y, HM, SIG);
            const double sec = seconds_since(t0);
            const size_t words = s.pp1.words.size() + s.ml1.words.size() + s.pp2.words.size() + s.ml2.words.size() + s.fp12_mul.words.size() +
                                 s.final_exp.words.size();
            printf("six proofs (trace generation + prove + verify, one at a time): %.2f s, %.1f MB, valid=%d linked=%d\n", sec, words * 8 / 1e6,
                   (int)s.valid, (int)s.linked);
            return s.valid && s.linked ? 0 : 1;
        }
        if (batch == 0) batch = 1;
        std::vector<SignatureOperands> sigs;
        if (operands) sigs = load_operands(operands, batch);
        else sigs.assign(batch, reference_vector());  // the reference's vector (B copies of it when no operand file is given)
        if (batch == 1) {  // latency: everything of one signature in flight -- one context per proof is all it can use
            if (!cfg.big_contexts) cfg.big_contexts = 1;
            if (!cfg.small_contexts) cfg.small_contexts = 5;
            if (!cfg.generator_threads) cfg.generator_threads = 6;
        }
        if (batch > 1) {  // measured on one MI355X: six FinalExp contexts (their commitments then go out in lane-form
                          // groups) and 12 small ones: 4.0 signatures/s; four and 16 (quad form throughout): 3.9
            if (!cfg.big_contexts) cfg.big_contexts = 6;
            if (!cfg.small_contexts) cfg.small_contexts = 12;
        }
        starkhip_driver::tune_host_allocator();
        std::unique_ptr<starkhip_driver::Pool> pool_owner(devices.empty() ? new starkhip_driver::Pool(cfg) : new starkhip_driver::Pool(devices, cfg));
        starkhip_driver::Pool& pool = *pool_owner;
        std::vector<SignatureProofs> proofs;
        double total = 0, best = 1e30;
        std::string step_ms;
        if (pipeline == 1) {
            for (size_t k = 0; k < warmup + steps; k++) {
                const auto t0 = std::chrono::steady_clock::now();
                proofs = starkhip_driver::prove_batch(pool, sigs, /*verify=*/false);
                const double sec = seconds_since(t0);
                if (k >= warmup) {
                    total += sec;
                    best = std::min(best, sec);
                    char buf[32];
                    snprintf(buf, sizeof buf, "%s%.1f", step_ms.empty() ? "" : ", ", sec * 1e3);
                    step_ms += buf;
                }
            }
        } else {
            // a steady stream: `pipeline` batches in flight; batch k + pipeline is submitted when batch k has been waited for.  The
            // time is that of `steps` consecutive batches in the middle of the stream (start-up and drain excluded by the warm-up
            // batches in front and the `pipeline - 1` batches still in flight behind).
            std::vector<std::vector<starkhip_driver::SignatureTickets>> flying;
            auto submit_batch = [&]() {
                std::vector<starkhip_driver::SignatureTickets> t;
                for (const SignatureOperands& s : sigs) t.push_back(starkhip_driver::submit_signature(pool, s));
                flying.push_back(std::move(t));
            };
            auto wait_batch = [&]() {
                std::vector<SignatureProofs> out;
                for (auto& t : flying.front()) out.push_back(starkhip_driver::wait_signature(pool, t, false));
                flying.erase(flying.begin());
                return out;
            };
            for (size_t k = 0; k < pipeline; k++) submit_batch();
            auto t_prev = std::chrono::steady_clock::now();
            for (size_t k = 0; k < warmup + steps; k++) {
                proofs = wait_batch();
                submit_batch();
                const double sec = seconds_since(t_prev);
                t_prev = std::chrono::steady_clock::now();
                if (k >= warmup) {
                    total += sec;
                    best = std::min(best, sec);
                    char buf[32];
                    snprintf(buf, sizeof buf, "%s%.1f", step_ms.empty() ? "" : ", ", sec * 1e3);
                    step_ms += buf;
                }
            }
            while (!flying.empty()) proofs = wait_batch();  // drain (untimed)
        }
        const double per_step = total / (steps ? steps : 1);
        // untimed: what the reference does after each prove (verify_stark_proof) and what its recursion enforces on the public inputs
        size_t verified = 0, ok = 0;
        const auto tv = std::chrono::steady_clock::now();
        std::vector<std::future<int>> checks;
        for (const SignatureProofs& s : proofs)
            for (const starkhip_driver::Proof* p : {&s.pp1, &s.ml1, &s.pp2, &s.ml2, &s.fp12_mul, &s.final_exp})
                checks.push_back(std::async(std::launch::async, [p] { return starkhip_verify(p->air, &p->config, p->words.data(), p->words.size()); }));
        for (auto& c : checks) verified += c.get() == STARKHIP_OK;
        for (size_t i = 0; i < proofs.size(); i++) ok += proofs[i].valid && proofs[i].linked && starkhip_driver::statement_holds(proofs[i], sigs[i]);
        if (timeline) {
            double t0 = 1e300;
            for (const SignatureProofs& s : proofs)
                for (const starkhip_driver::Proof* p : {&s.pp1, &s.ml1, &s.pp2, &s.ml2, &s.fp12_mul, &s.final_exp}) t0 = std::min(t0, p->info.t_submit);
            static const char* names[] = {"pp1", "ml1", "pp2", "ml2", "fp12_mul", "final_exp"};
            for (size_t i = 0; i < proofs.size(); i++) {
                const starkhip_driver::Proof* ps[] = {&proofs[i].pp1, &proofs[i].ml1, &proofs[i].pp2, &proofs[i].ml2, &proofs[i].fp12_mul, &proofs[i].final_exp};
                for (int k = 0; k < 6; k++) {
                    const starkhip_ticket_info_t& f = ps[k]->info;
                    fprintf(stderr, "%zu:%-9s submit %7.1f gen %7.1f..%7.1f prove %7.1f..%7.1f | up %5.1f lde %5.1f merkle %6.1f quot %5.1f qcom %5.1f open %5.1f fri_comb %5.1f "
                            "fri_com %5.1f pow %5.1f query %5.1f total %6.1f | host fs %5.1f\n", i, names[k],
                            (f.t_submit - t0) * 1e3, (f.t_generate_start - t0) * 1e3, (f.t_generate_end - t0) * 1e3, (f.t_prove_start - t0) * 1e3,
                            (f.t_done - t0) * 1e3, f.phase_ms[0], f.phase_ms[1], f.phase_ms[2], f.phase_ms[3], f.phase_ms[4], f.phase_ms[5], f.phase_ms[6], f.phase_ms[7],
                            f.phase_ms[8], f.phase_ms[9], f.phase_ms[10], f.host_ms[0]);
                }
            }
        }
        // FNV-1a over each proof's words, and which device slot proved it
        std::string digest_json, per_device_json;
        if (digests) {
            for (size_t i = 0; i < proofs.size(); i++) {
                const starkhip_driver::Proof* ps[] = {&proofs[i].pp1, &proofs[i].ml1, &proofs[i].pp2, &proofs[i].ml2, &proofs[i].fp12_mul, &proofs[i].final_exp};
                for (int k = 0; k < 6; k++) {
                    uint64_t h = 0xcbf29ce484222325ULL;
                    for (size_t w = 0; w < ps[k]->words.size(); w++) h = (h ^ ps[k]->words[w]) * 0x100000001b3ULL;
                    char buf[40];
                    snprintf(buf, sizeof buf, "%s\"%016llx\"", digest_json.empty() ? "" : ", ", (unsigned long long)h);
                    digest_json += buf;
                }
            }
        }
        for (size_t k = 0; k < pool.devices(); k++) {
            const starkhip_pool_stats_t one = pool.stats((int)k);
            char buf[96];
            snprintf(buf, sizeof buf, "%s{\"big\": %lu, \"small_requests\": %lu}", k ? ", " : "", one.big_commit_launches, one.small_commit_requests);
            per_device_json += buf;
        }
        const starkhip_pool_stats_t st = pool.stats();
        printf("{\"metric\": \"BLS signature checks/s, end to end from compiled host code (operands -> natives -> trace generation -> 6 STARK proofs each)\", "
               "\"value\": %.4f, \"unit\": \"signatures/s\", \"batch\": %zu, \"steps\": %zu, \"warmup\": %zu, \"ms_per_step\": %.1f, \"best_ms\": %.1f, \"step_ms\": [%s], "
               "\"batches_in_flight\": %zu, \"proofs_per_step\": %zu, \"proofs_verified_after_timing\": %zu, \"verify_s\": %.2f, \"signatures_valid_linked_bound\": %zu, "
               "\"commit_launches\": {\"big\": %lu, \"small_merged\": %lu, \"small_requests\": %lu, \"max_merged\": %lu}, \"pools\": %zu, \"commit_launches_per_pool\": [%s], \"hw_queues_late\": %d, \"proof_digests\": [%s], \"operands\": \"%s\"}\n",
               batch / per_step, batch, steps, warmup, per_step * 1e3, best * 1e3, step_ms.c_str(), pipeline, 6 * batch, verified, seconds_since(tv), ok, st.big_commit_launches,
               st.small_commit_launches, st.small_commit_requests, st.max_merged_commitments, pool.devices(), per_device_json.c_str(), starkhip_hw_queues_status(), digest_json.c_str(), operands ? operands : "reference vector (src/native.rs:1480-1498)");
        return verified == 6 * batch && ok == batch ? 0 : 1;
    } catch (const std::exception& e) {
        fprintf(stderr, "signature_demo: %s\n", e.what());
        return 2;
    }
}
