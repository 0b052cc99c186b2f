// Sanitizer harness for the proof pool's host logic (csrc/scheduler.cpp): generator threads, context workers, the commitment
// scheduler (gather window, merged launches, the two classes), job-to-context matching, failing jobs, shutdown with work queued.
// Built by `make tsan-test` / `make asan-test` WITHOUT a GPU against the stand-ins of csrc/host_only_stubs.cc
// (STARKHIP_FAKE_DEVICE=1: contexts exist, prove() sleeps and returns a blob that ends in the public inputs).  Test
// infrastructure only: it checks threading and memory, nothing about proofs.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <thread>
#include <vector>

#include "starkhip.h"

#define CHECK(cond)                                                      \
    do {                                                                 \
        if (!(cond)) {                                                   \
            fprintf(stderr, "tsan_pool: %s failed at line %d\n", #cond, __LINE__); \
            return 1;                                                    \
        }                                                                \
    } while (0)

static std::vector<uint32_t> limbs(size_t n, uint32_t seed) {
    std::vector<uint32_t> v(n);
    for (size_t i = 0; i < n; i++) v[i] = (i % 12 == 11) ? 0x09000000u : seed * 0x9E3779B1u + 13u * (uint32_t)i;
    return v;
}

// A pool per device behind one handle (starkhip_multipool_*): three pretended devices, one of them twice; jobs placed by the library
// (longest first) and by the caller, from several threads; every proof must come from the device of the slot its ticket names.
static int multi_device_pass() {
    const int devices[4] = {0, 2, 5, 2};
    starkhip_pool_config_t cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = 7;  // ignored
    cfg.big_contexts = 2;
    cfg.small_contexts = 3;
    cfg.generator_threads = 2;
    cfg.trace_threads = 1;
    cfg.warm_up = 1;
    cfg.gather_ms = 3.0f;
    void* mp = nullptr;
    const int bad_dev[2] = {0, 9};
    CHECK(starkhip_multipool_create(bad_dev, 2, &cfg, &mp) == STARKHIP_ERR_NO_DEVICE && mp == nullptr);  // the good pool is taken down again
    CHECK(starkhip_multipool_create(devices, 4, &cfg, &mp) == STARKHIP_OK);
    CHECK(starkhip_multipool_size(mp) == 4 && starkhip_multipool_device(mp, 3) == 2 && starkhip_multipool_pool(mp, 4) == nullptr);
    // one signature's worth of jobs as ONE batch: the FinalExp-class job is placed first and alone, the rest fill the other pools
    std::vector<starkhip_air_t> airs = {STARKHIP_AIR_PAIRING_PRECOMP, STARKHIP_AIR_MILLER_LOOP, STARKHIP_AIR_PAIRING_PRECOMP,
                                        STARKHIP_AIR_MILLER_LOOP,     STARKHIP_AIR_FP12_MUL,    STARKHIP_AIR_FINAL_EXP};
    std::vector<std::vector<uint32_t>> ops = {limbs(72, 1), limbs(96, 2), limbs(72, 3), limbs(96, 4), limbs(288, 5), limbs(144, 6)};
    std::vector<int> plan(airs.size());
    CHECK(starkhip_plan_lpt(airs.size(), airs.data(), 4, plan.data()) == STARKHIP_OK);
    CHECK(plan[5] == 0 && plan[1] == 1 && plan[3] == 2 && plan[4] == 3 && plan[0] == 3 && plan[2] == 1);  // 128 | 24 + 10.8 | 24 | 15.4 + 10.8
    std::vector<const uint32_t*> op_ptr;
    std::vector<size_t> op_len;
    for (auto& o : ops) { op_ptr.push_back(o.data()); op_len.push_back(o.size()); }
    std::vector<uint64_t> tickets(airs.size());
    std::vector<int> rcs(airs.size());
    CHECK(starkhip_multipool_submit_witness_batch(mp, airs.size(), airs.data(), op_ptr.data(), op_len.data(), STARKHIP_POW_SEARCH, tickets.data(), rcs.data()) == STARKHIP_OK);
    for (size_t i = 0; i < airs.size(); i++) CHECK(rcs[i] == STARKHIP_OK && starkhip_multipool_ticket_slot(mp, tickets[i]) == plan[i]);
    // more jobs from four threads at once, some placed by the caller; a bad batch entry is reported per job
    struct Sub { starkhip_air_t air; std::vector<uint32_t> ops; int slot; uint64_t ticket; int rc; };
    std::vector<Sub> subs;
    for (int k = 0; k < 12; k++) subs.push_back({k % 3 == 0 ? STARKHIP_AIR_MILLER_LOOP : STARKHIP_AIR_FP12_MUL, limbs(k % 3 == 0 ? 96 : 288, 40 + k), k % 4 == 3 ? k % 4 : -1, 0, 0});
    std::vector<std::thread> th;
    for (int w = 0; w < 4; w++)
        th.emplace_back([&, w] {
            for (size_t i = w; i < subs.size(); i += 4)
                subs[i].rc = starkhip_multipool_submit_witness(mp, subs[i].slot, subs[i].air, nullptr, subs[i].ops.data(), subs[i].ops.size(), STARKHIP_POW_SEARCH, &subs[i].ticket);
        });
    for (auto& t : th) t.join();
    uint64_t t_bad = 0;
    CHECK(starkhip_multipool_submit_witness(mp, 4, STARKHIP_AIR_FP12_MUL, nullptr, ops[4].data(), 288, STARKHIP_POW_SEARCH, &t_bad) == STARKHIP_ERR_BAD_SHAPE);  // no such slot
    // rows in "device memory" need a slot
    const size_t fe_cols = (size_t)starkhip_air_columns(STARKHIP_AIR_FINAL_EXP), fe_pis = (size_t)starkhip_air_public_inputs(STARKHIP_AIR_FINAL_EXP);
    std::vector<uint64_t> pis(fe_pis, 7), fake_rows(16);
    starkhip_config_t fe_cfg;
    CHECK(starkhip_config_for_air(STARKHIP_AIR_FINAL_EXP, &fe_cfg) == STARKHIP_OK);
    CHECK(starkhip_multipool_submit(mp, -1, STARKHIP_AIR_FINAL_EXP, &fe_cfg, fake_rows.data(), 8192, fe_cols, 1, 1, pis.data(), pis.size(), STARKHIP_POW_SEARCH, &t_bad) == STARKHIP_ERR_BAD_SHAPE);
    uint64_t fe_t[5];
    for (int k = 0; k < 5; k++)  // host rows, placed by the library: FinalExp-class jobs spread over the pools before any pool gets a second
        CHECK(starkhip_multipool_submit(mp, -1, STARKHIP_AIR_FINAL_EXP, &fe_cfg, fake_rows.data(), 8192, fe_cols, 1, 0, pis.data(), pis.size(), STARKHIP_POW_SEARCH, &fe_t[k]) == STARKHIP_OK);
    int failures = 0;
    auto collect = [&](starkhip_air_t air, uint64_t ticket) {
        uint64_t* proof = nullptr;
        size_t words = 0;
        starkhip_ticket_info_t info;
        const int slot = starkhip_multipool_ticket_slot(mp, ticket);
        if (slot < 0 || starkhip_multipool_wait(mp, ticket, &proof, &words, &info) != STARKHIP_OK) { failures++; return; }
        if (words != 4 + (size_t)starkhip_air_public_inputs(air) || proof[1] != (uint64_t)air || (int)(proof[3] >> 8) != starkhip_multipool_device(mp, (size_t)slot)) failures++;
        starkhip_free(proof);
    };
    std::thread w1([&] { for (size_t i = 0; i < airs.size(); i++) collect(airs[i], tickets[i]); });
    std::thread w2([&] { for (auto& s : subs) { if (s.rc != STARKHIP_OK || (s.slot >= 0 && starkhip_multipool_ticket_slot(mp, s.ticket) != s.slot)) failures++; else collect(s.air, s.ticket); } });
    for (int k = 0; k < 5; k++) collect(STARKHIP_AIR_FINAL_EXP, fe_t[k]);
    w1.join();
    w2.join();
    CHECK(failures == 0);
    uint64_t* none = nullptr;
    size_t nw = 0;
    CHECK(starkhip_multipool_wait(mp, tickets[0], &none, &nw, nullptr) == STARKHIP_ERR_BAD_SHAPE);       // waited for once
    CHECK(starkhip_multipool_wait(mp, 12345, &none, &nw, nullptr) == STARKHIP_ERR_BAD_SHAPE);            // not a ticket of this handle
    // every pool proved something, and the FinalExp-class jobs went 2 + 2 + 1 + 1 (one from the batch, five placed one by one)
    unsigned long big[4], total_big = 0;
    for (size_t s = 0; s < 4; s++) {
        starkhip_pool_stats_t st;
        CHECK(starkhip_pool_stats(starkhip_multipool_pool(mp, s), &st) == STARKHIP_OK);
        big[s] = st.big_commit_launches;
        total_big += big[s];
        CHECK(st.small_commit_requests > 0 || big[s] > 0);
    }
    CHECK(total_big == 6);
    for (size_t s = 0; s < 4; s++) CHECK(big[s] >= 1 && big[s] <= 2);
    // shutdown with work queued on several pools
    for (int k = 0; k < 8; k++) {
        uint64_t t = 0;
        CHECK(starkhip_multipool_submit_witness(mp, -1, STARKHIP_AIR_MILLER_LOOP, nullptr, ops[1].data(), 96, STARKHIP_POW_SEARCH, &t) == STARKHIP_OK);
    }
    starkhip_multipool_destroy(mp);
    uint64_t bs[5];
    starkhip_proof_blob_stats(bs);
    CHECK(bs[0] == 0 && bs[2] == 0);
    printf("multi-device: ok (FinalExp-class proofs per pool %lu %lu %lu %lu)\n", big[0], big[1], big[2], big[3]);
    return 0;
}

int main() {
    setenv("STARKHIP_FAKE_DEVICE", "1", 1);
    if (int rc = multi_device_pass()) return rc;
    for (unsigned pass = 0; pass < 4; pass++) {
        const unsigned policy = pass < 3 ? pass : 0;
        // the fourth pass: big commitments gathered into lane-form groups (what pools with five or more big contexts do)
        if (pass == 3) setenv("STARKHIP_POOL_BIG_LANE", "1", 1);
        starkhip_pool_config_t cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.big_contexts = 2;
        cfg.small_contexts = 5;
        cfg.generator_threads = 3;
        cfg.trace_threads = 2;
        cfg.commit_policy = policy;
        cfg.stream_priority = 1;
        cfg.warm_up = 1;
        cfg.gather_ms = 5.0f;
        void* pool = nullptr;
        CHECK(starkhip_pool_create(&cfg, &pool) == STARKHIP_OK);
        // witness jobs of three small AIRs and the toy one, submitted from several threads at once
        struct Sub { starkhip_air_t air; std::vector<uint32_t> ops; uint64_t pow; uint64_t ticket; int rc_submit; };
        std::vector<Sub> subs;
        for (int k = 0; k < 4; k++) subs.push_back({STARKHIP_AIR_FP12_MUL, limbs(288, 1 + k), STARKHIP_POW_SEARCH, 0, 0});
        for (int k = 0; k < 3; k++) subs.push_back({STARKHIP_AIR_PAIRING_PRECOMP, limbs(72, 11 + k), STARKHIP_POW_SEARCH, 0, 0});
        for (int k = 0; k < 2; k++) subs.push_back({STARKHIP_AIR_MILLER_LOOP, limbs(96, 21 + k), STARKHIP_POW_SEARCH, 0, 0});
        subs.push_back({STARKHIP_AIR_TEST_FIBONACCI, {3, 0, 5, 0}, STARKHIP_POW_SEARCH, 0, 0});
        subs.push_back({STARKHIP_AIR_FP12_MUL, limbs(288, 31), 0xBAD, 0, 0});    // fails before its commitment
        subs.push_back({STARKHIP_AIR_FP12_MUL, limbs(288, 32), 0xBAD2, 0, 0});   // fails after it
        std::vector<std::thread> th;
        for (size_t i = 0; i < subs.size(); i++)
            th.emplace_back([&, i] {
                subs[i].rc_submit = starkhip_pool_submit_witness(pool, subs[i].air, nullptr, subs[i].ops.data(), subs[i].ops.size(), subs[i].pow, &subs[i].ticket);
            });
        for (auto& t : th) t.join();
        // a FinalExp-class job from rows the caller owns (fake: only the shape is looked at) -- two of them: the second is the "last wave"
        const size_t fe_cols = (size_t)starkhip_air_columns(STARKHIP_AIR_FINAL_EXP), fe_pis = (size_t)starkhip_air_public_inputs(STARKHIP_AIR_FINAL_EXP);
        std::vector<uint64_t> pis(fe_pis, 7), fake_rows(16);
        starkhip_config_t fe_cfg;
        CHECK(starkhip_config_for_air(STARKHIP_AIR_FINAL_EXP, &fe_cfg) == STARKHIP_OK);
        uint64_t fe_t[3];
        for (int k = 0; k < 3; k++)
            CHECK(starkhip_pool_submit(pool, STARKHIP_AIR_FINAL_EXP, &fe_cfg, fake_rows.data(), 8192, fe_cols, 1, 1, pis.data(), pis.size(), STARKHIP_POW_SEARCH, &fe_t[k]) == STARKHIP_OK);
        // wrong shapes are refused at submit
        uint64_t t_bad = 0;
        CHECK(starkhip_pool_submit(pool, STARKHIP_AIR_FINAL_EXP, &fe_cfg, fake_rows.data(), 8192, fe_cols - 1, 1, 1, pis.data(), pis.size(), STARKHIP_POW_SEARCH, &t_bad) == STARKHIP_ERR_BAD_SHAPE);
        CHECK(starkhip_pool_submit_witness(pool, STARKHIP_AIR_FP12_MUL, nullptr, subs[0].ops.data(), 17, STARKHIP_POW_SEARCH, &t_bad) == STARKHIP_ERR_BAD_SHAPE);
        // waits, from two threads
        int failures = 0;
        auto wait_range = [&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++) {
                uint64_t* proof = nullptr;
                size_t words = 0;
                starkhip_ticket_info_t info;
                const int rc = subs[i].rc_submit == STARKHIP_OK ? starkhip_pool_wait(pool, subs[i].ticket, &proof, &words, &info) : subs[i].rc_submit;
                const int want = subs[i].pow == 0xBAD ? STARKHIP_ERR_BAD_SHAPE : subs[i].pow == 0xBAD2 ? STARKHIP_ERR_QUOTIENT_NOT_DIVISIBLE : STARKHIP_OK;
                if (rc != want) failures++;
                if (rc == STARKHIP_OK) {
                    const size_t n_pis = (size_t)starkhip_air_public_inputs(subs[i].air);
                    if (words != 4 + n_pis || proof[1] != (uint64_t)subs[i].air || info.t_done < info.t_prove_start) failures++;
                    starkhip_free(proof);
                }
            }
        };
        std::thread w1([&] { wait_range(0, subs.size() / 2); }), w2([&] { wait_range(subs.size() / 2, subs.size()); });
        w1.join();
        w2.join();
        CHECK(failures == 0);
        uint64_t* kept = nullptr;  // one proof stays with the caller beyond the pool's life
        for (int k = 0; k < 3; k++) {
            uint64_t* proof = nullptr;
            size_t words = 0;
            CHECK(starkhip_pool_wait(pool, fe_t[k], &proof, &words, nullptr) == STARKHIP_OK && words == 4 + fe_pis);
            if (k == 0) kept = proof;
            else starkhip_free(proof);
        }
        uint64_t* none = nullptr;
        size_t nw = 0;
        CHECK(starkhip_pool_wait(pool, fe_t[0], &none, &nw, nullptr) == STARKHIP_ERR_BAD_SHAPE);  // a ticket is waited for once
        starkhip_pool_stats_t st;
        CHECK(starkhip_pool_stats(pool, &st) == STARKHIP_OK);
        if (policy != 2) CHECK(st.big_commit_launches == 3 && st.small_commit_requests == 11);  // 10 good small jobs + the one that fails after its commitment
        // shutdown with work still queued: never waited for, still run to the end and freed
        for (int k = 0; k < 6; k++) {
            uint64_t t = 0;
            CHECK(starkhip_pool_submit_witness(pool, STARKHIP_AIR_FP12_MUL, nullptr, subs[0].ops.data(), 288, STARKHIP_POW_SEARCH, &t) == STARKHIP_OK);
        }
        // ... and witness jobs that are still queued for (or in) their RECORDING when the pool is destroyed: the contexts must not
        // leave before the generators have handed them over, and a caller blocked in wait on one of them must get its proof
        uint64_t t_late[3];
        for (int k = 0; k < 3; k++)
            CHECK(starkhip_pool_submit_witness(pool, k == 1 ? STARKHIP_AIR_PAIRING_PRECOMP : STARKHIP_AIR_MILLER_LOOP, nullptr,
                                               subs[k == 1 ? 4 : 7].ops.data(), k == 1 ? 72 : 96, STARKHIP_POW_SEARCH, &t_late[k]) == STARKHIP_OK);
        int late_rc = -100;
        size_t late_words = 0;
        std::thread late([&] {
            uint64_t* proof = nullptr;
            late_rc = starkhip_pool_wait(pool, t_late[2], &proof, &late_words, nullptr);
            if (late_rc == STARKHIP_OK) starkhip_free(proof);
        });
        std::this_thread::sleep_for(std::chrono::milliseconds(20));  // the waiter is inside pool_wait (its job's recording takes longer)
        starkhip_pool_destroy(pool);
        late.join();
        CHECK(late_rc == STARKHIP_OK && late_words == 4 + (size_t)starkhip_air_public_inputs(STARKHIP_AIR_MILLER_LOOP));
        uint64_t bs[5];
        starkhip_proof_blob_stats(bs);
        CHECK(bs[3] > 0);                    // warmed contexts served proofs from the arena
        CHECK(bs[0] == bs[1] && bs[0] <= 1);  // every idle blob went with its context; at most the kept proof's remains
        CHECK(kept[0] == 0xFA4EULL);          // and is still readable
        starkhip_free(kept);
        starkhip_proof_blob_stats(bs);
        CHECK(bs[0] == 0 && bs[2] == 0);
        printf("policy %u: ok (%lu small requests in %lu launches, up to %lu merged)\n", policy, st.small_commit_requests, st.small_commit_launches,
               st.max_merged_commitments);
    }
    return 0;
}
