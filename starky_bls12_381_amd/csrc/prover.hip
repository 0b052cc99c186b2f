// Host driver of the GPU prover: the MI355X-native replacement of starky::prover::prove as called at
// /root/reference/src/aggregate_proof.rs:59-65 (and :105, :138, :169).  Follows the transcript of
// SURVEY.md App. A.5 step by step; every heavy step is one of the kernels in kernels_*.hip, the host
// only runs the Fiat-Shamir challenger, two length-n synthetic divisions and the proof assembly.
#include <hip/hip_runtime.h>

#include <atomic>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <memory>
#include <set>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "airs.h"
#include "blob_arena.h"
#include "kernels.h"
#include "lde_ranges.h"
#include "trace_log.h"
#include "poseidon.h"
#include "proof.h"
#include "quotient_ops.h"
#include "quotient_plan.h"
#include "prover.h"
#include "scheduler.h"

#ifdef STARKHIP_ROCTX  // make ROCTX=1: phase ranges for rocprofv3 --marker-trace; the default build has no profiler-SDK dependency
#include <rocprofiler-sdk-roctx/roctx.h>
#endif

namespace starkhip {

// One open rocTX range at a time on the calling thread; closed on every way out of prove().  Without STARKHIP_ROCTX: nothing.
struct PhaseRanges {
#ifdef STARKHIP_ROCTX
    bool open = false;
    void next(const char* name) {
        if (open) roctxRangePop();
        roctxRangePushA(name);
        open = true;
    }
    ~PhaseRanges() {
        if (open) roctxRangePop();
    }
#else
    void next(const char*) {}
#endif
};


#define HIPCHK(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess) {                                                                            \
            fprintf(stderr, "starkhip: HIP error %s at %s:%d (%s)\n", hipGetErrorString(_e), __FILE__, __LINE__, #expr); \
            (void)hipDeviceSynchronize(); /* pending async copies target host buffers that are about to go out of scope */ \
            return _e == hipErrorOutOfMemory ? STARKHIP_ERR_OOM : STARKHIP_ERR_HIP;                        \
        }                                                                                                  \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() const { return (T*)p; }
};

struct Ctx {
    int device = 0;
    hipStream_t st = nullptr;       // the stream of the current proof: st_normal, or st_high for a proof the pool marks urgent
    hipStream_t st_normal = nullptr, st_high = nullptr;
    hipEvent_t ev[STARKHIP_N_PHASES + 1];
    float timings[STARKHIP_N_PHASES] = {0};
    hipEvent_t kev[6];            // the three heavy kernels bracketed on their own: leaf hash, quotient evaluation, trace LDE
    float ktimings[3] = {0};      // lde_columns, leaf_hash (trace), quotient_eval
    float htimings[2] = {0};      // host time inside the last prove: Fiat-Shamir hashing (the challenger's sequential sponge), other host arithmetic
    HashService* hs = nullptr;    // a pooled context's trace commitments are launched by the pool's scheduler (scheduler.h)
    hipEvent_t hash_ready = nullptr, hash_done = nullptr;
    HashService::Timing hash_timing;  // pooled: the commitment kernel's own start / stop on ITS launch stream, its form and group
    hipEvent_t wait_ev = nullptr;  // hipEventBlockingSync: see stream_wait()
    // Read-backs (caps, openings, FRI batches, the nonce) land in a page-locked arena and are copied to where prove() wants them when the
    // host next waits for the stream (read_back() / stream_wait()): hipMemcpyAsync into PAGEABLE memory does not return until the copy has
    // run, and the runtime waits for it spinning -- every context thread of a pool burned a CPU for as long as its proof's kernels ran
    // (0.84 CPU-seconds per FinalExp proof with eight in flight against 0.27 with the arena; bench.py: host.cpu_seconds_per_proof_by_role).
    void* rb = nullptr;
    size_t rb_cap = 0, rb_used = 0;
    struct Pending { void* dst; const void* src; size_t bytes; };
    std::vector<Pending> rb_pending;
    void* host_staging = nullptr;  // page-locked: a recording's parts gathered for one upload (prove(), layout 2); scattered columns (layout 3)
    size_t host_staging_cap = 0;
    hipEvent_t col_ev[2] = {nullptr, nullptr};  // layout 3: the two halves of host_staging, each free again when its copy has run
    std::set<int> blob_airs;  // AIRs this context has reserved page-locked proof blobs for (blob_arena.h)
    bool hash_requested = false;
    bool urgent = false;  // ctx_set_urgent
    // tuning (starkhip_set_option; defaults are the measured best)
    long opt_quotient_impl = 0;   // 0: tiled evaluator (quotient_plan.h), 1: op-stream interpreter (quotient_ops.h)
    long opt_quotient_waves = 65536, opt_quotient_slots = 0, opt_quotient_chunks = 0, opt_quotient_debug = 0, opt_zeta_on_coset = 0;
    // Shape-dependent tables and the per-AIR constraint plan are CACHED per context: a pooled context that alternates between
    // AIRs (a PairingPrecomp proof, then an FP12Mul one) finds both again instead of rebuilding the plan on the host and
    // re-allocating device buffers -- hipFree synchronises the whole device, i.e. waits for every other proof's kernels.
    struct Tables {
        int log_n = -1, rate = -1, qdb = -1;
        DevBuf tw_fwd, tw_inv, coset_scale, qtab, qshift_inv;
        DevBuf lde2_fwd, lde2_inv, lde2_cs, lde2_oh;  // kernels_lde.hip tables (log_n >= 8)
        DevBuf lde_wave;                               // ... and of its wave-resident kernel (log_n == 13)
    };
    struct PlanDev {  // tiled plan (quotient_plan.h) of one AIR on the device
        int air = -1;
        unsigned chunks = 0, want = 0;
        uint32_t recs = 0;
        DevBuf q_recs, q_streams, q_chunk_tile_off, q_tile_list, q_contrib_off, q_contribs, q_consts, q_apow;
    };
    std::vector<std::unique_ptr<Tables>> table_cache;
    std::vector<std::unique_ptr<PlanDev>> plan_cache;
    Tables* tab = nullptr;    // the current shape's (ensure_tables)
    PlanDev* plan = nullptr;  // the current AIR's (ensure_plan)
    long opt_leaf_hash_form = 0;     // 0: a lone context's commitments: row form for <= 4096 leaves, pair form for >= 32 768, quad form between; 1: quad always; 2: row always; 3: lane always; 4: pair always
#ifdef STARKHIP_LDE_V2_DEFAULT      // A/B builds (make variant NAME=ldev2 DEFS=-DSTARKHIP_LDE_V2_DEFAULT): pooled contexts cannot be given an option from outside
    long opt_lde_impl = 1;
#else
    long opt_lde_impl = 0;           // 0: 8192-row traces take lde_columns_wave_kernel; 1: lde_columns_v2_kernel for every shape (the cross-check)
#endif
    long opt_lde_closed_forms = 1;   // constant / unit-vector columns skip their transforms (kernels_lde.hip); 0: every column is transformed
    long opt_host_commit_leaves = 64; // trace commitments of at most this many leaves (and >= 64 columns) are hashed by host threads (0: never)
    std::vector<gl_t> host_lde;      // their LDE on the host
    // op-stream program (quotient_impl = 1; kept as the cross-check)
    int prog_air = -1;
    unsigned prog_chunks = 0;
    DevBuf d_ops, d_loads, d_chunk_off;  // compile_quotient_ops() + attach_cell_cache() output for prog_air
    unsigned prog_slots = 0;
    std::vector<uint32_t> chunk_k_after;
    // work buffers
    // `lde` is the one big buffer (19.3 GB for FinalExp).  Before the LDE kernel writes it, it holds everything that waits for that
    // kernel: the trace columns as its LAST quarter (the LDE goes out in launches that overwrite only columns already transformed:
    // run_lde_trace) and, at its start, the upload staging (row-major rows before the transpose, a recording's words before the
    // expansion).  Coefficients are the LDE kernel's scratch inside a column's own block and are not kept: openings and the FRI
    // combination read coset 0 of the LDE (kernels_fri.hip).  `values` is the 1/64 of the columns the last LDE launch reads (75 MB), a
    // whole trace only for rate_bits == 0, and starkhip_lde_batch's in-place values / coefficients.  Together 19.6 GB per FinalExp
    // context; rounds 1-3: values + coefficients + staging + LDE = 33.7 GB.
    DevBuf staging, values, lde, digests, pis, apow, chunk_scale, partial, qvals, qcoef, qlde, qdigests, zpow, gzpow, open_local,
        open_next, open_q, ext_apow, comb_partial, comb_out, fri_coef, fri_vals, fri_rows[16], fri_digests[16], scale_tab, pow_state,
        pow_best, qidx, gather_t, gather_q;
};

// Wait for everything enqueued on the context's stream -- SLEEPING, not spinning: the wait goes through an event created with
// hipEventBlockingSync (an interrupt-driven wait).  With several proofs in flight every context has a host thread waiting for
// its stream most of the time; hipStreamSynchronize spins by default (hipDeviceScheduleAuto on a many-core host), and spinning
// threads eat the CPUs -- in a container with a CPU quota, the quota -- that trace generation and the other proofs' Fiat-Shamir
// hashing need.  Per event, so nothing about the device's scheduling flags changes for other libraries in the process (RCCL).
// Waiting for an event WITHOUT a CPU: hipEventSynchronize on a hipEventBlockingSync event does not sleep on this runtime -- measured with
// eight proofs in flight, 0.80 of the 0.84 CPU-seconds a context thread spends per FinalExp proof were inside that call (it yields, so it
// only shows where CPUs are idle; where they are not, it takes them from the recordings, which run at nice 10).  The device phases it
// waits for are milliseconds long, so: look a few times, then sleep in steps that grow from 20 to 200 microseconds.
hipError_t event_wait_sleeping(hipEvent_t ev) {
#ifdef STARKHIP_RUNTIME_WAIT  // A/B builds: the runtime's own wait (rounds 3-4)
    return hipEventSynchronize(ev);
#endif
    for (int spin = 0; spin < 8; spin++) {
        const hipError_t q = hipEventQuery(ev);
        if (q != hipErrorNotReady) return q;
    }
    (void)hipGetLastError();  // hipErrorNotReady is not an error (and must not surface at the next launch)
    timespec ts = {0, 20000};
    for (;;) {
        nanosleep(&ts, nullptr);
        const hipError_t q = hipEventQuery(ev);
        if (q != hipErrorNotReady) return q;
        (void)hipGetLastError();
        if (ts.tv_nsec < 200000) ts.tv_nsec += ts.tv_nsec / 2;
    }
}

uint64_t thread_cpu_ns();                      // trace_tasks.cpp
std::atomic<uint64_t> g_wait_cpu_ns(0);       // CPU time the context threads spend INSIDE their waits for the device (should be next to nothing)
static hipError_t stream_wait(Ctx* c) {
    const uint64_t cpu0 = thread_cpu_ns();
    hipError_t e = hipEventRecord(c->wait_ev, c->st);
    if (e == hipSuccess) e = event_wait_sleeping(c->wait_ev);
    g_wait_cpu_ns.fetch_add(thread_cpu_ns() - cpu0);
    for (const Ctx::Pending& p : c->rb_pending)  // the read-backs requested since the last wait have landed in the arena
        if (e == hipSuccess) memcpy(p.dst, p.src, p.bytes);
    c->rb_pending.clear();
    c->rb_used = 0;
    return e;
}

// device -> host on the context's stream, complete after the next stream_wait(c); `dst` may be pageable
static hipError_t read_back(Ctx* c, void* dst, const void* src, size_t bytes, hipStream_t st) {
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (!c->rb || c->rb_used + need > c->rb_cap || st != c->st) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);  // does not fit: the direct (blocking) way
    void* slot = (char*)c->rb + c->rb_used;
    c->rb_used += need;
    const hipError_t e = hipMemcpyAsync(slot, src, bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) c->rb_pending.push_back({dst, slot, bytes});
    return e;
}

static int ensure_tables(Ctx* c, unsigned log_n, unsigned rate, unsigned qdb) {
    for (auto& t : c->table_cache)
        if (t->log_n == (int)log_n && t->rate == (int)rate && t->qdb == (int)qdb) {
            c->tab = t.get();
            return 0;
        }
    std::unique_ptr<Ctx::Tables> fresh(new Ctx::Tables());
    Ctx::Tables* T = fresh.get();
    struct Release {  // a half-built set of tables is not kept
        Ctx::Tables* t;
        ~Release() {
            if (!t) return;
            for (DevBuf* b : {&t->tw_fwd, &t->tw_inv, &t->coset_scale, &t->qtab, &t->qshift_inv, &t->lde2_fwd, &t->lde2_inv, &t->lde2_cs, &t->lde2_oh, &t->lde_wave}) b->release();
        }
    } guard{T};
    const unsigned log_N = log_n + rate;
    const size_t N = (size_t)1 << log_N, size = (size_t)1 << (log_n + qdb);
    HIPCHK(T->tw_fwd.ensure(N / 2 * 8 + 8));
    HIPCHK(T->tw_inv.ensure(N / 2 * 8 + 8));
    HIPCHK(T->coset_scale.ensure(N * 8));
    HIPCHK(T->qtab.ensure(4 * size * 8));
    HIPCHK(T->qshift_inv.ensure(size * 8));
    gl_t w = gl_root_of_unity(log_N);
    HIPCHK(launch_fill_powers(T->tw_fwd.as<gl_t>(), 1, w, N / 2, c->st));
    HIPCHK(launch_fill_powers(T->tw_inv.as<gl_t>(), 1, gl_inv(w), N / 2, c->st));
    HIPCHK(launch_fill_coset_scale(T->coset_scale.as<gl_t>(), log_n, rate, c->st));
    HIPCHK(launch_quotient_tables(T->qtab.as<gl_t>(), log_n, qdb, c->st));
    HIPCHK(launch_fill_powers(T->qshift_inv.as<gl_t>(), 1, gl_inv(GL_GENERATOR), size, c->st));
    if (lde_v2_supported(log_n)) {
        HIPCHK(T->lde2_fwd.ensure(lde_v2_tw_words(log_n) * 8));
        HIPCHK(T->lde2_inv.ensure(lde_v2_tw_words(log_n) * 8));
        HIPCHK(T->lde2_cs.ensure(N * 8));
        HIPCHK(T->lde2_oh.ensure(std::max<size_t>(1, lde_v2_oh_words(log_n, rate)) * 8));
        HIPCHK(lde_v2_upload_tables(log_n, rate, T->lde2_fwd.as<gl_t>(), T->lde2_inv.as<gl_t>(), T->lde2_cs.as<gl_t>(), T->lde2_oh.as<gl_t>(), c->st));
        if (lde_wave_supported(log_n)) {
            HIPCHK(T->lde_wave.ensure((lde_wave_table_words(rate) + 1) * 8));  // + the launches' column counter
            HIPCHK(lde_wave_upload_tables(rate, T->lde_wave.as<gl_t>(), c->st));
        }
    }
    T->log_n = log_n;
    T->rate = rate;
    T->qdb = qdb;
    guard.t = nullptr;
    c->table_cache.push_back(std::move(fresh));
    c->tab = T;
    return 0;
}

// IFFT + coset LDE of `cols` columns with the tables of ensure_tables(log_n, rate, .)
static hipError_t run_lde(Ctx* c, const gl_t* values, gl_t* coeffs, gl_t* lde, size_t cols, unsigned log_n, unsigned rate, int from_coeffs) {
    // 8192-row traces (FinalExp, ECCAgg): values -> LDE with nothing kept in between goes through the wave-resident kernel
    if (lde_wave_supported(log_n) && !coeffs && !from_coeffs && c->opt_lde_impl == 0) {
        // the launch's column counter: the last word of the table buffer, cleared in stream order before every launch
        unsigned* next = (unsigned*)(c->tab->lde_wave.as<gl_t>() + lde_wave_table_words(rate));
        if (hipError_t e = hipMemsetAsync(next, 0, sizeof(unsigned), c->st); e != hipSuccess) return e;
        return launch_lde_columns_wave(values, lde, cols, rate, c->tab->lde_wave.as<gl_t>(),
                                       (c->opt_lde_closed_forms && lde_v2_oh_words(log_n, rate)) ? c->tab->lde2_oh.as<gl_t>() : nullptr, next, c->st);
    }
    if (lde_v2_supported(log_n))
        return launch_lde_columns_v2(values, coeffs, lde, cols, log_n, rate, c->tab->lde2_fwd.as<gl_t>(), c->tab->lde2_inv.as<gl_t>(),
                                     c->tab->lde2_cs.as<gl_t>(),
                                     (c->opt_lde_closed_forms && lde_v2_oh_words(log_n, rate)) ? c->tab->lde2_oh.as<gl_t>() : nullptr, from_coeffs, c->st);
    return launch_lde_columns(values, coeffs, lde, cols, log_n, rate, c->tab->tw_fwd.as<gl_t>(), c->tab->tw_inv.as<gl_t>(), log_n + rate,
                              c->tab->coset_scale.as<gl_t>(), from_coeffs, c->st);
}

// The LDE of a trace whose columns are parked in the buffer the LDE goes to, as its last C n words (in_place).  lde_ranges.h has the
// launch plan and why it is safe: launches over 3/4, 3/16, 3/64 of the columns for R = 4, each overwriting only columns an earlier
// launch has transformed, and the last lde_tail_columns(C) columns -- 1/64 of them, 75 MB for FinalExp -- from a copy (`tail`).  Carried
// to the end the series would be log_R(C) launches, the last of them a few columns wide and each as long as one column takes; four
// launches cost 0.07 ms of 17.5 against one (same box, alternating builds).
static hipError_t run_lde_trace(Ctx* c, const gl_t* values, gl_t* lde, gl_t* tail, size_t C, unsigned log_n, unsigned rate, bool in_place) {
    if (!in_place) return run_lde(c, values, nullptr, lde, C, log_n, rate, 0);
    const size_t n = (size_t)1 << log_n, R = (size_t)1 << rate;
    for (const LdeLaunch& l : lde_launch_plan(C, rate)) {
        const gl_t* in = values + l.a * n;
        if (l.from_copy) {
            if (hipError_t e = hipMemcpyAsync(tail, in, (l.b - l.a) * n * 8, hipMemcpyDeviceToDevice, c->st); e != hipSuccess) return e;
            in = tail;
        }
        if (hipError_t e = run_lde(c, in, nullptr, lde + l.a * R * n, l.b - l.a, log_n, rate, 0); e != hipSuccess) return e;
    }
    return hipSuccess;
}

static int ensure_program(Ctx* c, const AirInfo& air, size_t quotient_points) {
    // enough (point-block x chunk) waves to fill 256 CUs several times over
    size_t blocks = (quotient_points + 63) / 64;
    size_t target_waves = (size_t)std::max(64L, c->opt_quotient_waves);  // measured on FinalExp: 8 K waves 61.3 ms, 16 K 56.4, 32 K 54.3, 64 K 53.5, 128 K 52.9
    unsigned want = (unsigned)std::min<size_t>(256, std::max<size_t>(1, (target_waves + blocks - 1) / blocks));
    want = (unsigned)std::min<size_t>(want, air.prog.group_off.size());
    if (c->prog_air == air.id && c->prog_chunks == want) return 0;
    QProgram Q = compile_quotient_ops(air.prog, want);
    want = (unsigned)Q.chunk_k_after.size();
    // per-wave LDS cell cache, OFF by default: measured on FinalExp (MI355X) 0 slots 40 ms, 16: 44, 32: 67, 48: 94 ms.
    // The kernel is bound by memory (253 GB fetched per launch, 6.1 TB/s) and its throughput is proportional to the waves
    // in flight; Belady replacement would hit 38 / 56 / 64 % with 16 / 32 / 64 slots, but the LDS those slots take costs more
    // occupancy than the hits return.  Option "quotient_slots" (0..64) keeps the path testable.
    c->prog_slots = (unsigned)std::min(64L, std::max(0L, c->opt_quotient_slots));
    attach_cell_cache(Q, c->prog_slots);
    HIPCHK(c->d_loads.ensure(Q.loads.size() * 4));
    HIPCHK(hipMemcpyAsync(c->d_loads.p, Q.loads.data(), Q.loads.size() * 4, hipMemcpyHostToDevice, c->st));
    c->chunk_k_after = Q.chunk_k_after;
    HIPCHK(c->d_ops.ensure(Q.ops.size() * sizeof(QOp)));
    HIPCHK(hipMemcpyAsync(c->d_ops.p, Q.ops.data(), Q.ops.size() * sizeof(QOp), hipMemcpyHostToDevice, c->st));
    HIPCHK(c->d_chunk_off.ensure(Q.chunk_batch.size() * 4));
    HIPCHK(hipMemcpyAsync(c->d_chunk_off.p, Q.chunk_batch.data(), Q.chunk_batch.size() * 4, hipMemcpyHostToDevice, c->st));
    HIPCHK(stream_wait(c));  // Q goes out of scope
    c->prog_air = air.id;
    c->prog_chunks = want;
    return 0;
}

// Tiled plan of `air` on the device.  Chunks: enough (64-point block x chunk) workgroups to fill 256 CUs several times over.
static int ensure_plan(Ctx* c, const AirInfo& air, size_t quotient_points) {
    const size_t blocks = (quotient_points + 63) / 64;
    unsigned want = (unsigned)std::min<size_t>(512, std::max<size_t>(1, (8192 + blocks - 1) / blocks));  // FinalExp: 4 chunks 29.8 ms, 8: 29.4, 16: 29.0, 32: 28.9
    if (c->opt_quotient_chunks > 0) want = (unsigned)c->opt_quotient_chunks;
    for (auto& pd : c->plan_cache)
        if (pd->air == air.id && pd->want == want) {
            c->plan = pd.get();
            return 0;
        }
    const QTPlan Q = build_quotient_plan(air.prog, want);
    std::unique_ptr<Ctx::PlanDev> fresh(new Ctx::PlanDev());
    Ctx::PlanDev* D = fresh.get();
    struct Release {
        Ctx::PlanDev* d;
        ~Release() {
            if (!d) return;
            for (DevBuf* b : {&d->q_recs, &d->q_streams, &d->q_chunk_tile_off, &d->q_tile_list, &d->q_contrib_off, &d->q_contribs, &d->q_consts, &d->q_apow}) b->release();
        }
    } guard{D};
    struct Up { DevBuf* b; const void* src; size_t bytes; };
    const std::vector<gl_t>& consts = air.prog.consts;
    const gl_t zero = 0;
    const Up ups[] = {{&D->q_recs, Q.recs.data(), Q.recs.size() * sizeof(QTRec)},
                      {&D->q_streams, Q.streams.data(), Q.streams.size() * sizeof(QTStream)},
                      {&D->q_chunk_tile_off, Q.chunk_tile_off.data(), Q.chunk_tile_off.size() * 4},
                      {&D->q_tile_list, Q.tile_list.empty() ? (const void*)&zero : (const void*)Q.tile_list.data(), std::max<size_t>(1, Q.tile_list.size()) * 4},
                      {&D->q_contrib_off, Q.contrib_off.data(), Q.contrib_off.size() * 4},
                      {&D->q_contribs, Q.contribs.empty() ? (const void*)&zero : (const void*)Q.contribs.data(), std::max<size_t>(1, Q.contribs.size()) * sizeof(QTContrib)},
                      {&D->q_consts, consts.empty() ? (const void*)&zero : (const void*)consts.data(), std::max<size_t>(1, consts.size()) * 8}};
    for (const Up& u : ups) {
        HIPCHK(u.b->ensure(u.bytes));
        HIPCHK(hipMemcpyAsync(u.b->p, u.src, u.bytes, hipMemcpyHostToDevice, c->st));
    }
    HIPCHK(D->q_apow.ensure(std::max<size_t>(1, air.prog.n_constraints) * 16));
    HIPCHK(stream_wait(c));  // Q goes out of scope
    D->air = air.id;
    D->want = want;
    D->chunks = Q.n_chunks;
    D->recs = (uint32_t)Q.recs.size();
    guard.d = nullptr;
    c->plan_cache.push_back(std::move(fresh));
    c->plan = D;
    return 0;
}

// digest buffer: level 0 (n_leaves nodes) followed by level 1, ... ; offset of level l in nodes
static inline size_t level_off(size_t n_leaves, unsigned l) { return 2 * n_leaves - (2 * n_leaves >> l); }
static inline size_t digest_words(size_t n_leaves) { return 8 * n_leaves; }

// Which leaf-hash form a LONE context uses (a pool's commitments go through its scheduler, which merges the small ones into quad
// launches): the quad form of a commitment with <= 4096 leaves is at most 256 waves on 1024 SIMDs, each a chain of up to 12 167
// sequential permutations, so the form with fewer instructions per wave and permutation wins (MillerLoop 119 -> ms, kernels_hash.hip).
static bool use_row_form(const Ctx* c, size_t n_cols, unsigned log_N) {
    if (c->opt_leaf_hash_form == 1) return false;
    if (c->opt_leaf_hash_form == 2) return true;
    if (c->opt_leaf_hash_form == 3 || c->opt_leaf_hash_form == 4) return false;
    return log_N <= 12 && n_cols >= 64;
}
// The pair form (two lanes per leaf, 256 registers per wave) fills the chip from 32 768 leaves on: 1 024 waves, one per SIMD.
static bool use_pair_form(const Ctx* c, size_t n_cols, unsigned log_N) {
    if (c->opt_leaf_hash_form == 4) return true;
    if (c->opt_leaf_hash_form != 0) return false;
    return log_N >= 15 && n_cols >= 64;
}

int ctx_create(int device, Ctx** out, int priority) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return STARKHIP_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return STARKHIP_ERR_NO_DEVICE;
    HIPCHK(hipSetDevice(device));
    Ctx* c = new Ctx();
    c->device = device;
    for (auto& e : c->ev) e = nullptr;
    for (auto& e : c->kev) e = nullptr;
    bool ok;
    if (priority) {  // +1: the highest stream priority of the device, -1: the lowest (pooled contexts, starkhip_pool_config_t)
        int least = 0, greatest = 0;
        ok = hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess &&
             hipStreamCreateWithPriority(&c->st_normal, hipStreamDefault, priority > 0 ? greatest : least) == hipSuccess;
    } else {
        ok = hipStreamCreate(&c->st_normal) == hipSuccess;
    }
    c->st = c->st_normal;
    for (auto& e : c->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    for (auto& e : c->kev) ok = ok && hipEventCreate(&e) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->hash_ready, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->hash_done, hipEventDisableTiming | hipEventBlockingSync) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->wait_ev, hipEventDisableTiming | hipEventBlockingSync) == hipSuccess;
    ok = ok && hipEventCreate(&c->hash_timing.t0) == hipSuccess && hipEventCreate(&c->hash_timing.t1) == hipSuccess;
    if (ok && hipHostMalloc(&c->rb, (size_t)8 << 20, hipHostMallocDefault) == hipSuccess) c->rb_cap = (size_t)8 << 20;  // (without it read-backs go the direct way)
    else c->rb = nullptr;
    if (!ok) {  // release whatever was created
        if (c->rb) (void)hipHostFree(c->rb);
        if (c->hash_ready) (void)hipEventDestroy(c->hash_ready);
        if (c->hash_done) (void)hipEventDestroy(c->hash_done);
        if (c->wait_ev) (void)hipEventDestroy(c->wait_ev);
        if (c->hash_timing.t0) (void)hipEventDestroy(c->hash_timing.t0);
        if (c->hash_timing.t1) (void)hipEventDestroy(c->hash_timing.t1);
        for (auto& e : c->ev)
            if (e) (void)hipEventDestroy(e);
        for (auto& e : c->kev)
            if (e) (void)hipEventDestroy(e);
        if (c->st_normal) (void)hipStreamDestroy(c->st_normal);
        delete c;
        return STARKHIP_ERR_HIP;
    }
    *out = c;
    return 0;
}

void ctx_destroy(Ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->st);
    for (auto& t : c->table_cache)
        for (DevBuf* b : {&t->tw_fwd, &t->tw_inv, &t->coset_scale, &t->qtab, &t->qshift_inv, &t->lde2_fwd, &t->lde2_inv, &t->lde2_cs, &t->lde2_oh, &t->lde_wave}) b->release();
    for (auto& d : c->plan_cache)
        for (DevBuf* b : {&d->q_recs, &d->q_streams, &d->q_chunk_tile_off, &d->q_tile_list, &d->q_contrib_off, &d->q_contribs, &d->q_consts, &d->q_apow}) b->release();
    DevBuf* bufs[] = {&c->d_ops, &c->d_loads, &c->d_chunk_off, &c->staging,
                      &c->values, &c->lde, &c->digests, &c->pis, &c->apow, &c->chunk_scale, &c->partial, &c->qvals, &c->qcoef,
                      &c->qlde, &c->qdigests, &c->zpow, &c->gzpow, &c->open_local, &c->open_next, &c->open_q, &c->ext_apow, &c->comb_partial,
                      &c->comb_out, &c->fri_coef, &c->fri_vals, &c->scale_tab, &c->pow_state, &c->pow_best, &c->qidx, &c->gather_t,
                      &c->gather_q};
    for (auto b : bufs) b->release();
    for (auto& b : c->fri_rows) b.release();
    for (auto& b : c->fri_digests) b.release();
    for (auto& e : c->ev) (void)hipEventDestroy(e);
    for (auto& e : c->kev) (void)hipEventDestroy(e);
    if (c->rb) (void)hipHostFree(c->rb);
    (void)hipEventDestroy(c->hash_ready);
    (void)hipEventDestroy(c->hash_done);
    (void)hipEventDestroy(c->wait_ev);
    (void)hipEventDestroy(c->hash_timing.t0);
    (void)hipEventDestroy(c->hash_timing.t1);
    for (auto& e : c->col_ev)
        if (e) (void)hipEventDestroy(e);
    if (c->host_staging) (void)hipHostFree(c->host_staging);
    blob_arena_drop(c);
    (void)hipStreamDestroy(c->st_normal);
    if (c->st_high) (void)hipStreamDestroy(c->st_high);
    delete c;
}
void ctx_attach_hash_service(Ctx* c, HashService* hs) { c->hs = hs; }
// The next proofs of this context run on a high-priority stream (urgent = true) or on its ordinary one.  Between proofs only:
// a context's stream is idle then.
int ctx_set_urgent(Ctx* c, bool urgent) {
    if (urgent && !c->st_high) {
        HIPCHK(hipSetDevice(c->device));
        int least = 0, greatest = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPCHK(hipStreamCreateWithPriority(&c->st_high, hipStreamDefault, greatest));
    }
    c->st = urgent ? c->st_high : c->st_normal;
    c->urgent = urgent;
    return STARKHIP_OK;
}
bool ctx_has_hash_service(Ctx* c) { return c->hs != nullptr; }
void ctx_hash_request_reset(Ctx* c) { c->hash_requested = false; }
bool ctx_hash_requested(Ctx* c) { return c->hash_requested; }

hipStream_t ctx_stream(Ctx* c) { return c->st; }
int ctx_set_option(Ctx* c, const char* name, long value) {
    if (!c || !name) return STARKHIP_ERR_BAD_SHAPE;
    const std::string k(name);
    if (k == "quotient_impl" && (value == 0 || value == 1)) c->opt_quotient_impl = value;
    else if (k == "quotient_waves" && value >= 64) { c->opt_quotient_waves = value; c->prog_air = -1; }
    else if (k == "quotient_slots" && value >= 0 && value <= 64) { c->opt_quotient_slots = value; c->prog_air = -1; }
#ifdef STARKHIP_DEBUG  // make DEBUG_KNOBS=1 only: modes 1..4, 8 switch arithmetic off (timing decomposition; the proof is then WRONG and
                       // prove() refuses to return it), 9 compares the two evaluators point by point on stderr
    else if (k == "quotient_debug" && value >= 0 && value <= 9) c->opt_quotient_debug = value;
#endif
    else if (k == "zeta_on_coset" && value >= 0) c->opt_zeta_on_coset = value;  // tests: substitute zeta = 7 w_n^(value - 1); the proof is not a transcript any more
    else if (k == "lde_closed_forms" && (value == 0 || value == 1)) c->opt_lde_closed_forms = value;
    else if (k == "lde_impl" && (value == 0 || value == 1)) c->opt_lde_impl = value;
    else if (k == "host_commit_leaves" && value >= 0 && value <= 4096) c->opt_host_commit_leaves = value;
    else if (k == "leaf_hash_form" && value >= 0 && value <= 4) c->opt_leaf_hash_form = value;
    else if (k == "quotient_chunks" && value >= 0 && value <= 4096) c->opt_quotient_chunks = value;  // plans are cached by (AIR, chunks)
    else return STARKHIP_ERR_BAD_SHAPE;
    return STARKHIP_OK;
}
size_t ctx_device_bytes(Ctx* c) {
    size_t total = 0;
    for (auto& t : c->table_cache)
        for (DevBuf* b : {&t->tw_fwd, &t->tw_inv, &t->coset_scale, &t->qtab, &t->qshift_inv, &t->lde2_fwd, &t->lde2_inv, &t->lde2_cs, &t->lde2_oh, &t->lde_wave}) total += b->cap;
    for (auto& d : c->plan_cache)
        for (DevBuf* b : {&d->q_recs, &d->q_streams, &d->q_chunk_tile_off, &d->q_tile_list, &d->q_contrib_off, &d->q_contribs, &d->q_consts, &d->q_apow}) total += b->cap;
    DevBuf* bufs[] = {&c->d_ops, &c->d_loads, &c->d_chunk_off, &c->staging, &c->values, &c->lde, &c->digests, &c->pis, &c->apow, &c->chunk_scale, &c->partial,
                      &c->qvals, &c->qcoef, &c->qlde, &c->qdigests, &c->zpow, &c->gzpow, &c->open_local, &c->open_next, &c->open_q, &c->ext_apow,
                      &c->comb_partial, &c->comb_out, &c->fri_coef, &c->fri_vals, &c->scale_tab, &c->pow_state, &c->pow_best, &c->qidx, &c->gather_t,
                      &c->gather_q};
    for (auto b : bufs) total += b->cap;
    for (auto& b : c->fri_rows) total += b.cap;
    for (auto& b : c->fri_digests) total += b.cap;
    return total;
}
size_t ctx_pinned_bytes(Ctx* c) { return c->host_staging_cap + c->rb_cap; }
const float* ctx_timings(Ctx* c) { return c->timings; }
const float* ctx_kernel_timings(Ctx* c) { return c->ktimings; }
const float* ctx_host_timings(Ctx* c) { return c->htimings; }
void ctx_commit_info(Ctx* c, int* form, unsigned* group) {
    *form = c->hash_timing.form;
    *group = c->hash_timing.group;
}

// (F(X) - F(z)) / (X - z), padded with one zero coefficient back to length n (plonky2 divide_by_linear + push(0))
static void divide_by_linear(const gl2_t* F, size_t n, gl2_t z, gl2_t* q) {
    gl2_t carry = gl2_zero();
    q[n - 1] = gl2_zero();
    for (size_t k = n; k-- > 1;) {
        carry = gl2_add(F[k], gl2_mul(carry, z));
        q[k - 1] = carry;
    }
}

namespace {
struct HostWatch {  // accumulates wall time of the host-side stretches of prove()
    double ms = 0;
    std::chrono::steady_clock::time_point t0;
    void start() { t0 = std::chrono::steady_clock::now(); }
    void stop() { ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace

int prove(Ctx* c, const AirInfo& air, const starkhip_config_t& cfg, const uint64_t* trace, size_t n_rows, int layout, int on_device,
          const uint64_t* pis_host, size_t n_pis, uint64_t pow_witness, uint64_t** proof_out, size_t* proof_words) {
    struct ReadBackGuard {  // an early return between a read_back() and its stream_wait() must not leave destinations of this call behind
        Ctx* c;
        ~ReadBackGuard() {
            c->rb_pending.clear();
            c->rb_used = 0;
        }
    } read_back_guard{c};
    const AirProgram& P = air.prog;
    unsigned log_n = 0;
    while (((size_t)1 << log_n) < n_rows) log_n++;
    if (n_rows < 2 || ((size_t)1 << log_n) != n_rows || n_pis != P.n_pis || cfg.num_challenges != 2 || log_n > 13) return STARKHIP_ERR_BAD_SHAPE;
    FriGeometry geo;
    if (!FriGeometry::make(cfg, log_n, &geo)) return STARKHIP_ERR_BAD_SHAPE;
    const unsigned r = cfg.rate_bits, cap_h = cfg.cap_height, log_N = log_n + r;
    const unsigned factor = P.degree > 1 ? P.degree - 1 : 1;
    unsigned qdb = 0;
    while ((1u << qdb) < factor) qdb++;
    if (qdb > r) return STARKHIP_ERR_BAD_SHAPE;
    for (size_t i = 0; i < n_pis; i++)
        if (pis_host[i] >= GL_P) return STARKHIP_ERR_BAD_SHAPE;
    const size_t n = n_rows, N = n << r, C = P.n_cols, Q = (size_t)factor * 2, size = n << qdb, ncap = (size_t)1 << cap_h;
    const size_t L = geo.arities.size();
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = c->st;
    int rc;
    if ((rc = ensure_tables(c, log_n, r, qdb))) return rc;
    const bool tiled = c->opt_quotient_impl == 0;
    if ((rc = tiled ? ensure_plan(c, air, size) : ensure_program(c, air, size))) return rc;
    const unsigned n_chunks = tiled ? c->plan->chunks : c->prog_chunks;

    // ---- buffers
    // The trace waits for the LDE INSIDE the buffer the LDE is written to, as its last C n words (trace_in_lde: run_lde_trace below);
    // a separate buffer only when there is no room beside it (rate_bits == 0, or a recording longer than the rest of the buffer).
    size_t park_words = 0;  // what the upload parks at the start of the LDE buffer, in 64-bit words
    if (layout == 2) {
        const TraceLog* log = (const TraceLog*)trace;
        park_words = (log->total_words() + log->total_records() + log->total_late_zeros() + 2 + 1) / 2;
    } else if (!on_device && layout == 0) {
        park_words = C * n;
    }
    const bool trace_in_lde = r >= 1 && park_words <= (((size_t)1 << r) - 1) * C * n && !(on_device && layout == 1);
    if (trace_in_lde) HIPCHK(c->values.ensure(lde_tail_columns(C) * n * 8));  // the columns the last LDE launch reads (run_lde_trace)
    else if (!(on_device && layout == 1)) HIPCHK(c->values.ensure(C * n * 8));
    HIPCHK(c->lde.ensure(std::max(C * N * 8, park_words * 8)));
    gl_t* const d_trace = trace_in_lde ? c->lde.as<gl_t>() + (N - n) * C : c->values.as<gl_t>();
    HIPCHK(c->digests.ensure(digest_words(N) * 8));
    HIPCHK(c->pis.ensure(std::max<size_t>(1, n_pis) * 8));
    HIPCHK(c->apow.ensure(2 * (AIR_MAX_GROUP + 1) * 8));
    HIPCHK(c->chunk_scale.ensure(2 * n_chunks * 8));
    HIPCHK(c->partial.ensure((size_t)n_chunks * 2 * size * 8));
    HIPCHK(c->qvals.ensure(2 * size * 8));
    HIPCHK(c->qcoef.ensure(Q * n * 8));
    HIPCHK(c->qlde.ensure(Q * N * 8));
    HIPCHK(c->qdigests.ensure(digest_words(N) * 8));
    HIPCHK(c->zpow.ensure(2 * n * 16));  // powers of zeta (the quotient polynomials' openings), then the coset-0 weights of zeta
    HIPCHK(c->gzpow.ensure(n * 16));     // the weights of g zeta
    HIPCHK(c->open_local.ensure(C * 16));
    HIPCHK(c->open_next.ensure(C * 16));
    HIPCHK(c->open_q.ensure(Q * 16));
    HIPCHK(c->ext_apow.ensure((C + Q) * 16));
    const size_t comb_ppc = 256, comb_chunks = (C + comb_ppc - 1) / comb_ppc;
    HIPCHK(c->comb_partial.ensure(comb_chunks * n * 16));
    HIPCHK(c->comb_out.ensure(2 * n * 16));
    HIPCHK(c->fri_coef.ensure(2 * N * 8));
    HIPCHK(c->fri_vals.ensure(2 * N * 8));
    HIPCHK(c->scale_tab.ensure(N * 8));
    HIPCHK(c->pow_state.ensure(12 * 8));
    HIPCHK(c->pow_best.ensure(8));
    HIPCHK(c->qidx.ensure(cfg.num_query_rounds * 4));

    int evi = 0;
    PhaseRanges ranges;  // rocTX ranges named like the phases of starkhip_last_timings (visible with rocprofv3 --marker-trace)
    HIPCHK(hipEventRecord(c->ev[evi++], st));
    ranges.next("starkhip:upload");

    // ---- phase 0: trace into column-major device memory (trace_rows_to_poly_values)
    const gl_t* d_values;
    if (layout == 2) {  // compact trace: upload the generator's write log and expand it here
        const TraceLog* log = (const TraceLog*)trace;
        const size_t nw = log->total_words(), nr = log->total_records(), nz = log->total_late_zeros();
        if (log->rows != n || log->cols != C) return STARKHIP_ERR_BAD_SHAPE;
        uint32_t* d_words = c->lde.as<uint32_t>();  // the recording's words wait at the start of the (still unused) LDE buffer
        uint32_t* d_offsets = d_words + nw;
        uint32_t* d_zeros = d_offsets + nr;
        HIPCHK(hipMemsetAsync(d_trace, 0, C * n * 8, st));
        {  // A log recorded by several threads comes in parts (trace_log.h): each part's words land at its base, its offsets
           // (already shifted by that base) and late zeros back to back.  The parts are gathered into ONE page-locked staging
           // buffer of the context and go up as ONE copy: a FinalExp recording has 53 parts x 3 arrays, and on a GPU that other
           // proofs keep busy every one of 160 dependent stream operations waits its turn (measured: 0.9 - 1.6 s of "upload" for a
           // proof whose copies queued behind other proofs' commitments, against 6 ms alone).
            const size_t total = nw + nr + nz;
            if (c->host_staging_cap < total * 4) {
                if (c->host_staging) (void)hipHostFree(c->host_staging);
                c->host_staging = nullptr;
                c->host_staging_cap = 0;
                const size_t want = total * 4 + total;  // + 25 %: the next recording of this AIR is about as long
                HIPCHK(hipHostMalloc(&c->host_staging, want, hipHostMallocDefault));
                c->host_staging_cap = want;
            }
            uint32_t* h = (uint32_t*)c->host_staging;
            size_t at_r = 0, at_z = 0;
            struct Piece { uint32_t* dst; const uint32_t* src; size_t words; };
            std::vector<Piece> pieces;
            log->for_each_part([&](const TraceLog& part) {
                if (!part.words.empty()) pieces.push_back({h + part.base, part.words.data(), part.words.size()});
                if (!part.offsets.empty()) pieces.push_back({h + nw + at_r, part.offsets.data(), part.offsets.size()});
                if (!part.late_zeros.empty()) pieces.push_back({h + nw + nr + at_z, part.late_zeros.data(), part.late_zeros.size()});
                at_r += part.offsets.size();
                at_z += part.late_zeros.size();
            });
            // 150 MB for FinalExp: gathered on a few threads (10 ms on one), pieces dealt round-robin
            const unsigned n_thr = total * 4 > ((size_t)32 << 20) ? 4 : 1;
            auto gather = [&](unsigned w) {
                for (size_t i = w; i < pieces.size(); i += n_thr) memcpy(pieces[i].dst, pieces[i].src, pieces[i].words * 4);
            };
            std::vector<std::thread> helpers;
            for (unsigned w = 1; w < n_thr; w++) {
                try {
                    helpers.emplace_back(gather, w);
                } catch (const std::system_error&) {
                    gather(w);  // no thread to be had: this one does that share too
                }
            }
            gather(0);
            for (std::thread& t : helpers) t.join();
            if (total) HIPCHK(hipMemcpyAsync(d_words, h, total * 4, hipMemcpyHostToDevice, st));
        }
        if (nr) HIPCHK(launch_expand_trace(d_words, d_offsets, nr, d_trace, n, st));
        if (nz) HIPCHK(launch_zero_cells(d_zeros, nz / 2, d_trace, n, st));
        d_values = d_trace;
    } else if (layout == 3) {
        // The literal argument of starky's prove(): `Vec<PolynomialValues<F>>`, one heap allocation per column
        // (/root/reference/src/aggregate_proof.rs:168-175) -- `trace` is a table of C column pointers.  C separate pageable copies of
        // 64 KB would each be staged by the runtime (73 527 of them for FinalExp); instead host threads gather runs of columns into
        // the two halves of the context's page-locked staging and every half goes up as one copy, the gather of the next half under
        // the copy of this one.  Column-major device memory is just the columns back to back.
        const uint64_t* const* cols = (const uint64_t* const*)trace;
        const size_t col_bytes = n * 8;
        if (c->host_staging_cap < 2 * col_bytes || c->host_staging_cap < ((size_t)32 << 20)) {
            const size_t want = std::max<size_t>(2 * col_bytes, (size_t)128 << 20);
            if (c->host_staging_cap < want) {
                if (c->host_staging) (void)hipHostFree(c->host_staging);
                c->host_staging = nullptr;
                c->host_staging_cap = 0;
                HIPCHK(hipHostMalloc(&c->host_staging, want, hipHostMallocDefault));
                c->host_staging_cap = want;
            }
        }
        for (auto& e : c->col_ev)
            if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync));
        const size_t half_bytes = c->host_staging_cap / 2, per_half = std::max<size_t>(1, half_bytes / col_bytes);
        bool used[2] = {false, false};
        unsigned h = 0;
        for (size_t c0 = 0; c0 < C; c0 += per_half, h ^= 1) {
            const size_t cnt = std::min(per_half, C - c0);
            char* dst = (char*)c->host_staging + (size_t)h * half_bytes;
            if (used[h]) HIPCHK(event_wait_sleeping(c->col_ev[h]));  // the copy that last read this half has run
            const unsigned n_thr = cnt * col_bytes > ((size_t)8 << 20) ? 4 : 1;
            auto gather = [&](unsigned w) {
                for (size_t i = w; i < cnt; i += n_thr) memcpy(dst + i * col_bytes, cols[c0 + i], col_bytes);
            };
            std::vector<std::thread> helpers;
            for (unsigned w = 1; w < n_thr; w++) {
                try {
                    helpers.emplace_back(gather, w);
                } catch (const std::system_error&) {
                    gather(w);
                }
            }
            gather(0);
            for (std::thread& t : helpers) t.join();
            HIPCHK(hipMemcpyAsync(d_trace + c0 * n, dst, cnt * col_bytes, hipMemcpyHostToDevice, st));
            HIPCHK(hipEventRecord(c->col_ev[h], st));
            used[h] = true;
        }
        d_values = d_trace;
    } else if (on_device && layout == 1) {
        d_values = trace;  // the caller's memory: read only
    } else if (on_device) {
        HIPCHK(launch_transpose(trace, d_trace, n, C, st));
        d_values = d_trace;
    } else if (layout == 1) {
        HIPCHK(hipMemcpyAsync(d_trace, trace, C * n * 8, hipMemcpyHostToDevice, st));
        d_values = d_trace;
    } else {
        // row-major host rows: up into the start of the LDE buffer (idle until the LDE kernel writes it), transposed from there
        HIPCHK(hipMemcpyAsync(c->lde.p, trace, C * n * 8, hipMemcpyHostToDevice, st));
        HIPCHK(launch_transpose(c->lde.as<gl_t>(), d_trace, n, C, st));
        d_values = d_trace;
    }
    HIPCHK(hipEventRecord(c->ev[evi++], st));
    ranges.next("starkhip:ifft_lde");

    // ---- phase 1: IFFT + LDE (PolynomialBatch::from_values, App. A.3)
    HIPCHK(hipEventRecord(c->kev[4], st));
    HIPCHK(run_lde_trace(c, d_values, c->lde.as<gl_t>(), c->values.as<gl_t>(), C, log_n, r, trace_in_lde));
    HIPCHK(hipEventRecord(c->kev[5], st));
    HIPCHK(hipEventRecord(c->ev[evi++], st));
    ranges.next("starkhip:trace_merkle");

    // ---- phase 2: Merkle tree over bit-reversed LDE rows
    HIPCHK(hipEventRecord(c->kev[0], st));
    if (N <= (size_t)c->opt_host_commit_leaves && C >= 64 && c->opt_leaf_hash_form == 0) {
        // A commitment of a few leaves is a latency chain on the GPU whatever the form: FP12Mul (16 rows at blow-up 2) has 32 leaves of
        // 7 536 sequential permutations -- 42 ms in the row form at 5.6 us per permutation on eight waves of a chip that holds 4 096.
        // The host permutation the challenger uses runs at 0.8 us, and 32 independent leaves spread over the process's CPUs: the LDE
        // (15 MB) comes down, host threads hash the leaves, the digests go back up and the tree is built on the device as usual.
        // Same function, same bytes (tests/test_gpu_airs.py: FP12Mul against the oracle).
        if (c->hs && !HashService::is_big(log_n, r)) {  // a pooled proof was announced to the scheduler's window: it is not coming
            c->hs->abandon_small();
            c->hash_requested = true;
        }
        c->hash_timing.form = 4;
        c->hash_timing.group = 1;
        c->host_lde.resize(C * N);
        HIPCHK(hipMemcpyAsync(c->host_lde.data(), c->lde.p, C * N * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(stream_wait(c));
        std::vector<gl_t> leaf_digests(4 * N);
        const gl_t* lde_h = c->host_lde.data();
        const unsigned n_thr = (unsigned)std::min<size_t>(N, std::max(1u, cpu_budget()));
        auto work = [&](unsigned w) {
            for (size_t j = w; j < N; j += n_thr) {
                size_t i = 0;  // leaf j holds the LDE row of natural point index bitrev(j) = k * R + s, stored coset-major at [s][k]
                for (unsigned b = 0; b < log_N; b++) i |= ((j >> b) & 1) << (log_N - 1 - b);
                const size_t s_ = i & (((size_t)1 << r) - 1), k_ = i >> r;
                gl_t state[12] = {0};
                const gl_t* col = lde_h + s_ * n + k_;
                for (size_t off = 0; off < C; off += 8) {
                    const size_t cnt = std::min<size_t>(8, C - off);
                    for (size_t e = 0; e < cnt; e++) state[e] = col[(off + e) * N];
                    poseidon_permute_host(state);
                }
                for (int e = 0; e < 4; e++) leaf_digests[4 * j + e] = state[e];
            }
        };
        std::vector<std::thread> helpers;
        for (unsigned w = 1; w < n_thr; w++) {
            try {
                helpers.emplace_back(work, w);
            } catch (const std::system_error&) {
                work(w);
            }
        }
        work(0);
        for (std::thread& t : helpers) t.join();
        HIPCHK(hipMemcpyAsync(c->digests.p, leaf_digests.data(), 4 * N * 8, hipMemcpyHostToDevice, st));
        HIPCHK(stream_wait(c));  // leaf_digests goes out of scope
    } else if (c->hs) {  // pooled: the scheduler decides when this commitment runs and which others share its launch
        c->hash_requested = true;
        HIPCHK(c->hs->hash(c->lde.as<gl_t>(), C, log_n, r, c->digests.as<gl_t>(), st, c->hash_ready, c->hash_done, !HashService::is_big(log_n, r), c->urgent,
                           &c->hash_timing));
    } else if (c->opt_leaf_hash_form == 3) {
        c->hash_timing.form = 3; c->hash_timing.group = 1;
        HIPCHK(launch_leaf_hash_lane(c->lde.as<gl_t>(), C, log_n, r, c->digests.as<gl_t>(), st));
    } else if (use_pair_form(c, C, log_N)) {
        c->hash_timing.form = 5; c->hash_timing.group = 1;
        HIPCHK(launch_leaf_hash_pair(c->lde.as<gl_t>(), C, log_n, r, c->digests.as<gl_t>(), st));
    } else if (use_row_form(c, C, log_N)) {
        c->hash_timing.form = 1; c->hash_timing.group = 1;
        HIPCHK(launch_leaf_hash_row(c->lde.as<gl_t>(), C, log_n, r, c->digests.as<gl_t>(), st));
    } else {
        c->hash_timing.form = 0; c->hash_timing.group = 1;
        HIPCHK(launch_leaf_hash(c->lde.as<gl_t>(), C, log_n, r, c->digests.as<gl_t>(), st));
    }
    HIPCHK(hipEventRecord(c->kev[1], st));
    HIPCHK(launch_merkle_levels(c->digests.as<gl_t>(), log_N, cap_h, st));
    std::vector<gl_t> trace_cap(4 * ncap), quot_cap(4 * ncap);
    HIPCHK(read_back(c, trace_cap.data(), c->digests.as<gl_t>() + 4 * level_off(N, log_N - cap_h), 4 * ncap * 8, st));
    HIPCHK(hipEventRecord(c->ev[evi++], st));
    ranges.next("starkhip:quotient");
    HIPCHK(stream_wait(c));

    HostWatch fs, host_other;  // Fiat-Shamir hashing / other host arithmetic of this proof
    Challenger ch;
    fs.start();
    ch.observe_many(trace_cap.data(), trace_cap.size());  // public inputs are NOT observed (App. A.5)
    gl_t alphas[2] = {ch.get(), ch.get()};
    fs.stop();

    // ---- phase 3: quotient polynomials (App. A.6)
    {
        if (n_pis) HIPCHK(hipMemcpyAsync(c->pis.p, pis_host, n_pis * 8, hipMemcpyHostToDevice, st));
        if (tiled) {
            // per-proof weights of the plan's records, then one pass over the LDE in LDS-staged column tiles
            HIPCHK(launch_quotient_weights(c->plan->q_recs.as<QTRec>(), c->plan->q_contrib_off.as<uint32_t>(), c->plan->q_contribs.as<QTContrib>(), c->plan->recs,
                                           c->plan->q_apow.as<gl_t>(), P.n_constraints, c->plan->q_consts.as<gl_t>(), c->pis.as<gl_t>(), alphas[0], alphas[1], st));
            HIPCHK(hipEventRecord(c->kev[2], st));
            HIPCHK(launch_quotient_tiles(c->plan->q_recs.as<QTRec>(), c->plan->q_streams.as<QTStream>(),
                                         c->plan->q_chunk_tile_off.as<uint32_t>(), c->plan->q_tile_list.as<uint32_t>(), n_chunks, c->lde.as<gl_t>(),
                                         c->tab->qtab.as<gl_t>(), c->partial.as<gl_t>(), log_n, r, qdb, (unsigned)C, (unsigned)((c->opt_quotient_debug <= 4 || c->opt_quotient_debug == 8) ? c->opt_quotient_debug : 0), st));
            HIPCHK(hipEventRecord(c->kev[3], st));
            HIPCHK(launch_quotient_tiles_combine(c->partial.as<gl_t>(), n_chunks, c->tab->qtab.as<gl_t>(), log_n, qdb, c->qvals.as<gl_t>(), st));
        } else {
            std::vector<gl_t> apow(2 * (AIR_MAX_GROUP + 1)), cscale(2 * n_chunks);
            for (int j = 0; j < 2; j++) {
                apow[j * (AIR_MAX_GROUP + 1)] = 1;
                for (unsigned m = 1; m <= AIR_MAX_GROUP; m++) apow[j * (AIR_MAX_GROUP + 1) + m] = gl_mul(apow[j * (AIR_MAX_GROUP + 1) + m - 1], alphas[j]);
                for (unsigned p = 0; p < n_chunks; p++) cscale[p * 2 + j] = gl_pow(alphas[j], c->chunk_k_after[p]);
            }
            HIPCHK(hipMemcpyAsync(c->apow.p, apow.data(), apow.size() * 8, hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(c->chunk_scale.p, cscale.data(), cscale.size() * 8, hipMemcpyHostToDevice, st));
            HIPCHK(hipEventRecord(c->kev[2], st));
            HIPCHK(launch_quotient_eval(c->d_ops.as<QOp>(), c->d_loads.as<uint32_t>(), c->prog_slots, c->d_chunk_off.as<uint32_t>(), n_chunks,
                                        c->pis.as<gl_t>(), c->lde.as<gl_t>(), c->tab->qtab.as<gl_t>(), c->apow.as<gl_t>(), alphas[0], alphas[1],
                                        c->partial.as<gl_t>(), log_n, r, qdb, st));
            HIPCHK(hipEventRecord(c->kev[3], st));
            HIPCHK(launch_quotient_combine(c->partial.as<gl_t>(), c->chunk_scale.as<gl_t>(), n_chunks, c->tab->qtab.as<gl_t>(), log_n, qdb,
                                           c->qvals.as<gl_t>(), st));
            HIPCHK(stream_wait(c));  // apow / cscale go out of scope
        }
        if (c->opt_quotient_debug == 9 && !tiled) {
            // development aid: the tiled evaluator's values against the interpreter's on this very proof (stderr)
            int rc2;
            if ((rc2 = ensure_plan(c, air, size))) return rc2;
            std::vector<gl_t> ref(2 * size), got(2 * size);
            HIPCHK(read_back(c, ref.data(), c->qvals.p, 2 * size * 8, st));
            HIPCHK(c->partial.ensure((size_t)std::max(n_chunks, c->plan->chunks) * 2 * size * 8));
            HIPCHK(launch_quotient_weights(c->plan->q_recs.as<QTRec>(), c->plan->q_contrib_off.as<uint32_t>(), c->plan->q_contribs.as<QTContrib>(), c->plan->recs,
                                           c->plan->q_apow.as<gl_t>(), P.n_constraints, c->plan->q_consts.as<gl_t>(), c->pis.as<gl_t>(), alphas[0], alphas[1], st));
            HIPCHK(launch_quotient_tiles(c->plan->q_recs.as<QTRec>(), c->plan->q_streams.as<QTStream>(),
                                         c->plan->q_chunk_tile_off.as<uint32_t>(), c->plan->q_tile_list.as<uint32_t>(), c->plan->chunks, c->lde.as<gl_t>(),
                                         c->tab->qtab.as<gl_t>(), c->partial.as<gl_t>(), log_n, r, qdb, (unsigned)C, 0, st));
            HIPCHK(launch_quotient_tiles_combine(c->partial.as<gl_t>(), c->plan->chunks, c->tab->qtab.as<gl_t>(), log_n, qdb, c->comb_partial.as<gl_t>(), st));
            HIPCHK(read_back(c, got.data(), c->comb_partial.p, 2 * size * 8, st));
            HIPCHK(stream_wait(c));
            size_t bad = 0;
            size_t last_blk = (size_t)-1;
            for (size_t i = 0; i < 2 * size; i++)
                if (ref[i] != got[i]) {
                    const size_t ii = i % size, k = ii >> qdb, spp = ii & (((size_t)1 << qdb) - 1), tt = spp * n + k, blk = tt / 64;
                    if (blk != last_blk && bad < 4096) fprintf(stderr, "quotient mismatch alpha %zu point-block %zu (sp %zu, k %zu..)\n", i / size, blk, spp, k & ~(size_t)63);
                    last_blk = blk;
                    bad++;
                }
            fprintf(stderr, "quotient compare: %zu of %zu values differ\n", bad, 2 * size);
        }
        // coset_ifft(7): inverse transform, scale by size^-1 and by 7^-i
        HIPCHK(launch_ntt_global(c->qvals.as<gl_t>(), 2, size, log_n + qdb, c->tab->tw_inv.as<gl_t>(), log_N, nullptr, c->tab->qshift_inv.as<gl_t>(),
                                 gl_inv((gl_t)size), st));
        // trim_to_len(n * factor) must succeed, then chunks of n: [alpha0: c0..cf-1, alpha1: c0..cf-1]
        std::vector<gl_t> tail;
        if (size > (size_t)factor * n) {
            tail.resize(2 * (size - factor * n));
            for (int j = 0; j < 2; j++)
                HIPCHK(read_back(c, tail.data() + j * (size - factor * n), c->qvals.as<gl_t>() + j * size + factor * n,
                                      (size - factor * n) * 8, st));
        }
        for (int j = 0; j < 2; j++)
            HIPCHK(hipMemcpyAsync(c->qcoef.as<gl_t>() + (size_t)j * factor * n, c->qvals.as<gl_t>() + j * size, (size_t)factor * n * 8,
                                  hipMemcpyDeviceToDevice, st));
        HIPCHK(hipEventRecord(c->ev[evi++], st));
        ranges.next("starkhip:quotient_commit");
        HIPCHK(stream_wait(c));
        for (gl_t v : tail)
            if (v != 0) return STARKHIP_ERR_QUOTIENT_NOT_DIVISIBLE;
    }

    // ---- phase 4: quotient commit (PolynomialBatch::from_coeffs)
    HIPCHK(run_lde(c, c->qcoef.as<gl_t>(), nullptr, c->qlde.as<gl_t>(), Q, log_n, r, 1));
    HIPCHK(launch_leaf_hash(c->qlde.as<gl_t>(), Q, log_n, r, c->qdigests.as<gl_t>(), st));
    HIPCHK(launch_merkle_levels(c->qdigests.as<gl_t>(), log_N, cap_h, st));
    HIPCHK(read_back(c, quot_cap.data(), c->qdigests.as<gl_t>() + 4 * level_off(N, log_N - cap_h), 4 * ncap * 8, st));
    HIPCHK(hipEventRecord(c->ev[evi++], st));
    ranges.next("starkhip:openings");
    HIPCHK(stream_wait(c));
    fs.start();
    ch.observe_many(quot_cap.data(), quot_cap.size());
    gl2_t zeta = ch.get_ext();
    fs.stop();
    if (c->opt_zeta_on_coset > 0)  // test hook ("zeta_on_coset" = k + 1): zeta = 7 w_n^k, a point of coset 0 itself (probability 2^-115 in a real transcript)
        zeta = gl2_make(gl_mul(GL_GENERATOR, gl_pow(gl_root_of_unity(log_n), (uint64_t)(c->opt_zeta_on_coset - 1) & (n - 1))), 0);
    if (gl2_eq(gl2_pow(zeta, n), gl2_one())) return STARKHIP_ERR_ZETA_IN_SUBGROUP;
    gl2_t gzeta = gl2_mul_base(zeta, gl_root_of_unity(log_n));

    // ---- phase 5: openings (App. A.7)
    std::vector<gl2_t> op_local(C), op_next(C), op_q(Q);
    // the trace polynomials from their values on coset 0 of the LDE (kernels_fri.hip: the context keeps no coefficients of them), the
    // quotient polynomials from their coefficients
    const gl_t shift_n = gl_pow(GL_GENERATOR, n);  // 7^n
    const gl2_t zh = gl2_sub(gl2_pow(zeta, n), gl2_make(shift_n, 0));
    // (zh = 0: zeta lies on the coset 7 H itself -- the weights are then an indicator vector, coset_weights_kernel; starky proves there too)
    const gl2_t w_scale = gl2_mul_base(zh, gl_inv(gl_mul((gl_t)n, shift_n)));
    HIPCHK(launch_ext_powers(c->zpow.as<gl2_t>(), zeta, n, st));
    HIPCHK(launch_coset_weights(c->zpow.as<gl2_t>() + n, c->gzpow.as<gl2_t>(), zeta, w_scale, log_n, st));
    HIPCHK(launch_openings(c->lde.as<gl_t>(), N, C, n, c->zpow.as<gl2_t>() + n, c->gzpow.as<gl2_t>(), c->open_local.as<gl2_t>(),
                           c->open_next.as<gl2_t>(), st));
    HIPCHK(launch_openings(c->qcoef.as<gl_t>(), n, Q, n, c->zpow.as<gl2_t>(), nullptr, c->open_q.as<gl2_t>(), nullptr, st));
    HIPCHK(read_back(c, op_local.data(), c->open_local.p, C * 16, st));
    HIPCHK(read_back(c, op_next.data(), c->open_next.p, C * 16, st));
    HIPCHK(read_back(c, op_q.data(), c->open_q.p, Q * 16, st));
    HIPCHK(hipEventRecord(c->ev[evi++], st));
    ranges.next("starkhip:fri_combine");
    HIPCHK(stream_wait(c));
    fs.start();  // 2 (2 C + Q) field elements through the sponge, one permutation per 8: the longest host stretch of a proof
    for (size_t i = 0; i < C; i++) ch.observe_ext(op_local[i]);
    for (size_t i = 0; i < Q; i++) ch.observe_ext(op_q[i]);
    for (size_t i = 0; i < C; i++) ch.observe_ext(op_next[i]);

    // ---- phase 6: FRI batch combination (prove_openings, App. A.8)
    gl2_t fri_alpha = ch.get_ext();
    fs.stop();
    std::vector<gl2_t> fin(n);
    {
        HIPCHK(launch_ext_powers(c->ext_apow.as<gl2_t>(), fri_alpha, C + Q, st));
        // sum_j alpha^j trace_j: the sum is taken on coset 0 of the LDE (n values a column) and turned into coefficients by ONE inverse
        // coset transform of its two words -- linear, so these are the coefficients of the reference's sum of coefficient vectors
        gl_t* comb_t = c->comb_out.as<gl_t>();                 // [2][n] words
        gl2_t* comb_q = c->comb_out.as<gl2_t>() + n;           // [n] extension elements
        HIPCHK(launch_fri_combine(c->lde.as<gl_t>(), N, C, n, c->ext_apow.as<gl2_t>(), comb_ppc, comb_chunks, c->comb_partial.as<gl2_t>(), st));
        HIPCHK(launch_ext_reduce(c->comb_partial.as<gl2_t>(), comb_chunks, n, comb_t, st));
        HIPCHK(launch_ntt_global(comb_t, 2, n, log_n, c->tab->tw_inv.as<gl_t>(), log_N, nullptr, c->tab->qshift_inv.as<gl_t>(), gl_inv((gl_t)n), st));
        HIPCHK(launch_fri_combine(c->qcoef.as<gl_t>(), n, Q, n, c->ext_apow.as<gl2_t>() + C, Q, 1, comb_q, st));  // alpha^(C+q) quotient_q
        std::vector<gl_t> F1w(2 * n);
        std::vector<gl2_t> F1(n), tailq(n), F0(n), q0(n), q1(n);
        HIPCHK(read_back(c, F1w.data(), comb_t, n * 16, st));
        HIPCHK(read_back(c, tailq.data(), comb_q, n * 16, st));
        HIPCHK(stream_wait(c));
        host_other.start();
        for (size_t k = 0; k < n; k++) F1[k] = gl2_make(F1w[k], F1w[n + k]);
        for (size_t k = 0; k < n; k++) F0[k] = gl2_add(F1[k], tailq[k]);
        divide_by_linear(F0.data(), n, zeta, q0.data());   // batch 0: trace ++ quotient at zeta
        divide_by_linear(F1.data(), n, gzeta, q1.data());  // batch 1: trace at g*zeta
        gl2_t shift = gl2_pow(fri_alpha, C);               // alpha^{|batch 1|}
        for (size_t k = 0; k < n; k++) fin[k] = gl2_add(gl2_mul(q0[k], shift), q1[k]);
        // upload as SoA, zero padded to N (lde(rate_bits))
        std::vector<gl_t> soa(2 * N, 0);
        for (size_t k = 0; k < n; k++) {
            soa[k] = fin[k].a0;
            soa[N + k] = fin[k].a1;
        }
        host_other.stop();
        HIPCHK(hipMemcpyAsync(c->fri_coef.p, soa.data(), 2 * N * 8, hipMemcpyHostToDevice, st));
        HIPCHK(stream_wait(c));
    }
    HIPCHK(hipEventRecord(c->ev[evi++], st));
    ranges.next("starkhip:fri_commit");

    // ---- phase 7: FRI commit phase (fri_committed_trees)
    std::vector<gl_t> fri_caps(L * 4 * ncap);
    std::vector<gl2_t> final_poly(geo.final_poly_len);
    {
        size_t len = N;
        unsigned log_len = log_N;
        gl_t shift = GL_GENERATOR;
        gl_t* coef = c->fri_coef.as<gl_t>();
        gl_t* vals = c->fri_vals.as<gl_t>();
        for (size_t l = 0; l <= L; l++) {
            if (l == L) break;
            // values on shift * <w_len>
            HIPCHK(hipMemcpyAsync(vals, coef, len * 8, hipMemcpyDeviceToDevice, st));
            HIPCHK(hipMemcpyAsync(vals + len, coef + len, len * 8, hipMemcpyDeviceToDevice, st));
            HIPCHK(launch_fill_powers(c->scale_tab.as<gl_t>(), 1, shift, len, st));
            HIPCHK(launch_ntt_global(vals, 2, len, log_len, c->tab->tw_fwd.as<gl_t>(), log_N, c->scale_tab.as<gl_t>(), nullptr, 1, st));
            const unsigned ab = geo.arities[l];
            const size_t n_leaves = len >> ab, width = 2 << ab;
            HIPCHK(c->fri_rows[l].ensure(len * 2 * 8));
            HIPCHK(c->fri_digests[l].ensure(digest_words(n_leaves) * 8));
            HIPCHK(launch_fri_leaves(vals, log_len, ab, c->fri_rows[l].as<gl_t>(), st));
            HIPCHK(launch_leaf_hash_rows(c->fri_rows[l].as<gl_t>(), width, n_leaves, c->fri_digests[l].as<gl_t>(), st));
            HIPCHK(launch_merkle_levels(c->fri_digests[l].as<gl_t>(), log_len - ab, cap_h, st));
            // leaves and digests stay on the device (the query phase gathers from them); only the cap feeds the transcript
            gl_t* cap = fri_caps.data() + l * 4 * ncap;
            HIPCHK(read_back(c, cap, c->fri_digests[l].as<gl_t>() + 4 * level_off(n_leaves, log_len - ab - cap_h), 4 * ncap * 8, st));
            HIPCHK(stream_wait(c));
            fs.start();
            ch.observe_many(cap, 4 * ncap);
            gl2_t beta = ch.get_ext();
            fs.stop();
            // fold coefficients; output goes to the other half of fri_vals' sibling buffer: reuse coef in place via temp
            HIPCHK(launch_fri_fold(coef, len, ab, beta, vals, st));  // vals now holds folded coefficients SoA [2][len >> ab]
            std::swap(coef, vals);
            len >>= ab;
            log_len -= ab;
            for (unsigned b = 0; b < ab; b++) shift = gl_sqr(shift);
        }
        // final polynomial: truncate to len >> rate_bits; the dropped coefficients must be zero
        std::vector<gl_t> fc(2 * len);
        HIPCHK(read_back(c, fc.data(), coef, len * 8, st));
        HIPCHK(read_back(c, fc.data() + len, coef + len, len * 8, st));
        HIPCHK(stream_wait(c));
        if ((len >> r) != geo.final_poly_len) return STARKHIP_ERR_BAD_SHAPE;
        for (size_t k = 0; k < len; k++) {
            if (k < geo.final_poly_len) final_poly[k] = gl2_make(fc[k], fc[len + k]);
            else if (fc[k] || fc[len + k]) return STARKHIP_ERR_QUOTIENT_NOT_DIVISIBLE;
        }
        fs.start();
        for (auto& e : final_poly) ch.observe_ext(e);
        fs.stop();
    }
    HIPCHK(hipEventRecord(c->ev[evi++], st));
    ranges.next("starkhip:pow");

    // ---- phase 8: proof of work (fri_proof_of_work): smallest nonce unless one is supplied
    if (pow_witness == STARKHIP_POW_SEARCH) {
        if (cfg.proof_of_work_bits == 0) {
            pow_witness = 0;
        } else {
            gl_t base[12];
            memcpy(base, ch.state, sizeof base);
            for (int i = 0; i < ch.n_in; i++) base[i] = ch.in[i];
            HIPCHK(hipMemcpyAsync(c->pow_state.p, base, sizeof base, hipMemcpyHostToDevice, st));
            unsigned long long best = ~0ULL;
            const uint64_t batch = 1ULL << 20;
            for (uint64_t start = 0; best == ~0ULL; start += batch) {
                if (start >= GL_P - batch) return STARKHIP_ERR_HIP;
                HIPCHK(hipMemcpyAsync(c->pow_best.p, &best, 8, hipMemcpyHostToDevice, st));
                HIPCHK(launch_pow_grind(c->pow_state.as<gl_t>(), ch.n_in, cfg.proof_of_work_bits, start, batch, c->pow_best.as<unsigned long long>(), st));
                HIPCHK(read_back(c, &best, c->pow_best.p, 8, st));
                HIPCHK(stream_wait(c));
            }
            pow_witness = best;
        }
    }
    ch.observe(pow_witness);
    (void)ch.get();  // pow_response
    HIPCHK(hipEventRecord(c->ev[evi++], st));
    ranges.next("starkhip:queries");

    // ---- phase 9: query rounds (fri_prover_query_rounds)
    ProofLayout pl;
    pl.C = C; pl.Q = Q; pl.log_n = log_n; pl.rate_bits = r; pl.cap_h = cap_h; pl.L = L; pl.n_queries = cfg.num_query_rounds;
    pl.final_len = geo.final_poly_len; pl.n_pis = n_pis; pl.arity_bits = cfg.arity_bits; pl.n_challenges = 2;
    pl.compute();
    if (!c->hs && !c->blob_airs.count(air.id)) {
        // a context on its own (not a pool's: those reserve at warm-up) gets its two page-locked blobs with its first proof of an AIR,
        // the proof that also grows the work buffers -- hipHostMalloc waits for the device like the hipMallocs before it
        static const bool pinned = [] { const char* e = getenv("STARKHIP_PINNED_PROOFS"); return !(e && *e == '0'); }();
        if (pinned) (void)blob_arena_add(c, pl.total * 8, 2);  // failure: malloc serves the proof
        c->blob_airs.insert(air.id);
    }
    uint64_t* out = blob_alloc(pl.total * 8);  // page-locked if the context has reserved blobs (blob_arena.h)
    if (!out) return STARKHIP_ERR_OOM;
    {
        const size_t nq = cfg.num_query_rounds;
        std::vector<uint32_t> xs(nq);
        for (size_t q = 0; q < nq; q++) xs[q] = (uint32_t)(ch.get() % N);
        // every round's leaves and Merkle paths are gathered on the device, already in proof layout
        const size_t stride = pl.query_words;
        const unsigned d0 = log_N - cap_h;
        int err = 0;
#define HIPCHK_FREE(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { fprintf(stderr, "starkhip: HIP error %s (%s)\n", hipGetErrorString(_e), #expr); err = 1; } } while (0)
        HIPCHK_FREE(c->gather_t.ensure(nq * stride * 8));
        gl_t* dq = c->gather_t.as<gl_t>();
        HIPCHK_FREE(hipMemcpyAsync(c->qidx.p, xs.data(), nq * 4, hipMemcpyHostToDevice, st));
        const uint32_t* dxs = c->qidx.as<uint32_t>();
        size_t off = 0;
        if (!err) {
            HIPCHK_FREE(launch_query_leaf_colmajor(c->lde.as<gl_t>(), C, log_n, r, dxs, nq, dq, stride, off, st)); off += C;
            HIPCHK_FREE(launch_query_path(c->digests.as<gl_t>(), N, d0, dxs, 0, nq, dq, stride, off, st)); off += 4 * d0;
            HIPCHK_FREE(launch_query_leaf_colmajor(c->qlde.as<gl_t>(), Q, log_n, r, dxs, nq, dq, stride, off, st)); off += Q;
            HIPCHK_FREE(launch_query_path(c->qdigests.as<gl_t>(), N, d0, dxs, 0, nq, dq, stride, off, st)); off += 4 * d0;
            size_t len = N;
            unsigned shift_bits = 0;
            for (size_t l = 0; l < L; l++) {
                const unsigned ab = geo.arities[l];
                const size_t width = 2 << ab, n_leaves = len >> ab;
                shift_bits += ab;
                HIPCHK_FREE(launch_query_leaf_rows(c->fri_rows[l].as<gl_t>(), width, dxs, shift_bits, nq, dq, stride, off, st)); off += width;
                HIPCHK_FREE(launch_query_path(c->fri_digests[l].as<gl_t>(), n_leaves, (unsigned)pl.layer_depth[l], dxs, shift_bits, nq, dq, stride, off, st));
                off += 4 * pl.layer_depth[l];
                len = n_leaves;
            }
            HIPCHK_FREE(hipMemcpyAsync(out + pl.off_queries, dq, nq * stride * 8, hipMemcpyDeviceToHost, st));
        }
        pl.write_header(out);
        memcpy(out + pl.off_trace_cap, trace_cap.data(), 4 * ncap * 8);
        memcpy(out + pl.off_quot_cap, quot_cap.data(), 4 * ncap * 8);
        memcpy(out + pl.off_local, op_local.data(), C * 16);
        memcpy(out + pl.off_next, op_next.data(), C * 16);
        memcpy(out + pl.off_quot_open, op_q.data(), Q * 16);
        if (L) memcpy(out + pl.off_fri_caps, fri_caps.data(), L * 4 * ncap * 8);
        HIPCHK_FREE(stream_wait(c));
#undef HIPCHK_FREE
        if (err || off != stride) {
            blob_free(out);
            return STARKHIP_ERR_HIP;
        }
        memcpy(out + pl.off_final, final_poly.data(), geo.final_poly_len * 16);
        out[pl.off_pow] = pow_witness;
        if (n_pis) memcpy(out + pl.off_pis, pis_host, n_pis * 8);
    }
    if (hipEventRecord(c->ev[evi++], st) != hipSuccess || stream_wait(c) != hipSuccess) {
        blob_free(out);
        return STARKHIP_ERR_HIP;
    }
    for (int i = 0; i < STARKHIP_N_PHASES - 1; i++) (void)hipEventElapsedTime(&c->timings[i], c->ev[i], c->ev[i + 1]);
    if (c->opt_quotient_debug >= 1 && c->opt_quotient_debug <= 8) {  // profiling build only: timings are valid, the proof is not
        blob_free(out);
        (void)hipEventElapsedTime(&c->timings[STARKHIP_N_PHASES - 1], c->ev[0], c->ev[STARKHIP_N_PHASES - 1]);
        (void)hipEventElapsedTime(&c->ktimings[2], c->kev[2], c->kev[3]);
        return STARKHIP_ERR_VERIFY;
    }
    (void)hipEventElapsedTime(&c->timings[STARKHIP_N_PHASES - 1], c->ev[0], c->ev[STARKHIP_N_PHASES - 1]);
    c->htimings[0] = (float)fs.ms;
    c->htimings[1] = (float)host_other.ms;
    (void)hipEventElapsedTime(&c->ktimings[0], c->kev[4], c->kev[5]);
    // pooled: the commitment kernel's own duration on the scheduler's launch stream (kev[0] .. kev[1] on this context's stream would
    // include the wait for its group to form)
    if (c->hs && c->hash_timing.form != 4) (void)hipEventElapsedTime(&c->ktimings[1], c->hash_timing.t0, c->hash_timing.t1);
    else (void)hipEventElapsedTime(&c->ktimings[1], c->kev[0], c->kev[1]);
    (void)hipEventElapsedTime(&c->ktimings[2], c->kev[2], c->kev[3]);
    *proof_out = out;
    *proof_words = pl.total;
    return STARKHIP_OK;
}

// Everything a proof of `air` (default rows, config `cfg`) will ask of this context, allocated NOW: shape tables, the constraint
// plan, every work buffer, the page-locked staging of a recording of `log_bytes`.  A pool warms its contexts with this before the
// first job: growing a buffer later means hipFree + hipMalloc (or hipHostFree + hipHostMalloc), and those wait for EVERY stream of
// the device -- measured in a batch of 8 signatures: a PairingPrecomp proof with 212 ms of device time held its context for 2.3 s
// because its buffers grew while four FinalExp proofs kept the device busy.
int ctx_reserve(Ctx* c, const AirInfo& air, const starkhip_config_t& cfg, size_t log_bytes, unsigned proof_blobs, bool device_traces) {
    const AirProgram& P = air.prog;
    const size_t n = air.default_rows;
    unsigned log_n = 0;
    while (((size_t)1 << log_n) < n) log_n++;
    FriGeometry geo;
    if (!FriGeometry::make(cfg, log_n, &geo)) return STARKHIP_ERR_BAD_SHAPE;
    const unsigned r = cfg.rate_bits;
    const unsigned factor = P.degree > 1 ? P.degree - 1 : 1;
    unsigned qdb = 0;
    while ((1u << qdb) < factor) qdb++;
    if (qdb > r) return STARKHIP_ERR_BAD_SHAPE;
    const size_t N = n << r, C = P.n_cols, Q = (size_t)factor * 2, size = n << qdb, L = geo.arities.size();
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = ensure_tables(c, log_n, r, qdb))) return rc;
    if ((rc = ensure_plan(c, air, size))) return rc;
    const unsigned n_chunks = c->plan->chunks;
    const size_t comb_chunks = (C + 255) / 256;
    ProofLayout pl;
    pl.C = C; pl.Q = Q; pl.log_n = log_n; pl.rate_bits = r; pl.cap_h = cfg.cap_height; pl.L = L; pl.n_queries = cfg.num_query_rounds;
    pl.final_len = geo.final_poly_len; pl.n_pis = P.n_pis; pl.arity_bits = cfg.arity_bits; pl.n_challenges = 2;
    pl.compute();
    struct Want { DevBuf* b; size_t bytes; };
    // (`values`: a trace waits for its LDE inside the LDE buffer, prove() phase 0, but for the tail of run_lde_trace; rate_bits == 0 makes
    // prove() grow it to a whole trace on demand)
    const Want wants[] = {{&c->values, r >= 1 ? lde_tail_columns(C) * n * 8 : C * n * 8}, {&c->lde, std::max(C * N * 8, log_bytes + 64)}, {&c->digests, digest_words(N) * 8},
                          {&c->pis, std::max<size_t>(1, P.n_pis) * 8}, {&c->apow, 2 * (AIR_MAX_GROUP + 1) * 8}, {&c->chunk_scale, 2 * (size_t)n_chunks * 8},
                          {&c->partial, (size_t)n_chunks * 2 * size * 8}, {&c->qvals, 2 * size * 8}, {&c->qcoef, Q * n * 8}, {&c->qlde, Q * N * 8},
                          {&c->qdigests, digest_words(N) * 8}, {&c->zpow, 2 * n * 16}, {&c->gzpow, n * 16}, {&c->open_local, C * 16}, {&c->open_next, C * 16},
                          {&c->open_q, Q * 16}, {&c->ext_apow, (C + Q) * 16}, {&c->comb_partial, comb_chunks * n * 16}, {&c->comb_out, 2 * n * 16},
                          {&c->fri_coef, 2 * N * 8}, {&c->fri_vals, 2 * N * 8}, {&c->scale_tab, N * 8}, {&c->pow_state, 12 * 8}, {&c->pow_best, 8},
                          {&c->qidx, cfg.num_query_rounds * 4}, {&c->gather_t, cfg.num_query_rounds * pl.query_words * 8}};
    for (const Want& w : wants) HIPCHK(w.b->ensure(w.bytes));
    size_t len = N;
    for (size_t l = 0; l < L; l++) {
        const unsigned ab = geo.arities[l];
        HIPCHK(c->fri_rows[l].ensure(len * 2 * 8));
        HIPCHK(c->fri_digests[l].ensure(digest_words(len >> ab) * 8));
        len >>= ab;
    }
    if (log_bytes && !device_traces && c->host_staging_cap < log_bytes) {
        if (c->host_staging) (void)hipHostFree(c->host_staging);
        c->host_staging = nullptr;
        c->host_staging_cap = 0;
        HIPCHK(hipHostMalloc(&c->host_staging, log_bytes, hipHostMallocDefault));
        c->host_staging_cap = log_bytes;
    }
    if (proof_blobs && !c->blob_airs.count(air.id)) {  // page-locked blobs for this AIR's proofs, once per context
        if (blob_arena_add(c, pl.total * 8, proof_blobs) != 0) return STARKHIP_ERR_OOM;
        c->blob_airs.insert(air.id);
    }
    return stream_wait(c) == hipSuccess ? STARKHIP_OK : STARKHIP_ERR_HIP;
}

// ---------------------------------------------------------------- kernel-level entry points (tests)
int lde_batch(Ctx* c, const uint64_t* values, size_t n_cols, unsigned log_n, unsigned rate_bits, uint64_t* coeffs_out, uint64_t* lde_out) {
    if (log_n < 1 || log_n > 13) return STARKHIP_ERR_BAD_SHAPE;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = ensure_tables(c, log_n, rate_bits, 0))) return rc;
    const size_t n = (size_t)1 << log_n, N = n << rate_bits;
    HIPCHK(c->values.ensure(n_cols * n * 8));
    HIPCHK(c->lde.ensure(n_cols * N * 8));
    HIPCHK(hipMemcpyAsync(c->values.p, values, n_cols * n * 8, hipMemcpyHostToDevice, c->st));
    // the LDE comes from the kernel prove() uses for this shape: for 8192 rows the wave-resident one, which keeps no coefficients --
    // those, when asked for, come from the other kernel afterwards (in place of the values, which the first run leaves untouched)
    const bool wave = lde_wave_supported(log_n) && c->opt_lde_impl == 0;
    std::vector<gl_t> tmp;
    if (wave) {
        HIPCHK(run_lde(c, c->values.as<gl_t>(), nullptr, c->lde.as<gl_t>(), n_cols, log_n, rate_bits, 0));
        if (lde_out) {
            tmp.resize(n_cols * N);
            HIPCHK(hipMemcpyAsync(tmp.data(), c->lde.p, n_cols * N * 8, hipMemcpyDeviceToHost, c->st));
            HIPCHK(stream_wait(c));
        }
    }
    if (!wave || coeffs_out) HIPCHK(run_lde(c, c->values.as<gl_t>(), c->values.as<gl_t>(), c->lde.as<gl_t>(), n_cols, log_n, rate_bits, 0));
    if (coeffs_out) HIPCHK(hipMemcpyAsync(coeffs_out, c->values.p, n_cols * n * 8, hipMemcpyDeviceToHost, c->st));  // in place
    HIPCHK(stream_wait(c));
    if (lde_out) {
        // device layout is coset-major; hand back NATURAL point order i = k * R + s
        if (!wave) {
            tmp.resize(n_cols * N);
            HIPCHK(hipMemcpy(tmp.data(), c->lde.p, n_cols * N * 8, hipMemcpyDeviceToHost));
        }
        const size_t R = (size_t)1 << rate_bits;
        for (size_t col = 0; col < n_cols; col++)
            for (size_t s = 0; s < R; s++)
                for (size_t k = 0; k < n; k++) lde_out[col * N + k * R + s] = tmp[col * N + s * n + k];
    }
    return STARKHIP_OK;
}

// micro-benchmark entry: the trace LDE of `n_cols` synthetic columns (powers of a generator: no constant or unit column, so every
// column is transformed unless const_per_64 says otherwise), `reps` launches timed with HIP events on the context's stream; average milliseconds per launch
int lde_bench(Ctx* c, size_t n_cols, unsigned log_n, unsigned rate_bits, unsigned reps, unsigned const_per_64, const uint64_t* device_values, float* ms_out, float* each_ms) {
    if (log_n < 1 || log_n > 13 || !n_cols) return STARKHIP_ERR_BAD_SHAPE;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = ensure_tables(c, log_n, rate_bits, 0))) return rc;
    const size_t n = (size_t)1 << log_n, N = n << rate_bits;
    HIPCHK(c->values.ensure(n_cols * n * 8));
    HIPCHK(c->lde.ensure(n_cols * N * 8));
    const gl_t* in = device_values ? (const gl_t*)device_values : c->values.as<gl_t>();  // the caller's own column-major matrix, or the synthetic one
    if (device_values) const_per_64 = 0;
    else HIPCHK(launch_fill_powers(c->values.as<gl_t>(), 3, GL_GENERATOR, n_cols * n, c->st));
    // `const_per_64` of every 64 columns constant (a FinalExp trace: 11 of 64 take a closed form), in runs of up to 12 as its Fp12 blocks are
    // (+ 256: unit vectors instead -- one 1 per column, at a different row each -- the other closed form: FinalExp's 8192 row selectors)
    const bool unit = (const_per_64 & 256u) != 0, prewarm = (const_per_64 & 1024u) != 0, touch = (const_per_64 & 2048u) != 0;  // + 1024 / + 2048 (with reps == 0): see below
    const_per_64 &= 255u;
    for (size_t c0 = 0; const_per_64 && c0 < n_cols; c0 += 64) {
        const size_t cnt = std::min<size_t>(const_per_64, n_cols - c0);
        HIPCHK(hipMemsetAsync(c->values.as<gl_t>() + c0 * n, unit ? 0 : 1, cnt * n * 8, c->st));
        for (size_t k = 0; unit && k < cnt; k++) HIPCHK(hipMemsetAsync(c->values.as<gl_t>() + (c0 + k) * n + ((c0 + k) * 37) % n, 1, 1, c->st));
    }
    const bool cold = reps == 0;  // reps == 0: ONE launch with no warm-up launch in front of it
    if (cold) reps = 1;
    reps = std::min(reps, 16u);
    std::vector<hipEvent_t> ev(reps + 1, nullptr);
    hipError_t err = hipSuccess;
    for (hipEvent_t& e : ev)
        if (err == hipSuccess) err = hipEventCreate(&e);
    if (err == hipSuccess && cold && touch) err = hipMemsetAsync(c->lde.p, 0, n_cols * N * 8, c->st);  // every page of the output written once just before
    if (err == hipSuccess && cold && prewarm)  // a few milliseconds of the same arithmetic on a small footprint, then the launch that is timed
        for (int k = 0; k < 4 && err == hipSuccess; k++) err = run_lde(c, in, nullptr, c->lde.as<gl_t>(), std::min<size_t>(n_cols, 4096), log_n, rate_bits, 0);
    if (err == hipSuccess) err = cold ? hipStreamSynchronize(c->st) : run_lde(c, in, nullptr, c->lde.as<gl_t>(), n_cols, log_n, rate_bits, 0);  // warm-up
    if (err == hipSuccess) err = hipEventRecord(ev[0], c->st);
    for (unsigned r = 0; r < reps && err == hipSuccess; r++) {
        err = run_lde(c, in, nullptr, c->lde.as<gl_t>(), n_cols, log_n, rate_bits, 0);
        if (err == hipSuccess) err = hipEventRecord(ev[r + 1], c->st);
    }
    if (err == hipSuccess) err = hipEventSynchronize(ev[reps]);
    float ms = 0;
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, ev[0], ev[reps]);
    for (unsigned r = 0; r < reps && err == hipSuccess && each_ms; r++) err = hipEventElapsedTime(&each_ms[r], ev[r], ev[r + 1]);
    for (hipEvent_t e : ev)
        if (e) (void)hipEventDestroy(e);
    HIPCHK(err);
    *ms_out = ms / reps;
    return STARKHIP_OK;
}

// kernel-level test entry: a recorded trace through expand_trace_kernel + zero_cells_kernel, handed back column-major [C][rows]
int expand_log(Ctx* c, const TraceLog* log, uint64_t* out_colmajor) {
    HIPCHK(hipSetDevice(c->device));
    const size_t nw = log->total_words(), nr = log->total_records(), nz = log->total_late_zeros();
    std::vector<uint32_t> h(nw + nr + nz);
    size_t at_r = 0, at_z = 0;
    log->for_each_part([&](const TraceLog& part) {
        std::copy(part.words.begin(), part.words.end(), h.begin() + part.base);
        std::copy(part.offsets.begin(), part.offsets.end(), h.begin() + nw + at_r);
        std::copy(part.late_zeros.begin(), part.late_zeros.end(), h.begin() + nw + nr + at_z);
        at_r += part.offsets.size();
        at_z += part.late_zeros.size();
    });
    const size_t cells = log->rows * log->cols;
    HIPCHK(c->values.ensure(cells * 8));
    HIPCHK(c->staging.ensure(std::max<size_t>(h.size(), 1) * 4));
    uint32_t* d = c->staging.as<uint32_t>();
    if (!h.empty()) HIPCHK(hipMemcpyAsync(d, h.data(), h.size() * 4, hipMemcpyHostToDevice, c->st));
    HIPCHK(hipMemsetAsync(c->values.p, 0, cells * 8, c->st));
    if (nr) HIPCHK(launch_expand_trace(d, d + nw, nr, c->values.as<gl_t>(), log->rows, c->st));
    if (nz) HIPCHK(launch_zero_cells(d + nw + nr, nz / 2, c->values.as<gl_t>(), log->rows, c->st));
    HIPCHK(hipMemcpyAsync(out_colmajor, c->values.p, cells * 8, hipMemcpyDeviceToHost, c->st));
    HIPCHK(stream_wait(c));
    return STARKHIP_OK;
}

int merkle_cap(Ctx* c, const uint64_t* lde_natural, size_t n_cols, unsigned log_N, unsigned cap_h, uint64_t* cap_out) {
    if (log_N < cap_h) return STARKHIP_ERR_BAD_SHAPE;
    HIPCHK(hipSetDevice(c->device));
    const size_t N = (size_t)1 << log_N;
    // treat the input as rate_bits = 0 (coset-major == natural)
    HIPCHK(c->lde.ensure(n_cols * N * 8));
    HIPCHK(c->digests.ensure(digest_words(N) * 8));
    HIPCHK(hipMemcpyAsync(c->lde.p, lde_natural, n_cols * N * 8, hipMemcpyHostToDevice, c->st));
    if (c->opt_leaf_hash_form == 3) HIPCHK(launch_leaf_hash_lane(c->lde.as<gl_t>(), n_cols, log_N, 0, c->digests.as<gl_t>(), c->st));
    else if (use_pair_form(c, n_cols, log_N)) HIPCHK(launch_leaf_hash_pair(c->lde.as<gl_t>(), n_cols, log_N, 0, c->digests.as<gl_t>(), c->st));
    else if (use_row_form(c, n_cols, log_N)) HIPCHK(launch_leaf_hash_row(c->lde.as<gl_t>(), n_cols, log_N, 0, c->digests.as<gl_t>(), c->st));
    else HIPCHK(launch_leaf_hash(c->lde.as<gl_t>(), n_cols, log_N, 0, c->digests.as<gl_t>(), c->st));
    HIPCHK(launch_merkle_levels(c->digests.as<gl_t>(), log_N, cap_h, c->st));
    HIPCHK(hipMemcpyAsync(cap_out, c->digests.as<gl_t>() + 4 * level_off(N, log_N - cap_h), ((size_t)4 << cap_h) * 8, hipMemcpyDeviceToHost, c->st));
    HIPCHK(stream_wait(c));
    return STARKHIP_OK;
}

int permute_batch(Ctx* c, uint64_t* states, size_t n) {
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(c->staging.ensure(n * 12 * 8));
    HIPCHK(hipMemcpyAsync(c->staging.p, states, n * 96, hipMemcpyHostToDevice, c->st));
    HIPCHK(launch_permute_batch(c->staging.as<gl_t>(), n, c->st));
    HIPCHK(hipMemcpyAsync(states, c->staging.p, n * 96, hipMemcpyDeviceToHost, c->st));
    HIPCHK(stream_wait(c));
    return STARKHIP_OK;
}

int field_ops(Ctx* c, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(c->staging.ensure(3 * n * 8));
    gl_t* d = c->staging.as<gl_t>();
    HIPCHK(hipMemcpyAsync(d, a, n * 8, hipMemcpyHostToDevice, c->st));
    HIPCHK(hipMemcpyAsync(d + n, b, n * 8, hipMemcpyHostToDevice, c->st));
    HIPCHK(launch_field_ops(op, d, d + n, d + 2 * n, n, c->st));
    HIPCHK(hipMemcpyAsync(out, d + 2 * n, n * 8, hipMemcpyDeviceToHost, c->st));
    HIPCHK(stream_wait(c));
    return STARKHIP_OK;
}

int host_alloc(Ctx* c, size_t bytes, void** out) {
    if (bytes == 0) return STARKHIP_ERR_BAD_SHAPE;
    HIPCHK(hipSetDevice(c->device));
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        return STARKHIP_ERR_OOM;
    }
    HIPCHK(e);
    *out = p;
    return STARKHIP_OK;
}

void host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

}  // namespace starkhip
