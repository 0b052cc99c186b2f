// Proof pool (starkhip_pool_* of include/starkhip.h) and the commitment scheduler it owns (scheduler.h).
//
// What the reference's caller does on one thread -- generate_trace, prove, verify, six times per signature
// (/root/reference/src/aggregate_proof.rs:23-179, :304-370) -- becomes jobs of a pool: generator threads record compact
// traces (trace_log.h), one host thread per prover context proves them, and a caller only submits and waits.
#include "scheduler.h"

#include "blob_arena.h"

#include <pthread.h>
#include <sched.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <new>
#include <unordered_map>

#include "kernels.h"
#include "prover.h"
#include "trace_log.h"

namespace starkhip {

// CPUs this process may actually use: the cgroup's quota where there is one (a container that sees 256 hardware threads may be
// entitled to 16 of them -- threads beyond the quota are not slower, they are THROTTLED, kernel launches included), else the
// affinity mask.
unsigned cpu_budget() {
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) hw = std::max(1, CPU_COUNT(&set));
    double quota = 0, period = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
        char q[64];
        if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0) quota = atof(q);
        fclose(f);
    } else {
        FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
        FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (fq && fp && fscanf(fq, "%lf", &quota) == 1 && fscanf(fp, "%lf", &period) == 1 && quota <= 0) quota = 0;
        if (fq) fclose(fq);
        if (fp) fclose(fp);
    }
    if (quota > 0 && period > 0) hw = std::min(hw, std::max(1u, (unsigned)(quota / period + 0.5)));
    // one process per GPU (torch.distributed.run exports LOCAL_WORLD_SIZE): the node's CPUs are shared by that many pools
    if (const char* lws = getenv("LOCAL_WORLD_SIZE")) {
        const long ranks = atol(lws);
        if (ranks > 1) hw = std::max(1u, hw / (unsigned)ranks);
    }
    return hw;
}

uint64_t thread_cpu_ns();                                 // trace_tasks.cpp
extern std::atomic<uint64_t> g_trace_worker_cpu_ns;       // CPU time of the recordings' helper threads
extern std::atomic<uint64_t> g_wait_cpu_ns;               // prover.hip: CPU time inside the context threads' waits for the device
static std::atomic<uint64_t> g_gen_cpu_ns(0), g_prove_cpu_ns(0);  // ... of the generator threads inside a recording, of the context threads inside prove()
void host_cpu_seconds(double out[3]) {
    out[0] = (double)(g_gen_cpu_ns.load() + g_trace_worker_cpu_ns.load()) * 1e-9;
    out[1] = (double)g_prove_cpu_ns.load() * 1e-9;
    out[2] = (double)g_wait_cpu_ns.load() * 1e-9;
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ------------------------------------------------------------------------------------------------ HashService
HashService::HashService(int device) : device_(device) { th_ = std::thread([this] { run(); }); }

HashService::~HashService() {
    {
        std::lock_guard<std::mutex> g(mu_);
        stop_ = true;
    }
    cv_.notify_all();
    th_.join();
}

void HashService::announce_small() {
    std::lock_guard<std::mutex> g(mu_);
    announced_++;
}
void HashService::abandon_small() {
    {
        std::lock_guard<std::mutex> g(mu_);
        if (announced_ > 0) announced_--;
    }
    cv_.notify_all();
}

void HashService::announce_big() {
    std::lock_guard<std::mutex> g(mu_);
    big_expected_++;
    big_active_++;
}
void HashService::set_big_queued(int n) {
    {
        std::lock_guard<std::mutex> g(mu_);
        big_queued_ = n;
    }
    cv_.notify_all();
}
void HashService::finish_big() {
    std::lock_guard<std::mutex> g(mu_);
    if (big_active_ > 0) big_active_--;
}
void HashService::abandon_big() {
    {
        std::lock_guard<std::mutex> g(mu_);
        if (big_expected_ > 0) big_expected_--;
    }
    cv_.notify_all();
}

HashService::Stats HashService::stats() {
    std::lock_guard<std::mutex> g(mu_);
    return stats_;
}

hipError_t HashService::hash(const gl_t* mat, size_t n_cols, unsigned log_n, unsigned rate_bits, gl_t* digests, hipStream_t st, hipEvent_t ready,
                             hipEvent_t done, bool announced, bool urgent, Timing* timing) {
    hipError_t e = hipEventRecord(ready, st);
    Req r;
    r.timing = timing;
    r.mat = mat; r.digests = digests; r.n_cols = n_cols; r.log_n = log_n; r.rate_bits = rate_bits; r.ready = ready; r.done = done;
    r.big = is_big(log_n, rate_bits);
    r.urgent = urgent;
    r.t_arrive = now_s();
    std::unique_lock<std::mutex> lk(mu_);
    if (announced && announced_ > 0) announced_--;
    if (r.big && big_expected_ > 0) big_expected_--;
    if (e != hipSuccess) {
        lk.unlock();
        cv_.notify_all();
        return e;
    }
    (r.big ? big_ : small_).push_back(&r);
    cv_.notify_all();
    // Both hand-overs between the caller's stream and the service's go through the HOST: the caller sleeps until its own work has
    // reached `ready` and says so; the service launches then; the caller sleeps until `done` and goes on enqueueing.  Round 5 used
    // hipStreamWaitEvent both ways -- and a thread of the HIP runtime then polls for as long as a cross-stream dependence is pending:
    // 0.92 of a core during the benchmark, 0.12 of its 0.33 CPU-seconds per proof (tools/experiments/runtime_spin_probe.hip: 65 % of a
    // core with such waits, none without; no runtime setting changed it).  The request has JOINED its window already (groups form while
    // the LDEs still run); what the host round trips cost is a few hundred microseconds of idle stream per commitment.
    lk.unlock();
    const hipError_t er = event_wait_sleeping(ready);
    lk.lock();
    r.ready_state = er == hipSuccess ? 1 : 2;
    if (er != hipSuccess) r.err = er;
    cv_done_.notify_all();
    cv_done_.wait(lk, [&] { return r.state != 0; });
    lk.unlock();
    if (r.state == 2) return r.err;
    return event_wait_sleeping(done);
}

hipError_t HashService::wait_ready(Req* r) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] { return r->ready_state != 0; });
    return r->ready_state == 1 ? hipSuccess : r->err;
}

void HashService::drain(std::vector<hipEvent_t>& evs) {
    for (hipEvent_t ev : evs) (void)hipEventSynchronize(ev);
    evs.clear();
}

void HashService::launch_big(Req* r, bool lane, unsigned group) {
    hipStream_t s = (r->urgent && st_high_) ? st_high_ : st_;
    hipError_t e = hipSuccess;
    if (lane) s = pick_small_stream(&e);  // lane-form grids are a quarter of the chip each: they must overlap, not queue in one stream
    if (e == hipSuccess) e = wait_ready(r);
    if (r->timing) {
        r->timing->form = lane ? 3 : 5;
        r->timing->group = group;
        if (e == hipSuccess && r->timing->t0) e = hipEventRecord(r->timing->t0, s);
    }
    // a big commitment on its own: the pair form (is_big() = 32 768 leaves or more: 1 024 waves of it fill the chip)
    if (e == hipSuccess) e = lane ? launch_leaf_hash_lane(r->mat, r->n_cols, r->log_n, r->rate_bits, r->digests, s)
                                  : launch_leaf_hash_pair(r->mat, r->n_cols, r->log_n, r->rate_bits, r->digests, s);
    if (e == hipSuccess && r->timing && r->timing->t1) e = hipEventRecord(r->timing->t1, s);
    if (e == hipSuccess) e = hipEventRecord(r->done, s);
    r->err = e;
    if (e == hipSuccess) track(running_big_, r->done);
}

// Done events of launches that may still be executing.  Only policy 1 (exclusive classes) ever waits for them, so only policy 1
// keeps them; events whose launch has completed are dropped first (a context re-records its event with its next proof: a stale
// entry would make drain() wait for that later proof, and a pool that proves one class only would grow the list for ever).
void HashService::track(std::vector<hipEvent_t>& evs, hipEvent_t done) {
    if (policy != 1) return;
    size_t keep = 0;
    for (hipEvent_t ev : evs) {
        if (ev == done) continue;
        const bool finished = hipEventQuery(ev) == hipSuccess;
        (void)hipGetLastError();  // hipErrorNotReady is not an error
        if (!finished) evs[keep++] = ev;
    }
    evs.resize(keep);
    evs.push_back(done);
}

// A stream with nothing pending (a merged launch must not wait in stream order behind an earlier window's latency chain); all
// busy: the next one in turn.
hipStream_t HashService::pick_small_stream(hipError_t* err) {
    *err = hipSuccess;
    for (int k = 0; k < N_SMALL_STREAMS; k++) {
        hipStream_t& s = small_st_[(next_small_st_ + k) % N_SMALL_STREAMS];
        if (!s) {
            *err = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            next_small_st_ = (next_small_st_ + k + 1) % N_SMALL_STREAMS;
            return s;
        }
        const bool idle = hipStreamQuery(s) == hipSuccess;
        (void)hipGetLastError();  // hipErrorNotReady from a query is not an error (and must not surface at the next launch)
        if (idle) {
            next_small_st_ = (next_small_st_ + k + 1) % N_SMALL_STREAMS;
            return s;
        }
    }
    hipStream_t s = small_st_[next_small_st_];
    next_small_st_ = (next_small_st_ + 1) % N_SMALL_STREAMS;
    return s;
}

// all pending small commitments, merged by shape: one launch per (columns, rows, rate), each on a stream of its own so that
// different AIRs' windows overlap
void HashService::launch_small(std::vector<Req*>& reqs) {
    std::map<std::tuple<size_t, unsigned, unsigned>, std::vector<Req*>> groups;
    for (Req* r : reqs) groups[std::make_tuple(r->n_cols, r->log_n, r->rate_bits)].push_back(r);
    for (auto& kv : groups) {
        std::vector<Req*>& g = kv.second;
        for (size_t at = 0; at < g.size(); at += LEAF_HASH_MAX_BATCH) {
            const size_t cnt = std::min<size_t>(LEAF_HASH_MAX_BATCH, g.size() - at);
            hipError_t e = hipSuccess;
            hipStream_t s = pick_small_stream(&e);
            LeafHashBatch B;
            for (size_t i = 0; i < cnt && e == hipSuccess; i++) {
                B.mat[i] = g[at + i]->mat;
                B.digests[i] = g[at + i]->digests;
                e = wait_ready(g[at + i]);
            }
            const bool row_form = row_leaves_ && (((size_t)1 << (g[0]->log_n + g[0]->rate_bits)) <= row_leaves_);
            for (size_t i = 0; i < cnt; i++)
                if (Timing* t = g[at + i]->timing) {
                    t->form = row_form ? 1 : 2;
                    t->group = (unsigned)cnt;
                    if (e == hipSuccess && t->t0) e = hipEventRecord(t->t0, s);
                }
            if (e == hipSuccess && row_form) {
                // the row form (16 lanes per leaf): shortest chain per leaf at 2.8 x the chip time.  Measured with every small commitment
                // of a pool in it: one signature 0.36 -> 0.38 s, a batch of 8 3.8 -> 3.3 signatures/s; only the tiny ones take it by default
                for (size_t i = 0; i < cnt && e == hipSuccess; i++)
                    e = launch_leaf_hash_row(B.mat[i], g[0]->n_cols, g[0]->log_n, g[0]->rate_bits, B.digests[i], s);
            } else if (e == hipSuccess) {
                e = launch_leaf_hash_multi(B, (unsigned)cnt, g[0]->n_cols, g[0]->log_n, g[0]->rate_bits, s);
            }
            for (size_t i = 0; i < cnt; i++) {
                Req* r = g[at + i];
                if (e == hipSuccess && r->timing && r->timing->t1) e = hipEventRecord(r->timing->t1, s);
                if (e == hipSuccess) e = hipEventRecord(r->done, s);
                r->err = e;
                if (e == hipSuccess) track(running_small_, r->done);
            }
            std::lock_guard<std::mutex> lock(mu_);
            stats_.small_launches += row_form ? cnt : 1;
            stats_.max_merged = std::max<unsigned long>(stats_.max_merged, row_form ? 1 : cnt);
        }
    }
}

// Somebody who has STARTED is on the way to the commitment (its upload and LDE are tens of milliseconds): the full bound.  Only jobs that
// have not started could still join -- a recording under way, or every context busy with a proof past its commitment: the soonest of them
// needs a recording's end, an upload and an LDE, so waiting much longer than that for a fuller group costs more than the group gains.
double HashService::big_wait_bound() const { return big_expected_ > 0 ? big_gather_ms_ : std::min(big_gather_ms_, big_queued_wait_ms_); }

void HashService::run() {
    pthread_setname_np(pthread_self(), "starkhip-hash");  // thread names: bench.py attributes host CPU time by them
    (void)hipSetDevice(device_);
    int least = 0, greatest = 0;
    if (hipStreamCreateWithFlags(&st_, hipStreamNonBlocking) != hipSuccess) st_ = nullptr;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess ||
        hipStreamCreateWithPriority(&st_high_, hipStreamNonBlocking, greatest) != hipSuccess)
        st_high_ = nullptr;
    std::unique_lock<std::mutex> lk(mu_);
    while (true) {
        cv_.wait(lk, [&] { return stop_ || !big_.empty() || !small_.empty(); });
        if (stop_ && big_.empty() && small_.empty()) break;
        // small window: ready when every announced small proof has arrived, or the oldest request has waited long enough
        bool small_ready = false;
        if (!small_.empty()) {
            const double waited = (now_s() - small_.front()->t_arrive) * 1e3;
            small_ready = announced_ <= 0 || waited >= gather_ms || stop_;
        }
        // Lane form (one lane per leaf: 176 issue slots per permutation against the pair form's 205 and the quad form's 272, but 512 waves
        // of 256 registers per commitment -- a quarter of the chip): FOUR side by side run at the issue limit, 344 ms for four against
        // 118 ms each in the pair form.  Launched one by one as they arrive they leave the chip half empty and fall into step behind each
        // other, so they go out in GROUPS: a group waits (bounded) while big proofs that have started have not reached their commitment;
        // a commitment that ends up alone goes out in the pair form.
        bool big_ready = !big_.empty();
        if (big_lane_ && !big_.empty()) {
            const double waited = (now_s() - big_.front()->t_arrive) * 1e3;
            // A group goes out full.  Short of four it goes out when nobody else can join soon -- no big proof is on its way to its
            // commitment and none is waiting to start -- or when the oldest request has waited its bound; a lone proof is not held up.
            // (A request joins as soon as its proof has ENQUEUED its LDE: the early launches of a staggered group hash while the late
            // ones' LDEs still run, which measured better than holding the group until every LDE has run -- profiles/r04_ab_experiments.txt.)
            const bool all_here = big_contexts_ > 0 && (int)big_.size() >= big_contexts_;  // every context's proof is waiting in this queue
            big_ready = big_.size() >= BIG_LANE_GROUP || (big_expected_ <= 0 && big_queued_ <= 0) || all_here || waited >= big_wait_bound() || stop_;
        }
        const bool take_big = big_ready && (!small_ready || !last_was_big_);
        if (take_big) {
            std::vector<Req*> group;
            const size_t want = big_lane_ ? BIG_LANE_GROUP : 1u;
            while (!big_.empty() && group.size() < want) {  // in arrival order
                group.push_back(big_.front());
                big_.pop_front();
            }
            std::vector<hipEvent_t> wait_for;
            wait_for.swap(running_small_);
            lk.unlock();
            if (policy == 1) drain(wait_for);  // exclusive classes: the small window has left the chip
            for (Req* r : group) launch_big(r, big_lane_ && group.size() >= 2, (unsigned)group.size());
            lk.lock();
            for (Req* r : group) r->state = r->err == hipSuccess ? 1 : 2;
            stats_.big_launches += group.size();
            last_was_big_ = true;
            cv_done_.notify_all();
            continue;
        }
        if (small_ready) {
            std::vector<Req*> reqs(small_.begin(), small_.end());
            small_.clear();
            std::vector<hipEvent_t> wait_for;
            wait_for.swap(running_big_);
            lk.unlock();
            if (policy == 1) drain(wait_for);
            lk.lock();
            // whatever arrived while the big commitment drained joins the window
            reqs.insert(reqs.end(), small_.begin(), small_.end());
            small_.clear();
            lk.unlock();
            launch_small(reqs);
            lk.lock();
            for (Req* r : reqs) r->state = r->err == hipSuccess ? 1 : 2;
            stats_.small_requests += reqs.size();
            last_was_big_ = false;
            cv_done_.notify_all();
            continue;
        }
        // requests are pending but their window is still gathering: wake up when something arrives or its time is up
        double left_ms = 1e9;
        if (!small_.empty()) left_ms = std::min(left_ms, gather_ms - (now_s() - small_.front()->t_arrive) * 1e3);
        if (big_lane_ && !big_.empty()) left_ms = std::min(left_ms, big_wait_bound() - (now_s() - big_.front()->t_arrive) * 1e3);
        // (system_clock deadline = pthread_cond_timedwait: ThreadSanitizer of gcc 11 does not know pthread_cond_clockwait, which a
        // steady-clock wait_for uses, and then reports the mutex as still held)
        cv_.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds((long)(std::max(0.1, left_ms) * 1e3)));
    }
    lk.unlock();
    for (hipStream_t s : {st_, st_high_})
        if (s) {
            (void)hipStreamSynchronize(s);
            (void)hipStreamDestroy(s);
        }
    for (hipStream_t& s : small_st_)
        if (s) {
            (void)hipStreamSynchronize(s);
            (void)hipStreamDestroy(s);
        }
}

// ------------------------------------------------------------------------------------------------ pool
extern void set_thread_trace_threads(int n);  // capi.cpp: trace_threads() of the calling thread (0 = the process-wide setting)

// What one proof of each AIR costs a pool, for placing jobs on the pools of a multi-device handle (longest processing time first):
// milliseconds per proof with the pool FULL of that AIR on one MI355X (tools/air_pool_cost.py, profiles/r06_air_pool_cost.json) -- a job's
// share of its device's time, not its latency.  One signature's six proofs add up to 214 ms, which is what a batch takes per signature
// (4.9 signatures/s).  Rounds 1-5 used the reference's CPU seconds (92 : 12.5 : 4.5 : 0.22, README.md:36-39; ECCAgg a guess of 3): the same
// order, but FP12Mul -- 32 leaves hashed on the host and a 24 ms sponge over 60 285 columns -- weighs more here than its 16 rows suggest.
// The same table as the Python plan (parallel.AIR_COST).
double air_cost(int air) {
    switch (air) {
        case STARKHIP_AIR_FINAL_EXP: return 128.0;
        case STARKHIP_AIR_MILLER_LOOP: return 24.0;
        case STARKHIP_AIR_PAIRING_PRECOMP: return 10.8;
        case STARKHIP_AIR_ECC_AGGREGATE: return 12.6;
        case STARKHIP_AIR_FP12_MUL: return 15.4;
        default: return 0.01;
    }
}

namespace {

enum JobKind { JOB_DENSE, JOB_COMPACT, JOB_WITNESS, JOB_COLUMNS };

struct Job {
    uint64_t id = 0;
    int air = 0;
    starkhip_config_t cfg;
    JobKind kind = JOB_DENSE;
    const uint64_t* trace = nullptr;  // dense: caller's matrix; compact: a TraceLog*
    size_t n_rows = 0, n_cols = 0;
    int layout = 0, on_device = 0;
    const uint64_t* pis = nullptr;
    size_t n_pis = 0;
    uint64_t pow = 0;
    std::vector<const uint64_t*> columns;  // JOB_COLUMNS: the caller's column pointers (the table is copied at submit, the columns are not)
    std::vector<uint32_t> operands;   // witness jobs
    void* own_log = nullptr;          // witness jobs: the recording, freed when proven
    std::vector<uint64_t> own_pis, own_rows;  // own_rows: the toy AIR's generator writes plain rows (it does not record)
    bool big = false;
    double cost = 0;  // relative proving cost (air_cost): what the job adds to its pool's load until it is done
    // result
    int state = 0;  // 0 queued for generation / proving, 1 running, 2 done
    int rc = STARKHIP_OK;
    uint64_t* proof = nullptr;
    size_t words = 0;
    float phase_ms[STARKHIP_N_PHASES] = {0};
    float kernel_ms[3] = {0};
    float host_ms[2] = {0};
    int leaf_hash_form = 0;
    unsigned leaf_hash_group = 1;
    double t[5] = {0, 0, 0, 0, 0};  // submit, generation start / end, proof start / end (seconds since the pool was created)
};

int witness_limbs(int air) {
    switch (air) {
        case STARKHIP_AIR_FP12_MUL: return 288;
        case STARKHIP_AIR_FINAL_EXP: return 144;
        case STARKHIP_AIR_MILLER_LOOP: return 96;
        case STARKHIP_AIR_PAIRING_PRECOMP: return 72;
        case STARKHIP_AIR_ECC_AGGREGATE: return 512 * 24 + 512;
        case STARKHIP_AIR_TEST_FIBONACCI: return 4;
        default: return -1;
    }
}

// the ONE starkhip_trace_* call of `air` on packed operands (layouts: starkhip_pool_submit_witness in starkhip.h)
int run_generator(int air, const uint32_t* w, size_t n_rows, uint64_t* pis, uint64_t* rows) {
    switch (air) {
        case STARKHIP_AIR_FP12_MUL: return starkhip_trace_fp12_mul(w, w + 144, nullptr, n_rows, pis);
        case STARKHIP_AIR_FINAL_EXP: return starkhip_trace_final_exp(w, nullptr, n_rows, pis);
        case STARKHIP_AIR_MILLER_LOOP: return starkhip_trace_miller_loop(w, w + 12, w + 24, w + 48, w + 72, nullptr, n_rows, pis);
        case STARKHIP_AIR_PAIRING_PRECOMP: return starkhip_trace_pairing_precomp(w, w + 24, w + 48, nullptr, n_rows, pis);
        case STARKHIP_AIR_ECC_AGGREGATE: {
            std::vector<uint8_t> bits(512);
            for (int i = 0; i < 512; i++) bits[i] = (uint8_t)(w[512 * 24 + i] != 0);
            return starkhip_trace_ecc_aggregate(w, bits.data(), nullptr, n_rows, pis);
        }
        case STARKHIP_AIR_TEST_FIBONACCI:
            return starkhip_trace_fibonacci((uint64_t)w[0] | ((uint64_t)w[1] << 32), (uint64_t)w[2] | ((uint64_t)w[3] << 32), rows, n_rows, pis);
        default: return STARKHIP_ERR_BAD_AIR;
    }
}

}  // namespace

struct Pool {
    int device = 0;
    unsigned pools_on_device = 1;  // > 1: a multi-device handle was given this ordinal several times (starkhip_pool_host_info)
    double t0 = 0;
    std::unique_ptr<HashService> hs;
    std::vector<Ctx*> big_ctx, small_ctx;
    std::mutex mu;
    std::condition_variable cv_gen, cv_big, cv_small, cv_done;
    std::deque<Job*> q_gen, q_big, q_small;
    std::unordered_map<uint64_t, Job*> jobs;
    uint64_t next_id = 1;
    bool stop = false;
    unsigned gen_threads = 0, trace_threads_cfg = 0, gen_running = 0, cpus = 1;
    size_t big_recordings_started = 0, big_proofs_done = 0;  // under mu
    size_t big_in_gen = 0;  // FinalExp-class witness jobs queued for, or in, their recording (under mu)
    double load = 0;        // sum of air_cost over the jobs that are not done (under mu): what a multi-device handle balances
    unsigned big_open = 0;  // FinalExp-class jobs that are not done (under mu)
    unsigned waiters = 0;   // callers inside pool_wait (under mu): pool_destroy lets them leave before it frees anything
    std::map<int, int> idle_big, idle_small;                 // idle contexts by the AIR they proved last (under mu)
    unsigned stream_priority = 0;
    bool warm_device_traces = false;  // warm_up == 2: the caller's traces are column-major device memory: no trace buffers are reserved
    int gen_nice = 10;  // STARKHIP_GEN_NICE: nice value of the generator threads (0: as the rest of the process)
#ifndef STARKHIP_GEN_AHEAD
#define STARKHIP_GEN_AHEAD 1
#endif
    static constexpr size_t gen_ahead = STARKHIP_GEN_AHEAD;  // FinalExp-class recordings made beyond the ones the contexts can take at once
    bool warm = false;        // contexts reserve the pipeline's AIRs when their threads start (pool_create waits for it)
    unsigned warmed = 0;
    int warm_rc = STARKHIP_OK;
    std::vector<std::thread> threads;

    double now() const { return now_s() - t0; }

    void tell_big_queued() {  // under mu
        if (hs) hs->set_big_queued((int)(q_big.size() + big_in_gen));
    }

    void finish(Job* j, int rc) {
        std::lock_guard<std::mutex> g(mu);
        j->rc = rc;
        j->state = 2;
        j->t[4] = now();
        load = std::max(0.0, load - j->cost);
        if (j->big && big_open > 0) big_open--;
        cv_done.notify_all();
    }

    // Expected length of a small proof, for ordering only: the permutations per leaf of its commitment; trivial recordings first.
    static unsigned long small_rank(const Job* j) {
        const AirInfo* a = air_get(j->air);
        if (!a) return 0;
        return (unsigned long)a->cols + (a->default_rows <= 64 ? 1000000ul : 0ul);
    }

    // Threads one recording may use.  The long pole -- a FinalExp-class recording -- gets three quarters of the CPU budget (its
    // 53 tasks scale to 16 threads: 212 ms on one, 25 on 16), a small AIR's a quarter of it split over the small recordings
    // under way; the prover threads' Fiat-Shamir hashing and the natives need the rest.
    int trace_threads_for_call(bool big_job) {
        if (trace_threads_cfg) return (int)trace_threads_cfg;
        if (big_job) return (int)std::min(16u, std::max(1u, cpus * 3 / 4));
        return (int)std::min(4u, std::max(1u, cpus / 4));
    }

    void generator_loop() {
        pthread_setname_np(pthread_self(), "starkhip-gen");
        // Recording is the work that can wait: whenever the process is short of CPUs (16 per GPU on the measured boxes, and a batch
        // starts with four FinalExp recordings' worth of threads), the threads that feed the GPU -- the contexts' own: gathering a
        // recording for its upload, the challenger's hashing between two kernels -- must run first.  Per-thread nice value, inherited
        // by the recording's worker threads; measured on a batch of 8: the first FinalExp proofs' upload phase (the gather of a 153 MB recording) 90 - 127 -> 10 - 13 ms,
        // 3.88 -> 3.93 signatures/s over three alternating pairs.
        if (gen_nice > 0) (void)setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), gen_nice);
        while (true) {
            Job* j;
            int tt;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_gen.wait(lk, [&] { return stop || !q_gen.empty(); });
                if (q_gen.empty()) return;
                // Order by need, not by arrival.  FinalExp-class recordings are the long pole of a signature, so the first ones
                // go first -- as many as there are contexts to prove them, plus one in reserve -- then the small AIRs' (their
                // proofs fill the chip beside the first FinalExp proofs), then the FinalExp traces that will wait for a context
                // anyway.  Few recordings run at once, each on many threads (trace_threads_for_call): the FIRST trace of each
                // class is ready after tens of milliseconds instead of all of them after hundreds.
                // Among the small AIRs' the LONGEST proof first (a small proof is a latency chain of cols / 8 permutations per
                // leaf: MillerLoop 12 167, FP12Mul 7 536, PairingPrecomp 3 672), so that the batch does not end on one; recordings
                // that cost nothing (FP12Mul: 16 rows) before all others -- their proofs are 2-wave chains that start at once.
                auto it = q_gen.end();
                const bool want_big = big_recordings_started < big_ctx.size() + gen_ahead;
                for (auto k = q_gen.begin(); k != q_gen.end(); ++k)
                    if ((*k)->big == want_big && (it == q_gen.end() || (!want_big && small_rank(*k) > small_rank(*it)))) it = k;
                if (it == q_gen.end())  // none of the wanted class: the best of the other
                    for (auto k = q_gen.begin(); k != q_gen.end(); ++k)
                        if (it == q_gen.end() || (want_big && small_rank(*k) > small_rank(*it))) it = k;
                j = *it;
                q_gen.erase(it);
                if (j->big) big_recordings_started++;
                tt = trace_threads_for_call(j->big);
                gen_running++;
                j->t[1] = now();
            }
            int rc = STARKHIP_OK;
            const uint64_t cpu0 = thread_cpu_ns();
            try {
                set_thread_trace_threads(tt);
                const AirInfo* a = air_get(j->air);
                j->own_pis.assign(a->pis, 0);
                if (j->air == STARKHIP_AIR_TEST_FIBONACCI) {  // plain rows
                    j->own_rows.assign((size_t)a->default_rows * a->cols, 0);
                    rc = run_generator(j->air, j->operands.data(), a->default_rows, j->own_pis.data(), j->own_rows.data());
                } else {
                    rc = starkhip_trace_log_begin(&j->own_log);
                    if (rc == STARKHIP_OK) {
                        rc = run_generator(j->air, j->operands.data(), a->default_rows, j->own_pis.data(), nullptr);
                        const int rc_end = starkhip_trace_log_end(j->own_log);
                        if (rc == STARKHIP_OK) rc = rc_end;
                    }
                }
                set_thread_trace_threads(0);
            } catch (const std::bad_alloc&) {
                rc = STARKHIP_ERR_OOM;
            } catch (const std::exception&) {
                rc = STARKHIP_ERR_BAD_SHAPE;
            }
            g_gen_cpu_ns.fetch_add(thread_cpu_ns() - cpu0);
            if (rc != STARKHIP_OK) {
                if (j->own_log) starkhip_trace_log_free(j->own_log);
                j->own_log = nullptr;
                {
                    std::lock_guard<std::mutex> g(mu);
                    gen_running--;
                    if (j->big && big_in_gen > 0) big_in_gen--;
                    tell_big_queued();
                    j->t[2] = now();
                }
                finish(j, rc);
                cv_big.notify_all();  // contexts that are shutting down re-check whether a generator may still feed them
                cv_small.notify_all();
                continue;
            }
            std::lock_guard<std::mutex> g(mu);
            gen_running--;
            j->t[2] = now();
            if (j->own_log) {
                j->kind = JOB_COMPACT;
                j->trace = (const uint64_t*)j->own_log;
                j->n_rows = ((const TraceLog*)j->own_log)->rows;
            } else {
                j->kind = JOB_DENSE;
                j->trace = j->own_rows.data();
                j->n_rows = air_get(j->air)->default_rows;
                j->layout = 0;
                j->on_device = 0;
            }
            j->pis = j->own_pis.data();
            j->n_pis = j->own_pis.size();
            if (j->big && big_in_gen > 0) big_in_gen--;
            (j->big ? q_big : q_small).push_back(j);
            cv_big.notify_all();
            cv_small.notify_all();
            (void)0;  // (the count of big jobs that have not started is unchanged: from recording to queued)
        }
    }

    void prover_loop(Ctx* c, bool big) {
        pthread_setname_np(pthread_self(), big ? "starkhip-ctx" : "starkhip-ctxs");
        std::deque<Job*>& q = big ? q_big : q_small;
        std::condition_variable& cv = big ? cv_big : cv_small;
        std::map<int, int>& idle = big ? idle_big : idle_small;
        // A context keeps the tables, the constraint plan and -- above all -- device buffers sized for the AIRs it has proven
        // (growing them means hipFree + hipMalloc, and hipFree waits for every kernel on the device).  So a waiting job goes to
        // a context that proved its AIR last if one is idle; a context takes another AIR only when no idle one matches it.
        int last_air = -1;
        bool urgent = false, announce_big = false;
        if (warm) {  // every context brings up what the BLS pipeline's AIRs of its class need, all contexts in parallel
            // proof blobs: a context's last proof is usually still with the caller when the next one ends, hence two per big context;
            // the small contexts' MillerLoop-sized blob (69 MB) also serves FP12Mul (42 MB) -- blob_alloc takes the smallest that fits
            static const struct { int air; size_t log_bytes; unsigned blobs; } BIG[] = {{STARKHIP_AIR_FINAL_EXP, (size_t)200 << 20, 2}},
                SMALL[] = {{STARKHIP_AIR_MILLER_LOOP, (size_t)110 << 20, 1}, {STARKHIP_AIR_PAIRING_PRECOMP, (size_t)44 << 20, 1}, {STARKHIP_AIR_FP12_MUL, (size_t)2 << 20, 0}};
            const char* pe = getenv("STARKHIP_PINNED_PROOFS");
            const bool pinned = !(pe && *pe == '0');
            int rc = STARKHIP_OK;
            auto one = [&](int air, size_t log_bytes, unsigned blobs) {
                const AirInfo* a = air_get(air);
                starkhip_config_t cfg;
                if (!a || starkhip_config_for_air((starkhip_air_t)air, &cfg) != STARKHIP_OK) return;
                try {
                    const int r = ctx_reserve(c, *a, cfg, log_bytes, pinned ? blobs : 0, warm_device_traces);
                    if (r != STARKHIP_OK) rc = r;
                } catch (const std::exception&) {
                    rc = STARKHIP_ERR_OOM;
                }
            };
            if (big) for (const auto& w : BIG) one(w.air, w.log_bytes, w.blobs);
            else for (const auto& w : SMALL) one(w.air, w.log_bytes, w.blobs);
            std::lock_guard<std::mutex> g(mu);
            if (rc != STARKHIP_OK && warm_rc == STARKHIP_OK) warm_rc = rc;
            warmed++;
            cv_done.notify_all();
        }
        while (true) {
            Job* j = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                idle[last_air]++;
                while (true) {
                    if (!q.empty()) {
                        auto it = q.end();
                        for (auto k = q.begin(); k != q.end(); ++k)
                            if ((*k)->air == last_air) { it = k; break; }
                        if (it == q.end())
                            for (auto k = q.begin(); k != q.end(); ++k) {
                                auto f = idle.find((*k)->air);
                                if (f != idle.end() && f->second != 0) continue;  // somebody idle knows this AIR better
                                if (it == q.end() || (!big && small_rank(*k) > small_rank(*it))) it = k;  // the longest proof first
                                if (big) break;
                            }
                        if (it != q.end()) {
                            j = *it;
                            q.erase(it);
                            break;
                        }
                    } else if (stop && q_gen.empty() && gen_running == 0) {
                        // (a witness job still queued for, or in, its recording lands in q_big / q_small later: "runs what is queued
                        // to the end first" holds for those too, so a context leaves only when no generator can hand it anything)
                        idle[last_air]--;
                        return;
                    }
                    // shutting down: the queue may drain through other contexts without another notification
                    if (stop) cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(20));
                    else cv.wait(lk);
                }
                idle[last_air]--;
                last_air = j->air;
                j->state = 1;
                j->t[3] = now();
                // stream_priority 1: the LAST wave of FinalExp-class proofs -- no more of them waiting than there are contexts --
                // is the tail every other proof has finished before; it runs on high-priority streams.  The first wave does not:
                // strict priority starves the small proofs (a MillerLoop upload measured at 2 s behind four urgent FinalExp proofs)
                // (big jobs still queued for, or in, their recording count as waiting: with submit_witness they trickle into q_big
                // one at a time, and q_big alone would make the FIRST wave look like the last)
                urgent = big && stream_priority == 1 && q.size() + big_in_gen < big_ctx.size();
                // announced BEFORE it stops counting as queued: between the two a waiting lane group would see nobody on the way
                // and go out short, and this proof's commitment would follow it alone
                announce_big = big && ctx_has_hash_service(c);
                if (announce_big) hs->announce_big();
                if (big) tell_big_queued();
            }
            if (big && stream_priority == 1) (void)ctx_set_urgent(c, urgent);
            int rc;
            const bool announce = !big && ctx_has_hash_service(c);
            if (announce) hs->announce_small();
            ctx_hash_request_reset(c);
            const uint64_t cpu0 = thread_cpu_ns();
            try {
                const AirInfo* a = air_get(j->air);
                rc = prove(c, *a, j->cfg, j->trace, j->n_rows, j->kind == JOB_COMPACT ? 2 : j->kind == JOB_COLUMNS ? 3 : j->layout, j->on_device, j->pis,
                           j->n_pis, j->pow, &j->proof, &j->words);
            } catch (const std::bad_alloc&) {
                rc = STARKHIP_ERR_OOM;
            } catch (const std::exception&) {
                rc = STARKHIP_ERR_BAD_SHAPE;
            }
            g_prove_cpu_ns.fetch_add(thread_cpu_ns() - cpu0);
            if (announce && !ctx_hash_requested(c)) hs->abandon_small();  // failed before its commitment: do not hold the window open
            if (announce_big && !ctx_hash_requested(c)) hs->abandon_big();
            if (announce_big) hs->finish_big();
            if (big) {
                std::lock_guard<std::mutex> g(mu);
                if (big_recordings_started > 0) big_recordings_started--;  // a context is free again: the next FinalExp-class recording moves up
            }
            memcpy(j->phase_ms, ctx_timings(c), sizeof j->phase_ms);
            memcpy(j->kernel_ms, ctx_kernel_timings(c), sizeof j->kernel_ms);
            memcpy(j->host_ms, ctx_host_timings(c), sizeof j->host_ms);
            ctx_commit_info(c, &j->leaf_hash_form, &j->leaf_hash_group);
            if (j->own_log) {
                starkhip_trace_log_free(j->own_log);
                j->own_log = nullptr;
                j->trace = nullptr;
            }
            finish(j, rc);
        }
    }
};

int pool_create(const starkhip_pool_config_t& cfg_in, Pool** out, unsigned cpu_share) {
    starkhip_pool_config_t cfg = cfg_in;
    std::unique_ptr<Pool> p(new Pool());
    p->device = cfg.device;
    p->t0 = now_s();
    const unsigned n_big = cfg.big_contexts ? cfg.big_contexts : 3, n_small = cfg.small_contexts ? cfg.small_contexts : 16;
    // recording is host work the GPU waits for, but the CPU budget is shared with the prover threads (Fiat-Shamir hashing, kernel
    // launches): a quarter of the budget in recordings at once (at least 3), each on a few threads (trace_threads_for_call)
    p->cpus = std::max(1u, cpu_budget() / std::max(1u, cpu_share));  // cpu_share: pools of one multi-device handle share the process's CPUs
    // (the floor of three is capped by the budget itself: eight pools of a multi-device handle on sixteen CPUs plan with two each, and
    // three generator threads apiece would be 24 recording threads on those sixteen)
    p->gen_threads = cfg.generator_threads ? cfg.generator_threads : std::min(12u, std::max(std::min(3u, p->cpus), p->cpus / 4));
    p->trace_threads_cfg = cfg.trace_threads;
    int rc = STARKHIP_OK;
    for (unsigned i = 0; i < n_big + n_small && rc == STARKHIP_OK; i++) {
        Ctx* c = nullptr;
        const bool is_big = i < n_big;
        const int prio = cfg.stream_priority == 3 ? (is_big ? 1 : 0) : cfg.stream_priority == 2 ? (is_big ? 0 : 1) : 0;
        rc = ctx_create(cfg.device, &c, prio);
        if (rc == STARKHIP_OK) (i < n_big ? p->big_ctx : p->small_ctx).push_back(c);
    }
    if (rc != STARKHIP_OK) {
        for (Ctx* c : p->big_ctx) ctx_destroy(c);
        for (Ctx* c : p->small_ctx) ctx_destroy(c);
        return rc;
    }
    p->hs.reset(new HashService(cfg.device));
    p->stream_priority = cfg.stream_priority;
    bool big_lane = false;
    size_t row_leaves = 64;  // a commitment this small is a handful of waves in either form: the shorter chain costs nothing (FP12Mul: 32 leaves)
    {
        const char* n = getenv("STARKHIP_GEN_NICE");
        if (n && *n) p->gen_nice = atoi(n);
        const char* bl = getenv("STARKHIP_POOL_BIG_LANE");
        big_lane = (bl && *bl) ? *bl == '1' : p->big_ctx.size() >= 5;  // with four or fewer in flight the quad form is faster (5.65 against 4.05 proofs/s)
        const char* rl = getenv("STARKHIP_POOL_ROW_LEAVES");
        if (rl && *rl) row_leaves = (size_t)atol(rl);
    }
    p->warm = cfg.warm_up != 0;
    p->warm_device_traces = cfg.warm_up == 2;
    if (cfg.gather_ms > 0) p->hs->gather_ms = cfg.gather_ms;
    p->hs->policy = (int)cfg.commit_policy;
    p->hs->row_leaves_ = row_leaves;
    p->hs->big_lane_ = big_lane;
    p->hs->big_contexts_ = (int)p->big_ctx.size();
    {
        const char* lg = getenv("STARKHIP_POOL_LANE_GROUP");
        if (lg && *lg && atoi(lg) >= 2 && atoi(lg) <= 8) p->hs->BIG_LANE_GROUP = (unsigned)atoi(lg);
        const char* bg = getenv("STARKHIP_POOL_BIG_GATHER_MS");
        if (bg && *bg && atof(bg) > 0) p->hs->big_gather_ms_ = atof(bg);
    }
    if (cfg.commit_policy != 2) {  // 2: no commitment scheduling at all -- every context launches its own (A/B measurements)
        for (Ctx* c : p->big_ctx) ctx_attach_hash_service(c, p->hs.get());
        for (Ctx* c : p->small_ctx) ctx_attach_hash_service(c, p->hs.get());
    }
    Pool* raw = p.get();
    for (unsigned i = 0; i < p->gen_threads; i++) p->threads.emplace_back([raw] { raw->generator_loop(); });
    for (Ctx* c : p->big_ctx) p->threads.emplace_back([raw, c] { raw->prover_loop(c, true); });
    for (Ctx* c : p->small_ctx) p->threads.emplace_back([raw, c] { raw->prover_loop(c, false); });
    if (p->warm) {
        std::unique_lock<std::mutex> lk(p->mu);
        p->cv_done.wait(lk, [&] { return p->warmed == n_big + n_small; });
        const int wrc = p->warm_rc;
        lk.unlock();
        if (wrc != STARKHIP_OK) {
            pool_destroy(p.release());
            return wrc;
        }
    }
    *out = p.release();
    return STARKHIP_OK;
}

void pool_destroy(Pool* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(p->mu);
        p->stop = true;
    }
    p->cv_gen.notify_all();
    p->cv_big.notify_all();
    p->cv_small.notify_all();
    for (std::thread& t : p->threads) t.join();  // queued jobs are still run to completion: their callers may be waiting
    {   // every job is done now, so every caller blocked in pool_wait is on its way out: let them go before anything is freed
        std::unique_lock<std::mutex> lk(p->mu);
        p->cv_done.wait(lk, [&] { return p->waiters == 0; });
    }
    for (Ctx* c : p->big_ctx) ctx_destroy(c);
    for (Ctx* c : p->small_ctx) ctx_destroy(c);
    p->hs.reset();
    for (auto& kv : p->jobs) {
        blob_free(kv.second->proof);
        if (kv.second->own_log) starkhip_trace_log_free(kv.second->own_log);
        delete kv.second;
    }
    delete p;
}

static int pool_enqueue(Pool* p, Job* j, uint64_t* ticket) {
    const AirInfo* a = air_get(j->air);
    unsigned log_n = 0;
    const size_t rows = j->kind == JOB_WITNESS ? a->default_rows : j->n_rows;
    while (((size_t)1 << log_n) < rows) log_n++;
    j->big = HashService::is_big(log_n, j->cfg.rate_bits);
    std::lock_guard<std::mutex> g(p->mu);
    if (p->stop) {
        delete j;
        return STARKHIP_ERR_BAD_SHAPE;
    }
    j->id = p->next_id++;
    j->t[0] = p->now();
    j->cost = air_cost(j->air);
    p->load += j->cost;
    if (j->big) p->big_open++;
    p->jobs[j->id] = j;
    *ticket = j->id;
    if (j->kind == JOB_WITNESS) {
        if (j->big) p->big_in_gen++;
        p->q_gen.push_back(j);
        p->cv_gen.notify_one();
    } else {
        (j->big ? p->q_big : p->q_small).push_back(j);
        (j->big ? p->cv_big : p->cv_small).notify_all();
    }
    if (j->big) p->tell_big_queued();
    return STARKHIP_OK;
}

int pool_submit(Pool* p, int air, const starkhip_config_t* cfg, const uint64_t* trace, size_t n_rows, size_t n_cols, int layout, int on_device,
                const uint64_t* pis, size_t n_pis, uint64_t pow, uint64_t* ticket) {
    const AirInfo* a = air_get(air);
    if (!a) return STARKHIP_ERR_BAD_AIR;
    if (!cfg || !trace || !ticket || (n_pis && !pis) || (layout != 0 && layout != 1) || n_cols != a->cols) return STARKHIP_ERR_BAD_SHAPE;
    Job* j = new (std::nothrow) Job();
    if (!j) return STARKHIP_ERR_OOM;
    j->air = air; j->cfg = *cfg; j->kind = JOB_DENSE; j->trace = trace; j->n_rows = n_rows; j->n_cols = n_cols; j->layout = layout;
    j->on_device = on_device; j->pis = pis; j->n_pis = n_pis; j->pow = pow;
    return pool_enqueue(p, j, ticket);
}

int pool_submit_columns(Pool* p, int air, const starkhip_config_t* cfg, const uint64_t* const* columns, size_t n_rows, size_t n_cols,
                        const uint64_t* pis, size_t n_pis, uint64_t pow, uint64_t* ticket) {
    const AirInfo* a = air_get(air);
    if (!a) return STARKHIP_ERR_BAD_AIR;
    if (!cfg || !columns || !ticket || (n_pis && !pis) || n_cols != a->cols) return STARKHIP_ERR_BAD_SHAPE;
    for (size_t i = 0; i < n_cols; i++)
        if (!columns[i]) return STARKHIP_ERR_BAD_SHAPE;
    Job* j = new (std::nothrow) Job();
    if (!j) return STARKHIP_ERR_OOM;
    try {
        j->columns.assign(columns, columns + n_cols);
    } catch (const std::bad_alloc&) {
        delete j;
        return STARKHIP_ERR_OOM;
    }
    j->air = air; j->cfg = *cfg; j->kind = JOB_COLUMNS; j->trace = (const uint64_t*)j->columns.data(); j->n_rows = n_rows; j->n_cols = n_cols;
    j->pis = pis; j->n_pis = n_pis; j->pow = pow;
    return pool_enqueue(p, j, ticket);
}

int pool_submit_compact(Pool* p, int air, const starkhip_config_t* cfg, const void* log, const uint64_t* pis, size_t n_pis, uint64_t pow,
                        uint64_t* ticket) {
    const AirInfo* a = air_get(air);
    if (!a) return STARKHIP_ERR_BAD_AIR;
    const TraceLog* l = (const TraceLog*)log;
    if (!cfg || !l || !ticket || (n_pis && !pis) || l->cols != a->cols || !l->rows) return STARKHIP_ERR_BAD_SHAPE;
    Job* j = new (std::nothrow) Job();
    if (!j) return STARKHIP_ERR_OOM;
    j->air = air; j->cfg = *cfg; j->kind = JOB_COMPACT; j->trace = (const uint64_t*)l; j->n_rows = l->rows; j->n_cols = l->cols;
    j->pis = pis; j->n_pis = n_pis; j->pow = pow;
    return pool_enqueue(p, j, ticket);
}

int pool_submit_witness(Pool* p, int air, const starkhip_config_t* cfg, const uint32_t* operands, size_t n_limbs, uint64_t pow, uint64_t* ticket) {
    const AirInfo* a = air_get(air);
    if (!a) return STARKHIP_ERR_BAD_AIR;
    if (!operands || !ticket || witness_limbs(air) < 0 || (size_t)witness_limbs(air) != n_limbs) return STARKHIP_ERR_BAD_SHAPE;
    Job* j = new (std::nothrow) Job();
    if (!j) return STARKHIP_ERR_OOM;
    j->air = air;
    if (cfg) j->cfg = *cfg;
    else if (int rc = starkhip_config_for_air((starkhip_air_t)air, &j->cfg)) { delete j; return rc; }
    j->kind = JOB_WITNESS; j->pow = pow;
    j->operands.assign(operands, operands + n_limbs);
    return pool_enqueue(p, j, ticket);
}

int pool_wait(Pool* p, uint64_t ticket, uint64_t** proof, size_t* words, starkhip_ticket_info_t* info) {
    Job* j;
    {
        std::unique_lock<std::mutex> lk(p->mu);
        auto it = p->jobs.find(ticket);
        if (it == p->jobs.end()) return STARKHIP_ERR_BAD_SHAPE;
        j = it->second;
        p->waiters++;
        p->cv_done.wait(lk, [&] { return j->state == 2; });
        p->jobs.erase(ticket);
        p->waiters--;
        if (p->stop) p->cv_done.notify_all();
    }
    const int rc = j->rc;
    if (info) {
        memcpy(info->phase_ms, j->phase_ms, sizeof info->phase_ms);
        memcpy(info->kernel_ms, j->kernel_ms, sizeof info->kernel_ms);
        memcpy(info->host_ms, j->host_ms, sizeof info->host_ms);
        info->t_submit = j->t[0]; info->t_generate_start = j->t[1]; info->t_generate_end = j->t[2]; info->t_prove_start = j->t[3];
        info->t_done = j->t[4];
        info->leaf_hash_form = j->leaf_hash_form;
        info->leaf_hash_group = j->leaf_hash_group;
    }
    if (rc == STARKHIP_OK && proof && words) {
        *proof = j->proof;
        *words = j->words;
    } else {
        blob_free(j->proof);
        if (proof) *proof = nullptr;
        if (words) *words = 0;
    }
    delete j;
    return rc;
}

// What the pool holds: device memory of all its contexts, their page-locked staging; per FinalExp-class context for sizing.
// Read between proofs (the contexts grow their buffers only inside prove()).
int pool_reservation(Pool* p, starkhip_pool_reservation_t* out) {
    memset(out, 0, sizeof *out);
    for (Ctx* c : p->big_ctx) {
        out->device_bytes += ctx_device_bytes(c);
        out->pinned_host_bytes += ctx_pinned_bytes(c);
        out->big_context_device_bytes = std::max<uint64_t>(out->big_context_device_bytes, ctx_device_bytes(c));
    }
    for (Ctx* c : p->small_ctx) {
        out->device_bytes += ctx_device_bytes(c);
        out->pinned_host_bytes += ctx_pinned_bytes(c);
        out->small_context_device_bytes = std::max<uint64_t>(out->small_context_device_bytes, ctx_device_bytes(c));
    }
    out->big_contexts = (unsigned)p->big_ctx.size();
    out->small_contexts = (unsigned)p->small_ctx.size();
    return STARKHIP_OK;
}

int pool_host_info(Pool* p, starkhip_pool_host_info_t* out) {
    memset(out, 0, sizeof *out);
    out->cpu_budget = p->cpus;
    out->generator_threads = p->gen_threads;
    out->trace_threads_big = (unsigned)p->trace_threads_for_call(true);
    out->trace_threads_small = (unsigned)p->trace_threads_for_call(false);
    out->prover_threads = (unsigned)(p->big_ctx.size() + p->small_ctx.size());
    out->device = p->device;
    out->pools_on_device = p->pools_on_device;
    return STARKHIP_OK;
}

int pool_stats(Pool* p, starkhip_pool_stats_t* out) {
    const HashService::Stats s = p->hs->stats();
    out->big_commit_launches = s.big_launches;
    out->small_commit_launches = s.small_launches;
    out->small_commit_requests = s.small_requests;
    out->max_merged_commitments = s.max_merged;
    return STARKHIP_OK;
}

// ------------------------------------------------------------------------------------------------ many devices, one caller
// The reference's caller is ONE process that issues its proves from one thread (/root/reference/src/aggregate_proof.rs:304-370,
// :402-414).  For that caller to use a node of GPUs it needs no process group and no collective -- the proofs are independent
//  -- only a pool per device and a rule that says which pool a job goes to.  The rule is the Python plan's
// (signature.plan_batch): longest processing time first -- a job goes to the pool with the least outstanding cost (air_cost), a batch is
// placed in order of decreasing cost, so every device gets whole FinalExp proofs first and the small proofs fill the gaps.
struct MultiPool {
    std::vector<Pool*> pools;
    std::vector<int> devices;
    std::mutex mu;  // placement + enqueue are one step: two submitting threads see each other's jobs
};

static const unsigned TICKET_SLOT_SHIFT = 48;  // ticket of a multi-device handle = (slot + 1) << 48 | the pool's own ticket

int multipool_create(const int* devices, size_t n, const starkhip_pool_config_t& cfg, MultiPool** out) {
    if (!devices || n == 0 || n > 64) return STARKHIP_ERR_BAD_SHAPE;
    std::unique_ptr<MultiPool> mp(new MultiPool());
    // The pools come up side by side (a warmed FinalExp pool allocates 160 GB and builds its plans: seconds per device)
    std::vector<Pool*> made(n, nullptr);
    std::vector<int> rcs(n, STARKHIP_OK);
    std::vector<std::thread> th;
    for (size_t i = 0; i < n; i++)
        th.emplace_back([&, i] {
            starkhip_pool_config_t c = cfg;
            c.device = devices[i];
            try {
                rcs[i] = pool_create(c, &made[i], (unsigned)n);
            } catch (const std::bad_alloc&) {
                rcs[i] = STARKHIP_ERR_OOM;
            } catch (const std::exception&) {
                rcs[i] = STARKHIP_ERR_HIP;
            }
        });
    for (std::thread& t : th) t.join();
    int rc = STARKHIP_OK;
    for (size_t i = 0; i < n; i++)
        if (rcs[i] != STARKHIP_OK && rc == STARKHIP_OK) rc = rcs[i];
    if (rc != STARKHIP_OK) {
        for (Pool* p : made)
            if (p) pool_destroy(p);
        return rc;
    }
    mp->pools = made;
    mp->devices.assign(devices, devices + n);
    // one ordinal given several times: a rehearsal of the multi-device control flow on one card.  Legitimate (the tests do it), but its
    // figures must never pass for N devices: every pool reports how many share its device, and the process says so once.
    bool shared = false;
    for (size_t i = 0; i < n; i++) {
        unsigned same = 0;
        for (size_t k = 0; k < n; k++) same += devices[k] == devices[i];
        made[i]->pools_on_device = same;
        shared = shared || same > 1;
    }
    static std::atomic<bool> said(false);
    if (shared && !said.exchange(true))
        fprintf(stderr, "starkhip: starkhip_multipool_create was given the same device ordinal more than once -- the pools share that GPU "
                        "(a rehearsal, not a measurement of %zu devices)\n", n);
    *out = mp.release();
    return STARKHIP_OK;
}

void multipool_destroy(MultiPool* mp) {
    if (!mp) return;
    std::vector<std::thread> th;  // every pool runs what it has queued to the end: side by side
    for (Pool* p : mp->pools) th.emplace_back([p] { pool_destroy(p); });
    for (std::thread& t : th) t.join();
    delete mp;
}

size_t multipool_size(const MultiPool* mp) { return mp->pools.size(); }
Pool* multipool_pool(MultiPool* mp, size_t slot) { return slot < mp->pools.size() ? mp->pools[slot] : nullptr; }
int multipool_device(const MultiPool* mp, size_t slot) { return slot < mp->devices.size() ? mp->devices[slot] : -1; }

// the pool a job of `air` goes to (under mp->mu): for a FinalExp-class job the pool with the fewest of them open, then -- and for every
// other job -- the least outstanding cost, then the lowest slot
static size_t multipool_pick(MultiPool* mp, int air) {
    const AirInfo* a = air_get(air);
    starkhip_config_t cfg;
    bool big = false;
    if (a && starkhip_config_for_air((starkhip_air_t)air, &cfg) == STARKHIP_OK) {
        unsigned log_n = 0;
        while (((size_t)1 << log_n) < (size_t)a->default_rows) log_n++;
        big = HashService::is_big(log_n, cfg.rate_bits);
    }
    size_t best = 0;
    double best_load = 0;
    unsigned best_big = 0;
    for (size_t i = 0; i < mp->pools.size(); i++) {
        Pool* p = mp->pools[i];
        double load;
        unsigned open;
        {
            std::lock_guard<std::mutex> g(p->mu);
            load = p->load;
            open = p->big_open;
        }
        const bool better = i == 0 || (big && open != best_big ? open < best_big : load < best_load);
        if (better) {
            best = i;
            best_load = load;
            best_big = open;
        }
    }
    return best;
}

template <class Submit>
static int multipool_place(MultiPool* mp, int air, int slot, uint64_t* ticket, Submit submit) {
    if (!ticket || slot >= (int)mp->pools.size()) return STARKHIP_ERR_BAD_SHAPE;
    std::lock_guard<std::mutex> g(mp->mu);
    const size_t at = slot >= 0 ? (size_t)slot : multipool_pick(mp, air);
    uint64_t inner = 0;
    const int rc = submit(mp->pools[at], &inner);
    if (rc == STARKHIP_OK) *ticket = ((uint64_t)(at + 1) << TICKET_SLOT_SHIFT) | inner;
    return rc;
}

int multipool_submit(MultiPool* mp, int slot, int air, const starkhip_config_t* cfg, const uint64_t* trace, size_t n_rows, size_t n_cols, int layout,
                     int on_device, const uint64_t* pis, size_t n_pis, uint64_t pow, uint64_t* ticket) {
    if (on_device && slot < 0) return STARKHIP_ERR_BAD_SHAPE;  // device memory belongs to one device: the caller says which
    return multipool_place(mp, air, slot, ticket,
                           [&](Pool* p, uint64_t* t) { return pool_submit(p, air, cfg, trace, n_rows, n_cols, layout, on_device, pis, n_pis, pow, t); });
}
int multipool_submit_columns(MultiPool* mp, int slot, int air, const starkhip_config_t* cfg, const uint64_t* const* columns, size_t n_rows, size_t n_cols,
                             const uint64_t* pis, size_t n_pis, uint64_t pow, uint64_t* ticket) {
    return multipool_place(mp, air, slot, ticket,
                           [&](Pool* p, uint64_t* t) { return pool_submit_columns(p, air, cfg, columns, n_rows, n_cols, pis, n_pis, pow, t); });
}
int multipool_submit_compact(MultiPool* mp, int slot, int air, const starkhip_config_t* cfg, const void* log, const uint64_t* pis, size_t n_pis,
                             uint64_t pow, uint64_t* ticket) {
    return multipool_place(mp, air, slot, ticket, [&](Pool* p, uint64_t* t) { return pool_submit_compact(p, air, cfg, log, pis, n_pis, pow, t); });
}
int multipool_submit_witness(MultiPool* mp, int slot, int air, const starkhip_config_t* cfg, const uint32_t* operands, size_t n_limbs, uint64_t pow,
                             uint64_t* ticket) {
    return multipool_place(mp, air, slot, ticket, [&](Pool* p, uint64_t* t) { return pool_submit_witness(p, air, cfg, operands, n_limbs, pow, t); });
}

// A whole batch of witness jobs, placed longest first (ties in the caller's order).  All or nothing is not promised: tickets[i] == 0 and
// rcs[i] != OK for a job that was refused; the return value is the first failure.
int multipool_submit_witness_batch(MultiPool* mp, size_t n, const int* airs, const uint32_t* const* operands, const size_t* n_limbs, uint64_t pow,
                                   uint64_t* tickets, int* rcs) {
    if (!airs || !operands || !n_limbs || !tickets) return STARKHIP_ERR_BAD_SHAPE;
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return air_cost(airs[a]) > air_cost(airs[b]); });
    int first = STARKHIP_OK;
    for (size_t i : order) {
        tickets[i] = 0;
        const int rc = multipool_submit_witness(mp, -1, airs[i], nullptr, operands[i], n_limbs[i], pow, &tickets[i]);
        if (rcs) rcs[i] = rc;
        if (rc != STARKHIP_OK && first == STARKHIP_OK) first = rc;
    }
    return first;
}

int multipool_ticket_slot(const MultiPool* mp, uint64_t ticket) {
    const uint64_t s = ticket >> TICKET_SLOT_SHIFT;
    return (s >= 1 && s <= mp->pools.size()) ? (int)(s - 1) : -1;
}

int multipool_wait(MultiPool* mp, uint64_t ticket, uint64_t** proof, size_t* words, starkhip_ticket_info_t* info) {
    const int slot = multipool_ticket_slot(mp, ticket);
    if (slot < 0) return STARKHIP_ERR_BAD_SHAPE;
    return pool_wait(mp->pools[(size_t)slot], ticket & (((uint64_t)1 << TICKET_SLOT_SHIFT) - 1), proof, words, info);
}

// the plan alone, for tests and for callers that want to see it: slot per job of a batch placed on `n_pools` idle pools
void plan_lpt(size_t n, const int* airs, size_t n_pools, int* slots) {
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return air_cost(airs[a]) > air_cost(airs[b]); });
    std::vector<double> load(n_pools, 0.0);
    for (size_t i : order) {
        size_t best = 0;
        for (size_t k = 1; k < n_pools; k++)
            if (load[k] < load[best]) best = k;
        slots[i] = (int)best;
        load[best] += air_cost(airs[i]);
    }
}

}  // namespace starkhip
