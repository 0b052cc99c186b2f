// In-library scheduling of many proofs on one GPU (starkhip_pool_*, include/starkhip.h).
//
// The reference's caller issues its proves one after the other on one thread (/root/reference/src/aggregate_proof.rs:304-370)
// and leaves all parallelism to rayon inside prove().  On a GPU the proofs of one or many signature checks are what fills the
// chip, so the library itself keeps several in flight: a pool of prover contexts with one host thread each, generator threads
// that record traces, and ONE place that decides when a Merkle commitment -- the kernel that owns the chip -- is launched:
//
//  * HashService: every trace commitment of a pooled context goes through it.  The FinalExp-class commitment is a one-shot
//    grid of exactly two 188-register waves per SIMD (kernels_hash.hip): a foreign long-lived wave on a SIMD pushes one of
//    them into a second round and doubles the launch.  The 1024-row AIRs' commitments are the opposite: 128 .. 256 waves of
//    up to 12 167 sequential permutations, 7/8 of the chip idle.  So: commitments of the small class that arrive together are
//    launched as ONE merged grid (leaf_hash_multi_kernel, grid.y = proof); optionally (policy 1) the two classes never
//    overlap -- a big commitment then starts when the small window has drained and vice versa.
#pragma once
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "gl.h"

namespace starkhip {

unsigned cpu_budget();  // CPUs this process may really use (cgroup quota, affinity mask; scheduler.cpp)
hipError_t event_wait_sleeping(hipEvent_t ev);  // prover.hip: a few queries, then sleeps of 20 .. 200 microseconds between them

class HashService {
  public:
    explicit HashService(int device);
    ~HashService();
    HashService(const HashService&) = delete;
    HashService& operator=(const HashService&) = delete;

    // A pooled proof of the small class announces itself when it starts and calls hash() (or abandon()) exactly once: the
    // service holds a small window open while announced proofs have not arrived yet (bounded by `gather_ms`).
    void announce_small();
    void announce_big();   // a FinalExp-class proof has started: its commitment will come (lane-form groups wait for it)
    void abandon_big();
    void finish_big();     // that proof has ended
    void abandon_small();
    // Leaf digests of the coset-major LDE `mat` (kernels_hash.hip: launch_leaf_hash) into `digests`, ordered after everything
    // enqueued on `st` so far; when this returns, `st` has been made to wait for the launch (the caller goes on enqueueing).
    // `ready` / `done` are events owned by the caller's context.
    // `timing` (optional): two timing-enabled events of the caller's, recorded on the LAUNCH stream right before and after the kernel
    // that hashes this commitment (its own duration, not the wait for its group), and how it went out: form 0 = quad, 1 = row,
    // 2 = one grid merged with other proofs' commitments (quad form), 3 = lane; group = commitments launched side by side with it.
    struct Timing {
        hipEvent_t t0 = nullptr, t1 = nullptr;
        int form = 0;
        unsigned group = 1;
    };
    hipError_t hash(const gl_t* mat, size_t n_cols, unsigned log_n, unsigned rate_bits, gl_t* digests, hipStream_t st, hipEvent_t ready,
                    hipEvent_t done, bool announced, bool urgent = false, Timing* timing = nullptr);

    static bool is_big(unsigned log_n, unsigned rate_bits) { return log_n + rate_bits >= 15; }  // >= 2048 waves: fills every SIMD twice
    double gather_ms = 25.0;  // how long a small window waits for announced proofs that have not reached their commitment
    // 0 (default): small commitments that arrive together share a launch, and a big one starts whenever it arrives;
    // 1: in addition the two classes never overlap (a big commitment waits for the small window to drain and vice versa).
    // Measured on one MI355X: a batch of 8 signatures 3.5 against 3.2 signatures/s, one signature 0.42
    // against 0.47 s -- a lone wave on a SIMD runs about twice as fast as one of two, so a FinalExp commitment that starts
    // beside a MillerLoop latency chain loses less than it would by waiting for it.
    int policy = 0;
    unsigned BIG_LANE_GROUP = 4;  // STARKHIP_POOL_LANE_GROUP: commitments per lane-form group (four fill the chip: two waves of 256 registers per SIMD)
    double big_gather_ms_ = 1000.0; // lane form: how long a group of big commitments waits at most for proofs that HAVE STARTED to join (a group of three
                                    // wastes a quarter of a 350 ms launch: full groups measured 6.46 against 6.2 - 6.3 proofs/s with a 150 ms bound)
#ifndef STARKHIP_QUEUED_WAIT_MS
#define STARKHIP_QUEUED_WAIT_MS 300.0
#endif
    double big_queued_wait_ms_ = STARKHIP_QUEUED_WAIT_MS;  // ... and for jobs that have not started (queued, or being recorded) when nobody who has started is on the
                                         // way: a recording plus upload plus LDE -- what the soonest of them needs -- not the full bound (a short batch,
                                         // or the tail of one, would otherwise hold a commitment for several proof lengths)
    void set_big_queued(int n);     // the pool's count of big jobs that have not started yet (queued, or their trace being recorded)
    bool big_lane_ = false;  // big commitments in groups, a group of two or more in the lane form (pools with five or more big contexts;
                             // STARKHIP_POOL_BIG_LANE=0 / 1 overrides)
    int big_expected_ = 0;   // big proofs that have started and not yet reached their commitment
    int big_queued_ = 0;     // big jobs of the pool that have not started (under mu_)
    int big_contexts_ = 0;   // the pool's number of big contexts (set once): when all of them wait here, nobody else can come
    int big_active_ = 0;     // big proofs being proved (before, in or after their commitment)  // STARKHIP_POOL_BIG_LANE=1: big commitments in the lane form (one lane per leaf)
    size_t row_leaves_ = 64;  // STARKHIP_POOL_ROW_LEAVES: small commitments of at most this many leaves go out in the row form, one launch each (0: never)

    struct Stats {
        unsigned long big_launches = 0, small_launches = 0, small_requests = 0, max_merged = 0;
    };
    Stats stats();

  private:
    struct Req {
        const gl_t* mat;
        gl_t* digests;
        size_t n_cols;
        unsigned log_n, rate_bits;
        hipEvent_t ready, done;
        bool big, urgent = false;
        int state = 0;  // 0 queued, 1 launched, 2 failed
        int ready_state = 0;  // 0: the work that produces `mat` is still running (`ready` not reached), 1: it has run, 2: it failed (under mu_)
        hipError_t err = hipSuccess;
        double t_arrive = 0;
        Timing* timing = nullptr;
    };
    void run();
    double big_wait_bound() const;  // under mu_: how long the oldest queued big commitment waits at most for its group to fill
    hipError_t wait_ready(Req* r);  // the service thread, before it launches r's kernel: sleeps until r's caller has seen `ready`
    void launch_big(Req* r, bool lane, unsigned group);
    void launch_small(std::vector<Req*>& reqs);
    void drain(std::vector<hipEvent_t>& evs);
    void track(std::vector<hipEvent_t>& evs, hipEvent_t done);

    int device_;
    hipStream_t st_ = nullptr, st_high_ = nullptr;  // big commitments: ordinary / urgent (high-priority stream)
    // merged launches run side by side: each goes to a stream that is idle, so a window never queues behind an earlier one
    static const int N_SMALL_STREAMS = 12;
    hipStream_t small_st_[N_SMALL_STREAMS] = {};
    unsigned next_small_st_ = 0;
    hipStream_t pick_small_stream(hipError_t* err);
    std::mutex mu_;
    std::condition_variable cv_, cv_done_;
    std::deque<Req*> big_, small_;
    int announced_ = 0;  // small proofs that have started and not yet asked for their commitment
    bool stop_ = false, last_was_big_ = false;
    std::vector<hipEvent_t> running_big_, running_small_;  // done events of launches that may still be executing
    Stats stats_;
    std::thread th_;
};

}  // namespace starkhip
