// Recording one trace on several host threads (trace_log.h, starkhip_trace_set_threads).
// The reference's generators (e.g. /root/reference/src/final_exponentiate.rs:240-279, src/miller_loop.rs:87-146) fill their
// gadget blocks one after the other while carrying a running value; once those running values are known from the native
// chain (milliseconds), each block's rows depend only on its operands.  A generator therefore describes its work as tasks
// over disjoint cells; when it records and more than one thread is allowed, the tasks are filled into logs of their own and
// taken over in task order (nothing copied), so the result is the same for any thread count > 1 and any timing.
#include <pthread.h>
#include <atomic>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include <time.h>

#include "gadgets.h"

namespace starkhip {

// CPU time of the calling thread, and the process-wide tally of what recording costs on the helper threads below (they end with their
// recording, so nothing else can read their clocks afterwards): starkhip_host_cpu_seconds
uint64_t thread_cpu_ns() {
    timespec ts;
    if (clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts) != 0) return 0;
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
std::atomic<uint64_t> g_trace_worker_cpu_ns(0);

void fill_tasks(Trace& t, size_t n_tasks, const std::function<void(Trace&, size_t)>& fill) {
    const int threads = t.log ? trace_threads() : 1;
    if (threads <= 1 || n_tasks <= 1) {
        for (size_t k = 0; k < n_tasks; k++) fill(t, k);
        return;
    }
    std::vector<TraceLog> parts(n_tasks);
    std::atomic<size_t> next(0);
    std::mutex mu;
    std::string failure;
    auto worker = [&] {
        for (size_t k; (k = next.fetch_add(1)) < n_tasks;) {
            try {
                parts[k].reset(t.rows, t.cols);
                Trace part(nullptr, t.rows, t.cols, &parts[k]);
                fill(part, k);
            } catch (const std::exception& e) {
                std::lock_guard<std::mutex> g(mu);
                failure = e.what();
            }
        }
    };
    std::vector<std::thread> pool;
    auto helper = [&] {
        pthread_setname_np(pthread_self(), "starkhip-rec");
        worker();
        g_trace_worker_cpu_ns.fetch_add(thread_cpu_ns());  // a fresh thread: its clock started at zero
    };
    try {
        for (int w = 1; w < threads && (size_t)w < n_tasks; w++) pool.emplace_back(helper);
    } catch (const std::system_error&) {
        // no more threads to be had: the ones that started and this one share the tasks
    }
    worker();
    for (std::thread& th : pool) th.join();
    if (!failure.empty()) throw std::runtime_error(failure);
    for (TraceLog& part : parts) t.log->adopt(std::move(part));
}

}  // namespace starkhip
