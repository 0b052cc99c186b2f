// Compact trace: instead of filling n_rows x C cells on the host and moving them over PCIe
// (FinalExp: 4.8 GB), the trace generators can RECORD their writes.  The gadget fillers write limb vectors
// (`put(row, col, limbs)`), and most of those are the same vector on consecutive rows -- a 12-row gadget block repeats
// its inputs on every row, FinalExp repeats every intermediate Fp12 on all 8192 rows: 95 % of the cells equal the cell
// above.  A record is one limb vector and the run of rows it occupies:
//     words: col, first_row, n_rows_in_run, n, v[0..n)          (uint32; every cell of these AIRs is a u32 limb or bit)
// A put that repeats the previous row's vector at the same column extends that record instead of adding one.  Zeros at
// either end of a vector are dropped (the expanded matrix starts zeroed); no generator writes a cell twice with
// different non-zero values, and the one place a cell is set and then cleared is handled in set()
// (tests/test_trace_log_cpu.py replays every AIR's log and checks all of this against the dense trace).
// The device expands the records straight into the column-major matrix the prover wants (kernels_trace.hip).
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <stdexcept>
#include <vector>

namespace starkhip {

// The word vectors of finished logs, kept for the next recording.  A FinalExp recording writes 142 MB of records into vectors that grow
// by doubling -- 100 000 page faults and as many pages unmapped again when the log is freed: 0.25 s of kernel time beside 0.41 s of
// recording (tools/experiments/recording_cpu_probe.cpp), every time, because the allocator hands blocks of this size straight back to the system.  A log
// returns its vectors here instead (capacity kept, at most TRACE_LOG_POOL_BYTES in all) and a new log takes the largest one.
struct TraceLogWordPool {
    static constexpr size_t TRACE_LOG_POOL_BYTES = (size_t)3 << 30;
    enum Kind { WORDS = 0, OFFSETS = 1, OPEN = 2 };  // a list per use: the three differ by an order of magnitude in size
    std::mutex mu;
    std::vector<std::vector<uint32_t>> free_list[3];
    size_t bytes = 0;
    static TraceLogWordPool& instance() {
        static TraceLogWordPool* p = new TraceLogWordPool();  // never destroyed: logs may be freed during process exit
        return *p;
    }
    std::vector<uint32_t> take(Kind k) {
        std::lock_guard<std::mutex> g(mu);
        std::vector<std::vector<uint32_t>>& fl = free_list[k];
        if (fl.empty()) return {};
        size_t best = 0;
        for (size_t i = 1; i < fl.size(); i++)
            if (fl[i].capacity() > fl[best].capacity()) best = i;
        std::vector<uint32_t> v = std::move(fl[best]);
        fl[best] = std::move(fl.back());
        fl.pop_back();
        bytes -= v.capacity() * sizeof(uint32_t);
        v.clear();
        return v;
    }
    void give(Kind k, std::vector<uint32_t>&& v) {
        const size_t b = v.capacity() * sizeof(uint32_t);
        if (b < ((size_t)64 << 10)) return;  // small blocks stay with the allocator
        std::lock_guard<std::mutex> g(mu);
        if (bytes + b > TRACE_LOG_POOL_BYTES || free_list[k].size() >= 4096) return;  // over the limit: freed as before
        bytes += b;
        free_list[k].push_back(std::move(v));
    }
};

struct TraceLog {
    size_t rows, cols;
    std::vector<uint32_t> words;    // the records, back to back
    std::vector<uint32_t> offsets;  // start of each record in `words`
    std::vector<uint32_t> open;     // per column: offset + 1 of the latest record that starts there (0 = none)
    std::vector<uint32_t> late_zeros;  // (col, row) pairs zeroed after everything else (see set())

    TraceLog() : rows(0), cols(0) {}
    TraceLog(TraceLog&&) = default;
    TraceLog& operator=(TraceLog&&) = default;
    TraceLog(const TraceLog&) = default;
    TraceLog& operator=(const TraceLog&) = default;
    ~TraceLog() {
        TraceLogWordPool& pool = TraceLogWordPool::instance();
        pool.give(TraceLogWordPool::WORDS, std::move(words));
        pool.give(TraceLogWordPool::OFFSETS, std::move(offsets));
        pool.give(TraceLogWordPool::OPEN, std::move(open));
    }
    void reset(size_t r, size_t c) {
        rows = r;
        cols = c;
        TraceLogWordPool& pool = TraceLogWordPool::instance();
        if (words.capacity() == 0) words = pool.take(TraceLogWordPool::WORDS);
        if (offsets.capacity() == 0) offsets = pool.take(TraceLogWordPool::OFFSETS);
        if (open.capacity() == 0) open = pool.take(TraceLogWordPool::OPEN);
        words.clear();
        offsets.clear();
        late_zeros.clear();
        parts.clear();
        base = 0;
        open.assign(c, 0);
    }

    void put(size_t row, size_t col, const uint32_t* v, size_t n) { put_rows(row, 1, col, v, n); }
    // the same vector on rows row .. row + n_rows - 1: exactly the record n_rows consecutive put()s leave (a generator that knows a
    // block is constant over the rows -- FinalExp's input and its 32 intermediate Fp12, on all 8192 rows -- says so in one call)
    void put_rows(size_t row, size_t n_rows, size_t col, const uint32_t* v, size_t n) {
        while (n && v[n - 1] == 0) n--;
        while (n && v[0] == 0) { v++; col++; n--; }
        if (!n || !n_rows) return;
        if (row + n_rows > rows || col + n > cols) throw std::runtime_error("trace_log: write outside the trace");
        const uint32_t o = open[col];
        if (o) {
            uint32_t* r = &words[o - 1];
            if (r[3] == n && r[1] + r[2] == row && memcmp(r + 4, v, n * sizeof(uint32_t)) == 0) {
                r[2] += (uint32_t)n_rows;
                return;
            }
        }
        if (words.size() + 4 + n > 0xFFFFFFF0u) throw std::runtime_error("trace_log: log too large");
        offsets.push_back((uint32_t)words.size());
        open[col] = (uint32_t)words.size() + 1;
        words.push_back((uint32_t)col);
        words.push_back((uint32_t)row);
        words.push_back((uint32_t)n_rows);
        words.push_back((uint32_t)n);
        words.insert(words.end(), v, v + n);
    }
    // Logs of the same trace recorded separately (other threads filling other gadgets), taken over whole: nothing is
    // copied, a part keeps its own `words` and its `offsets` are shifted by `base`, its position in the concatenation
    // (this log's own words first, then the parts in order).  Runs are not joined across parts; the expansion does not
    // depend on the order of records.
    std::vector<TraceLog> parts;
    uint32_t base = 0;
    size_t total_words() const {
        size_t n = words.size();
        for (const TraceLog& p : parts) n += p.words.size();
        return n;
    }
    size_t total_records() const {
        size_t n = offsets.size();
        for (const TraceLog& p : parts) n += p.offsets.size();
        return n;
    }
    size_t total_late_zeros() const {
        size_t n = late_zeros.size();
        for (const TraceLog& p : parts) n += p.late_zeros.size();
        return n;
    }
    template <class F>
    void for_each_part(F f) const {  // f(const TraceLog&): this log's own records, then every part's
        f(*this);
        for (const TraceLog& p : parts) f(p);
    }
    void adopt(TraceLog&& part) {
        if (part.rows != rows || part.cols != cols || !part.parts.empty()) throw std::runtime_error("trace_log: adopt of a different trace");
        const size_t at = total_words();
        if (at + part.words.size() > 0xFFFFFFF0u) throw std::runtime_error("trace_log: log too large");
        part.base = (uint32_t)at;
        for (uint32_t& o : part.offsets) o += part.base;
        TraceLogWordPool::instance().give(TraceLogWordPool::OPEN, std::move(part.open));
        part.open = std::vector<uint32_t>();
        parts.push_back(std::move(part));
        std::fill(open.begin(), open.end(), 0);  // no run of this log may be extended past records that came later
    }
    void set(size_t row, size_t col, uint64_t v) {
        if (v >> 32) throw std::runtime_error("trace_log: cell value does not fit 32 bits");
        if (row >= rows || col >= cols) throw std::runtime_error("trace_log: write outside the trace");
        if (v == 0) {
            // the one overwriting idiom of the fillers: "selector = 1 on rows a..b", then "selector(b) = 0".  Take the
            // row back from the run that just wrote it; a clear inside the latest run is kept as a late zero, applied
            // after the expansion.
            const uint32_t o = open[col];
            if (o) {
                uint32_t* r = &words[o - 1];
                if (r[3] == 1 && r[2] > 0 && r[1] + r[2] - 1 == row) {
                    r[2]--;
                    return;
                }
                if (row >= r[1] && row < r[1] + r[2]) {  // clears a cell inside the latest run at this column
                    late_zeros.push_back((uint32_t)col);
                    late_zeros.push_back((uint32_t)row);
                }
            }
            return;  // otherwise a first write of zero (a clear bit of a decomposition): the matrix starts zeroed
        }
        const uint32_t w = (uint32_t)v;
        put(row, col, &w, 1);
    }
};

}  // namespace starkhip
