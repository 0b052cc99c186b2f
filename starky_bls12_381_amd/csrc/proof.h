// Proof blob layout helpers (layout documented in include/starkhip.h; field order of
// starky's StarkProofWithPublicInputs).
#pragma once
#include <stdint.h>

#include <vector>

#include "../../include/starkhip.h"
#include "gl.h"

namespace starkhip {

struct FriGeometry {
    unsigned log_n, rate_bits, cap_h, log_N;
    std::vector<unsigned> arities;  // reduction_arity_bits
    size_t final_poly_len;
    // plonky2 FriReductionStrategy::ConstantArityBits(arity_bits, final_poly_bits), App. A.8
    static bool make(const starkhip_config_t& cfg, unsigned log_n, FriGeometry* g) {
        g->log_n = log_n;
        g->rate_bits = cfg.rate_bits;
        g->cap_h = cfg.cap_height;
        g->log_N = log_n + cfg.rate_bits;
        g->arities.clear();
        unsigned db = log_n;
        if (cfg.arity_bits == 0) return false;
        while (db > cfg.final_poly_bits && db + cfg.rate_bits >= cfg.cap_height + cfg.arity_bits) {
            g->arities.push_back(cfg.arity_bits);
            db -= cfg.arity_bits;
        }
        unsigned total = 0;
        for (unsigned a : g->arities) total += a;
        if (g->log_N < g->cap_h || total > g->log_N - g->cap_h) return false;
        g->final_poly_len = (size_t)1 << db;
        return true;
    }
};

struct ProofLayout {
    size_t C, Q, log_n, rate_bits, cap_h, L, n_queries, final_len, n_pis, arity_bits, n_challenges;
    size_t ncap;     // 2^cap_h
    size_t log_N;
    // offsets (in words)
    size_t off_trace_cap, off_quot_cap, off_local, off_next, off_quot_open, off_fri_caps, off_queries, query_words, off_final,
        off_pow, off_pis, total;
    std::vector<size_t> layer_depth;  // siblings per FRI layer

    void compute() {
        ncap = (size_t)1 << cap_h;
        log_N = log_n + rate_bits;
        size_t o = 16;
        off_trace_cap = o; o += 4 * ncap;
        off_quot_cap = o; o += 4 * ncap;
        off_local = o; o += 2 * C;
        off_next = o; o += 2 * C;
        off_quot_open = o; o += 2 * Q;
        off_fri_caps = o; o += L * 4 * ncap;
        off_queries = o;
        size_t d0 = log_N - cap_h;
        query_words = C + 4 * d0 + Q + 4 * d0;
        layer_depth.clear();
        size_t lg = log_N;
        for (size_t l = 0; l < L; l++) {
            lg -= arity_bits;
            size_t d = lg - cap_h;
            layer_depth.push_back(d);
            query_words += 2 * ((size_t)1 << arity_bits) + 4 * d;
        }
        o += n_queries * query_words;
        off_final = o; o += 2 * final_len;
        off_pow = o; o += 1;
        off_pis = o; o += n_pis;
        total = o;
    }
    void write_header(uint64_t* h) const {
        h[0] = STARKHIP_PROOF_MAGIC; h[1] = C; h[2] = Q; h[3] = log_n; h[4] = rate_bits; h[5] = cap_h; h[6] = L;
        h[7] = n_queries; h[8] = final_len; h[9] = n_pis; h[10] = arity_bits; h[11] = n_challenges;
        h[12] = h[13] = h[14] = h[15] = 0;
    }
    bool read_header(const uint64_t* h, size_t words) {
        if (words < 16 || h[0] != STARKHIP_PROOF_MAGIC) return false;
        C = h[1]; Q = h[2]; log_n = h[3]; rate_bits = h[4]; cap_h = h[5]; L = h[6]; n_queries = h[7]; final_len = h[8];
        n_pis = h[9]; arity_bits = h[10]; n_challenges = h[11];
        if (log_n > 32 || rate_bits > 8 || cap_h > 16 || L > 16 || arity_bits > 8 || log_n + rate_bits < cap_h) return false;
        if (L * arity_bits + cap_h > log_n + rate_bits) return false;
        compute();
        return total == words;
    }
};

}  // namespace starkhip
