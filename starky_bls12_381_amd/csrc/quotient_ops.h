// Device form of an AIR constraint program: the flat GROUP/TERM code of air_ir.h re-cut into fixed
// 16-byte ops so the quotient kernel can fetch four ops with ONE s_load_dwordx16 and issue their four
// trace-cell loads back to back, one batch ahead of the arithmetic (kernels_quotient.hip).
//
// Every op carries exactly one cell reference (ops that need none point at column 0, a harmless load):
//   NOP                 padding up to a batch boundary
//   GROUP(kind)         G <- mask[kind]; t0 = t1 = 0
//   GATE(ref)           G <- G * cell        (or G * (1 - cell) when ref carries REF_COMPL)
//   FACTOR(ref)         v <- cell            (or v * cell when PREV is set): non-final factor of a product term
//   TERM(ref, ck, k)    u <- cell, v * cell (PREV) or 1 (NOCELL); body <- body (+|-) u  or  body + u * k;
//                       with FOLD (last term of its constraint): t_j <- t_j * alpha_j + body, body <- 0
//   ENDGROUP(m)         acc_j <- acc_j * alpha_j^m + G * t_j
// which is the per-group Horner fold documented in air_ir.h, i.e. the reference's
// ConstraintConsumer accumulation acc = acc * alpha + constraint.
#pragma once
#include <stdint.h>

#include <vector>

#include "air_ir.h"

namespace starkhip {

enum : uint32_t { QOP_NOP = 0, QOP_GROUP = 1, QOP_GATE = 2, QOP_FACTOR = 3, QOP_TERM = 4, QOP_ENDGROUP = 5 };
enum : uint32_t {
    QOP_FOLD = 1u << 3,    // TERM: last term of the constraint
    QOP_PREV = 1u << 4,    // FACTOR / TERM: v holds the product of the earlier factors
    QOP_NOCELL = 1u << 5,  // TERM: no cell factor (the term is a constant)
    QOP_CK_SHIFT = 6,      // TERM: [8:6] = CK_*
    QOP_KIND_SHIFT = 9,    // GROUP: [10:9] = KIND_*
    QOP_SIMPLE = 1u << 11, // TERM: coefficient (+1, -1 or a constant) times ONE cell: the evaluator's short path
    QOP_IDX_SHIFT = 16     // ENDGROUP: m; TERM with CK_PI / CK_NEG_PI: public input index
};
static const unsigned QOP_BATCH = 4;          // ops per scalar fetch
static const unsigned QOP_UNROLL = 4;         // batches per unrolled kernel step (= depth of the global-load ring)
static const unsigned QOP_GUARD_BATCHES = 8;  // NOP batches behind the last chunk (the kernel reads up to 5 batches ahead)
enum : uint32_t { QREF_SLOT_MASK = 63u, QREF_FROM_LDS = 1u << 8, QREF_STORE = 1u << 9 };  // op.ref after attach_cell_cache()

struct QOp {
    uint32_t hdr;
    uint32_t ref;  // cellref of air_ir.h; after attach_cell_cache(): slot / QREF_* flags / REF_COMPL
    uint64_t k;    // TERM with CK_CONST: the coefficient (canonical)
};
static_assert(sizeof(QOp) == 16, "QOp is fetched as 4 dwords");

struct QProgram {
    std::vector<QOp> ops;                // chunk after chunk, each padded to a whole number of batches, + 2 guard batches
    std::vector<uint32_t> chunk_batch;   // [n_chunks + 1] first batch of each chunk
    std::vector<uint32_t> chunk_k_after; // constraints that follow the chunk (its fold is scaled by alpha^that)
    std::vector<uint32_t> loads;         // attach_cell_cache(): per op, the cellref its (always issued) global load fetches
};

// Cut `P` at group boundaries into at most `want` chunks of about equal op count.
inline QProgram compile_quotient_ops(const AirProgram& P, unsigned want) {
    // pass 1: ops of every group
    std::vector<std::vector<QOp>> groups;
    const std::vector<uint32_t>& code = P.code;
    size_t i = 0;
    while (i < code.size() && code[i] != 0) {
        const uint32_t gw = code[i++];
        const uint32_t kind = (gw >> 4) & 3u, ng = (gw >> 8) & 255u, m = gw >> 16;
        std::vector<QOp> g;
        g.push_back({QOP_GROUP | (kind << QOP_KIND_SHIFT), 0, 0});
        for (uint32_t j = 0; j < ng; j++) g.push_back({QOP_GATE, code[i++], 0});
        for (uint32_t c = 0; c < m; c++) {
            uint32_t tw;
            do {
                tw = code[i++];
                const uint32_t nf = tw & 3u, ck = (tw >> 2) & 7u, idx = tw >> 6;
                for (uint32_t f = 0; f + 1 < nf; f++) g.push_back({QOP_FACTOR | (f ? QOP_PREV : 0u), code[i++], 0});
                QOp t;
                t.hdr = QOP_TERM | (ck << QOP_CK_SHIFT) | ((tw & 32u) ? QOP_FOLD : 0u) | (nf >= 2 ? QOP_PREV : 0u) | (nf == 0 ? QOP_NOCELL : 0u);
                if (nf == 1 && (ck == CK_PLUS || ck == CK_MINUS || ck == CK_CONST)) t.hdr |= QOP_SIMPLE;
                t.ref = nf ? code[i++] : 0;
                t.k = 0;
                if (ck == CK_CONST) t.k = P.consts[idx];
                else if (ck == CK_PI || ck == CK_NEG_PI) t.hdr |= idx << QOP_IDX_SHIFT;
                g.push_back(t);
            } while (!(tw & 32u));
        }
        g.push_back({QOP_ENDGROUP | (m << QOP_IDX_SHIFT), 0, 0});
        groups.push_back(std::move(g));
    }
    size_t total = 0;
    for (auto& g : groups) total += g.size();

    QProgram Q;
    const size_t n_groups = groups.size();
    if (want > n_groups) want = (unsigned)n_groups;
    if (want == 0) want = 1;
    size_t g = 0, done = 0;
    for (unsigned p = 0; p < want && g < n_groups; p++) {
        Q.chunk_batch.push_back((uint32_t)(Q.ops.size() / QOP_BATCH));
        const size_t target = total * (p + 1) / want;
        size_t g_end = g;
        do {
            done += groups[g_end].size();
            Q.ops.insert(Q.ops.end(), groups[g_end].begin(), groups[g_end].end());
            g_end++;
        } while (g_end < n_groups && (p + 1 == want || (done < target && (n_groups - g_end) > (want - 1 - p))));
        while (Q.ops.size() % (QOP_UNROLL * QOP_BATCH)) Q.ops.push_back({QOP_NOP, 0, 0});  // whole unrolled steps
        const uint32_t k_end = g_end < n_groups ? P.group_k0[g_end] : P.n_constraints;
        Q.chunk_k_after.push_back(P.n_constraints - k_end);
        g = g_end;
    }
    Q.chunk_batch.push_back((uint32_t)(Q.ops.size() / QOP_BATCH));
    for (unsigned z = 0; z < QOP_GUARD_BATCHES * QOP_BATCH; z++) Q.ops.push_back({QOP_NOP, 0, 0});  // the kernel prefetches ahead
    return Q;
}

// Per-wave cell cache in LDS (kernels_quotient.hip): the order in which a chunk touches trace cells is known here, so
// the slot assignment is done ahead of time with Belady's rule (evict the entry whose next use is farthest) instead
// of a hardware-style policy: 40 slots remove ~2/3 of the cell loads of the four AIRs (an LRU of that size, ~1/2).
// Timing model of the kernel, in batches: the global load of an op in batch b is issued 4 batches early, its value is
// written to the slot while batch b is evaluated, and LDS reads for batch b+1 were issued before that -- so a slot
// filled in batch b serves ops from batch b+2 on; ops in between that want the same cell load it from memory again.
// After the pass: Q.loads[i] = cellref to load for op i (0 = column 0, a hot dummy: every op issues exactly one load so
// the kernel's s_waitcnt counts are static), and op.ref = [5:0] slot, QREF_FROM_LDS, QREF_STORE, REF_COMPL.
inline void attach_cell_cache(QProgram& Q, unsigned n_slots) {
    const size_t n_ops = Q.ops.size();
    Q.loads.assign(n_ops, 0);
    const uint32_t KEYMASK = REF_COL_MASK | REF_NEXT;
    auto uses_cell = [](const QOp& o) {
        const uint32_t op = o.hdr & 7u;
        return op == QOP_GATE || op == QOP_FACTOR || (op == QOP_TERM && !(o.hdr & QOP_NOCELL));
    };
    const size_t INF = (size_t)-1;
    std::vector<size_t> next_use(n_ops, INF);
    for (size_t c = 0; c + 1 < Q.chunk_batch.size(); c++) {
        const size_t lo = (size_t)Q.chunk_batch[c] * QOP_BATCH, hi = (size_t)Q.chunk_batch[c + 1] * QOP_BATCH;
        std::map<uint32_t, size_t> last;
        for (size_t i = hi; i-- > lo;) {
            if (!uses_cell(Q.ops[i])) continue;
            const uint32_t key = Q.ops[i].ref & KEYMASK;
            auto it = last.find(key);
            next_use[i] = it == last.end() ? INF : it->second;
            last[key] = i;
        }
        // every chunk starts with an empty cache (its waves start cold)
        std::map<uint32_t, unsigned> where;
        std::vector<uint32_t> slot_key(n_slots, 0);
        std::vector<size_t> slot_next(n_slots, INF), slot_ready(n_slots, 0);
        std::vector<char> slot_used(n_slots, 0);
        for (size_t i = lo; i < hi; i++) {
            QOp& o = Q.ops[i];
            const uint32_t compl_bit = o.ref & REF_COMPL;
            if (!uses_cell(o)) {
                o.ref = 0;
                continue;
            }
            const uint32_t key = o.ref & KEYMASK;
            const size_t b = i / QOP_BATCH, nu = next_use[i];
            auto it = n_slots ? where.find(key) : where.end();
            if (it != where.end()) {
                const unsigned s = it->second;
                slot_next[s] = nu;
                if (slot_ready[s] <= b) {
                    o.ref = s | QREF_FROM_LDS | compl_bit;  // hit
                    continue;
                }
                Q.loads[i] = key;  // being filled right now: load again
                o.ref = compl_bit;
                continue;
            }
            Q.loads[i] = key;
            o.ref = compl_bit;
            if (!n_slots || nu == INF) continue;  // never needed again in this chunk
            unsigned victim = 0;
            bool found_free = false;
            for (unsigned s = 0; s < n_slots; s++)
                if (!slot_used[s]) {
                    victim = s;
                    found_free = true;
                    break;
                }
            if (!found_free) {
                for (unsigned s = 1; s < n_slots; s++)
                    if (slot_next[s] > slot_next[victim]) victim = s;  // INF (dead) entries first
                if (slot_next[victim] != INF && slot_next[victim] <= nu) continue;  // everything cached is needed sooner: bypass
                where.erase(slot_key[victim]);
            }
            slot_used[victim] = 1;
            slot_key[victim] = key;
            slot_next[victim] = nu;
            slot_ready[victim] = b + 2;
            where[key] = victim;
            o.ref = victim | QREF_STORE | compl_bit;
        }
    }
    for (size_t i = 0; i < n_ops; i++)
        if (i >= (size_t)Q.chunk_batch.back() * QOP_BATCH) Q.ops[i].ref = 0;  // guard batches
}

}  // namespace starkhip
