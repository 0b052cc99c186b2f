// Tiled form of an AIR constraint program for the quotient kernel (kernels_quotient.hip).
//
// What the reference computes per LDE point is  acc_j = sum_k mask(kind_k) * c_k * alpha_j^(K-1-k)  (the ConstraintConsumer
// fold of S::eval_packed_generic, e.g. /root/reference/src/final_exponentiate.rs:907-1136).  Field
// arithmetic is exact, so the sum may be regrouped freely as long as every constraint keeps its own power of alpha:
//
//  * constraints with the same (kind, gates) anywhere in the program form a SUPERGROUP with common factor
//    mask(kind) * G, G = product of the gate cells (or 1 - cell);
//  * inside a supergroup the bodies collapse to  T_j = sum_m  w_j(m) * m  over the distinct monomials m (one trace cell for
//    92 % of the terms, a product of two or three cells otherwise, or the constant 1) with per-proof weights
//        w_j(m) = sum over the terms (k, coefficient) with monomial m of  coefficient * alpha_j^(K-1-k)
//    (coefficient = +-1, a table constant, or +-public_input) -- computed once per proof on the device;
//  * the sum over m is cut by COLUMN TILE: a tile is QT_TILE_COLS consecutive trace columns whose 64-point slice a
//    workgroup stages in LDS once; a PIECE = (supergroup, tile) holds the monomials whose first cell lies in the tile, and
//    contributes  mask * G * T_j(piece)  to acc_j.  Every LDE cell is thus read from HBM once per point (plus the gate
//    cells and the few factors that live in another tile, which are loaded directly).
//
// FinalExp: 360 800 constraints / 1.10 M terms -> 15 982 supergroups, 0.68 M monomial records, ~30 K pieces, 1 149 tiles.
//
// Layout for the kernel: the tiles of the program are cut into `n_chunks` contiguous ranges of about equal cost; a
// workgroup of QT_WAVES waves handles (64 points) x (one chunk) and walks the chunk's tiles in step (one barrier per tile);
// inside a tile the pieces are dealt to the waves (largest first), big pieces are split -- the contribution is linear in T.
// Each wave reads its own stream of 32-byte wave-uniform records and a stream of piece descriptors.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "air_ir.h"

namespace starkhip {

static const unsigned QT_TILE_COLS = 64;    // columns per LDS tile
static const unsigned QT_ONES_SLOT = 64;     // = QT_TILE_COLS: one more "column" of every tile buffer that holds 1 in every row, so that a constant
                                            // term is a PLAIN record (a cell read like any other) instead of a special one
static const unsigned QT_TILE_ROWS = 66;    // 64 points + the successor of the last one (next-row reads are "lane + 1") + 8 bytes so
                                            // that every column starts on a 16-byte boundary (direct-to-LDS loads write 16 bytes per lane)
static const unsigned QT_WAVES = 8;         // evaluating waves per workgroup: four per SIMD with two workgroups on a CU (for n >= 64 they stage the tiles themselves)
static const unsigned QT_MAX_PIECE = 192;   // records per piece after splitting
// Records one accumulation chain may hold before its fold: a piece, or the pieces of one supergroup a wave carries across tile
// boundaries.  Each record adds a (32-bit half) x (22-bit limb) product < 2^54 to six 64-bit sums, so S < 960 * 2^54 =
// 0.9375 * 2^64; qt_fold_sums' word chains add at most 2 * 2^54 + 2^44 + 2^32 to one S (X = S0 + lo(S1) 2^22,
// Y = S3 + hi(S1) 2^22 + lo(S2) 2^12 + lo(S4) 2^22 + hi(X)), i.e. < (0.9375 + 2^-9) * 2^64: no wrap.  At 1024 records the sums
// themselves still fit but X and Y can wrap -- planner, host replay and kernel all use THIS bound.
static const unsigned QT_MAX_CHAIN = 960;
static const unsigned QT_FOLD_MAX_MONOS = 8;  // supergroups of at most this many monomials, all of them single cells, give one gate to their monomials (pass 1b)
static const unsigned QT_MAX_RUN = 252;     // longest plain run one special record announces (a multiple of four that fits eight bits)
static const unsigned QT_LIMB_BITS = 22;    // weights are split into three limbs of 22 bits

// record control word
enum : uint32_t {
    QT_OFF_MASK = 0xFFFFu,     // [15:0] byte offset of the cell inside the LDS tile: slot * QT_TILE_ROWS * 8 (+ 8 for the next row)
    QT_NEXT = 1u << 16,        // the cell is taken from the next row (already in the offset for the LDS path; SMALL_N kernels and
                               // direct loads look at the flag)
    QT_SRC_ONE = 1u << 17,     // x = 1 (constant term)
    QT_SRC_GLOBAL = 1u << 18,  // x = direct load of column `aux` (a factor outside the tile)
    QT_SETV = 1u << 19,        // v = x (SETV alone) or v = v * x (SETV | MULV); no accumulation
    QT_MULV = 1u << 20,        // x = v * x before accumulating
    QT_END = 1u << 21,         // last record of its piece: acc_j += mask * G * T_j, next piece descriptor
    QT_TILE = 1u << 22,        // no cell: the wave is done with this tile (barrier, next tile)
    QT_STOP = 1u << 23,        // end of stream
    QT_RUN_SHIFT = 24,         // [31:24] of a SPECIAL record (any flag below set): the number of plain records (no flag at all) that
                               // follow it before the next special one, at most QT_MAX_RUN -- the kernel evaluates such a run four
                               // records at a time without looking at their control words' flags; plain records carry 0 here
    QT_DESC = QT_SRC_ONE | QT_SETV | QT_NEXT,  // the descriptor of the piece that starts here (QTPiece in the weight words: w[0 .. 3] = gate, w[4] = ctl);
                               // a no-op as a record (QT_NEXT means nothing for x = 1, so the combination is free)
    QT_SPECIAL = QT_SRC_ONE | QT_SRC_GLOBAL | QT_SETV | QT_MULV | QT_END | QT_TILE | QT_STOP,
    QT_ODD_SOURCE = QT_SRC_ONE | QT_SRC_GLOBAL | QT_SETV | QT_MULV | QT_TILE | QT_STOP
};

struct QTRec {
    uint32_t ctl;
    uint32_t aux;   // QT_SRC_GLOBAL: column
    uint32_t w[6];  // limbs of the weights for alpha_0 then alpha_1 (written per proof by quotient_weights_kernel)
};
static_assert(sizeof(QTRec) == 32, "records are fetched as 8 dwords");

// piece descriptor: ctl = kind | n_gates << 2 | complement mask << 5 | n_foreign << 9
// gate[0 .. n_gates) are the gate cells; gate[n_gates .. n_gates + n_foreign) are FOREIGN cells: the single cells of tiny pieces of the same
// supergroup that lived in other tiles and were absorbed into this one (a piece end costs as much as fifteen records; 10 594 of FinalExp's
// 32 850 closed a piece of exactly one record).  All of them are requested when the piece starts, so a foreign cell is in a register
// long before its record (QT_SRC_GLOBAL with aux = QT_AUX_SLOT | slot) is evaluated.
struct QTPiece {
    uint32_t ctl;
    uint32_t gate[4];  // cellrefs (column | REF_NEXT)
    uint32_t pad[3];
};
// A special record that is not QT_SRC_GLOBAL announces in `aux` (bits 15:0) how many FAST PAIRS follow the plain run it announces: a fast
// pair is a degree-2 monomial with both cells in the tile -- record (QT_SETV, cell a) then record (QT_MULV, cell b, weights), no other flag --
// and the kernel evaluates a run of them two pairs at a time without looking at their control words' flags (generic steps cost ~ 80 issue
// slots per record, a pair in the fast path ~ 40).  Records inside a fast run carry no run length and no announcement.
static const uint32_t QT_AUX_PAIRS_MASK = 0xFFFFu;
// ... and in bits 30:16 how many DIRECT PAIRS follow the fast pairs: (QT_SETV, cell a of the tile) then (QT_SRC_GLOBAL | QT_MULV, column b, weights),
// no other flag, not a descriptor slot -- the kernel requests both factors b of two such pairs at once and multiplies without looking at a
// flag.  Always an even number, and only announced behind an even number of fast pairs (the kernel's loops take two pairs per turn).
static const uint32_t QT_AUX_DPAIRS_SHIFT = 16, QT_AUX_DPAIRS_MASK = 0x7FFFu;
static const uint32_t QT_AUX_SLOT = 0x80000000u;   // aux of a QT_SRC_GLOBAL record: the cell is descriptor slot (aux & 3), not column aux
static const unsigned QT_FOREIGN_SHIFT = 9;
static_assert(sizeof(QTPiece) == 32, "piece descriptors are fetched as 8 dwords");

// one term's share of a record's weight: coefficient * alpha^e
struct QTContrib {
    uint32_t e;    // exponent K - 1 - k
    uint32_t coef; // [2:0] CK_*  [31:3] index (constant table / public input)
};

struct QTStream {
    uint32_t rec_off, piece_off;
};

struct QTPlan {
    uint32_t n_cols = 0, n_constraints = 0, n_chunks = 0;
    std::vector<QTRec> recs;              // all streams, chunk after chunk, wave after wave
    std::vector<QTPiece> pieces;
    std::vector<QTStream> streams;        // [n_chunks][QT_WAVES]
    std::vector<uint32_t> chunk_tile_off; // [n_chunks + 1] into tile_list
    std::vector<uint32_t> tile_list;      // tile indices (first column / QT_TILE_COLS) in walking order
    std::vector<uint32_t> contrib_off;    // [recs.size() + 1]
    std::vector<QTContrib> contribs;
    // statistics
    size_t n_absorbed = 0;  // tiny pieces whose cells ride in a bigger piece's descriptor
    size_t n_supergroups = 0, n_pieces = 0, n_piece_ends = 0, n_cell_records = 0, n_direct_loads = 0;  // n_piece_ends <= n_pieces: pieces carried across a tile boundary end once
    uint64_t cost_sum_max = 0, cost_sum_mean = 0, rec_sum_max = 0, rec_sum_total = 0, tile_phases = 0;  // per tile: the busiest wave's cost / the mean over the waves (model units)
};

namespace qt_detail {

struct Term {
    uint32_t sg;
    uint32_t cells[3];  // sorted; unused = 0xFFFFFFFF
    uint32_t e, coef;
};

inline uint32_t tile_of(uint32_t cellref) { return (cellref & REF_COL_MASK) / QT_TILE_COLS; }

}  // namespace qt_detail

// Build the plan.  `want_chunks` >= 1; the result may have fewer (never more than the number of non-empty tiles).
inline QTPlan build_quotient_plan(const AirProgram& P, unsigned want_chunks) {
    using namespace qt_detail;
    const uint32_t K = P.n_constraints;
    const uint32_t NONE = 0xFFFFFFFFu;
    // ---- pass 1: terms with their supergroup
    std::unordered_map<std::string, uint32_t> sg_index;
    std::vector<std::vector<uint32_t>> sg_gates;
    std::vector<uint32_t> sg_kind;
    std::vector<Term> terms;
    terms.reserve(P.code.size() / 2);
    const std::vector<uint32_t>& code = P.code;
    size_t i = 0;
    uint32_t k = 0;
    while (i < code.size() && code[i] != 0) {
        const uint32_t gw = code[i++];
        const uint32_t kind = (gw >> 4) & 3u, ng = (gw >> 8) & 255u, m = gw >> 16;
        if (ng > 4) throw std::runtime_error("quotient_plan: more than four gates");
        std::string key((const char*)&kind, 4);
        key.append((const char*)&code[i], ng * 4);  // gates are stored sorted by the builder
        uint32_t sg;
        auto it = sg_index.find(key);
        if (it == sg_index.end()) {
            sg = (uint32_t)sg_gates.size();
            sg_index.emplace(std::move(key), sg);
            sg_gates.emplace_back(code.begin() + i, code.begin() + i + ng);
            sg_kind.push_back(kind);
        } else {
            sg = it->second;
        }
        i += ng;
        for (uint32_t c = 0; c < m; c++, k++) {
            uint32_t tw;
            do {
                tw = code[i++];
                const uint32_t nf = tw & 3u, ck = (tw >> 2) & 7u, idx = tw >> 6;
                Term t;
                t.sg = sg;
                t.cells[0] = t.cells[1] = t.cells[2] = NONE;
                for (uint32_t f = 0; f < nf; f++) t.cells[f] = code[i++] & (REF_COL_MASK | REF_NEXT);
                std::sort(t.cells, t.cells + nf);
                t.e = K - 1 - k;
                t.coef = ck | (idx << 3);
                if (ck == CK_CONST && P.consts[idx] == 0) continue;  // explicit zero term of an identically-zero constraint
                terms.push_back(t);
            } while (!(tw & 32u));
        }
    }
    if (k != K) throw std::runtime_error("quotient_plan: constraint count mismatch");

    // ---- pass 1b: fold ONE gate of the tiny supergroups into their monomials.  mask * g * G' * sum_m w(m) m  =  mask * G' * sum_m w(m) (g m):
    // the same constraint powers, regrouped under the supergroup without g.  A piece end (two folds, the gate product, the next
    // descriptor) costs as much as twenty-three plain records (measured per tile: profiles/r06_quotient_cost_fit.txt), and FinalExp has
    // 4 409 supergroups "selector_i * (k_i + sum of five cells)" and 2 384 "g0 * a * b * (cell + k)" of six and two records each: 28 % of
    // its piece ends for 4 % of its records, all of them in one tile (that of the shared cells).  Folded, the first family is 4 409 x
    // (one plain record + five pairs) in the SELECTORS' tiles -- which otherwise keep three of seven waves busy -- and the second merges
    // four to one.  Only supergroups whose monomials are all single cells or constants are folded (the products stay pairs), and only
    // uncomplemented gates ((1 - c) m would be two monomials).
    {
        const size_t n_sg0 = sg_gates.size();
        std::vector<std::vector<uint32_t>> sg_monos(n_sg0);  // distinct first cells (NONE = the constant) -- enough for "tiny and linear"
        std::vector<uint8_t> linear(n_sg0, 1);
        for (const Term& t : terms) {
            if (t.cells[1] != NONE) linear[t.sg] = 0;
            std::vector<uint32_t>& mm = sg_monos[t.sg];
            if (mm.size() <= QT_FOLD_MAX_MONOS && std::find(mm.begin(), mm.end(), t.cells[0]) == mm.end()) mm.push_back(t.cells[0]);
        }
        auto key_without = [&](uint32_t sg, size_t drop) {
            std::string key((const char*)&sg_kind[sg], 4);
            for (size_t g = 0; g < sg_gates[sg].size(); g++)
                if (g != drop) key.append((const char*)&sg_gates[sg][g], 4);
            return key;
        };
        // candidates: (supergroup, gate index) -> key of the target; a target is the better the more tiny supergroups reach it
        std::unordered_map<std::string, uint32_t> reach;
        std::vector<uint32_t> tiny;
        for (uint32_t sg = 0; sg < n_sg0; sg++) {
            if (!linear[sg] || sg_gates[sg].empty() || sg_monos[sg].empty() || sg_monos[sg].size() > QT_FOLD_MAX_MONOS) continue;
            bool any = false;
            for (size_t g = 0; g < sg_gates[sg].size(); g++)
                if (!(sg_gates[sg][g] & REF_COMPL)) {
                    reach[key_without(sg, g)]++;
                    any = true;
                }
            if (any) tiny.push_back(sg);
        }
        std::vector<uint32_t> fold_gate(n_sg0, NONE), fold_to(n_sg0, NONE);
        for (uint32_t sg : tiny) {
            size_t best = NONE;
            uint32_t best_reach = 0;
            for (size_t g = 0; g < sg_gates[sg].size(); g++) {
                if (sg_gates[sg][g] & REF_COMPL) continue;
                const uint32_t r = reach[key_without(sg, g)];
                if (r > best_reach) {
                    best_reach = r;
                    best = g;
                }
            }
            // alone in its target the supergroup would only trade its gate for a longer monomial: fold when at least two meet, or when
            // the target is an existing supergroup (then this one's piece end disappears)
            std::string key = key_without(sg, best);
            const bool target_exists = sg_index.find(key) != sg_index.end();
            if (best_reach < 2 && !target_exists) continue;
            uint32_t to;
            auto it = sg_index.find(key);
            if (it == sg_index.end()) {
                to = (uint32_t)sg_gates.size();
                std::vector<uint32_t> gs;
                for (size_t g = 0; g < sg_gates[sg].size(); g++)
                    if (g != best) gs.push_back(sg_gates[sg][g]);
                sg_index.emplace(std::move(key), to);
                sg_gates.push_back(std::move(gs));
                sg_kind.push_back(sg_kind[sg]);
            } else {
                to = it->second;
            }
            fold_gate[sg] = sg_gates[sg][best] & (REF_COL_MASK | REF_NEXT);
            fold_to[sg] = to;
        }
        for (Term& t : terms) {
            if (t.sg >= n_sg0 || fold_to[t.sg] == NONE) continue;
            t.cells[t.cells[0] == NONE ? 0 : 1] = fold_gate[t.sg];  // linear: at most one cell so far
            if (t.cells[1] != NONE && t.cells[1] < t.cells[0]) std::swap(t.cells[0], t.cells[1]);
            t.sg = fold_to[t.sg];
        }
    }
#ifdef STARKHIP_DEBUG
    if (getenv("STARKHIP_PLAN_STATS")) {  // development aid: supergroups by (terms, gates)
        std::vector<uint32_t> n_terms(sg_gates.size(), 0), max_deg(sg_gates.size(), 0);
        for (const Term& t : terms) {
            n_terms[t.sg]++;
            uint32_t d = 0;
            for (int f = 0; f < 3; f++) d += t.cells[f] != NONE;
            max_deg[t.sg] = std::max(max_deg[t.sg], d);
        }
        size_t hist[6][5][4] = {{{0}}};
        for (size_t g = 0; g < sg_gates.size(); g++) {
            const uint32_t b = n_terms[g] <= 1 ? 0 : n_terms[g] <= 2 ? 1 : n_terms[g] <= 4 ? 2 : n_terms[g] <= 8 ? 3 : n_terms[g] <= 16 ? 4 : 5;
            hist[b][std::min<size_t>(4, sg_gates[g].size())][std::min<uint32_t>(3, max_deg[g])]++;
        }
        const char* names[6] = {"1", "2", "3-4", "5-8", "9-16", ">16"};
        for (int b = 0; b < 6; b++)
            for (int g = 0; g < 5; g++)
                for (int d = 0; d < 4; d++)
                    if (hist[b][g][d]) fprintf(stderr, "supergroups with %s terms, %d gates, max degree %d: %zu\n", names[b], g, d, hist[b][g][d]);
    }
#endif

    // ---- pass 2: merge terms with the same (supergroup, monomial) into records; order by (tile, supergroup, monomial)
    // the tile of a monomial is the tile of its first cell; constants go with the supergroup's first gate (or tile 0)
    auto term_tile = [&](const Term& t) -> uint32_t {
        if (t.cells[0] != NONE) {
            // prefer the tile that holds most factors
            uint32_t t0 = tile_of(t.cells[0]);
            if (t.cells[1] != NONE && t.cells[2] != NONE && tile_of(t.cells[1]) == tile_of(t.cells[2])) return tile_of(t.cells[1]);
            return t0;
        }
        return sg_gates[t.sg].empty() ? 0u : tile_of(sg_gates[t.sg][0]);
    };
    std::vector<uint32_t> order(terms.size());
    std::vector<uint32_t> ttile(terms.size());
    for (size_t j = 0; j < terms.size(); j++) {
        order[j] = (uint32_t)j;
        ttile[j] = term_tile(terms[j]);
    }
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        const Term &x = terms[a], &y = terms[b];
        if (ttile[a] != ttile[b]) return ttile[a] < ttile[b];
        if (x.sg != y.sg) return x.sg < y.sg;
        for (int f = 0; f < 3; f++)
            if (x.cells[f] != y.cells[f]) return x.cells[f] < y.cells[f];
        return x.e > y.e;
    });

    struct Mono {
        uint32_t cells[3];
        uint32_t c_begin, c_end;  // contributions
    };
    struct Piece {
        uint32_t sg, tile;
        uint32_t m_begin, m_end;  // monomials
        uint32_t cost;      // issue slots (model) of its records, + PIECE_COST
        uint32_t n_recs = 0;  // records (what the fold's overflow bound counts)
        std::vector<uint32_t> foreign;  // monomials (single cells of other tiles) absorbed from tiny pieces of the same supergroup
    };
    std::vector<Mono> monos;
    std::vector<QTContrib> contribs;
    std::vector<Piece> pieces;
    size_t Q_absorbed = 0;
    contribs.reserve(terms.size());
    for (size_t j = 0; j < order.size(); j++) {
        const Term& t = terms[order[j]];
        const uint32_t tl = ttile[order[j]];
        const bool same_piece = !pieces.empty() && pieces.back().sg == t.sg && pieces.back().tile == tl;
        const bool same_mono = same_piece && !monos.empty() && monos.back().cells[0] == t.cells[0] && monos.back().cells[1] == t.cells[1] &&
                               monos.back().cells[2] == t.cells[2] && pieces.back().m_end == monos.size();
        if (!same_piece) pieces.push_back({t.sg, tl, (uint32_t)monos.size(), (uint32_t)monos.size(), 0, 0, {}});
        if (!same_mono) {
            monos.push_back({{t.cells[0], t.cells[1], t.cells[2]}, (uint32_t)contribs.size(), (uint32_t)contribs.size()});
            pieces.back().m_end = (uint32_t)monos.size();
        }
        contribs.push_back({t.e, t.coef});
        monos.back().c_end = (uint32_t)contribs.size();
    }
    auto mono_records = [&](const Mono& m) -> uint32_t { return m.cells[0] == NONE ? 1u : m.cells[1] == NONE ? 1u : m.cells[2] == NONE ? 2u : 3u; };
    // Inside a piece the order of the monomials is free (a sum): single cells first (they form the plain runs), then the degree-2
    // monomials with both cells in the piece's tile (the fast pairs), then those with one cell in it (the direct pairs), then the rest.
    {
        auto klass = [&](const Mono& m, uint32_t tile) -> int {
            if (m.cells[1] == NONE) return 0;   // one cell of the tile, or the constant (the tile's column of ones)
            if (m.cells[1] != NONE && m.cells[2] == NONE && tile_of(m.cells[0]) == tile && tile_of(m.cells[1]) == tile) return 1;
            if (m.cells[1] != NONE && m.cells[2] == NONE && (tile_of(m.cells[0]) == tile || tile_of(m.cells[1]) == tile)) return 2;  // direct pairs
            return 3;
        };
        for (const Piece& p : pieces)
            std::stable_sort(monos.begin() + p.m_begin, monos.begin() + p.m_end,
                             [&](const Mono& a, const Mono& b) { return klass(a, p.tile) < klass(b, p.tile); });
    }
    // Tiny pieces go into the largest piece of their supergroup: a piece of one to three single-cell monomials costs a whole piece end
    // (fold, gate product, descriptor, ~ 280 issue slots) for <= 3 records; as FOREIGN cells of the big piece they cost one record each and a
    // descriptor slot (gates + foreign cells <= 4), and are requested with the gates when that piece starts.
    {
        std::vector<std::vector<uint32_t>> by_sg(sg_gates.size());
        for (size_t j = 0; j < pieces.size(); j++) by_sg[pieces[j].sg].push_back((uint32_t)j);
        std::vector<uint8_t> dead(pieces.size(), 0);
        for (size_t sg = 0; sg < by_sg.size(); sg++) {
            const std::vector<uint32_t>& ps = by_sg[sg];
            if (ps.size() < 2) continue;
            uint32_t big = ps[0];
            auto n_monos = [&](uint32_t pi) { return pieces[pi].m_end - pieces[pi].m_begin; };
            for (uint32_t pi : ps)
                if (n_monos(pi) > n_monos(big)) big = pi;
            // (the constant monomial of a supergroup used to get a piece of its own in the tile of the first gate: it needs no cell at
            // all and goes along for nothing -- 8 178 of FinalExp's single-record pieces were that)
            size_t room = 4 - sg_gates[sg].size();
            for (uint32_t pi : ps) {
                if (pi == big || n_monos(pi) > 3) continue;
                size_t cells_needed = 0;
                bool ok = true;
                for (uint32_t b = pieces[pi].m_begin; b < pieces[pi].m_end; b++) {
                    if (monos[b].cells[1] != NONE) ok = false;          // a product: stays where its factors are
                    else if (monos[b].cells[0] != NONE) cells_needed++;
                }
                if (!ok || cells_needed > room) continue;
                for (uint32_t b = pieces[pi].m_begin; b < pieces[pi].m_end; b++) pieces[big].foreign.push_back(b);
                room -= cells_needed;
                dead[pi] = 1;
                Q_absorbed++;
            }
        }
        std::vector<Piece> kept;
        kept.reserve(pieces.size());
        for (size_t j = 0; j < pieces.size(); j++)
            if (!dead[j]) kept.push_back(std::move(pieces[j]));
        pieces.swap(kept);
    }
    // Costs by kind of record, in cycles of an evaluator wave's clock between two tile barriers, fitted to the per-tile clocks of
    // the profiling variant over FinalExp's 8 043 (tile, wave) phases (tools/quotient_wave_prof.py + tools/plan_streams.cpp ->
    // tools/fit_plan_costs.py, R^2 = 0.99; profiles/r06_quotient_cost_fit.txt): four plain records in the straight-line block 690,
    // a plain record through the generic step 356, two fast pairs 830, two direct pairs 1 058, any other special record ~ 1 350, a
    // piece end 4 010 (2 600 once the descriptor rides in the record stream), a tile change 2 500.  (The round-4 figures -- 18 issue
    // slots per record, 240 per piece end -- had a piece end at thirteen records; it costs twenty-three, so tiles were dealt unevenly:
    // the busiest wave of a tile worked 1.31 x the mean.)
    const uint32_t PIECE_COST = 2600, REC_COST = 172, PAIR_COST = 415, DPAIR_COST = 530, GENERIC_COST = 1350, TILE_COST = 2500;
    auto measure = [&](Piece& p) {
        uint32_t recs = 0, cost = 0;
        for (uint32_t b = p.m_begin; b < p.m_end; b++) {
            const Mono& m = monos[b];
            const uint32_t r = mono_records(m);
            recs += r;
            if (m.cells[1] == NONE) cost += REC_COST;
            else if (m.cells[2] == NONE && tile_of(m.cells[0]) == p.tile && tile_of(m.cells[1]) == p.tile) cost += PAIR_COST;
            else if (m.cells[2] == NONE && (tile_of(m.cells[0]) == p.tile || tile_of(m.cells[1]) == p.tile)) cost += DPAIR_COST;
            else cost += GENERIC_COST * r;
        }
        recs += (uint32_t)p.foreign.size();
        cost += GENERIC_COST * (uint32_t)p.foreign.size();
        p.n_recs = recs;
        p.cost = PIECE_COST + cost;
    };
    for (Piece& p : pieces) measure(p);
    // Split pieces (the contribution is linear in T, so a piece may be cut anywhere): into parts of at most QT_MAX_PIECE records, and
    // so that no part is longer than a wave's share of its tile -- a tile with three pieces (the 128 tiles of FinalExp's selector
    // columns: one transition piece of 128 records, a first-row and a last-row piece of 64) otherwise keeps three waves busy and four
    // waiting at the barrier.  A part that goes on from the previous tile in the same wave has no end of its own (pass 4), so in such
    // runs of tiles the split costs nothing; elsewhere it costs a piece end, hence no part below half of one.
    {
        std::vector<uint64_t> tile_total(pieces.empty() ? 0 : pieces.back().tile + 1, 0);
        for (const Piece& p : pieces) tile_total[p.tile] += p.cost;
        std::vector<Piece> cut;
        cut.reserve(pieces.size() + pieces.size() / 4);
        for (const Piece& p : pieces) {
            const uint64_t share = std::max<uint64_t>(tile_total[p.tile] / QT_WAVES, 2 * PIECE_COST);
            const uint32_t body = p.cost - PIECE_COST, own = p.n_recs - (uint32_t)p.foreign.size();
            uint32_t parts = std::max<uint32_t>(1, (own + QT_MAX_PIECE - 1) / QT_MAX_PIECE);
            if (body > share) parts = std::max<uint32_t>(parts, std::min<uint32_t>((uint32_t)((body + share - 1) / share), std::max<uint32_t>(1, body / (PIECE_COST / 2))));
            // cut at monomial boundaries into `parts` runs of about equal record count
            uint32_t b = p.m_begin, start = p.m_begin, recs = 0, made = 0;
            const uint32_t per = (own + parts - 1) / parts;
            for (; b < p.m_end; b++) {
                const uint32_t r = mono_records(monos[b]);
                if (recs && (recs + r > QT_MAX_PIECE || (recs + r > per && made + 1 < parts))) {
                    cut.push_back({p.sg, p.tile, start, b, 0, 0, {}});
                    made++;
                    start = b;
                    recs = 0;
                }
                recs += r;
            }
            cut.push_back({p.sg, p.tile, start, p.m_end, 0, 0, p.foreign});  // the absorbed cells go with the last part
        }
        pieces.swap(cut);
    }
    for (Piece& p : pieces) measure(p);

    // ---- pass 3: tiles in walking order, cut into chunks of about equal cost
    std::vector<uint32_t> tile_first_piece;  // index into pieces of each non-empty tile (+ sentinel)
    std::vector<uint32_t> tiles;
    std::vector<uint64_t> tile_cost;
    for (size_t j = 0; j < pieces.size(); j++) {
        if (j == 0 || pieces[j].tile != pieces[j - 1].tile) {
            tiles.push_back(pieces[j].tile);
            tile_first_piece.push_back((uint32_t)j);
            tile_cost.push_back(TILE_COST * QT_WAVES);
        }
        tile_cost.back() += pieces[j].cost;
    }
    tile_first_piece.push_back((uint32_t)pieces.size());
    uint64_t total_cost = 0;
    for (uint64_t c : tile_cost) total_cost += c;

    QTPlan Q;
    Q.n_cols = P.n_cols;
    Q.n_constraints = K;
    {  // supergroups in use (pass 1b empties the tiny ones it folds and may open new targets)
        std::vector<uint8_t> used(sg_gates.size(), 0);
        for (const Term& t : terms) used[t.sg] = 1;
        Q.n_supergroups = (size_t)std::count(used.begin(), used.end(), 1);
    }
    Q.n_pieces = pieces.size();
    Q.n_absorbed = Q_absorbed;
    if (tiles.empty()) {  // a program without terms: one empty chunk
        Q.n_chunks = 1;
        Q.chunk_tile_off = {0, 0};
        for (unsigned w = 0; w < QT_WAVES; w++) {
            Q.streams.push_back({(uint32_t)Q.recs.size(), (uint32_t)Q.pieces.size()});
            Q.recs.push_back({QT_STOP, 0, {0, 0, 0, 0, 0, 0}});
            Q.pieces.push_back({0, {0, 0, 0, 0}, {0, 0, 0}});
        }
        Q.contrib_off.assign(Q.recs.size() + 1, 0);
        return Q;
    }
    unsigned n_chunks = std::max(1u, std::min<unsigned>(want_chunks, (unsigned)tiles.size()));
    std::vector<uint32_t> chunk_first_tile;
    {
        uint64_t done = 0;
        size_t t = 0;
        for (unsigned c = 0; c < n_chunks && t < tiles.size(); c++) {
            chunk_first_tile.push_back((uint32_t)t);
            const uint64_t target = total_cost * (c + 1) / n_chunks;
            do {
                done += tile_cost[t];
                t++;
            } while (t < tiles.size() && (c + 1 == n_chunks || (done < target && tiles.size() - t > n_chunks - 1 - c)));
        }
        n_chunks = (unsigned)chunk_first_tile.size();
        chunk_first_tile.push_back((uint32_t)tiles.size());
    }
    Q.n_chunks = n_chunks;

    // ---- pass 4: deal the pieces of each tile to the waves and emit the streams
    std::vector<std::vector<uint32_t>> contrib_of_rec;  // parallel to Q.recs while building: [begin, end) pairs
    std::vector<uint32_t> rec_c_begin, rec_c_end;
    auto push_rec = [&](uint32_t ctl, uint32_t aux, uint32_t cb, uint32_t ce) {
        Q.recs.push_back({ctl, aux, {0, 0, 0, 0, 0, 0}});
        rec_c_begin.push_back(cb);
        rec_c_end.push_back(ce);
    };
    // streams are built wave by wave, so first decide the assignment for every tile of the chunk
    for (unsigned c = 0; c < n_chunks; c++) {
        const uint32_t t_lo = chunk_first_tile[c], t_hi = chunk_first_tile[c + 1];
        Q.chunk_tile_off.push_back((uint32_t)Q.tile_list.size());
        for (uint32_t t = t_lo; t < t_hi; t++) Q.tile_list.push_back(tiles[t]);
        // assignment[tile - t_lo][wave] = piece indices
        std::vector<std::vector<std::vector<uint32_t>>> assign(t_hi - t_lo, std::vector<std::vector<uint32_t>>(QT_WAVES));
        // carry_in[tile - t_lo][wave]: the wave's first piece of this tile has the supergroup of its last piece of the previous
        // tile and simply goes on accumulating -- no fold, no gate product, no descriptor at the tile boundary (a supergroup
        // spans 2.4 tiles on average and a piece end costs as much as a dozen records)
        std::vector<std::vector<uint8_t>> carry_in(t_hi - t_lo, std::vector<uint8_t>(QT_WAVES, 0));
        std::vector<uint32_t> chain_recs(QT_WAVES, 0);  // records accumulated by the piece a wave ends the previous tile with
        auto piece_recs = [&](uint32_t p) { return pieces[p].n_recs; };
        for (uint32_t t = t_lo; t < t_hi; t++) {
            std::vector<uint32_t> ps;
            for (uint32_t p = tile_first_piece[t]; p < tile_first_piece[t + 1]; p++) ps.push_back(p);
            std::stable_sort(ps.begin(), ps.end(), [&](uint32_t a, uint32_t b) { return pieces[a].cost > pieces[b].cost; });
            uint64_t load[QT_WAVES] = {0};
            std::vector<std::vector<uint32_t>>& lists = assign[t - t_lo];
            std::vector<uint32_t> next_chain(QT_WAVES, 0);
            std::vector<uint8_t> taken(ps.size(), 0);
            // first: every wave that can continue a supergroup of its previous tile gets that piece as its first one ...
            if (t > t_lo) {
                for (unsigned w = 0; w < QT_WAVES; w++) {
                    std::vector<uint32_t>& prev = assign[t - 1 - t_lo][w];
                    if (prev.empty()) continue;
                    const bool first_pinned = carry_in[t - 1 - t_lo][w];
                    bool done = false;
                    for (size_t a = 0; a < ps.size() && !done; a++) {  // largest first
                        if (taken[a]) continue;
                        for (size_t b = 0; b < prev.size() && !done; b++) {
                            if (pieces[ps[a]].sg != pieces[prev[b]].sg) continue;
                            if (b == 0 && first_pinned && prev.size() > 1) continue;  // that piece must stay first
                            const uint32_t so_far = (b == 0 && first_pinned) ? chain_recs[w] : piece_recs(prev[b]);
                            if (so_far + piece_recs(ps[a]) > QT_MAX_CHAIN) continue;   // the fold's 64-bit chains (QT_MAX_CHAIN)
                            std::swap(prev[b], prev.back());                          // ends the previous tile
                            lists[w].push_back(ps[a]);                                // starts this one
                            load[w] += pieces[ps[a]].cost - PIECE_COST;               // it has no piece end of its own
                            taken[a] = 1;
                            carry_in[t - t_lo][w] = 1;
                            next_chain[w] = so_far + piece_recs(ps[a]);
                            done = true;
                        }
                    }
                }
            }
            // ... then the rest, largest first, to the least loaded wave
            for (size_t a = 0; a < ps.size(); a++) {
                if (taken[a]) continue;
                unsigned best = 0;
                for (unsigned w = 1; w < QT_WAVES; w++)
                    if (load[w] < load[best]) best = w;
                load[best] += pieces[ps[a]].cost;
                lists[best].push_back(ps[a]);
            }
            // the chain a wave may continue from this tile: its last piece -- which is only known once the next tile has
            // chosen it; a carried-in first piece that is alone in its list keeps its count
            for (unsigned w = 0; w < QT_WAVES; w++) chain_recs[w] = (carry_in[t - t_lo][w] && assign[t - t_lo][w].size() == 1) ? next_chain[w] : 0;
            uint64_t mx = 0, sum = 0;
            for (unsigned w = 0; w < QT_WAVES; w++) {
                mx = std::max(mx, load[w]);
                sum += load[w];
            }
            Q.cost_sum_max += mx;
            Q.cost_sum_mean += sum / QT_WAVES;
            uint64_t rmx = 0;
            for (unsigned w = 0; w < QT_WAVES; w++) {
                uint64_t r = 0;
                for (uint32_t p : lists[w]) r += piece_recs(p);
                rmx = std::max(rmx, r);
                Q.rec_sum_total += r;
            }
            Q.rec_sum_max += rmx;
            Q.tile_phases++;
        }
        for (unsigned w = 0; w < QT_WAVES; w++) {
            Q.streams.push_back({(uint32_t)Q.recs.size(), (uint32_t)Q.pieces.size()});
            const size_t stream_first = Q.recs.size();
            // A piece's DESCRIPTOR rides in the record stream, as a no-op record (QT_DESC: x = 1, v = x, no accumulation) in front of the
            // piece: its weight words hold the gate cells and the piece control word.  The kernel has it in registers four steps before
            // its turn like any record, so the piece's gate cells are requested without a dependent global load in between (a descriptor
            // array of its own cost one L2 round trip per piece end, waited for on the spot: the fit had a piece end at 4 010 cycles).  A run
            // of plain records needs a special record in front of it that announces its length anyway, and nothing precedes the first
            // record of a stream.
            bool descriptor_due = true;  // the next piece that does not go on from the previous tile opens with its descriptor
            for (uint32_t t = t_lo; t < t_hi; t++) {
                const uint32_t tile = tiles[t];
                const std::vector<uint32_t>& mine = assign[t - t_lo][w];
                for (size_t pi = 0; pi < mine.size(); pi++) {
                    const uint32_t p = mine[pi];
                    const Piece& pc = pieces[p];
                    const std::vector<uint32_t>& gates = sg_gates[pc.sg];
                    const bool continues = pi == 0 && carry_in[t - t_lo][w];                                   // no descriptor of its own
                    const bool goes_on = pi + 1 == mine.size() && t + 1 < t_hi && carry_in[t + 1 - t_lo][w];  // no end of its own
                    // a piece that goes on from the previous tile has no descriptor of its own: its foreign cells are loaded directly instead
                    size_t foreign_cells = 0;
                    for (uint32_t b : pc.foreign) foreign_cells += monos[b].cells[0] != NONE;
                    const bool slots = !continues && gates.size() + foreign_cells <= 4;
                    if (continues && descriptor_due) throw std::runtime_error("quotient_plan: a stream's first piece cannot continue another");
                    if (!continues) {
                        QTPiece d = {0, {0, 0, 0, 0}, {0, 0, 0}};
                        uint32_t compl_mask = 0;
                        for (size_t g = 0; g < gates.size(); g++) {
                            d.gate[g] = gates[g] & (REF_COL_MASK | REF_NEXT);
                            if (gates[g] & REF_COMPL) compl_mask |= 1u << g;
                        }
                        uint32_t n_foreign = 0;
                        if (slots)
                            for (uint32_t b : pc.foreign)
                                if (monos[b].cells[0] != NONE) d.gate[gates.size() + n_foreign++] = monos[b].cells[0] & (REF_COL_MASK | REF_NEXT);
                        d.ctl = sg_kind[pc.sg] | ((uint32_t)gates.size() << 2) | (compl_mask << 5) | (n_foreign << QT_FOREIGN_SHIFT);
                        Q.recs.push_back({QT_DESC, 0, {d.gate[0], d.gate[1], d.gate[2], d.gate[3], d.ctl, 0}});
                        rec_c_begin.push_back(0);
                        rec_c_end.push_back(0);
                        descriptor_due = false;
                        Q.pieces.push_back(d);
                        Q.n_direct_loads += gates.size() + n_foreign;
                        Q.n_piece_ends++;
                    }
                    const size_t n_own = pc.m_end - pc.m_begin;
                    size_t slot_at = gates.size();
                    for (size_t fi = 0; fi < pc.foreign.size(); fi++) {
                        const Mono& mo = monos[pc.foreign[fi]];
                        const uint32_t cell = mo.cells[0];
                        const bool last = n_own == 0 && fi + 1 == pc.foreign.size() && !goes_on;
                        if (cell == NONE) {  // the constant term: the tile's column of ones
                            push_rec(QT_ONES_SLOT * QT_TILE_ROWS * 8 | (last ? QT_END : 0u), 0, mo.c_begin, mo.c_end);
                            continue;
                        }
                        uint32_t ctl = QT_SRC_GLOBAL | ((cell & REF_NEXT) ? QT_NEXT : 0u) | (last ? QT_END : 0u);
                        push_rec(ctl, slots ? (QT_AUX_SLOT | (uint32_t)slot_at++) : (cell & REF_COL_MASK), mo.c_begin, mo.c_end);
                        if (!slots) Q.n_direct_loads++;
                    }
                    // emission order: as stored (single cells, fast pairs, the rest) -- except that a piece whose last monomial would be a
                    // fast pair ends on one of its single cells instead: the record that carries QT_END goes through the generic step,
                    // and a plain record does that cheaply while a pair's two records do not
                    std::vector<uint32_t> order_m;
                    order_m.reserve(n_own);
                    for (uint32_t b = pc.m_begin; b < pc.m_end; b++) order_m.push_back(b);
                    if (n_own >= 2 && !goes_on) {
                        const Mono& tail = monos[order_m.back()];
                        const bool tail_is_pair = tail.cells[1] != NONE && tail.cells[2] == NONE && tile_of(tail.cells[0]) == tile && tile_of(tail.cells[1]) == tile;
                        const Mono& head = monos[order_m.front()];
                        if (tail_is_pair && head.cells[1] == NONE) {
                            order_m.erase(order_m.begin());
                            order_m.push_back(pc.m_begin);
                        }
                    }
                    for (size_t oi = 0; oi < order_m.size(); oi++) {
                        const uint32_t b = order_m[oi];
                        const Mono& mo = monos[b];
                        const bool last = oi + 1 == order_m.size() && !goes_on;
                        if (mo.cells[0] == NONE) {  // the constant term: the tile's column of ones
                            push_rec(QT_ONES_SLOT * QT_TILE_ROWS * 8 | (last ? QT_END : 0u), 0, mo.c_begin, mo.c_end);
                            continue;
                        }
                        // factors inside the tile first (LDS), direct loads after; the last factor carries the weights
                        uint32_t f[3], nf = 0;
                        for (int q = 0; q < 3; q++)
                            if (mo.cells[q] != NONE && tile_of(mo.cells[q]) == tile) f[nf++] = mo.cells[q];
                        for (int q = 0; q < 3; q++)
                            if (mo.cells[q] != NONE && tile_of(mo.cells[q]) != tile) f[nf++] = mo.cells[q];
                        for (uint32_t q = 0; q < nf; q++) {
                            const uint32_t cell = f[q];
                            uint32_t ctl = 0, aux = 0;
                            if (tile_of(cell) == tile) {
                                const uint32_t slot = (cell & REF_COL_MASK) - tile * QT_TILE_COLS;
                                ctl = slot * QT_TILE_ROWS * 8 + ((cell & REF_NEXT) ? 8u : 0u);
                                if (cell & REF_NEXT) ctl |= QT_NEXT;
                                aux = 0;
                                Q.n_cell_records++;
                            } else {
                                ctl = QT_SRC_GLOBAL | ((cell & REF_NEXT) ? QT_NEXT : 0u);
                                aux = cell & REF_COL_MASK;
                                Q.n_direct_loads++;
                            }
                            const bool final_factor = q + 1 == nf;
                            if (nf > 1) {
                                if (q == 0) ctl |= QT_SETV;
                                else if (!final_factor) ctl |= QT_SETV | QT_MULV;
                                else ctl |= QT_MULV;
                            }
                            if (final_factor && last) ctl |= QT_END;
                            push_rec(ctl, aux, final_factor ? mo.c_begin : 0, final_factor ? mo.c_end : 0);
                        }
                    }
                }
                if (descriptor_due && Q.recs.size() == stream_first) push_rec(QT_SRC_ONE | QT_SETV, 0, 0, 0);  // an empty first tile: the stream still opens with a special record
                push_rec(QT_TILE, 0, 0, 0);
                // the wave enters tile t + 1 behind this record and stages its share of tile t + 2 then: that tile's first column
                Q.recs.back().w[0] = t + 2 < t_hi ? tiles[t + 2] * QT_TILE_COLS : 0xFFFFFFFFu;
            }
            push_rec(QT_STOP, 0, 0, 0);
            // announcements: every special record says how many plain records follow it (QT_RUN_SHIFT) and, in aux, how many fast pairs
            // follow those; a plain run longer than QT_MAX_RUN is cut by more no-ops (the last of them carries the pairs)
            {
                const size_t n_in = Q.recs.size() - stream_first;
                auto in = [&](size_t i) -> const QTRec& { return Q.recs[stream_first + i]; };
                auto is_plain = [&](size_t i) { return (in(i).ctl & QT_SPECIAL) == 0; };
                auto is_pair_a = [&](size_t i) { return (in(i).ctl & QT_SPECIAL) == QT_SETV; };   // first factor, a cell of the tile
                auto is_pair_b = [&](size_t i) { return (in(i).ctl & QT_SPECIAL) == QT_MULV; };   // second factor, a cell of the tile, not the piece's last record
                auto is_dpair_b = [&](size_t i) { return (in(i).ctl & QT_SPECIAL) == (QT_SRC_GLOBAL | QT_MULV) && !(in(i).aux & QT_AUX_SLOT); };   // second factor: a direct load
                std::vector<QTRec> out;
                std::vector<uint32_t> ocb, oce;
                out.reserve(n_in + 8);
                auto emit = [&](const QTRec& rec, uint32_t cb, uint32_t ce) {
                    out.push_back(rec);
                    ocb.push_back(cb);
                    oce.push_back(ce);
                };
                size_t i = 0;
                while (i < n_in) {
                    QTRec rec = in(i);
                    if (is_plain(i)) throw std::runtime_error("quotient_plan: a plain record nobody announced");
                    // what follows this special record: plains, then fast pairs
                    size_t p_end = i + 1;
                    while (p_end < n_in && is_plain(p_end)) p_end++;
                    size_t q_end = p_end;
                    while (q_end + 1 < n_in && is_pair_a(q_end) && is_pair_b(q_end + 1)) q_end += 2;
                    uint32_t plains = (uint32_t)(p_end - i - 1), pairs = (uint32_t)((q_end - p_end) / 2);
                    const bool can_announce_pairs = !(rec.ctl & QT_SRC_GLOBAL);   // its aux is a column (or a descriptor slot)
                    if (!can_announce_pairs || pairs > QT_AUX_PAIRS_MASK) pairs = 0;
                    // direct pairs behind the fast pairs (two per turn of the kernel's loop, and its fast-pair loop leaves an odd pair to the
                    // generic step, which would forget the announcement)
                    uint32_t dpairs = 0;
                    if (can_announce_pairs && pairs % 2 == 0 && p_end + 2 * (size_t)pairs == q_end) {
                        size_t d_end = q_end;
                        while (d_end + 1 < n_in && is_pair_a(d_end) && is_dpair_b(d_end + 1)) d_end += 2;
                        dpairs = std::min<uint32_t>((uint32_t)((d_end - q_end) / 2) & ~1u, QT_AUX_DPAIRS_MASK & ~1u);
                    }
                    uint32_t first = std::min<uint32_t>(plains, QT_MAX_RUN);
                    rec.ctl |= first << QT_RUN_SHIFT;
                    if (plains <= QT_MAX_RUN && can_announce_pairs) rec.aux = pairs | (dpairs << QT_AUX_DPAIRS_SHIFT);
                    emit(rec, rec_c_begin[stream_first + i], rec_c_end[stream_first + i]);
                    size_t at = i + 1;
                    uint32_t left = plains - first;
                    for (uint32_t done = 0; at < p_end; at++) {
                        if (done == QT_MAX_RUN) {
                            const uint32_t more = std::min<uint32_t>(left, QT_MAX_RUN);
                            left -= more;
                            emit({QT_SRC_ONE | QT_SETV | (more << QT_RUN_SHIFT), left == 0 ? (pairs | (dpairs << QT_AUX_DPAIRS_SHIFT)) : 0u, {0, 0, 0, 0, 0, 0}}, 0, 0);
                            done = 0;
                        }
                        emit(in(at), rec_c_begin[stream_first + at], rec_c_end[stream_first + at]);
                        done++;
                    }
                    // the announced pairs go out as they are (their own control words announce nothing); unannounced ones are
                    // ordinary special records and get their turn in this loop
                    const size_t fast_end = p_end + 2 * (size_t)pairs + 2 * (size_t)dpairs;   // (the direct pairs likewise)
                    for (; at < fast_end; at++) emit(in(at), rec_c_begin[stream_first + at], rec_c_end[stream_first + at]);
                    i = at;
                }
                // Factors outside the tile that no announced pair run covers (odd pairs, pairs that end a piece, degree-3 monomials) were
                // loaded where they are used: one exposed HBM / L2 round trip per record (the fit has such a record at ~ 8 000 cycles, 7 % of
                // the evaluators' time for 2 619 records).  Where the piece's descriptor has a free slot -- gates + absorbed cells < 4 -- the
                // cell goes there instead: requested with the gates when the piece starts, in a register long before its record's turn.
                {
                    const size_t NO_DESC = (size_t)-1;
                    size_t desc = NO_DESC;
                    uint32_t a_plain = 0, a_fast = 0, a_dfast = 0;  // what the kernel's straight-line loops will take without looking
                    for (size_t k = 0; k < out.size(); k++) {
                        QTRec& r = out[k];
                        const uint32_t ctl = r.ctl;
                        if ((ctl & QT_SPECIAL) == 0) {
                            if (a_plain) a_plain--;
                            continue;
                        }
                        if (a_plain == 0 && a_fast) { a_fast--; continue; }
                        if (a_plain == 0 && a_dfast) { a_dfast--; continue; }
                        a_plain = ctl >> QT_RUN_SHIFT;
                        a_fast = (ctl & QT_SRC_GLOBAL) ? 0u : 2u * (r.aux & QT_AUX_PAIRS_MASK);
                        a_dfast = (ctl & QT_SRC_GLOBAL) ? 0u : 2u * ((r.aux >> QT_AUX_DPAIRS_SHIFT) & QT_AUX_DPAIRS_MASK);
                        if ((ctl & (QT_DESC | QT_SRC_GLOBAL | QT_MULV | QT_END)) == QT_DESC) {
                            desc = k;
                            continue;
                        }
                        if (ctl & (QT_TILE | QT_STOP)) continue;  // a piece that goes on in the next tile keeps its descriptor
                        if ((ctl & QT_SRC_GLOBAL) && !(r.aux & QT_AUX_SLOT) && desc != NO_DESC) {
                            QTRec& d = out[desc];
                            const uint32_t pctl = d.w[4], ng = (pctl >> 2) & 7u, nf = (pctl >> QT_FOREIGN_SHIFT) & 7u;
                            const uint32_t ref = (r.aux & REF_COL_MASK) | ((ctl & QT_NEXT) ? REF_NEXT : 0u);
                            uint32_t slot = 4;
                            for (uint32_t q = ng; q < ng + nf; q++)
                                if (d.w[q] == ref) slot = q;  // the same cell twice in one piece: one slot
                            if (slot == 4 && ng + nf < 4) {
                                slot = ng + nf;
                                d.w[slot] = ref;
                                d.w[4] = pctl + (1u << QT_FOREIGN_SHIFT);
                                Q.n_direct_loads++;
                            }
                            if (slot < 4) {
                                r.aux = QT_AUX_SLOT | slot;
                                if (Q.n_direct_loads) Q.n_direct_loads--;
                            }
                        }
                        if (ctl & QT_END) desc = NO_DESC;
                    }
                }
                Q.recs.resize(stream_first);
                rec_c_begin.resize(stream_first);
                rec_c_end.resize(stream_first);
                Q.recs.insert(Q.recs.end(), out.begin(), out.end());
                rec_c_begin.insert(rec_c_begin.end(), ocb.begin(), ocb.end());
                rec_c_end.insert(rec_c_end.end(), oce.begin(), oce.end());
            }
            // the kernel fetches a few records / one descriptor beyond the end of a stream
            for (int z = 0; z < 4; z++) push_rec(QT_STOP, 0, 0, 0);
            Q.pieces.push_back({0, {0, 0, 0, 0}, {0, 0, 0}});
        }
    }
    Q.chunk_tile_off.push_back((uint32_t)Q.tile_list.size());
    for (int z = 0; z < 64; z++) push_rec(QT_STOP, 0, 0, 0);  // the kernel's record ring is filled up to 32 records ahead
    // contributions in record order (CSR)
    Q.contrib_off.resize(Q.recs.size() + 1);
    Q.contribs.reserve(contribs.size());
    for (size_t r = 0; r < Q.recs.size(); r++) {
        Q.contrib_off[r] = (uint32_t)Q.contribs.size();
        Q.contribs.insert(Q.contribs.end(), contribs.begin() + rec_c_begin[r], contribs.begin() + rec_c_end[r]);
    }
    Q.contrib_off[Q.recs.size()] = (uint32_t)Q.contribs.size();
    return Q;
}

// ---- host replay (tests): exactly what the kernel does for ONE point, with the accumulators the kernel uses
// local / next: the frame; masks = {1, z_last, L_first, L_last}.  weights are derived here from alphas and public inputs.
inline void quotient_plan_weights_host(const AirProgram& P, QTPlan& Q, const gl_t alphas[2], const gl_t* pis) {
    std::vector<gl_t> apow[2];
    for (int j = 0; j < 2; j++) {
        apow[j].resize(std::max<uint32_t>(1, P.n_constraints));
        apow[j][0] = 1;
        for (uint32_t e = 1; e < P.n_constraints; e++) apow[j][e] = gl_mul(apow[j][e - 1], alphas[j]);
    }
    for (size_t r = 0; r < Q.recs.size(); r++) {
        if (Q.contrib_off[r] == Q.contrib_off[r + 1]) continue;  // no weights: markers, no-ops, and the descriptors, whose words hold the gate cells
        for (int j = 0; j < 2; j++) {
            gl_t w = 0;
            for (uint32_t c = Q.contrib_off[r]; c < Q.contrib_off[r + 1]; c++) {
                const uint32_t ck = Q.contribs[c].coef & 7u, idx = Q.contribs[c].coef >> 3;
                gl_t coef = ck == CK_PLUS ? 1 : ck == CK_MINUS ? GL_P - 1 : ck == CK_CONST ? P.consts[idx] : ck == CK_PI ? pis[idx] : gl_neg(pis[idx]);
                w = gl_add(w, gl_mul(coef, apow[j][Q.contribs[c].e]));
            }
            const uint32_t M = (1u << QT_LIMB_BITS) - 1;
            Q.recs[r].w[3 * j + 0] = (uint32_t)(w & M);
            Q.recs[r].w[3 * j + 1] = (uint32_t)((w >> QT_LIMB_BITS) & M);
            Q.recs[r].w[3 * j + 2] = (uint32_t)(w >> (2 * QT_LIMB_BITS));
        }
    }
}

// value of the six partial sums S[i][l] (x limb i in {lo, hi}, weight limb l) as a field element
inline gl_t qt_fold_sums_host(const uint64_t S[6]) {
    gl_t r = 0;
    for (int i = 0; i < 2; i++)
        for (int l = 0; l < 3; l++) {
            gl_t v = gl_from_u64(S[3 * i + l]);
            v = gl_mul(v, gl_pow(2, 32 * i + QT_LIMB_BITS * l));
            r = gl_add(r, v);
        }
    return r;
}

inline bool quotient_plan_eval_host(const QTPlan& Q, const gl_t* local, const gl_t* next, const gl_t masks[4], gl_t acc[2]) {
    acc[0] = acc[1] = 0;
    for (unsigned c = 0; c < Q.n_chunks; c++)
        for (unsigned w = 0; w < QT_WAVES; w++) {
            const QTStream& st = Q.streams[c * QT_WAVES + w];
            const QTRec* rec = &Q.recs[st.rec_off];
            QTPiece cur = {0, {0, 0, 0, 0}, {0, 0, 0}};  // the descriptor of the piece under way, from the stream
            const QTPiece* pc = &cur;
            bool have_desc = false;
            uint32_t ti = Q.chunk_tile_off[c];
            uint64_t S[2][6] = {{0}};
            unsigned n_in_piece = 0;
            bool in_product = false;
            gl_t v = 1;
            uint32_t announced = 0;  // plain records the last special record said would follow
            uint32_t fast = 0;       // records of announced fast pairs still to come (two per pair): evaluated without a look at their flags
            uint32_t dfast = 0;      // the same for the announced direct pairs behind them
            if ((rec->ctl & QT_SPECIAL) == 0) return false;  // a stream opens with a special record (the kernel knows no run length before it)
            for (;; rec++) {
                const uint32_t ctl = rec->ctl;
                if ((ctl & QT_SPECIAL) == 0) {  // plain: must have been announced, and carries no run length itself
                    if (announced == 0 || (ctl >> QT_RUN_SHIFT) != 0) return false;
                    announced--;
                } else if (announced == 0 && fast != 0) {
                    // inside a fast run: exactly (SETV, cell in the tile) then (MULV, cell in the tile), nothing announced, not a piece's end
                    const uint32_t want = (fast & 1u) ? QT_MULV : QT_SETV;
                    if ((ctl & QT_SPECIAL) != want || (ctl >> QT_RUN_SHIFT) != 0 || rec->aux != 0) return false;
                    fast--;
                } else if (announced == 0 && dfast != 0) {
                    // inside a direct run: (SETV, cell in the tile) then (SRC_GLOBAL | MULV, a column), nothing announced, not a piece's end
                    if ((ctl >> QT_RUN_SHIFT) != 0) return false;
                    if (dfast & 1u) {
                        if ((ctl & QT_SPECIAL) != (QT_SRC_GLOBAL | QT_MULV) || (rec->aux & QT_AUX_SLOT)) return false;
                    } else if ((ctl & QT_SPECIAL) != QT_SETV || rec->aux != 0) return false;
                    dfast--;
                } else {
                    if (announced != 0) return false;  // a special record inside an announced run: the kernel would not look at its flags
                    announced = ctl >> QT_RUN_SHIFT;
                    if (announced > QT_MAX_RUN) return false;
                    if (!(ctl & QT_SRC_GLOBAL)) {
                        const uint32_t pairs = rec->aux & QT_AUX_PAIRS_MASK, dpairs = (rec->aux >> QT_AUX_DPAIRS_SHIFT) & QT_AUX_DPAIRS_MASK;
                        if ((rec->aux >> 31) || (dpairs & 1u) || (dpairs && (pairs & 1u))) return false;
                        fast = 2 * pairs;
                        dfast = 2 * dpairs;
                    }
                }
                if (ctl & QT_STOP) break;
                if (ctl & QT_TILE) {
                    if (in_product) return false;  // a piece may go on in the next tile (same supergroup), a monomial may not
                    ti++;
                    if (rec->w[0] != (ti + 1 < Q.chunk_tile_off[c + 1] ? Q.tile_list[ti + 1] * QT_TILE_COLS : 0xFFFFFFFFu)) return false;  // the tile to stage next
                    continue;
                }
                if ((ctl & (QT_DESC | QT_SRC_GLOBAL | QT_MULV | QT_END)) == QT_DESC) {  // the next piece's descriptor
                    if (n_in_piece || in_product || have_desc) return false;
                    cur.gate[0] = rec->w[0], cur.gate[1] = rec->w[1], cur.gate[2] = rec->w[2], cur.gate[3] = rec->w[3];
                    cur.ctl = rec->w[4];
                    if (rec->w[5]) return false;
                    have_desc = true;
                    v = 1;
                    continue;
                }
                if (ti >= Q.chunk_tile_off[c + 1]) return false;
                const uint32_t tile = Q.tile_list[ti];
                gl_t x;
                if (ctl & QT_SRC_ONE) x = 1;
                else if ((ctl & QT_SRC_GLOBAL) && (rec->aux & QT_AUX_SLOT)) {  // a foreign cell: descriptor slot of the piece under way
                    const uint32_t slot = rec->aux & 3u, ng = (pc->ctl >> 2) & 7u, nf = (pc->ctl >> QT_FOREIGN_SHIFT) & 7u;
                    if ((rec->aux & ~(QT_AUX_SLOT | 3u)) != 0 || slot < ng || slot >= ng + nf || ng + nf > 4) return false;
                    if (!!(pc->gate[slot] & REF_NEXT) != !!(ctl & QT_NEXT)) return false;
                    x = ((pc->gate[slot] & REF_NEXT) ? next : local)[pc->gate[slot] & REF_COL_MASK];
                } else if (ctl & QT_SRC_GLOBAL) x = ((ctl & QT_NEXT) ? next : local)[rec->aux];
                else {
                    const uint32_t off = ctl & QT_OFF_MASK, slot = off / (QT_TILE_ROWS * 8), row = (off % (QT_TILE_ROWS * 8)) / 8;
                    if (row > 1 || slot > QT_ONES_SLOT || (row == 1) != !!(ctl & QT_NEXT)) return false;
                    if (slot == QT_ONES_SLOT) {
                        if (row) return false;
                        x = 1;
                    } else {
                        const uint32_t col = tile * QT_TILE_COLS + slot;
                        if (col >= Q.n_cols) return false;
                        x = (row ? next : local)[col];
                    }
                }
                if (ctl & QT_MULV) x = gl_mul(v, x);
                if (ctl & QT_SETV) {
                    v = x;
                    if (ctl & QT_END) return false;
                    // (x = 1 with SETV alone is the planner's no-op -- a stream's opening record, or the record that cuts a run longer
                    // than QT_MAX_RUN --, not the first factor of a product)
                    if ((ctl & (QT_SRC_ONE | QT_MULV)) != QT_SRC_ONE) in_product = true;
                    continue;
                }
                in_product = false;
                const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32);
                for (int j = 0; j < 2; j++)
                    for (int l = 0; l < 3; l++) {
                        if (rec->w[3 * j + l] >> QT_LIMB_BITS) return false;
                        S[j][l] += (uint64_t)x0 * rec->w[3 * j + l];
                        S[j][3 + l] += (uint64_t)x1 * rec->w[3 * j + l];
                    }
                if (++n_in_piece > QT_MAX_CHAIN) return false;  // the device fold's chains would wrap (see QT_MAX_CHAIN)
                if (!have_desc) return false;  // a record that accumulates belongs to a piece, and a piece opens with its descriptor
                if (ctl & QT_END) {
                    const uint32_t kind = pc->ctl & 3u, ng = (pc->ctl >> 2) & 7u, cm = (pc->ctl >> 5) & 15u;
                    gl_t G = masks[kind];
                    for (uint32_t g = 0; g < ng; g++) {
                        gl_t gv = ((pc->gate[g] & REF_NEXT) ? next : local)[pc->gate[g] & REF_COL_MASK];
                        if (cm & (1u << g)) gv = gl_sub(1, gv);
                        G = gl_mul(G, gv);
                    }
                    for (int j = 0; j < 2; j++) {
                        acc[j] = gl_add(acc[j], gl_mul(G, qt_fold_sums_host(S[j])));
                        for (int l = 0; l < 6; l++) S[j][l] = 0;
                    }
                    n_in_piece = 0;
                    have_desc = false;
                }
            }
            if (n_in_piece || announced || fast) return false;
        }
    return true;
}

}  // namespace starkhip
