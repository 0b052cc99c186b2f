// Per (chunk, tile, wave) census of the tiled quotient plan's record streams (csrc/quotient_plan.h), as CSV on stdout: what every
// evaluator wave does between two tile barriers, counted on the streams the kernel reads.  Together with the per-tile clocks of the
// profiling variant (tools/quotient_wave_prof.py --tiles) this calibrates the planner's cost model.
//   g++ -O2 -std=c++17 -Iinclude -Istarky_bls12_381_amd/csrc tools/plan_streams.cpp -o build/plan_streams -Lstarky_bls12_381_amd -lstarkhip -Wl,-rpath,$PWD/starky_bls12_381_amd
//   build/plan_streams 5 16 > /tmp/final_exp_streams.csv        (AIR id, chunks)
#include <stdio.h>
#include <stdlib.h>

#include "airs.h"
#include "quotient_plan.h"

using namespace starkhip;

int main(int argc, char** argv) {
    const int air_id = argc > 1 ? atoi(argv[1]) : STARKHIP_AIR_FINAL_EXP;
    const unsigned chunks = argc > 2 ? (unsigned)atoi(argv[2]) : 16;
    const AirInfo* a = air_get(air_id);
    if (!a) return 1;
    const QTPlan Q = build_quotient_plan(a->prog, chunks);
    printf("chunk,tile_in_chunk,tile,wave,records,plain4,plain_tail,fast_pairs2,fast_pair_odd,dpairs2,special,piece_ends,ends_on_plain,direct,slot_cells,setv_mulv,noop\n");
    for (unsigned c = 0; c < Q.n_chunks; c++)
        for (unsigned w = 0; w < QT_WAVES; w++) {
            const QTStream& st = Q.streams[c * QT_WAVES + w];
            const QTRec* rec = &Q.recs[st.rec_off];
            unsigned ti = 0;
            // what the kernel's control flow does with the stream
            uint32_t n_plain = 0, n_pairs = 0, n_dpairs = 0;
            struct Cnt { unsigned records = 0, plain4 = 0, plain_tail = 0, fast2 = 0, fast_odd = 0, dp2 = 0, special = 0, ends = 0, ends_plain = 0, direct = 0, slot = 0, mulv = 0, noop = 0; } k;
            for (;; ) {
                // straight-line loops
                while (n_plain >= 4) { k.plain4++; k.records += 4; rec += 4; n_plain -= 4; }
                if (n_plain == 0) while (n_pairs >= 2) { k.fast2++; k.records += 4; rec += 4; n_pairs -= 2; }
                if (n_plain == 0 && n_pairs == 0) while (n_dpairs >= 2) { k.dp2++; k.records += 4; rec += 4; n_dpairs -= 2; }
                // generic step
                k.records++;
                if (n_plain != 0) { n_plain--; k.plain_tail++; rec++; continue; }
                const uint32_t ctl = rec->ctl, aux = rec->aux;
                n_plain = ctl >> QT_RUN_SHIFT;
                n_pairs = (ctl & QT_SRC_GLOBAL) ? 0u : (aux & QT_AUX_PAIRS_MASK);
                n_dpairs = (ctl & QT_SRC_GLOBAL) ? 0u : ((aux >> QT_AUX_DPAIRS_SHIFT) & QT_AUX_DPAIRS_MASK);
                rec++;
                if (ctl & (QT_TILE | QT_STOP)) {
                    if (ctl & QT_TILE) {
                        printf("%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u,%u\n", c, ti, Q.tile_list[Q.chunk_tile_off[c] + ti], w, k.records, k.plain4, k.plain_tail, k.fast2, k.fast_odd,
                               k.dp2, k.special, k.ends, k.ends_plain, k.direct, k.slot, k.mulv, k.noop);
                        k = Cnt();
                        ti++;
                        if (ti >= Q.chunk_tile_off[c + 1] - Q.chunk_tile_off[c]) break;
                        continue;
                    }
                    break;
                }
                if ((ctl & QT_ODD_SOURCE) == 0 && (ctl & QT_END)) k.ends_plain++;   // a plain cell that ends its piece
                else if ((ctl & QT_SPECIAL) == QT_SETV || (ctl & QT_SPECIAL) == QT_MULV) k.fast_odd++;   // a pair's record outside an announced run
                else if ((ctl & (QT_SRC_ONE | QT_SETV)) == (QT_SRC_ONE | QT_SETV) && !(ctl & QT_MULV)) k.noop++;
                else k.special++;
                if (ctl & QT_END) k.ends++;
                if (ctl & QT_SRC_GLOBAL) (aux & QT_AUX_SLOT) ? k.slot++ : k.direct++;
                if (ctl & QT_MULV) k.mulv++;
            }
        }
    return 0;
}
