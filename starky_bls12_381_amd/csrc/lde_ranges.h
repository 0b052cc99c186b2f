// The launch plan of an LDE whose input columns are parked inside the buffer it writes (prover.hip: run_lde_trace).  Host only, no
// device state: prover.hip launches from it, starkhip_lde_launch_ranges hands it to the CPU test that checks its invariant.
//
// Parked column c' lies at words [(R - 1) C n + c' n, + n) of the LDE buffer (R = 2^rate_bits cosets, C columns of n rows); the LDE
// block of column c is [R c n, R (c + 1) n), i.e. it covers the parked columns R c - (R - 1) C + j, j < R (those >= 0).  Workgroups of
// one launch run in no particular order, so a launch over [a, b) may cover only parked columns an EARLIER launch has read (c' < a):
// R b <= a + (R - 1) C.  The geometric series this gives (3/4, 3/16, 3/64 .. of the columns for R = 4) stops when at most
// lde_tail_columns(C) are left: those are copied aside and read from the copy by one last launch, whose blocks may then cover any
// parked column.
#pragma once
#include <stddef.h>

#include <algorithm>
#include <vector>

namespace starkhip {

inline size_t lde_tail_columns(size_t C) { return std::max<size_t>(C / 32, 64); }

struct LdeLaunch {
    size_t a, b;     // columns [a, b)
    bool from_copy;  // reads the copy of the parked columns [a, C) instead of the parked columns themselves
};

inline std::vector<LdeLaunch> lde_launch_plan(size_t C, unsigned rate_bits) {
    std::vector<LdeLaunch> plan;
    const size_t R = (size_t)1 << rate_bits;
    size_t a = 0;
    if (R >= 2)
        while (C - a > lde_tail_columns(C)) {
            const size_t b = (a + (R - 1) * C) / R;  // > a while C - a >= 2
            plan.push_back({a, b, false});
            a = b;
        }
    if (a < C) plan.push_back({a, C, true});
    return plan;
}

}  // namespace starkhip
