#!/usr/bin/env python3
"""Statistics of the flat constraint programs (air_ir.h): how many distinct GROUP shapes each AIR has once
column numbers are replaced by first-appearance indices.  Used to size the shape-specialised quotient evaluator."""
import hashlib
import os
import sys
from collections import Counter

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import starky_bls12_381_amd as S  # noqa: E402


def load(air):
    b = np.asarray(S.air_program(air), dtype=np.uint64)
    nconsts, ncode = int(b[5]), int(b[6])
    return b[8 + nconsts:8 + nconsts + (ncode + 1) // 2].view(np.uint32)[:ncode].tolist()


def groups(code):
    i = 0
    while i < len(code):
        gw = code[i]
        if gw == 0:
            return
        start = i
        i += 1
        ng, m = (gw >> 8) & 255, gw >> 16
        i += ng
        for _ in range(m):
            while True:
                tw = code[i]
                i += 1 + (tw & 3)
                if tw & 32:
                    break
        yield code[start:i]


def shape_of(g):
    cmap = {}
    sig = [g[0]]
    i = 1
    ng, m = (g[0] >> 8) & 255, g[0] >> 16

    def ref(r):
        col = r & 0xFFFFFF
        if col not in cmap:
            cmap[col] = len(cmap)
        return (r & ~0xFFFFFF) | cmap[col]
    for _ in range(ng):
        sig.append(ref(g[i]))
        i += 1
    nt = 0
    for _ in range(m):
        while True:
            tw = g[i]
            i += 1
            sig.append(tw)
            for _r in range(tw & 3):
                sig.append(ref(g[i]))
                i += 1
            nt += 1
            if tw & 32:
                break
    return tuple(sig), nt, list(cmap.keys())


if __name__ == "__main__":
    for air, name in [(S.AIR_FINAL_EXP, "finalexp"), (S.AIR_MILLER_LOOP, "miller"), (S.AIR_PAIRING_PRECOMP, "precomp"), (S.AIR_FP12_MUL, "fp12mul")]:
        code = load(air)
        cnt, terms, params = Counter(), {}, {}
        for g in groups(code):
            sig, nt, cols = shape_of(g)
            h = hashlib.md5(repr(sig).encode()).hexdigest()
            cnt[h] += 1
            terms[h] = nt
            params[h] = len(cols)
        tot = sum(cnt[h] * terms[h] for h in cnt)
        print(name, "groups", sum(cnt.values()), "distinct shapes", len(cnt), "sum terms over shapes", sum(terms.values()),
              "max params", max(params.values()))
        cov, acc = 0, []
        for h, c in sorted(cnt.items(), key=lambda kv: -kv[1] * terms[kv[0]])[:40]:
            cov += c * terms[h]
            acc.append((c, terms[h], params[h], round(cov / tot, 3)))
        print("  (count, terms, params, cumulative term coverage):", acc)
