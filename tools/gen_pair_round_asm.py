#!/usr/bin/env python3
"""Generates csrc/pair_round_asm.inc: the rounds of the PAIR form of the Poseidon permutation -- lanes l and l + 32 of a wave share one
permutation (lane l holds state elements 0 .. 5, lane l + 32 elements 6 .. 11), one 256-register wave per SIMD: what a LONE commitment of
>= 32 768 leaves takes (poseidon_dev.h, kernels_hash.hip: leaf_hash_pair_kernel).  Built with the instruction
model, list scheduler and interpreter of tools/gen_lane_round_asm.py / gen_row_round_asm.py; every block is executed on two lanes
against the rounds in Python integers before it is printed.

  full round      six S-boxes per lane; the circulant layer on the matrix pipe as in the lane form, but with a DENSE weight tile: the
                  two lane halves supply the two K halves of v_mfma_i32_32x32x32_i8 (elements 0 .. 5 and 6 .. 11 of the same column),
                  TWO byte planes per instruction, and the result rows chosen so that outputs 0 .. 5 land in the lower half-wave and
                  6 .. 11 in the upper one
  merged four     four partial rounds at once (poseidon_merged.h): the three dot products are partial sums over a lane's own six
                  elements added across the pair with v_permlane32_swap_b32; the dense 12 x 12 layer needs the partner's six elements
                  (12 swaps) and computes six outputs per lane
  partial round   the two plain ones: S-box on element 0 (lower half only), layer on the matrix pipe

    python tools/gen_pair_round_asm.py > starky_bls12_381_amd/csrc/pair_round_asm.inc
"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_row_round_asm as G  # noqa: E402
import gen_lane_round_asm as L  # noqa: E402

Ins, v, vp, sp = G.Ins, G.v, G.vp, G.sp
P, M32, M64 = G.P, G.M32, G.M64
ABLATE = os.environ.get("PAIR_ABLATE", "")         # timing experiments only (wrong results): "seeds", "rows"
# slots between an LDS load and its first use that the scheduler aims for (a wave alone on its SIMD has nobody to hide a longer wait)
ROUND_LOAD_LATENCY = int(os.environ.get("PAIR_LOAD_LATENCY", 16))
TRIPLE_LOAD_LATENCY = int(os.environ.get("PAIR_TRIPLE_LOAD_LATENCY", os.environ.get("PAIR_LOAD_LATENCY", 56)))   # the merged rounds; measured with triples: 16 -> 123.9 ms, 28 -> 122.0, 56 -> 121.6
L.VALU_RAW = int(os.environ.get("PAIR_VALU_RAW", L.VALU_RAW))
L.SGPR_RAW = int(os.environ.get("PAIR_SGPR_RAW", L.SGPR_RAW))
G.NL = 2   # lane 0 = lower half-wave (elements 0 .. 5), lane 1 = upper half-wave (elements 6 .. 11)

T, S = L.T, L.S                  # state: six pairs v[80:91]; S-box outputs / hi halves: six pairs v[104:115]
PT = 116                         # the partner's six elements (pairs v[116:127]) -- merged fours
O = L.O                          # dense layer outputs: six pairs v[176:187]
TMP = 188                        # copies for the swaps (four pairs)
MASK_LO = 70                     # SGPR pair in: lanes 0 .. 31
NE = 6                           # elements per lane


def swap(prog, a, b):
    """v_permlane32_swap_b32 a, b: rows 2, 3 of a <-> rows 0, 1 of b"""
    ins = Ins("v_permlane32_swap_b32 %s, %s" % (v(a), v(b)), [a, b], [a, b], dpp=True, sem=("swap32", a, b))
    prog.append(ins)


def partners(prog, d0, d1, s0, s1):
    """d0 = the PARTNER lane's s1, d1 = the partner lane's s0 (64-bit pairs), in both lanes.  Two copies and four swaps:
    swap(X, Y) sends the upper lane's X to the lower lane's Y and the lower lane's Y to the upper lane's X; with X = copy of s0's dword
    and Y = copy of s1's dword, swap(X, Y) then swap(Y, X) leaves the partner's s1 in X and the partner's s0 in Y -- in BOTH lanes."""
    prog.append(Ins("v_mov_b64 %s, %s" % (vp(d0), vp(s0)), [s0, s0 + 1], [d0, d0 + 1], sem=("mov64", d0, s0)))
    prog.append(Ins("v_mov_b64 %s, %s" % (vp(d1), vp(s1)), [s1, s1 + 1], [d1, d1 + 1], sem=("mov64", d1, s1)))
    for h in (0, 1):
        swap(prog, d0 + h, d1 + h)
        swap(prog, d1 + h, d0 + h)


# ---------------------------------------------------------------- full / partial round on the matrix pipe
# v_mfma_i32_32x32x32_i8, D[row][col] = sum_K A[row][K] B[K][col]: lane (col = l & 31, half = l >> 5) of the B operand holds 16 K-values of
# its column, which meet the 16 K-values lane (row, half) of the A operand holds; result register i of lane (col, half) is row
# (i & 3) + 8 (i >> 2) + 4 half (tools/experiments/mfma_mds_probe.hip).  Here column = permutation, the two halves hold its elements
# 0 .. 5 and 6 .. 11, and ONE instruction multiplies TWO byte planes: the 16 K-values of a lane are
#     dword d = 0 .. 2:  (e[2d].b_p, e[2d+1].b_p, e[2d].b_p+1, e[2d+1].b_p+1)   -- one v_perm_b32 of the two words' dwords
#     dword 3:           (1, 64, 127, 127)                                      -- the constants' carrier, as in the lane form
# and weight row (g, p') = output g against plane p + p' is M[g][6 half + 2 d + (i & 1)] where (i >> 1) == p', zero elsewhere.  Rows are
# assigned so that result registers 0 .. 5 of a lane are its own six outputs for plane p and 6 .. 11 the same for plane p + 1:
# row(i, half) -> g = 6 half + i % 6, p' = i / 6.  Four instructions per round instead of the lane form's eight, 12 byte permutes instead of 48.
def pair_mfma(prog, q):
    a, b, d = L.AW[q % 2], L.BP[q], DT[q]
    L.load(prog, a + 3, 1, L.A_RCB, 256 * q, ("rcb", q))
    ins = Ins("v_mfma_i32_32x32x32_i8 v[%d:%d], v[%d:%d], v[%d:%d], 0" % (d, d + 15, a, a + 3, b, b + 3),
              [a, a + 1, a + 2, a + 3, b, b + 1, b + 2, b + 3, L.MFMA_PIPE], list(range(d, d + 12)) + [L.MFMA_PIPE], sem=("pmfma", d, a, b, q))
    ins.is_mfma = True
    ins.junk = set(range(d + 12, d + 16))
    prog.append(ins)


DT = [128 + 12 * q for q in range(4)]   # result tiles (sixteen registers written, twelve results; the junk rows are the next tile's first four)


def circulant_pair(prog, in_base, out_base, first_out=0):
    """out[i] = RC + sum_j M[g][j] in[j], g = 6 * half + i, for this lane's outputs first_out .. 5"""
    for half in range(2):
        w = [in_base + 2 * e + half for e in range(NE)]
        for q2, sel in ((0, "A"), (1, "B")):
            q = 2 * half + q2
            for d in range(3):
                L.perm(prog, L.BP[q] + d, w[2 * d + 1], w[2 * d], sel)
            for d in range(3):
                L.xor80(prog, L.BP[q] + d)
            pair_mfma(prog, q)
    for half in range(2):
        for i in range(first_out, NE):
            ad = L.AD[i % 2]
            u = L.UT[(2 * half + i) % 4]
            t0, t1 = DT[2 * half], DT[2 * half + 1]
            L.lshl_add(prog, ad, t0 + NE + i, 8, t0 + i)
            L.lshl_add(prog, u, t1 + NE + i, 8, t1 + i)
            dst = (out_base if half == 0 else L.HIP) + 2 * i
            prog.append(Ins("v_mad_u64_u32 %s, %s, %s, s%d, %s" % (vp(dst), sp(L.SINK), v(u), L.S_64K, vp(ad)), [u, ad, ad + 1], [dst, dst + 1],
                            sem=("mad", dst, None, u, ("const", 65536), ad)))
    for i in range(first_out, NE):
        L.fold_hi_lo(prog, out_base + 2 * i, out_base + 2 * i, L.HIP + 2 * i, i % 2)


def block_full_pair(first_out=0):
    prog = []
    for e in range(NE):
        L.sbox(prog, S + 2 * e, T + 2 * e, e % 2)
    circulant_pair(prog, S, T, first_out)
    return prog


def block_partial_pair():
    """element 0 (the lower lane's first) through the S-box; the upper lane keeps its element 6"""
    prog = []
    L.sbox(prog, S, T, 0)
    for h in (0, 1):
        G.cndmask(prog, T + h, T + h, S + h, MASK_LO)
    circulant_pair(prog, T, T)
    return prog


# ---------------------------------------------------------------- pieces of the merged partial rounds
# LDS rows per lane half (the kernel gives the two halves different base addresses): dot rows of the half's own six elements, dense rows
# of 16 dwords (own six, partner's six, the x-coefficients)
def dot6(prog, A, B, coef_off, seed_regs, key, wide=False):
    """A / B = seed + sum over the lane's own six elements; `wide`: eight dwords of the row are loaded (the caller uses dword 6)"""
    cr = L.COEFR + 16 * (key[1] % 2)
    L.load(prog, cr, 4, L.A_COEF, coef_off, (key, 0))
    L.load(prog, cr + 4, 4 if wide else 2, L.A_COEF, coef_off + 16, (key, 1))
    for j in range(NE):
        L.madc(prog, A, T + 2 * j, ("v", cr + j), seed=seed_regs if j == 0 else None)
        L.madc(prog, B, T + 2 * j + 1, ("v", cr + j), seed=seed_regs + 2 if j == 0 else None)
    return cr


def pair_sums(prog, A, B, k):
    """A and B (64-bit pairs) += the partner lane's A and B, in both lanes"""
    ca, cb = TMP + 4 * k, TMP + 4 * k + 2
    partners(prog, ca, cb, A, B)       # ca = partner's B, cb = partner's A
    G.add64(prog, A, A, cb)
    G.add64(prog, B, B, ca)


# ---------------------------------------------------------------- FOUR partial rounds at once (gen_lane_round_asm.py: block_four)
# per lane half, 480 bytes: rows 0 of M, N2, N3 against the half's own six elements (8 dwords each; N2[0][0] in dword 6 of the third), then
# per local output r (g = 6 half + r) sixteen dwords: N4[g][own six], N4[g][the partner's six, neighbours crossed], N3[g][0], N2[g][0], M[g][0], 0
F_DOT0, F_DOT1, F_DOT2, F_ROW = 0, 32, 64, 96
KQ = S + 4          # the third scalar seed: v[108:111] (x1 is in v[104:105], the rest of the S-box output area is idle here)


def block_four_pair():
    prog = []
    L.sbox(prog, S, T, 0)                                      # x1 (meaningful in the lower lane)
    for h in (0, 1):
        G.cndmask(prog, T + h, T + h, S + h, MASK_LO)        # T is ut now
    L.load(prog, L.SEEDR, 4, L.A_K12, 0, ("kf", 0))           # k1, k2, k3: in the lower half's table, zero in the upper half's
    L.load(prog, L.SEEDR + 4, 4, L.A_K12, 16, ("kf", 1))
    L.load(prog, KQ, 4, L.A_K12, 32, ("kf", 2))
    dot6(prog, L.ACC, L.ACC + 2, F_DOT0, L.SEEDR, ("dot", 0))
    pair_sums(prog, L.ACC, L.ACC + 2, 0)
    L.fold_to(prog, L.YY, L.ACC, L.ACC + 2, 0)
    L.sbox(prog, L.YY + 2, L.YY, 1)                            # x2, in both lanes
    dot6(prog, L.ACC + 4, L.ACC + 6, F_DOT1, L.SEEDR + 4, ("dot", 1))
    pair_sums(prog, L.ACC + 4, L.ACC + 6, 1)
    L.madc(prog, L.ACC + 4, L.YY + 2, 25)
    L.madc(prog, L.ACC + 6, L.YY + 3, 25)
    L.fold_to(prog, L.YY, L.ACC + 4, L.ACC + 6, 1)
    L.sbox(prog, L.YY + 4, L.YY, 0)                            # x3
    cr = dot6(prog, L.ACC, L.ACC + 2, F_DOT2, KQ, ("dot", 2), wide=True)
    pair_sums(prog, L.ACC, L.ACC + 2, 0)
    L.madc(prog, L.ACC, L.YY + 2, ("v", cr + 6))               # N2[0][0] x2
    L.madc(prog, L.ACC + 2, L.YY + 3, ("v", cr + 6))
    L.madc(prog, L.ACC, L.YY + 4, 25)                          # M[0][0] x3
    L.madc(prog, L.ACC + 2, L.YY + 5, 25)
    L.fold_to(prog, L.YY, L.ACC, L.ACC + 2, 0)
    L.sbox(prog, L.YY + 6, L.YY, 1)                            # x4
    for e in range(0, NE, 2):                                  # the partner's six elements: PT + 2 e = its element e ^ 1
        partners(prog, PT + 2 * e, PT + 2 * e + 2, T + 2 * e, T + 2 * e + 2)
    for r in range(NE):
        sd = L.SEEDR + 8 + 4 * (r % 2)
        L.load(prog, sd, 4, L.A_K3, 16 * r, ("k4", r))
        A, B = L.ACC + 4 * (r % 2), L.ACC + 4 * (r % 2) + 2
        cr = L.COEFR + 16 * (r % 2)
        for q in range(4):
            L.load(prog, cr + 4 * q, 4, L.A_COEF, F_ROW + 64 * r + 16 * q, (("row", r), q))
        for j in range(12):
            src = T + 2 * j if j < NE else PT + 2 * ((j - NE) ^ 1)
            L.madc(prog, A, src, ("v", cr + j), seed=sd if j == 0 else None)
            L.madc(prog, B, src + 1, ("v", cr + j), seed=sd + 2 if j == 0 else None)
        for q in range(3):                                     # N3[g][0] x2 + N2[g][0] x3 + M[g][0] x4
            L.madc(prog, A, L.YY + 2 + 2 * q, ("v", cr + 12 + q))
            L.madc(prog, B, L.YY + 3 + 2 * q, ("v", cr + 12 + q))
        L.fold_big(prog, O + 2 * r, A, B, r % 2)
    for e in range(NE):
        prog.append(Ins("v_mov_b64 %s, %s" % (vp(T + 2 * e), vp(O + 2 * e)), [O + 2 * e, O + 2 * e + 1], [T + 2 * e, T + 2 * e + 1], sem=("mov64", T + 2 * e, O + 2 * e)))
    return prog


def test_four_pair(order):
    for _ in range(30):
        state = [L.rnd() for _ in range(12)]
        c1, c2, c3, c4 = [[random.getrandbits(64) % P for _ in range(12)] for _ in range(4)]
        M, N2, N3, N4, k1, k2, k3, k4 = L.merged_tables4(c1, c2, c3, c4)
        want = state
        for c in (c1, c2, c3, c4):
            want = G.reference_round(want, c, True)
        vregs = fresh()
        set_state(vregs, state)
        mem = vregs["mem"]
        mem[("kf", 0)], mem[("kf", 1)], mem[("kf", 2)] = pair4(k1, 0), pair4(k2, 0), pair4(k3, 0)
        for r in range(NE):
            mem[("k4", r)] = pair4(k4[r], k4[NE + r])
            rows = []
            for l in range(2):
                g = NE * l + r
                rows.append([N4[g][(NE * l + j) % 12] for j in range(12)] + [N3[g][0], N2[g][0], M[g][0], 0])
            for q in range(4):
                mem[(("row", r), q)] = [[rows[0][4 * q + i], rows[1][4 * q + i]] for i in range(4)]
        for d, row, extra in ((0, M[0], 0), (1, N2[0], 0), (2, N3[0], N2[0][0])):
            co = [[row[e], row[NE + e]] for e in range(NE)] + [[extra, extra], [0, 0]]
            mem[(("dot", d), 0)] = co[0:4]
            mem[(("dot", d), 1)] = co[4:8] if d == 2 else co[4:6]
        run_pair(order, vregs, {MASK_LO: [1, 0]})
        assert get_state(vregs) == want


# ---------------------------------------------------------------- scheduling: the lane generator's, plus the swap's two wait states
def schedule_pair(prog):
    """gen_lane_round_asm.schedule, then a swap's operands written at least three slots before it (two wait states; the builtin gets
    s_nop 1 from the compiler, kernels_lde.hip): the lane scheduler does not know the instruction, so s_nop fills what is missing."""
    out = []
    for ins in L.schedule(prog):
        if getattr(ins, "dpp", False):
            gap = 0
            for back in (1, 2):
                if len(out) >= back and (ins.reads & out[-back].writes):
                    gap = max(gap, 3 - back)
            for _ in range(gap):
                out.append(Ins("s_nop 0", [], []))
        out.append(ins)
    return out


# ---------------------------------------------------------------- interpreter additions (two lanes)
def run_pair(order, vregs, sregs):
    for ins in order:
        k = ins.sem[0] if ins.sem else None
        if k == "swap32":
            _, a, b = ins.sem
            va, vb = vregs[a][:], vregs[b][:]
            vregs[a] = [va[0], vb[0]]
            vregs[b] = [va[1], vb[1]]
        elif k == "perm":
            _, d, s0, s1, sel = ins.sem
            out = []
            for l in range(2):
                src = (vregs[s1][l] & M32) | ((vregs[s0][l] & M32) << 32)
                o = 0
                for i in range(4):
                    o |= ((src >> (8 * ((L.SEL_VALUE[sel] >> (8 * i)) & 0xFF))) & 0xFF) << (8 * i)
                out.append(o)
            vregs[d] = out
        elif k == "xor80":
            vregs[ins.sem[1]] = [x ^ 0x80808080 for x in vregs[ins.sem[1]]]
        elif k == "lshladd":
            _, d, a, sh, b = ins.sem
            vregs[d] = [((vregs[a][l] << sh) + vregs[b][l]) & M32 for l in range(2)]
        elif k == "pmfma":
            _, d, a, b, q = ins.sem
            assert vregs[b + 3] == [L.B_CONST, L.B_CONST]
            assert vregs[a + 3] == [0xC0DE00 + q] * 2, ("A tuple holds another instruction's constants", q)
            res = [[0] * 2 for _ in range(16)]
            for pp in range(2):
                plane = 2 * q + pp
                by = []
                for l in range(2):
                    for e in range(NE):
                        x = (vregs[b + e // 2][l] >> (8 * ((e & 1) + 2 * pp))) & 0xFF
                        by.append(x - 256 if x >= 128 else x)
                rc = vregs["rcbytes"][plane]
                for l in range(2):
                    for i in range(NE):
                        g = NE * l + i
                        val = sum(L.mds_coef(g, j) * by[j] for j in range(12)) + (rc[g] & 0x7F) + 64 * (2 * (rc[g] >> 7) + 40) + 2 * 127 * 127
                        assert 0 <= val < (1 << 17)
                        res[NE * pp + i][l] = val
            for i in range(12):
                vregs[d + i] = res[i]
            for i in range(12, 16):
                vregs[d + i] = [0xDEAD0000 + i] * 2
        else:
            G.run([ins], vregs, sregs)


def fresh():
    vregs = {r: [random.getrandbits(32), random.getrandbits(32)] for r in range(52, 256)}
    vregs[L.AD[0] + 1] = [0, 0]
    vregs[L.AD[1] + 1] = [0, 0]
    vregs["mem"] = {}
    return vregs


def set_state(vregs, state):
    for e in range(NE):
        vregs[T + 2 * e] = [state[e] & M32, state[NE + e] & M32]
        vregs[T + 2 * e + 1] = [state[e] >> 32, state[NE + e] >> 32]


def get_state(vregs):
    out = [0] * 12
    for e in range(NE):
        for l in range(2):
            out[NE * l + e] = (vregs[T + 2 * e][l] | (vregs[T + 2 * e + 1][l] << 32)) % P
    return out


def test_round_pair(order, partial, first_out=0):
    for _ in range(40):
        state = [L.rnd() for _ in range(12)]
        rc = [random.getrandbits(64) % P for _ in range(12)]
        vregs = fresh()
        for k in range(4):
            vregs[L.BP[k] + 3] = [L.B_CONST] * 2
        set_state(vregs, state)
        RC = L.mfma_round_constants(rc)
        vregs["rcbytes"] = [[(RC[g] >> (8 * b)) & 0xFF for g in range(12)] for b in range(8)]
        for q in range(4):
            vregs["mem"][("rcb", q)] = [[0xC0DE00 + q] * 2]
        run_pair(order, vregs, {MASK_LO: [1, 0]})
        got, want = get_state(vregs), G.reference_round(state, rc, partial)
        for l in range(2):
            for i in range(first_out, NE):
                assert got[NE * l + i] == want[NE * l + i], (partial, l, i)


def pair4(lo_val, hi_val):
    """a 64-bit constant as two 64-bit addends (low half, 0, high half, 0), per lane"""
    return [[lo_val & M32, hi_val & M32], [0, 0], [lo_val >> 32, hi_val >> 32], [0, 0]]


def check_swaps(order):
    for i, ins in enumerate(order):
        if getattr(ins, "dpp", False):
            for back in (1, 2):
                if i - back >= 0:
                    assert not (ins.reads & order[i - back].writes), ("a swap's operand written fewer than three slots before it", i, ins.text)


def main():
    random.seed(7)
    blocks = (("STARKHIP_PAIR_FULL_ROUND_ASM", block_full_pair(), lambda o: test_round_pair(o, False), "full round: six S-boxes per lane, circulant layer on the matrix pipe"),
              ("STARKHIP_PAIR_LAST_ROUND_ASM", block_full_pair(2), lambda o: test_round_pair(o, False, 2), "last full round before an absorb: the lane's outputs 2 .. 5 only (the capacity is the upper lane's)"),
              ("STARKHIP_PAIR_PARTIAL_ROUND_ASM", block_partial_pair(), lambda o: test_round_pair(o, True), "partial round"),
              ("STARKHIP_PAIR_FOUR_ASM", block_four_pair(), test_four_pair, "four partial rounds at once (poseidon_merged.h)"))
    done, slots = [], {}
    for name, prog, tester, what in blocks:
        L.LOAD_LATENCY = TRIPLE_LOAD_LATENCY if name == "STARKHIP_PAIR_FOUR_ASM" else ROUND_LOAD_LATENCY
        order = schedule_pair(prog)
        L.check_hazards(order)
        check_swaps(order)
        if not ABLATE:
            tester(order)
        done.append((name, order, what))
        slots[name] = len(order)
    per_wave = 7 * slots["STARKHIP_PAIR_FULL_ROUND_ASM"] + slots["STARKHIP_PAIR_LAST_ROUND_ASM"] + 5 * slots["STARKHIP_PAIR_FOUR_ASM"] + 2 * slots["STARKHIP_PAIR_PARTIAL_ROUND_ASM"]
    print("// generated by tools/gen_pair_round_asm.py -- do not edit.  The PAIR form: lanes l and l + 32 share a permutation (elements 0 .. 5 / 6 .. 11).")
    print("// Physical registers: state v[%d:%d] (in and out), LDS addresses v%d (k3) v%d (k12) v%d (coefficient rows) v%d (the matrix-pipe rounds' constants)," % (T, T + 11, L.A_K3, L.A_K12, L.A_COEF, L.A_RCB))
    print("// zeros v%d v%d, s[%d:%d] = the lower half-wave's lane mask; v%d .. v255 and s%d .. s%d are clobbered." % (L.AD[0] + 1, L.AD[1] + 1, MASK_LO, MASK_LO + 1, 92, L.SINK, L.FCS[1] + 1))
    print("// Per wave and 32 permutations: 7 x %d + %d + 5 x %d + 2 x %d = %d issue slots = %.1f per permutation (quad form: 271.6, lane form: 177.1)." %
          (slots["STARKHIP_PAIR_FULL_ROUND_ASM"], slots["STARKHIP_PAIR_LAST_ROUND_ASM"], slots["STARKHIP_PAIR_FOUR_ASM"], slots["STARKHIP_PAIR_PARTIAL_ROUND_ASM"], per_wave, per_wave / 32.0))
    for name, order, what in done:
        L.emit(name, order, what)
    for i in range(3):
        print('#define STARKHIP_PAIR_STATE%d "+{v[%d:%d]}"' % (i, T + 4 * i, T + 4 * i + 3))
    print('#define STARKHIP_PAIR_A_K3 "{v%d}"' % L.A_K3)
    print('#define STARKHIP_PAIR_A_K12 "{v%d}"' % L.A_K12)
    print('#define STARKHIP_PAIR_A_COEF "{v%d}"' % L.A_COEF)
    print('#define STARKHIP_PAIR_A_RCB "{v%d}"' % L.A_RCB)
    print('#define STARKHIP_PAIR_ZA "{v%d}"' % (L.AD[0] + 1))
    print('#define STARKHIP_PAIR_ZB "{v%d}"' % (L.AD[1] + 1))
    print('#define STARKHIP_PAIR_MASK_LO "{s[%d:%d]}"' % (MASK_LO, MASK_LO + 1))
    bound = set(range(T, T + 12)) | {L.AD[0] + 1, L.AD[1] + 1}
    vs = [r for r in range(T + 12, 256) if r not in bound]
    ss = list(range(L.SINK, L.FCS[1] + 2)) + list(range(L.FC2[0], L.FC2[1] + 2))
    print("#define STARKHIP_PAIR_CLOBBERS %s" % ", ".join(['"v%d"' % r for r in vs] + ['"s%d"' % r for r in ss]))
    # the matrix-pipe blocks: weight tile (dword 3 is loaded inside: in / out), the B tuples' constant dwords (their other dwords: clobbered)
    for k in range(2):
        for d in range(3):
            print('#define STARKHIP_PAIR_AW%d%d "{v%d}"' % (k, d, L.AW[k] + d))
        print('#define STARKHIP_PAIR_AW%d3 "+{v%d}"' % (k, L.AW[k] + 3))
    for k in range(4):
        print('#define STARKHIP_PAIR_BC%d "{v%d}"' % (k, L.BP[k] + 3))
    for name, reg in (("SEL_A", L.S_SEL["A"]), ("SEL_B", L.S_SEL["B"]), ("X80", L.S_X80), ("K64K", L.S_64K)):
        print('#define STARKHIP_PAIR_S_%s "{s%d}"' % (name, reg))
    print("#define STARKHIP_PAIR_MFMA_CLOBBERS %s" % ", ".join(['"v%d"' % (L.BP[k] + d) for k in range(4) for d in range(3)]))


if __name__ == "__main__":
    main()
