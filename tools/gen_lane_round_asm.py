#!/usr/bin/env python3
"""Generates csrc/lane_round_asm.inc: the rounds of the LANE form of the Poseidon permutation (one lane per leaf, the whole state in
the lane's registers; poseidon_dev.h) as scheduled inline-asm blocks on fixed physical registers.

The lane form costs the fewest instructions per permutation -- nothing is repeated across lanes: ~ 12.8 K slots against the quad
form's 4346 x 4 lane-slots -- but from C++ hipcc makes 14.4 K VALU + 4.7 K wait states + 335 s_waitcnt of it.  Here
the twelve independent S-boxes of a full round fill each other's flag hand-offs, round constants and the merged layers' coefficients
(uniform over the wave) come from one LDS image by broadcast loads issued a row ahead, and the waits are COUNTED (LDS returns in
order: s_waitcnt lgkmcnt(k) with k = the loads issued since the one needed).

Shares the instruction model, the list scheduler and the interpreter with tools/gen_row_round_asm.py.

    python tools/gen_lane_round_asm.py > starky_bls12_381_amd/csrc/lane_round_asm.inc
"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_row_round_asm as G  # noqa: E402

Ins, v, vp, sp = G.Ins, G.v, G.vp, G.sp
P, M32, M64, CIRC = G.P, G.M32, G.M64, G.CIRC
G.NL = 1  # one lane is enough: nothing crosses lanes here

# ---------------------------------------------------------------- register map (VGPRs 76 .. 255)
A_K3, A_K12, A_COEF, A_SEED = 76, 77, 78, 79   # in: LDS addresses (k3[t], k12[t], coefficient rows, rc[r + 1])
T = 80            # state: 12 pairs v[80:103] (in and out)
S = 104           # S-box outputs / u': 12 pairs
SEEDR = 128       # seed ring: 4 x (lo64, hi64) = 16 registers
COEFR = 144       # coefficient ring: 2 rows x 16
O = 176           # dense layer outputs: 12 pairs
ACC = 200         # two outputs in flight: (A, B) pairs each
FOLD = 208        # two folds in flight: FT pair + CV each (208:209, 210 / 212:213, 214)
XT = 216          # S-box temporaries: 2 sets x (x2, x3, x4) pairs
SLOTS = [228, 236]
AD = [244, 246]   # addend pairs (ad, zero): 245 and 247 hold zero (inputs)
YY = 248          # folded dot products, x2, x3 (pairs 248, 250, 252)
SINK, FLAGS, FCS = 42, [44, 52], [60, 62]
G.SINK = SINK


class Slot:
    def __init__(self, k):
        b = SLOTS[k]
        self.P0, self.M, self.P3, self.t, self.AD = b, b + 2, b + 4, b + 6, AD[k]
        f = FLAGS[k]
        self.CM, self.BR, self.BR2, self.CY = f, f + 2, f + 4, f + 6


def load(prog, first, count, addr, off, key):
    op = {4: "ds_read_b128", 2: "ds_read_b64", 1: "ds_read_b32"}[count]
    rng = "v[%d:%d]" % (first, first + count - 1) if count > 1 else v(first)
    prog.append(Ins("%s %s, %s offset:%d" % (op, rng, v(addr), off), [addr], list(range(first, first + count)), sem=("ldsload", first, count, key)))
    prog[-1].is_load = True
    prog[-1].boost = True


def madc(prog, acc, src, coef, seed=None):
    add = acc if seed is None else seed
    if isinstance(coef, tuple):
        prog.append(Ins("v_mad_u64_u32 %s, %s, %s, %s, %s" % (vp(acc), sp(SINK), v(src), v(coef[1]), vp(add)), [src, coef[1], add, add + 1], [acc, acc + 1],
                        sem=("mad", acc, None, src, coef[1], add)))
    else:
        prog.append(Ins("v_mad_u64_u32 %s, %s, %s, %d, %s" % (vp(acc), sp(SINK), v(src), coef, vp(add)), [src, add, add + 1], [acc, acc + 1],
                        sem=("mad", acc, None, src, ("const", coef), add)))


def fold_to(prog, dst, A, B, k):
    FT, CV, FC = FOLD + 4 * k, FOLD + 4 * k + 2, FCS[k]
    prog.append(Ins("v_mad_u64_u32 %s, %s, %s, -1, %s" % (vp(FT), sp(SINK), v(B + 1), vp(A)), [B + 1, A, A + 1], [FT, FT + 1], sem=("mad", FT, None, B + 1, "eps", A)))
    prog.append(Ins("v_add_co_u32 %s, %s, %s, %s" % (v(FT + 1), sp(FC), v(FT + 1), v(B)), [FT + 1, B], [FT + 1], swrites=[FC], sem=("addco", FT + 1, FC, FT + 1, B)))
    prog.append(Ins("v_addc_co_u32 %s, %s, 0, 0, %s" % (v(CV), sp(SINK), sp(FC)), [], [CV], sreads=[FC], sem=("addc", CV, None, None, None, FC)))
    prog.append(Ins("v_mad_u64_u32 %s, %s, %s, -1, %s" % (vp(dst), sp(SINK), v(CV), vp(FT)), [CV, FT, FT + 1], [dst, dst + 1], sem=("mad", dst, None, CV, "eps", FT)))


def sbox(prog, dst, x, k):
    """dst = x^7; temporaries set k, both multiply slots (x^3 and x^4 side by side)"""
    a, b = Slot(0), Slot(1)
    x2, x3, x4 = XT + 6 * k, XT + 6 * k + 2, XT + 6 * k + 4
    xx = (x, x + 1)
    G.mul(prog, x2, xx, xx, a if k == 0 else b)
    G.mul(prog, x4, (x2, x2 + 1), (x2, x2 + 1), a)
    G.mul(prog, x3, (x2, x2 + 1), xx, b)
    G.mul(prog, dst, (x3, x3 + 1), (x4, x4 + 1), a if k == 0 else b)


def circulant(prog, first_out, in_base, out_base):
    """out[r] = seed[r] + sum_i CIRC[i] in[(i + r) % 12] (+ 8 in[0] for r = 0), r = first_out .. 11; seeds from LDS at A_SEED"""
    for r in range(first_out, 12):
        sd = SEEDR + 4 * (r % 4)
        load(prog, sd, 4, A_SEED, 16 * r, ("seed", r))
        A, B = ACC + 4 * (r % 2), ACC + 4 * (r % 2) + 2
        for i in range(12):
            j = (i + r) % 12
            k = CIRC[i] + (8 if r == 0 and i == 0 else 0)
            madc(prog, A, in_base + 2 * j, k, seed=sd if i == 0 else None)
            madc(prog, B, in_base + 2 * j + 1, k, seed=sd + 2 if i == 0 else None)
        fold_to(prog, out_base + 2 * r, A, B, r % 2)


def block_full(first_out=0):
    prog = []
    for e in range(12):
        sbox(prog, S + 2 * e, T + 2 * e, e % 2)
    circulant(prog, first_out, S, T)
    return prog


def block_partial():
    prog = []
    sbox(prog, T, T, 0)     # element 0 in place
    # the layer reads T and must not overwrite it while later outputs still need it: outputs go to O, then back
    circulant(prog, 0, T, O)
    for e in range(12):
        prog.append(Ins("v_mov_b64 %s, %s" % (vp(T + 2 * e), vp(O + 2 * e)), [O + 2 * e, O + 2 * e + 1], [T + 2 * e, T + 2 * e + 1], sem=("mov64", T + 2 * e, O + 2 * e)))
    return prog


def dot(prog, A, B, coef_off, seed_regs, key):
    """A / B = seed + sum_j coef[j] * halves of T[j]; coefficient row (12 words) from LDS at A_COEF + coef_off"""
    cr = COEFR + 16 * (key[1] % 2)
    for q in range(3):
        load(prog, cr + 4 * q, 4, A_COEF, coef_off + 16 * q, (key, q))
    if key[0] == "row":
        load(prog, cr + 12, 4, A_COEF, coef_off + 48, (key, 3))
    for j in range(12):
        madc(prog, A, T + 2 * j, ("v", cr + j), seed=seed_regs if j == 0 else None)
        madc(prog, B, T + 2 * j + 1, ("v", cr + j), seed=seed_regs + 2 if j == 0 else None)
    return cr


ROW_OFF, M0_OFF, N20_OFF = 0, 12 * 64, 12 * 64 + 48   # LaneTables: row[12][16], m0[12], n20[12] contiguous


# ---------------------------------------------------------------- FOUR partial rounds at once
# With M the MDS matrix, Mz = M with row 0 zeroed, N_k = M Mz^(k-1), u the state at the start of partial round r (constants added),
# x1 = u0^7, ut = (x1, u1 .. u11), c1 .. c4 the constants of rounds r + 1 .. r + 4 (c?z: element 0 zeroed):
#     y1  = (M ut)[0] + k1                                          x2 = y1^7      k1 = c1[0]
#     y2  = (N2 ut)[0] + M[0][0] x2 + k2                             x3 = y2^7      k2 = (M c1z)[0] + c2[0]
#     y3  = (N3 ut)[0] + N2[0][0] x2 + M[0][0] x3 + k3               x4 = y3^7      k3 = (N2 c1z)[0] + (M c2z)[0] + c3[0]
#     out = N4 ut + N3[:,0] x2 + N2[:,0] x3 + M[:,0] x4 + k4                        k4 = N3 c1z + N2 c2z + M c3z + c4
# N4's entries are below 2^29 and a row of it, with its three x-coefficients, sums to less than 0.83 * 2^32: the two accumulators of
# an output (products with the 32-bit halves of the inputs) stay below 2^64 -- but no longer below 2^57, so their fold takes the
# multiply-add's carry (fold_big).  Five merges would need 37-bit coefficients.  Per round 825 / 4 slots against 732 / 3.
FC2 = [72, 74]    # scalar pairs: carry out of fold_big's first multiply-add
N30_OFF = N20_OFF + 48     # LaneTables: n30[16] = row 0 of N3, then N2[0][0]
KQ = S                     # the third scalar seed's registers (the S-box output area is idle in this block): v[104:107]


def merged_tables4(c1, c2, c3, c4):
    M = [[CIRC[(j - i) % 12] + (8 if i == 0 and j == 0 else 0) for j in range(12)] for i in range(12)]
    Mz = [[0] * 12 if i == 0 else M[i][:] for i in range(12)]

    def mm(a, b):
        return [[sum(a[i][k] * b[k][j] for k in range(12)) for j in range(12)] for i in range(12)]

    def mv(a, x):
        return [sum(a[i][j] * x[j] for j in range(12)) % P for i in range(12)]
    N2 = mm(M, Mz)
    N3 = mm(N2, Mz)
    N4 = mm(N3, Mz)
    c1z, c2z, c3z = [0] + c1[1:], [0] + c2[1:], [0] + c3[1:]
    k1 = c1[0]
    k2 = (mv(M, c1z)[0] + c2[0]) % P
    k3 = (mv(N2, c1z)[0] + mv(M, c2z)[0] + c3[0]) % P
    a, b, c = mv(N3, c1z), mv(N2, c2z), mv(M, c3z)
    k4 = [(a[i] + b[i] + c[i] + c4[i]) % P for i in range(12)]
    for g in range(12):
        assert (sum(N4[g]) + N3[g][0] + N2[g][0] + M[g][0]) * M32 + M32 < (1 << 64) - (1 << 32)  # what fold_big needs: B < 2^64 - 2^32 (B_hi + carry must not wrap)
    return M, N2, N3, N4, k1, k2, k3, k4


def fold_big(prog, dst, A, B, k):
    """dst = A + B 2^32 mod p (some representative) for 64-bit A and B with B < 2^64 - 2^32 (here both are below 0.83 * 2^64):
        A + B 2^32 = A_lo + (A_hi + B_lo) 2^32 + B_hi 2^64 = (s : A_lo) + (B_hi + c) eps   mod p,   s + c 2^32 = A_hi + B_lo
    -- one addition with carry-out in place, the carry into B_hi (which cannot wrap), one multiply-add whose own carry-out is worth eps
    once more (after it the sum is below (B_hi + c) eps < 2^64 - 2^32, so that last correction cannot overflow).  Five instructions; A is
    consumed."""
    FT, CV, FC, C2 = FOLD + 4 * k, FOLD + 4 * k + 2, FCS[k], FC2[k]
    prog.append(Ins("v_add_co_u32 %s, %s, %s, %s" % (v(A + 1), sp(FC), v(A + 1), v(B)), [A + 1, B], [A + 1], swrites=[FC], sem=("addco", A + 1, FC, A + 1, B)))
    prog.append(Ins("v_addc_co_u32 %s, %s, %s, 0, %s" % (v(CV), sp(SINK), v(B + 1), sp(FC)), [B + 1], [CV], sreads=[FC], sem=("addc", CV, None, B + 1, None, FC)))
    prog.append(Ins("v_mad_u64_u32 %s, %s, %s, -1, %s" % (vp(FT), sp(C2), v(CV), vp(A)), [CV, A, A + 1], [FT, FT + 1], swrites=[C2], sem=("mad", FT, C2, CV, "eps", A)))
    prog.append(Ins("v_addc_co_u32 %s, %s, 0, 0, %s" % (v(CV), sp(SINK), sp(C2)), [], [CV], sreads=[C2], sem=("addc", CV, None, None, None, C2)))
    prog.append(Ins("v_mad_u64_u32 %s, %s, %s, -1, %s" % (vp(dst), sp(SINK), v(CV), vp(FT)), [CV, FT, FT + 1], [dst, dst + 1], sem=("mad", dst, None, CV, "eps", FT)))


def block_four():
    prog = []
    sbox(prog, T, T, 0)                                       # x1 replaces element 0: T is ut
    load(prog, SEEDR, 4, A_K12, 0, ("kf", 0))
    load(prog, SEEDR + 4, 4, A_K12, 16, ("kf", 1))
    load(prog, KQ, 4, A_K12, 32, ("kf", 2))
    dot(prog, ACC, ACC + 2, M0_OFF, SEEDR, ("dot", 0))
    fold_to(prog, YY, ACC, ACC + 2, 0)
    sbox(prog, YY + 2, YY, 1)                                  # x2
    dot(prog, ACC + 4, ACC + 6, N20_OFF, SEEDR + 4, ("dot", 1))
    madc(prog, ACC + 4, YY + 2, 25)                            # M[0][0] x2
    madc(prog, ACC + 6, YY + 3, 25)
    fold_to(prog, YY, ACC + 4, ACC + 6, 1)
    sbox(prog, YY + 4, YY, 0)                                  # x3
    cr = dot(prog, ACC, ACC + 2, N30_OFF, KQ, ("dot", 2))
    load(prog, cr + 12, 1, A_COEF, N30_OFF + 48, (("dot", 2), 3))
    madc(prog, ACC, YY + 2, ("v", cr + 12))                    # N2[0][0] x2
    madc(prog, ACC + 2, YY + 3, ("v", cr + 12))
    madc(prog, ACC, YY + 4, 25)                                # M[0][0] x3
    madc(prog, ACC + 2, YY + 5, 25)
    fold_to(prog, YY, ACC, ACC + 2, 0)
    sbox(prog, YY + 6, YY, 1)                                  # x4
    for r in range(12):
        sd = SEEDR + 8 + 4 * (r % 2)
        load(prog, sd, 4, A_K3, 16 * r, ("k4", r))
        A, B = ACC + 4 * (r % 2), ACC + 4 * (r % 2) + 2
        cr = dot(prog, A, B, ROW_OFF + 64 * r, sd, ("row", r))
        for q in range(3):                                     # N3[r][0] x2 + N2[r][0] x3 + M[r][0] x4
            madc(prog, A, YY + 2 + 2 * q, ("v", cr + 12 + q))
            madc(prog, B, YY + 3 + 2 * q, ("v", cr + 12 + q))
        fold_big(prog, O + 2 * r, A, B, r % 2)
    for e in range(12):
        prog.append(Ins("v_mov_b64 %s, %s" % (vp(T + 2 * e), vp(O + 2 * e)), [O + 2 * e, O + 2 * e + 1], [T + 2 * e, T + 2 * e + 1], sem=("mov64", T + 2 * e, O + 2 * e)))
    return prog


# ---------------------------------------------------------------- the circulant layer on the matrix pipe
# One permutation per lane: the layer  out[i] = rc[i] + sum_j M[i][j] s[j]  over all 64 lanes IS a (12 x 12) x (12 x 64) product of a
# matrix of 6-bit weights with 64-bit words, i.e. eight products with the words' BYTE PLANES (bytes are exact in the i8 pipe, the
# 12-term sums stay below 2^17): v_mfma_i32_32x32x32_i8 with the weights as the A tile and byte plane b of the state as the B tile.
# Lane l of the B operand holds 16 K-values of column l & 31; lanes l and l + 32 are two different permutations here, so the weight
# tile is block diagonal: rows whose results land in the lower lane half (rows 0-3, 8-11, 16-19 = output g = (row & 3) + 4 (row >> 3))
# carry M[g][.] against the lower half's K-values and zeros against the upper half's, rows 4-7, 12-15, 20-23 the other way round
# (tools/experiments/mfma_mds_probe.hip checks this map with exact integer data on all 64 lanes).  Result register g of a lane is then
#   S_b[g] = sum_j M[g][j] sbyte_b(s[j]) + (what the four spare K-values add),
# sbyte = byte XOR 0x80 read as signed = byte - 128 (the pipe's operands are signed).  The spare K-values 12 .. 15 carry the constants:
# the B side holds (1, 64, 127, 127) in every lane, the A side -- per row, round and plane, one dword per lane from LDS -- holds
# (c7, 2 m + 40, 127, 127) with c7 + 128 m = byte b of a 64-bit constant RC[g]: together + RC byte + 34 818, which makes every S_b[g]
# non-negative (34 818 >= 128 * 272, the largest row sum) and lets the host fold the offsets into RC (kernels_hash.hip).
# Recombination: lo = S0 + S1 2^8 + S2 2^16 + S3 2^24 (two v_lshl_add_u32 and one multiply-add by 2^16), hi likewise from planes 4 .. 7,
# then the fold lo + hi 2^32 the multiply-add form already uses.  Per round 24 XORs + 48 byte permutes + 8 LDS dwords + 8 MFMA (which
# cost the vector issue < 1 slot each, measured by the probe) + 72 + 48 instead of 288 multiply-adds + 48 + 12 LDS rows.
AW = [52, 56]            # in: the weight tile's dwords 0 .. 2, twice (v52-54, v56-58); dword 3 (v55, v59) is loaded per plane
BP = [60, 64, 68, 72]    # the B tuples of planes b mod 4; dwords 0 .. 2 are written here, dword 3 (v63, v67, v71, v75) in: 0x7F7F4001
A_RCB = 79               # in: LDS address of this lane's dword in the round's constant table (plane b at byte offset 256 b)
DT = [128 + 12 * b for b in range(8)]   # result tiles, one per plane: sixteen registers are written, the first twelve are results; the four
                         # junk registers are the next tile's first four, which the next MFMA (issued later, one pipe: it completes later)
                         # overwrites with its results.  v128 .. v227
HIP = S                  # hi halves (12 pairs): the S-box outputs' registers in a full round (every permute has read them by then)
ST = 248                 # byte-transpose temporaries (8)
UT = [228, 229, 230, 231]     # 32-bit partial sums
MFOLD = 232              # the folds' temporaries here (the usual ones lie under the tiles): 232 .. 238
S_SEL = {"A": 64, "B": 65, "C": 66, "D": 67}   # SGPRs in: v_perm_b32 selectors
S_X80, S_64K = 68, 69    # SGPRs in: 0x80808080, 65536
MFMA_PIPE = 9999         # a pseudo register that chains the MFMAs (they share one pipe: issued closer than its occupancy they would stall the wave)
MFMA_RESULT, MFMA_SPACING, MFMA_OPERAND, MFMA_WAR = 20, 9, 3, 6   # slots: result -> first VALU read; MFMA -> next MFMA; VALU write -> MFMA operand; MFMA operand read -> overwrite
SEL_VALUE = {"A": 0x05010400, "B": 0x07030602, "C": 0x05040100, "D": 0x07060302}
B_CONST = 0x7F7F4001
K_OFFSET = 2 * 127 * 127 + 64 * 40


def mds_coef(r, j):
    return CIRC[(j - r) % 12] + (8 if r == 0 and j == 0 else 0)


def perm(prog, dst, s0, s1, sel):
    prog.append(Ins("v_perm_b32 %s, %s, %s, s%d" % (v(dst), v(s0), v(s1), S_SEL[sel]), [s0, s1], [dst], sem=("perm", dst, s0, s1, sel)))


def xor80(prog, reg):
    prog.append(Ins("v_xor_b32 %s, s%d, %s" % (v(reg), S_X80, v(reg)), [reg], [reg], sem=("xor80", reg)))


def lshl_add(prog, dst, a, sh, b):
    prog.append(Ins("v_lshl_add_u32 %s, %s, %d, %s" % (v(dst), v(a), sh, v(b)), [a, b], [dst], sem=("lshladd", dst, a, sh, b)))


def mfma(prog, k, plane):
    a, b, d = AW[k % 2], BP[k], DT[plane]
    load(prog, a + 3, 1, A_RCB, 256 * plane, ("rcb", plane))
    # (the junk registers d + 12 .. d + 15 are not listed as written: nothing reads them, and the MFMA that owns them is ordered behind
    # this one through the pipe's pseudo register)
    ins = Ins("v_mfma_i32_32x32x32_i8 v[%d:%d], v[%d:%d], v[%d:%d], 0" % (d, d + 15, a, a + 3, b, b + 3),
              [a, a + 1, a + 2, a + 3, b, b + 1, b + 2, b + 3, MFMA_PIPE], list(range(d, d + 12)) + [MFMA_PIPE], sem=("mfma", d, a, b, plane))
    ins.is_mfma = True
    ins.junk = set(range(d + 12, d + 16))
    prog.append(ins)


def fold_hi_lo(prog, dst, A, B, k):
    """fold_to with its temporaries at MFOLD"""
    FT, CV, FC = MFOLD + 4 * k, MFOLD + 4 * k + 2, FCS[k]
    prog.append(Ins("v_mad_u64_u32 %s, %s, %s, -1, %s" % (vp(FT), sp(SINK), v(B + 1), vp(A)), [B + 1, A, A + 1], [FT, FT + 1], sem=("mad", FT, None, B + 1, "eps", A)))
    prog.append(Ins("v_add_co_u32 %s, %s, %s, %s" % (v(FT + 1), sp(FC), v(FT + 1), v(B)), [FT + 1, B], [FT + 1], swrites=[FC], sem=("addco", FT + 1, FC, FT + 1, B)))
    prog.append(Ins("v_addc_co_u32 %s, %s, 0, 0, %s" % (v(CV), sp(SINK), sp(FC)), [], [CV], sreads=[FC], sem=("addc", CV, None, None, None, FC)))
    prog.append(Ins("v_mad_u64_u32 %s, %s, %s, -1, %s" % (vp(dst), sp(SINK), v(CV), vp(FT)), [CV, FT, FT + 1], [dst, dst + 1], sem=("mad", dst, None, CV, "eps", FT)))


def circulant_mfma(prog, in_base, out_base):
    """out[i] = RC[i] + sum_j M[i][j] in[j] for all twelve outputs; RC comes in through the table at A_RCB.  The low dwords of the inputs
    make planes 0 .. 3, the high dwords planes 4 .. 7.  All eight products are issued before anything is recombined (a tile per plane),
    so the wait for the last one is filled with the recombination of the first ones; the low halves of the results go to `out_base`,
    the high halves to HIP, both only after every input has been read (in_base may be out_base or HIP)."""
    for half in range(2):
        for q in range(3):
            w = [in_base + 2 * (4 * q + k) + half for k in range(4)]
            t = ST + 4 * (q % 2)
            perm(prog, t + 0, w[1], w[0], "A")   # (w0.b0, w1.b0, w0.b1, w1.b1)
            perm(prog, t + 1, w[1], w[0], "B")   # (w0.b2, w1.b2, w0.b3, w1.b3)
            perm(prog, t + 2, w[3], w[2], "A")
            perm(prog, t + 3, w[3], w[2], "B")
            perm(prog, BP[0] + q, t + 2, t + 0, "C")   # byte 0 of w0 .. w3
            perm(prog, BP[1] + q, t + 2, t + 0, "D")   # byte 1
            perm(prog, BP[2] + q, t + 3, t + 1, "C")   # byte 2
            perm(prog, BP[3] + q, t + 3, t + 1, "D")   # byte 3
            for k in range(4):
                xor80(prog, BP[k] + q)
        for k in range(4):
            mfma(prog, k, 4 * half + k)
    for half in range(2):
        for i in range(12):
            ad = AD[i % 2]
            u = UT[(2 * half + i) % 4]
            d0, d1, d2, d3 = (DT[4 * half + k] + i for k in range(4))
            lshl_add(prog, ad, d1, 8, d0)
            lshl_add(prog, u, d3, 8, d2)
            dst = (out_base if half == 0 else HIP) + 2 * i
            prog.append(Ins("v_mad_u64_u32 %s, %s, %s, s%d, %s" % (vp(dst), sp(SINK), v(u), S_64K, vp(ad)), [u, ad, ad + 1], [dst, dst + 1],
                            sem=("mad", dst, None, u, ("const", 65536), ad)))
    for i in range(12):
        fold_hi_lo(prog, out_base + 2 * i, out_base + 2 * i, HIP + 2 * i, i % 2)


def block_full_mfma():
    prog = []
    for e in range(12):
        sbox(prog, S + 2 * e, T + 2 * e, e % 2)
    circulant_mfma(prog, S, T)
    return prog


def block_partial_mfma():
    prog = []
    sbox(prog, T, T, 0)       # element 0 in place; the layer reads T and writes T (every input is in the B tuples before any output exists)
    circulant_mfma(prog, T, T)
    return prog


# ---------------------------------------------------------------- scheduling with load latency, counted waits
LOAD_LATENCY = 16
VALU_RAW, SGPR_RAW = 1, 3   # slots: VALU result -> VALU read; VALU-written SGPR (carry) -> VALU read (W1).  Experiments change them (gen_pair_round_asm.py)


def schedule_lane(prog):
    """gen_row_round_asm.schedule_with plus the matrix pipe's distances (MFMA_*)."""
    n = len(prog)
    preds = [[] for _ in range(n)]
    last_w, last_sw, readers, sreaders = {}, {}, {}, {}
    is_mfma = [getattr(x, "is_mfma", False) for x in prog]
    is_load = [getattr(x, "is_load", False) for x in prog]
    for i, ins in enumerate(prog):
        for r in ins.reads:
            if r in last_w:
                j = last_w[r]
                d = VALU_RAW
                if is_load[j]:
                    d = LOAD_LATENCY
                elif is_mfma[j]:
                    d = MFMA_SPACING if r == MFMA_PIPE else MFMA_RESULT
                elif is_mfma[i]:
                    d = MFMA_OPERAND
                preds[i].append((j, d))
        for r in ins.sreads:
            if r in last_sw:
                preds[i].append((last_sw[r], SGPR_RAW))                 # W1
        for w in ins.writes:
            if w in last_w:
                j = last_w[w]
                preds[i].append((j, MFMA_RESULT if is_mfma[j] and w != MFMA_PIPE else 1))
            for j in readers.get(w, []):
                if j != i:
                    preds[i].append((j, MFMA_WAR if is_mfma[j] else 1 if is_load[i] else 2))   # W3
        for w in ins.swrites:
            if w in last_sw:
                preds[i].append((last_sw[w], 1))
            for j in sreaders.get(w, []):
                if j != i:
                    preds[i].append((j, 1))
        for r in ins.reads:
            readers.setdefault(r, []).append(i)
        for r in ins.sreads:
            sreaders.setdefault(r, []).append(i)
        for w in ins.writes:
            last_w[w] = i
            readers[w] = [i] if w in ins.reads else []
        for w in ins.swrites:
            last_sw[w] = i
            sreaders[w] = []
    succs = [[] for _ in range(n)]
    for i in range(n):
        for j, d in preds[i]:
            succs[j].append((i, d))
    prio = [0] * n
    for i in reversed(range(n)):
        prio[i] = 1 + max([prio[k] + d - 1 for k, d in succs[i]] + [0])
        if getattr(prog[i], "boost", False):
            prio[i] += 1000
    pos, order, slot, remaining = {}, [], 0, set(range(n))
    while remaining:
        ready = [i for i in remaining if all(j in pos and pos[j] + d <= slot for j, d in preds[i])]
        if ready:
            i = max(ready, key=lambda k: (prio[k], -k))
            pos[i] = slot
            order.append(prog[i])
            remaining.discard(i)
        else:
            order.append(Ins("s_nop 0", [], []))
        slot += 1
    return order


def schedule(prog):
    # a consumer of a loaded register is kept LOAD_LATENCY slots behind the load (there is other work); the waits are counted below
    order = schedule_lane(prog)
    out, pending = [], []          # pending: loads in issue order: (registers, position of issue)
    for ins in order:
        if getattr(ins, "is_load", False):
            pending.append((set(ins.writes), len(out)))
            assert len(pending) <= 15
            out.append(ins)
            continue
        need = -1
        touched = ins.reads | ins.writes
        for i, (regs, _) in enumerate(pending):
            if regs & touched:
                need = i
        if need >= 0:
            # the wait also covers the later loads that were issued long enough ago to be back: one wait per row instead of one per load
            while need + 1 < len(pending) and pending[need + 1][1] <= len(out) - LOAD_LATENCY:
                need += 1
            left = len(pending) - 1 - need
            out.append(Ins("s_waitcnt lgkmcnt(%d)" % left, [], []))
            pending = pending[need + 1:]
        out.append(ins)
    return out


def check_mfma_distances(order):
    """the final order (waits included, they only add slots) keeps every matrix-pipe distance"""
    last_mfma_write, last_mfma_read, last_valu_write, last_mfma = {}, {}, {}, None
    junk_since = {}   # registers an MFMA fills with junk rows: dead until something writes them again
    for i, ins in enumerate(order):
        m = getattr(ins, "is_mfma", False)
        for r in ins.reads:
            assert r not in junk_since, ("reads an MFMA's junk row", i, ins.text)
        for w in ins.writes:
            if w in junk_since:
                assert m or i - junk_since[w] >= MFMA_RESULT, ("writes where an MFMA's junk row is still to land", i, ins.text)
                del junk_since[w]
        for r in ins.reads:
            if r in last_mfma_write and r != MFMA_PIPE:
                assert i - last_mfma_write[r] >= MFMA_RESULT, ("MFMA result read too early", i, ins.text)
            if m and r in last_valu_write and r != MFMA_PIPE:
                assert i - last_valu_write[r] >= MFMA_OPERAND, ("MFMA operand written too late", i, ins.text)
        for w in ins.writes:
            if w == MFMA_PIPE:
                continue
            if w in last_mfma_read and not getattr(ins, "is_load", False):
                assert i - last_mfma_read[w] >= MFMA_WAR, ("MFMA operand overwritten too early", i, ins.text)
            if w in last_mfma_write:
                assert i - last_mfma_write[w] >= MFMA_RESULT, ("MFMA result overwritten too early", i, ins.text)
        if m:
            assert last_mfma is None or i - last_mfma >= MFMA_SPACING, ("MFMAs too close", i)
            last_mfma = i
            for w in getattr(ins, "junk", ()):
                junk_since[w] = i
            for r in ins.reads:
                last_mfma_read[r] = i
            for w in ins.writes:
                last_mfma_write[w] = i
        else:
            for w in ins.writes:
                last_valu_write[w] = i
                last_mfma_write.pop(w, None)


def check_hazards(order):
    check_mfma_distances(order)
    real = [o for o in order]
    for i, ins in enumerate(real):
        for back in (1, 2):
            if i - back < 0:
                continue
            assert not (ins.sreads & real[i - back].swrites), ("W1", i, ins.text)
        if i >= 1 and not ins.text.startswith(("ds_read", "s_waitcnt")):
            assert not (ins.writes & (real[i - 1].reads - real[i - 1].writes)), ("W3", i, ins.text)


# ---------------------------------------------------------------- interpreter of the matrix-pipe blocks' extra instructions
def run_lane(order, vregs, sregs):
    """gen_row_round_asm.run plus v_perm_b32, v_xor_b32, v_lshl_add_u32, and the MFMA as ONE LANE sees it (its own twelve K-values
    against the weight rows that land in it; the cross-lane map itself is checked on the device, tools/experiments/mfma_mds_probe.hip)."""
    for ins in order:
        k = ins.sem[0] if ins.sem else None
        if k == "perm":
            _, d, s0, s1, sel = ins.sem
            src = (vregs[s1][0] & M32) | ((vregs[s0][0] & M32) << 32)
            out = 0
            for i in range(4):
                out |= ((src >> (8 * ((SEL_VALUE[sel] >> (8 * i)) & 0xFF))) & 0xFF) << (8 * i)
            vregs[d] = [out]
        elif k == "xor80":
            vregs[ins.sem[1]] = [vregs[ins.sem[1]][0] ^ 0x80808080]
        elif k == "lshladd":
            _, d, a, sh, b = ins.sem
            vregs[d] = [((vregs[a][0] << sh) + vregs[b][0]) & M32]
        elif k == "mfma":
            _, d, a, b, plane = ins.sem
            assert vregs[b + 3][0] == B_CONST, "B tuple's constant dword"
            assert vregs[a + 3][0] == 0xC0DE00 + plane, ("A tuple holds another plane's constants", plane, vregs[a + 3][0])
            by = []
            for q in range(3):
                for i in range(4):
                    x = (vregs[b + q][0] >> (8 * i)) & 0xFF
                    by.append(x - 256 if x >= 128 else x)
            rc = vregs["rcbytes"][plane]
            for g in range(12):
                val = sum(mds_coef(g, j) * by[j] for j in range(12)) + (rc[g] & 0x7F) + 64 * (2 * (rc[g] >> 7) + 40) + 2 * 127 * 127
                assert 0 <= val < (1 << 17)
                vregs[d + g] = [val]
            for g in range(12, 16):  # the junk rows: anything
                vregs[d + g] = [0xDEAD0000 + g]
        else:
            G.run([ins], vregs, sregs)


# ---------------------------------------------------------------- tests
def mfma_round_constants(rc):
    """the 64-bit constants whose bytes ride in the weight tile: RC[g] = rc[g] - (K - 128 rowsum[g]) * 0x0101010101010101 mod p"""
    ones = 0x0101010101010101
    return [(rc[g] - (K_OFFSET - 128 * sum(mds_coef(g, j) for j in range(12))) * ones) % P for g in range(12)]


def test_round_mfma(order, partial):
    for _ in range(40):
        state = [rnd() for _ in range(12)]
        rc = [random.getrandbits(64) % P for _ in range(12)]
        vregs = fresh()
        for r in range(52, 76):
            vregs[r] = [random.getrandbits(32)]
        for k in range(4):
            vregs[BP[k] + 3] = [B_CONST]
        set_state(vregs, state)
        RC = mfma_round_constants(rc)
        vregs["rcbytes"] = [[(RC[g] >> (8 * b)) & 0xFF for g in range(12)] for b in range(8)]
        for b in range(8):
            vregs["mem"][("rcb", b)] = [[0xC0DE00 + b]]
        run_lane(order, vregs, {})
        want = G.reference_round(state, rc, partial)
        assert get_state(vregs) == want, partial


def rnd():
    return random.choice([0, 1, P - 1, P, M64, random.getrandbits(64), random.getrandbits(64)])


def set_state(vregs, state):
    for e in range(12):
        vregs[T + 2 * e], vregs[T + 2 * e + 1] = [state[e] & M32], [state[e] >> 32]


def get_state(vregs):
    return [(vregs[T + 2 * e][0] | (vregs[T + 2 * e + 1][0] << 32)) % P for e in range(12)]


def fresh():
    vregs = {r: [random.getrandbits(32)] for r in range(60, 256)}
    vregs[AD[0] + 1] = [0]
    vregs[AD[1] + 1] = [0]
    vregs["mem"] = {}
    return vregs


def pair4(c):
    return [[c & M32], [0], [c >> 32], [0]]


def test_round(order, partial, first_out=0):
    for _ in range(40):
        state = [rnd() for _ in range(12)]
        rc = [random.getrandbits(64) % P for _ in range(12)]
        vregs = fresh()
        set_state(vregs, state)
        for r in range(12):
            vregs["mem"][("seed", r)] = pair4(rc[r])
        G.run(order, vregs, {})
        want = G.reference_round(state, rc, partial)
        got = get_state(vregs)
        for e in range(first_out, 12):
            assert got[e] == want[e], (partial, e)


def test_four(order):
    for _ in range(30):
        state = [rnd() for _ in range(12)]
        c1, c2, c3, c4 = [[random.getrandbits(64) % P for _ in range(12)] for _ in range(4)]
        M, N2, N3, N4, k1, k2, k3, k4 = merged_tables4(c1, c2, c3, c4)
        want = state
        for c in (c1, c2, c3, c4):
            want = G.reference_round(want, c, True)
        vregs = fresh()
        set_state(vregs, state)
        mem = vregs["mem"]
        mem[("kf", 0)], mem[("kf", 1)], mem[("kf", 2)] = pair4(k1), pair4(k2), pair4(k3)
        for r in range(12):
            mem[("k4", r)] = pair4(k4[r])
            row = [N4[r][j] for j in range(12)] + [N3[r][0], N2[r][0], M[r][0], 0]
            for q in range(4):
                mem[(("row", r), q)] = [[x] for x in row[4 * q:4 * q + 4]]
        for q in range(3):
            mem[(("dot", 0), q)] = [[M[0][j]] for j in range(4 * q, 4 * q + 4)]
            mem[(("dot", 1), q)] = [[N2[0][j]] for j in range(4 * q, 4 * q + 4)]
            mem[(("dot", 2), q)] = [[N3[0][j]] for j in range(4 * q, 4 * q + 4)]
        mem[(("dot", 2), 3)] = [[N2[0][0]]]
        G.run(order, vregs, {})
        assert get_state(vregs) == want


def emit(name, order, what):
    n_wait = sum(1 for o in order if o.text.startswith("s_waitcnt"))
    n_nop = sum(1 for o in order if o.text.startswith("s_nop"))
    n_lds = sum(1 for o in order if o.text.startswith("ds_read"))
    print("// %s: %d instructions (%d LDS loads, %d s_waitcnt, %d s_nop)" % (what, len(order), n_lds, n_wait, n_nop))
    print("#define %s \\" % name)
    for i, o in enumerate(order):
        last = i == len(order) - 1
        print('    "%s%s"%s' % (o.text, "" if last else "\\n\\t", "" if last else " \\"))


def main():
    random.seed(5)
    print("// generated by tools/gen_lane_round_asm.py -- do not edit.  Physical registers: state v[%d:%d] (in and out), LDS addresses v%d (k3) v%d (k12)" %
          (T, T + 23, A_K3, A_K12))
    print("// v%d (coefficient rows) v%d (next round's constants), zeros v%d v%d; v%d .. v255 and s%d .. s%d are clobbered." %
          (A_COEF, A_SEED, AD[0] + 1, AD[1] + 1, S, SINK, FCS[1] + 1))
    for name, prog, tester, what in (
            ("STARKHIP_LANE_FULL_ROUND_ASM", block_full(), lambda o: test_round(o, False), "full round: twelve S-boxes, circulant layer"),
            ("STARKHIP_LANE_LAST_ROUND_ASM", block_full(8), lambda o: test_round(o, False, 8), "last full round before an absorb: the capacity outputs only"),
            ("STARKHIP_LANE_PARTIAL_ROUND_ASM", block_partial(), lambda o: test_round(o, True), "partial round"),
            ("STARKHIP_LANE_FOUR_ASM", block_four(), test_four, "four partial rounds at once"),
            ("STARKHIP_LANE_FULL_ROUND_MFMA_ASM", block_full_mfma(), lambda o: test_round_mfma(o, False), "full round, circulant layer on the matrix pipe"),
            ("STARKHIP_LANE_PARTIAL_ROUND_MFMA_ASM", block_partial_mfma(), lambda o: test_round_mfma(o, True), "partial round, circulant layer on the matrix pipe")):
        order = schedule(prog)
        check_hazards(order)
        tester(order)
        emit(name, order, what)
    for i in range(3):
        print('#define STARKHIP_LANE_STATE%d "+{v[%d:%d]}"' % (i, T + 8 * i, T + 8 * i + 7))
    print('#define STARKHIP_LANE_A_K3 "{v%d}"' % A_K3)
    print('#define STARKHIP_LANE_A_K12 "{v%d}"' % A_K12)
    print('#define STARKHIP_LANE_A_COEF "{v%d}"' % A_COEF)
    print('#define STARKHIP_LANE_A_SEED "{v%d}"' % A_SEED)
    print('#define STARKHIP_LANE_ZA "{v%d}"' % (AD[0] + 1))
    print('#define STARKHIP_LANE_ZB "{v%d}"' % (AD[1] + 1))
    bound = set(range(T, T + 24)) | {AD[0] + 1, AD[1] + 1}
    vs = [r for r in range(S, 256) if r not in bound]
    ss = list(range(SINK, FCS[1] + 2)) + list(range(FC2[0], FC2[1] + 2))
    print("#define STARKHIP_LANE_CLOBBERS %s" % ", ".join(['"v%d"' % r for r in vs] + ['"s%d"' % r for r in ss]))
    # the matrix-pipe blocks: weight tiles (dword 3 of each is loaded inside: in / out), the B tuples' constant dwords, the constant
    # table's address, the selectors and constants in scalar registers; the B tuples' other dwords are clobbered on top of the rest
    for k in range(2):
        for d in range(3):
            print('#define STARKHIP_LANE_AW%d%d "{v%d}"' % (k, d, AW[k] + d))
        print('#define STARKHIP_LANE_AW%d3 "+{v%d}"' % (k, AW[k] + 3))
    for k in range(4):
        print('#define STARKHIP_LANE_BC%d "{v%d}"' % (k, BP[k] + 3))
    print('#define STARKHIP_LANE_A_RCB "{v%d}"' % A_RCB)
    for name, reg in (("SEL_A", S_SEL["A"]), ("SEL_B", S_SEL["B"]), ("SEL_C", S_SEL["C"]), ("SEL_D", S_SEL["D"]), ("X80", S_X80), ("K64K", S_64K)):
        print('#define STARKHIP_LANE_S_%s "{s%d}"' % (name, reg))
    for name, val in sorted(SEL_VALUE.items()):
        print("#define STARKHIP_LANE_SEL_%s_VALUE 0x%08xu" % (name, val))
    print("#define STARKHIP_LANE_B_CONST 0x%08xu" % B_CONST)
    print("#define STARKHIP_LANE_K_OFFSET %du" % K_OFFSET)
    print("#define STARKHIP_LANE_MFMA_CLOBBERS %s" % ", ".join(['"v%d"' % (BP[k] + d) for k in range(4) for d in range(3)]))


if __name__ == "__main__":
    main()
