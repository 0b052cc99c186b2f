#!/usr/bin/env python3
"""Generates csrc/row_round_asm.inc: one whole round of the ROW form of the Poseidon permutation (poseidon_dev.h:
poseidon_permute_row) as ONE scheduled inline-asm block on fixed physical registers -- S-box, circulant layer, fold.

The row form exists for commitments too small to fill the chip: a lone wave issues one instruction per ~5 cycles whatever it
depends on, and every wait state that hipcc fills with s_nop is a lost slot.  The wait states this code meets on gfx950:
  W1  a VALU may read an SGPR (carry / borrow flag) two instructions after the VALU that wrote it, not earlier;
  W2  a DPP move may read a VGPR two instructions after it was written;
  W3  a VGPR may not be written in the slot right after an instruction that read it.
A list scheduler orders each block under those rules.  What makes the difference is what it is given to fill the slots with:
  * full round: x^3 and x^4 of the S-box are independent multiplies;
  * partial round: only lane 0's element passes the S-box, and  M s' = M (s with element 0 zeroed) + (column 0 of M) x0^7, so the
    whole circulant layer over the other eleven elements is issued UNDER lane 0's three dependent multiplies, and x0^7 enters at
    the end as one broadcast (three DPP moves per half) and two multiply-adds with a per-lane coefficient.
Inline asm cannot name the halves of an operand pair, so every operand is bound to a physical register ("{v231}") and the text
uses register names; temporaries are clobbers.  Every block is executed here by an interpreter of the instructions used, on random
16-lane rows, against the round computed with Python integers.

    python tools/gen_row_round_asm.py > starky_bls12_381_amd/csrc/row_round_asm.inc
"""
import random

P = 0xFFFFFFFF00000001
M64 = (1 << 64) - 1
M32 = (1 << 32) - 1
CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
NL = 16


class Ins:
    def __init__(self, text, reads, writes, sreads=(), swrites=(), dpp=False, sem=None):
        self.text, self.reads, self.writes = text, set(reads), set(writes)
        self.sreads, self.swrites, self.dpp, self.sem = set(sreads), set(swrites), dpp, sem


def v(n):
    return "v%d" % n


def vp(n):
    assert n % 2 == 0, "64-bit VGPR operands are even-aligned on gfx950"
    return "v[%d:%d]" % (n, n + 1)


def sp(n):
    return "s[%d:%d]" % (n, n + 1)


# ---------------------------------------------------------------- register map (VGPRs 176 .. 255, SGPRs 40 .. 99)
ADDR = 175                     # in: LDS address of this lane's round-constant row (RcPair rc[32])
S_LO, S_HI = 176, 177          # in: the lane's state word; out: the new one (v[176:177])
SEED_A, SEED_B = 178, 180      # in: the next round's constant, low half and high half, as 64-bit addends; out: the same for the round
                               # after (the block reloads v[178:181] from LDS as soon as it has consumed them: the load's latency
                               # passes under the rest of the round instead of in front of the next one)
C0, COL0 = 182, 183            # in: 17 (+ 8 on lane 0); column 0 of the MDS matrix at this lane (partial rounds)
ZA, ZB = 185, 187              # in: zero (upper halves of the two multiply slots' addend pairs v[184:185], v[186:187])
ACC_A, ACC_B = 188, 190        # the layer's two accumulators
FT, CV = 192, 194              # fold: T pair, carry value
X2, X3, X4, X7, SEL, YN, BC = 196, 198, 200, 202, 204, 206, 208   # S-box values (pairs)
SZ = 210                       # the state with lane 0 zeroed (pair)
ROT = 212                      # rotated operands of the layer: 24 registers 212 .. 235 (+ z, w copies 236 .. 239)
ZC, WC = 236, 238
SLOT = [240, 248]              # two multiply slots of eight registers
MASK0 = 40                     # in: lane-0 mask (bit 0 of every row)
SINK = 42
FLAG = [44, 52]                # per slot: CM, BR, BR2, CY pairs
FC = 60
MASKE = 62                     # in: even lanes (the uniform S-box of the merged triple)
# the merged triple's operands and extra temporaries
N3K = 140                      # in: 12 registers, coefficient of the operand rotated by k: N3[e][(e + k) mod 12]
R1, R2, B2, B3 = 152, 153, 154, 155      # in: M[0][e], N2[0][e], N2[e][0], M[e][0]
N3C0, L0M, L0N = 156, 157, 158           # in: N3[e][0]; M[0][0] and N2[0][0] on lane 0, 0 elsewhere
K1, K2, K3 = 160, 164, 168     # in / out: the triple's constants, each (low half, high half) as two 64-bit addends; k1, k2 on lane 0 only
D1A, D1B, D2A, D2B = 122, 124, 126, 128  # the two dot products' accumulators
RT = 130                       # all-reduce: rotated copies (two pairs)
YY, XS2, XS3 = 134, 136, 138   # y (folded), x2, x3


class Slot:
    def __init__(self, k):
        b = SLOT[k]
        self.P0, self.M, self.P3, self.t, self.AD = b, b + 2, b + 4, b + 6, (184, 186)[k]
        f = FLAG[k]
        self.CM, self.BR, self.BR2, self.CY = f, f + 2, f + 4, f + 6


def mul(prog, dst, a, b, s):
    """dst = a * b mod p (any representative): gl_dev.h's gl_mul_nc, 13 instructions.  a, b: (lo, hi) registers; dst: even pair."""
    a0, a1 = a
    b0, b1 = b
    R, AD = dst, s.AD
    prog += [
        Ins("v_mad_u64_u32 %s, %s, %s, %s, 0" % (vp(s.P0), sp(SINK), v(a0), v(b0)), [a0, b0], [s.P0, s.P0 + 1], sem=("mad", s.P0, None, a0, b0, None)),
        Ins("v_mov_b32 %s, %s" % (v(AD), v(s.P0 + 1)), [s.P0 + 1], [AD], sem=("mov", AD, s.P0 + 1)),
        Ins("v_mad_u64_u32 %s, %s, %s, %s, %s" % (vp(s.M), sp(SINK), v(a0), v(b1), vp(AD)), [a0, b1, AD, AD + 1], [s.M, s.M + 1], sem=("mad", s.M, None, a0, b1, AD)),
        Ins("v_mad_u64_u32 %s, %s, %s, %s, %s" % (vp(s.M), sp(s.CM), v(a1), v(b0), vp(s.M)), [a1, b0, s.M, s.M + 1], [s.M, s.M + 1], swrites=[s.CM],
            sem=("mad", s.M, s.CM, a1, b0, s.M)),
        Ins("v_mov_b32 %s, %s" % (v(AD), v(s.M + 1)), [s.M + 1], [AD], sem=("mov", AD, s.M + 1)),
        Ins("v_mad_u64_u32 %s, %s, %s, %s, %s" % (vp(s.P3), sp(SINK), v(a1), v(b1), vp(AD)), [a1, b1, AD, AD + 1], [s.P3, s.P3 + 1], sem=("mad", s.P3, None, a1, b1, AD)),
        # D = (l1 : l0) - h1 - cin, in place over P0
        Ins("v_subb_co_u32 %s, %s, %s, %s, %s" % (v(s.P0), sp(s.BR), v(s.P0), v(s.P3 + 1), sp(s.CM)), [s.P0, s.P3 + 1], [s.P0], sreads=[s.CM], swrites=[s.BR],
            sem=("subb", s.P0, s.BR, s.P0, s.P3 + 1, s.CM)),
        Ins("v_subb_co_u32 %s, %s, %s, 0, %s" % (v(s.P0 + 1), sp(s.BR2), v(s.M), sp(s.BR)), [s.M], [s.P0 + 1], sreads=[s.BR], swrites=[s.BR2],
            sem=("subb", s.P0 + 1, s.BR2, s.M, None, s.BR)),
        Ins("v_mad_u64_u32 %s, %s, %s, -1, %s" % (vp(R), sp(s.CY), v(s.P3), vp(s.P0)), [s.P3, s.P0, s.P0 + 1], [R, R + 1], swrites=[s.CY],
            sem=("mad", R, s.CY, s.P3, "eps", s.P0)),
        Ins("v_subb_co_u32 %s, %s, 0, 0, %s" % (v(s.t), sp(SINK), sp(s.BR2)), [], [s.t], sreads=[s.BR2], sem=("subb", s.t, None, None, None, s.BR2)),
        Ins("v_addc_co_u32 %s, %s, %s, 0, %s" % (v(s.t), sp(SINK), v(s.t), sp(s.CY)), [s.t], [s.t], sreads=[s.CY], sem=("addc", s.t, None, s.t, None, s.CY)),
        Ins("v_mad_i64_i32 %s, %s, %s, -1, %s" % (vp(R), sp(SINK), v(s.t), vp(R)), [s.t, R, R + 1], [R, R + 1], sem=("madi", R, s.t)),
        Ins("v_add_u32 %s, %s, %s" % (v(R + 1), v(s.t), v(R + 1)), [s.t, R + 1], [R + 1], sem=("add", R + 1, s.t, R + 1)),
    ]


def dpp(prog, dst, src, ctrl, bank=0xF, bound=True, kind=None):
    text = "v_mov_b32_dpp %s, %s %s row_mask:0xf bank_mask:0x%x%s" % (v(dst), v(src), ctrl, bank, " bound_ctrl:1" if bound else "")
    reads = [src] if bound and bank == 0xF else [src, dst]   # lanes that are not written keep the old value
    prog.append(Ins(text, reads, [dst], dpp=True, sem=("dpp", dst, src, kind, bank, bound)))


def shl(prog, dst, src, k):
    dpp(prog, dst, src, "row_shl:%d" % k, kind=("shl", k))


def mirror(prog, reg):  # lanes 12 .. 15 <- lanes 0 .. 3
    dpp(prog, reg, reg, "row_shr:12", bank=0x8, bound=False, kind=("shr", 12))


def quad(prog, dst, src, sel):
    dpp(prog, dst, src, "quad_perm:[%d,%d,%d,%d]" % tuple(sel), kind=("quad", tuple(sel)))


def cndmask(prog, dst, a, b, mask):
    """dst = mask ? b : a; a, b: register numbers or the literal 0"""
    ta = "0" if a is None else v(a)
    tb = "0" if b is None else v(b)
    prog.append(Ins("v_cndmask_b32 %s, %s, %s, %s" % (v(dst), ta, tb, sp(mask)), [r for r in (a, b) if r is not None], [dst], sem=("cnd", dst, a, b, mask)))


def madc(prog, acc, src, coef, seed=None):
    """acc (pair) = src * coef + (seed or acc); coef: an inline constant or ('v', register)"""
    add = acc if seed is None else seed
    if isinstance(coef, tuple):
        prog.append(Ins("v_mad_u64_u32 %s, %s, %s, %s, %s" % (vp(acc), sp(SINK), v(src), v(coef[1]), vp(add)), [src, coef[1], add, add + 1], [acc, acc + 1],
                        sem=("mad", acc, None, src, coef[1], add)))
    else:
        prog.append(Ins("v_mad_u64_u32 %s, %s, %s, %d, %s" % (vp(acc), sp(SINK), v(src), coef, vp(add)), [src, add, add + 1], [acc, acc + 1],
                        sem=("mad", acc, None, src, ("const", coef), add)))


def layer(prog, lo, hi):
    """ACC_A / ACC_B = seeds + sum_k CIRC[k] * (halves of element (e + k) mod 12).  lo / hi are mirrored IN PLACE."""
    r = ROT
    madc(prog, ACC_A, lo, ("v", C0), seed=SEED_A)
    madc(prog, ACC_B, hi, ("v", C0), seed=SEED_B)
    prefetch(prog)
    mirror(prog, lo)
    mirror(prog, hi)
    base = (lo, hi)
    for group, copy in ((0, (ZC, ZC + 1)), (1, (WC, WC + 1)), (2, None)):
        ks = (1, 2, 3, 4) if group < 2 else (1, 2, 3)
        for k in ks:
            kk = 4 * group + k
            dl, dh = (copy if k == 4 else (r, r + 1))
            if k != 4:
                r += 2
            shl(prog, dl, base[0], k)
            shl(prog, dh, base[1], k)
            madc(prog, ACC_A, dl, CIRC[kk])
            madc(prog, ACC_B, dh, CIRC[kk])
        if copy is not None:
            mirror(prog, copy[0])
            mirror(prog, copy[1])
            base = copy


def fold(prog):
    """v[S_LO:S_HI] = ACC_A + ACC_B * 2^32 mod p (combine_lohi_nc)"""
    prog.append(Ins("v_mad_u64_u32 %s, %s, %s, -1, %s" % (vp(FT), sp(SINK), v(ACC_B + 1), vp(ACC_A)), [ACC_B + 1, ACC_A, ACC_A + 1], [FT, FT + 1],
                    sem=("mad", FT, None, ACC_B + 1, "eps", ACC_A)))
    prog.append(Ins("v_add_co_u32 %s, %s, %s, %s" % (v(FT + 1), sp(FC), v(FT + 1), v(ACC_B)), [FT + 1, ACC_B], [FT + 1], swrites=[FC], sem=("addco", FT + 1, FC, FT + 1, ACC_B)))
    prog.append(Ins("v_addc_co_u32 %s, %s, 0, 0, %s" % (v(CV), sp(SINK), sp(FC)), [], [CV], sreads=[FC], sem=("addc", CV, None, None, None, FC)))
    prog.append(Ins("v_mad_u64_u32 %s, %s, %s, -1, %s" % (vp(S_LO), sp(SINK), v(CV), vp(FT)), [CV, FT, FT + 1], [S_LO, S_LO + 1], sem=("mad", S_LO, None, CV, "eps", FT)))


def seeds_ready(prog):
    """the previous block's (or the caller's) load of v[178:181] has landed"""
    prog.append(Ins("s_waitcnt lgkmcnt(0)", [], [SEED_A, SEED_A + 1, SEED_B, SEED_B + 1], sem=("nopsem",)))


def prefetch(prog):
    prog.append(Ins("ds_read_b128 v[%d:%d], v%d offset:%%[off]" % (SEED_A, SEED_B + 1, ADDR), [ADDR], [SEED_A, SEED_A + 1, SEED_B, SEED_B + 1], sem=("prefetch",)))
    prog[-1].boost = True


def fold_to(prog, dst, A, B):
    """v[dst:dst+1] = A + B * 2^32 mod p"""
    prog.append(Ins("v_mad_u64_u32 %s, %s, %s, -1, %s" % (vp(FT), sp(SINK), v(B + 1), vp(A)), [B + 1, A, A + 1], [FT, FT + 1], sem=("mad", FT, None, B + 1, "eps", A)))
    prog.append(Ins("v_add_co_u32 %s, %s, %s, %s" % (v(FT + 1), sp(FC), v(FT + 1), v(B)), [FT + 1, B], [FT + 1], swrites=[FC], sem=("addco", FT + 1, FC, FT + 1, B)))
    prog.append(Ins("v_addc_co_u32 %s, %s, 0, 0, %s" % (v(CV), sp(SINK), sp(FC)), [], [CV], sreads=[FC], sem=("addc", CV, None, None, None, FC)))
    prog.append(Ins("v_mad_u64_u32 %s, %s, %s, -1, %s" % (vp(dst), sp(SINK), v(CV), vp(FT)), [CV, FT, FT + 1], [dst, dst + 1], sem=("mad", dst, None, CV, "eps", FT)))


def layer_dense(prog, lo, hi):
    """ACC_A / ACC_B = K3 + sum_k n3k[k] * (halves of element (e + k) mod 12): any 12 x 12 layer, per-lane coefficients"""
    r = ROT
    madc(prog, ACC_A, lo, ("v", N3K + 0), seed=K3)
    madc(prog, ACC_B, hi, ("v", N3K + 0), seed=K3 + 2)
    mirror(prog, lo)
    mirror(prog, hi)
    base = (lo, hi)
    for group, copy in ((0, (ZC, ZC + 1)), (1, (WC, WC + 1)), (2, None)):
        ks = (1, 2, 3, 4) if group < 2 else (1, 2, 3)
        for k in ks:
            kk = 4 * group + k
            dl, dh = (copy if k == 4 else (r, r + 1))
            if k != 4:
                r += 2
            shl(prog, dl, base[0], k)
            shl(prog, dh, base[1], k)
            madc(prog, ACC_A, dl, ("v", N3K + kk))
            madc(prog, ACC_B, dh, ("v", N3K + kk))
        if copy is not None:
            mirror(prog, copy[0])
            mirror(prog, copy[1])
            base = copy


def add64(prog, dst, a, b):
    prog.append(Ins("v_lshl_add_u64 %s, %s, 0, %s" % (vp(dst), vp(a), vp(b)), [a, a + 1, b, b + 1], [dst, dst + 1], sem=("add64", dst, a, b)))


def allreduce(prog, acc):
    """the row's sum of a 64-bit accumulator, in every lane"""
    for k in (8, 4, 2, 1):
        for h in (0, 1):
            dpp(prog, RT + h, acc + h, "row_ror:%d" % k, kind=("ror", k))
        add64(prog, acc, acc, RT)


def sbox_uniform(prog, dst, y, slot):
    """dst = y^7 for a value every lane holds: even lanes form x^3, odd lanes x^4, each takes the other factor from its neighbour"""
    yy = (y, y + 1)
    mul(prog, X2, yy, yy, slot)
    for h in (0, 1):
        cndmask(prog, SEL + h, X2 + h, y + h, MASKE)            # even lanes: y, odd lanes: y^2
    mul(prog, X4, (X2, X2 + 1), (SEL, SEL + 1), slot)            # even: y^3, odd: y^4
    for h in (0, 1):
        quad(prog, YN + h, X4 + h, (1, 0, 3, 2))
    mul(prog, dst, (X4, X4 + 1), (YN, YN + 1), slot)


def block_triple():
    """three partial rounds (poseidon_merged.h): in: the state with the first round's constants added; out: the state three rounds
    later with the following round's constants added"""
    prog = []
    prog.append(Ins("s_waitcnt lgkmcnt(0)", [], list(range(K1, K3 + 4)), sem=("nopsem",)))
    a, b = Slot(0), Slot(1)
    x = (S_LO, S_HI)
    cndmask(prog, SZ, S_LO, None, MASK0)
    cndmask(prog, SZ + 1, S_HI, None, MASK0)
    # the parts of both dot products and of the dense layer that do not wait for x1
    madc(prog, D1A, SZ, ("v", R1), seed=K1)
    madc(prog, D1B, SZ + 1, ("v", R1), seed=K1 + 2)
    madc(prog, D2A, SZ, ("v", R2), seed=K2)
    madc(prog, D2B, SZ + 1, ("v", R2), seed=K2 + 2)
    layer_dense(prog, SZ, SZ + 1)
    for j, K in enumerate((K1, K2, K3)):
        prog.append(Ins("ds_read_b128 v[%d:%d], v%d offset:%%[off%d]" % (K, K + 3, ADDR, j + 1), [ADDR], [K, K + 1, K + 2, K + 3], sem=("prefetch3", K, j)))
        prog[-1].boost = True
    # x1 = u0^7 on lane 0 (lane 1 forms x^4 while lane 0 forms x^3)
    mul(prog, X2, x, x, a)
    for h in (0, 1):
        quad(prog, X3 + h, X2 + h, (0, 0, 2, 3))
        cndmask(prog, SEL + h, X3 + h, x[h], MASK0)
    mul(prog, X4, (X3, X3 + 1), (SEL, SEL + 1), a)
    for h in (0, 1):
        quad(prog, YN + h, X4 + h, (1, 1, 2, 3))
    mul(prog, X7, (X4, X4 + 1), (YN, YN + 1), a)          # lane 0: x1
    # y1 = (M ut)[0] + k1
    madc(prog, D1A, X7, ("v", L0M))
    madc(prog, D1B, X7 + 1, ("v", L0M))
    allreduce(prog, D1A)
    allreduce(prog, D1B)
    fold_to(prog, YY, D1A, D1B)
    # x1 to every lane for the dense layer (its values in lanes 1 .. 3 are not x1: take lane 0 explicitly)
    for h in (0, 1):
        quad(prog, BC + h, X7 + h, (0, 0, 0, 0))
        dpp(prog, BC + h, BC + h, "row_shr:4", bank=0x2, bound=False, kind=("shr", 4))
        dpp(prog, BC + h, BC + h, "row_shr:8", bank=0x4, bound=False, kind=("shr", 8))
    madc(prog, ACC_A, BC, ("v", N3C0))
    madc(prog, ACC_B, BC + 1, ("v", N3C0))
    # y2 = (N2 ut)[0] + M00 x2 + k2: everything but the x2 term is summed over the row while x2 is being computed
    madc(prog, D2A, X7, ("v", L0N))
    madc(prog, D2B, X7 + 1, ("v", L0N))
    allreduce(prog, D2A)
    allreduce(prog, D2B)
    sbox_uniform(prog, XS2, YY, b)
    madc(prog, D2A, XS2, 25)
    madc(prog, D2B, XS2 + 1, 25)
    fold_to(prog, YY, D2A, D2B)
    madc(prog, ACC_A, XS2, ("v", B2))
    madc(prog, ACC_B, XS2 + 1, ("v", B2))
    sbox_uniform(prog, XS3, YY, b)
    madc(prog, ACC_A, XS3, ("v", B3))
    madc(prog, ACC_B, XS3 + 1, ("v", B3))
    fold_to(prog, S_LO, ACC_A, ACC_B)
    return prog


def block_full():
    prog = []
    seeds_ready(prog)
    a, b = Slot(0), Slot(1)
    x = (S_LO, S_HI)
    mul(prog, X2, x, x, a)
    mul(prog, X4, (X2, X2 + 1), (X2, X2 + 1), a)
    mul(prog, X3, (X2, X2 + 1), x, b)
    mul(prog, X7, (X3, X3 + 1), (X4, X4 + 1), a)
    layer(prog, X7, X7 + 1)
    fold(prog)
    return prog


def block_partial():
    prog = []
    seeds_ready(prog)
    a = Slot(0)
    x = (S_LO, S_HI)
    # the layer over the state with lane 0 zeroed: independent of the S-box
    cndmask(prog, SZ, S_LO, None, MASK0)
    cndmask(prog, SZ + 1, S_HI, None, MASK0)
    layer(prog, SZ, SZ + 1)
    # lane 0's x^7: lane 1 forms x^4 while lane 0 forms x^3
    mul(prog, X2, x, x, a)
    for h in (0, 1):
        quad(prog, X3 + h, X2 + h, (0, 0, 2, 3))          # lanes 0 and 1 read lane 0's x^2
        cndmask(prog, SEL + h, X3 + h, x[h], MASK0)        # lane 0: x, others: x^2
    mul(prog, X4, (X3, X3 + 1), (SEL, SEL + 1), a)         # lane 0: x^3, lane 1: x^4
    for h in (0, 1):
        quad(prog, YN + h, X4 + h, (1, 1, 2, 3))           # lane 0 reads lane 1
    mul(prog, X7, (X4, X4 + 1), (YN, YN + 1), a)           # lane 0: x^7
    # x^7 of lane 0 to lanes 0 .. 11, times column 0 of the matrix
    for h in (0, 1):
        quad(prog, BC + h, X7 + h, (0, 0, 0, 0))
        dpp(prog, BC + h, BC + h, "row_shr:4", bank=0x2, bound=False, kind=("shr", 4))
        dpp(prog, BC + h, BC + h, "row_shr:8", bank=0x4, bound=False, kind=("shr", 8))
    madc(prog, ACC_A, BC, ("v", COL0))
    madc(prog, ACC_B, BC + 1, ("v", COL0))
    fold(prog)
    return prog


# ---------------------------------------------------------------- scheduler
def schedule(prog):
    return schedule_with(prog, None)


def schedule_with(prog, adjust):
    """adjust(producer, consumer, distance) -> distance: a hook for read-after-write distances (load latencies of other generators)"""
    n = len(prog)
    preds = [[] for _ in range(n)]
    last_w, last_sw, readers, sreaders = {}, {}, {}, {}
    for i, ins in enumerate(prog):
        for r in ins.reads:
            if r in last_w:
                d = 3 if ins.dpp else 1                                  # W2
                preds[i].append((last_w[r], adjust(prog[last_w[r]], ins, d) if adjust else d))
        for r in ins.sreads:
            if r in last_sw:
                preds[i].append((last_sw[r], 3))                        # W1
        for w in ins.writes:
            if w in last_w:
                preds[i].append((last_w[w], 1))
            for j in readers.get(w, []):
                if j != i:
                    preds[i].append((j, 1 if ins.text.startswith("ds_read") else 2))   # W3 (an LDS load lands much later anyway)
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


def check_hazards(order, inputs):
    """inputs: registers the caller's code may have written right before the block"""
    for i, ins in enumerate(order):
        for back in (1, 2):
            if i - back < 0:
                if ins.dpp:
                    assert not (ins.reads & inputs), ("W2 at the block's start", i, ins.text)
                continue
            p = order[i - back]
            assert not (ins.sreads & p.swrites), ("W1", i, ins.text)
            if ins.dpp:
                assert not (ins.reads & p.writes), ("W2", i, ins.text)
        if i >= 1 and not ins.text.startswith(("ds_read", "s_waitcnt")):
            assert not (ins.writes & (order[i - 1].reads - order[i - 1].writes)), ("W3", i, ins.text)


# ---------------------------------------------------------------- interpreter (one row of 16 lanes)
def run(order, vregs, sregs):
    def V(r):
        return vregs.setdefault(r, [0] * NL)

    def S(r):
        return sregs.setdefault(r, [0] * NL)
    for ins in order:
        if ins.sem is None:
            continue
        k = ins.sem[0]
        if k == "nopsem":
            continue
        if k == "ldsload":  # ("ldsload", first register, count, key): registers <- vregs["mem"][key] (a list of per-lane lists)
            _, first, count, key = ins.sem
            for q in range(count):
                vregs[first + q] = list(vregs["mem"][key][q])
            continue
        if k == "prefetch3":
            _, K, j = ins.sem
            for q in range(4):
                vregs[K + q] = list(vregs["next_k"][j][q])
            continue
        if k == "add64":
            _, d, a, b = ins.sem
            lo, hi = [0] * NL, [0] * NL
            for l in range(NL):
                x = ((V(a)[l] | (V(a + 1)[l] << 32)) + (V(b)[l] | (V(b + 1)[l] << 32))) & M64
                lo[l], hi[l] = x & M32, x >> 32
            vregs[d], vregs[d + 1] = lo, hi
            continue
        if k == "prefetch":
            for j, r in enumerate((SEED_A, SEED_A + 1, SEED_B, SEED_B + 1)):
                vregs[r] = list(vregs["next_seeds"][j])
            continue
        if k == "mad":
            _, d, cout, a, b, c = ins.sem
            lo, hi, co = [0] * NL, [0] * NL, [0] * NL
            for l in range(NL):
                bb = M32 if b == "eps" else (b[1] if isinstance(b, tuple) else V(b)[l])
                add = 0 if c is None else V(c)[l] | (V(c + 1)[l] << 32)
                x = V(a)[l] * bb + add
                co[l] = x >> 64
                lo[l], hi[l] = x & M32, (x >> 32) & M32
            vregs[d], vregs[d + 1] = lo, hi
            if cout is not None:
                sregs[cout] = co
            else:
                assert not any(co), ("a multiply-add whose carry-out nobody reads overflowed", ins.text)
        elif k == "mov":
            vregs[ins.sem[1]] = V(ins.sem[2])[:]
        elif k == "mov64":
            vregs[ins.sem[1]], vregs[ins.sem[1] + 1] = V(ins.sem[2])[:], V(ins.sem[2] + 1)[:]
        elif k == "subb":
            _, d, bout, a, b, bin_ = ins.sem
            out, bo = [0] * NL, [0] * NL
            for l in range(NL):
                x = (0 if a is None else V(a)[l]) - (0 if b is None else V(b)[l]) - S(bin_)[l]
                bo[l] = 1 if x < 0 else 0
                out[l] = x & M32
            vregs[d] = out
            if bout is not None:
                sregs[bout] = bo
        elif k == "addc":
            _, d, _, a, _, cin = ins.sem
            vregs[d] = [((0 if a is None else V(a)[l]) + S(cin)[l]) & M32 for l in range(NL)]
        elif k == "addco":
            _, d, cout, a, b = ins.sem
            x = [V(a)[l] + V(b)[l] for l in range(NL)]
            vregs[d] = [t & M32 for t in x]
            sregs[cout] = [t >> 32 for t in x]
        elif k == "madi":
            _, d, t = ins.sem
            lo, hi = [0] * NL, [0] * NL
            for l in range(NL):
                tv = V(t)[l] - (1 << 32) if V(t)[l] >> 31 else V(t)[l]
                x = ((V(d)[l] | (V(d + 1)[l] << 32)) - tv) & M64
                lo[l], hi[l] = x & M32, x >> 32
            vregs[d], vregs[d + 1] = lo, hi
        elif k == "add":
            _, d, a, b = ins.sem
            vregs[d] = [(V(a)[l] + V(b)[l]) & M32 for l in range(NL)]
        elif k == "dpp":
            _, d, s, kind, bank, bound = ins.sem
            src, old = V(s)[:], V(d)[:]
            out = old[:]
            for l in range(NL):
                if not (bank >> (l // 4)) & 1:
                    continue
                if kind[0] == "shl":
                    j = l + kind[1]
                elif kind[0] == "shr":
                    j = l - kind[1]
                elif kind[0] == "ror":
                    j = (l - kind[1]) % NL
                else:
                    j = (l & ~3) + kind[1][l & 3]
                if 0 <= j < NL:
                    out[l] = src[j]
                elif bound:
                    out[l] = 0
            vregs[d] = out
        elif k == "cnd":
            _, d, a, b, m = ins.sem
            av = [0] * NL if a is None else V(a)[:]
            bv = [0] * NL if b is None else V(b)[:]
            vregs[d] = [bv[l] if S(m)[l] else av[l] for l in range(NL)]


def reference_round(state, rc_next, partial):
    s = [pow(x, 7, P) if (not partial or e == 0) else x % P for e, x in enumerate(state)]
    out = []
    for e in range(12):
        acc = sum(CIRC[k] * s[(e + k) % 12] for k in range(12)) + (8 * s[0] if e == 0 else 0)
        out.append((acc + rc_next[e]) % P)
    return out


def test(order, partial):
    for _ in range(100):
        state = [random.choice([0, 1, P - 1, P, M64, random.getrandbits(64), random.getrandbits(64)]) for _ in range(12)]
        rc = [random.getrandbits(64) % P for _ in range(12)]
        vregs = {r: [random.getrandbits(32) for _ in range(NL)] for r in range(176, 256)}
        vregs[S_LO] = [x & M32 for x in state] + [random.getrandbits(32) for _ in range(4)]
        vregs[S_HI] = [x >> 32 for x in state] + [random.getrandbits(32) for _ in range(4)]
        vregs[SEED_A] = [c & M32 for c in rc] + [0] * 4
        vregs[SEED_A + 1] = [0] * NL
        vregs[SEED_B] = [c >> 32 for c in rc] + [0] * 4
        vregs[SEED_B + 1] = [0] * NL
        vregs[C0] = [25] + [17] * 15
        vregs[COL0] = [25] + [CIRC[(12 - e) % 12] for e in range(1, 12)] + [0] * 4
        vregs[ZA] = [0] * NL
        vregs[ZB] = [0] * NL
        sregs = {MASK0: [1] + [0] * 15}
        vregs["next_seeds"] = [[random.getrandbits(32) for _ in range(NL)] for _ in range(4)]
        run(order, vregs, sregs)
        for j, r in enumerate((SEED_A, SEED_A + 1, SEED_B, SEED_B + 1)):
            assert vregs[r] == vregs["next_seeds"][j]
        want = reference_round(state, rc, partial)
        for e in range(12):
            got = vregs[S_LO][e] | (vregs[S_LO + 1][e] << 32)
            assert got % P == want[e], (partial, e)


def merged_tables(c1, c2, c3):
    """poseidon_merged.h on Python integers: M, N2, N3 and k1, k2, k3 for the constants of the three following rounds"""
    M = [[CIRC[(j - i) % 12] + (8 if i == 0 and j == 0 else 0) for j in range(12)] for i in range(12)]
    Mz = [[0] * 12 if i == 0 else M[i][:] for i in range(12)]

    def mm(a, b):
        return [[sum(a[i][k] * b[k][j] for k in range(12)) for j in range(12)] for i in range(12)]

    def mv(a, x):
        return [sum(a[i][j] * x[j] for j in range(12)) % P for i in range(12)]
    N2 = mm(M, Mz)
    N3 = mm(N2, Mz)
    c1z, c2z = [0] + c1[1:], [0] + c2[1:]
    k1 = c1[0]
    k2 = (mv(M, c1z)[0] + c2[0]) % P
    a, b = mv(N2, c1z), mv(M, c2z)
    k3 = [(a[i] + b[i] + c3[i]) % P for i in range(12)]
    return M, N2, N3, k1, k2, k3


def test_triple(order):
    for _ in range(60):
        state = [random.choice([0, 1, P - 1, P, M64, random.getrandbits(64), random.getrandbits(64)]) for _ in range(12)]
        c1, c2, c3 = [[random.getrandbits(64) % P for _ in range(12)] for _ in range(3)]
        M, N2, N3, k1, k2, k3 = merged_tables(c1, c2, c3)
        assert max(max(r) for r in N3) < 1 << 21
        # reference: three plain partial rounds
        want = state
        for c in (c1, c2, c3):
            want = reference_round(want, c, True)
        vregs = {r: [random.getrandbits(32) for _ in range(NL)] for r in range(120, 256)}
        vregs[S_LO] = [x & M32 for x in state] + [random.getrandbits(32) for _ in range(4)]
        vregs[S_HI] = [x >> 32 for x in state] + [random.getrandbits(32) for _ in range(4)]
        for k in range(12):
            vregs[N3K + k] = [N3[e][(e + k) % 12] for e in range(12)] + [0] * 4
        vregs[R1] = [M[0][e] for e in range(12)] + [0] * 4
        vregs[R2] = [N2[0][e] for e in range(12)] + [0] * 4
        vregs[B2] = [N2[e][0] for e in range(12)] + [0] * 4
        vregs[B3] = [M[e][0] for e in range(12)] + [0] * 4
        vregs[N3C0] = [N3[e][0] for e in range(12)] + [0] * 4
        vregs[L0M] = [M[0][0]] + [0] * 15
        vregs[L0N] = [N2[0][0]] + [0] * 15
        for K, val in ((K1, [k1] + [0] * 15), (K2, [k2] + [0] * 15), (K3, k3 + [0] * 4)):
            vregs[K], vregs[K + 1] = [x & M32 for x in val], [0] * NL
            vregs[K + 2], vregs[K + 3] = [x >> 32 for x in val], [0] * NL
        vregs[ZA] = [0] * NL
        vregs[ZB] = [0] * NL
        sregs = {MASK0: [1] + [0] * 15, MASKE: [1, 0] * 8}
        vregs["next_k"] = [[[random.getrandbits(32) for _ in range(NL)] for _ in range(4)] for _ in range(3)]
        run(order, vregs, sregs)
        for e in range(12):
            got = vregs[S_LO][e] | (vregs[S_LO + 1][e] << 32)
            assert got % P == want[e], ("triple", e)
        for j, K in enumerate((K1, K2, K3)):
            for q in range(4):
                assert vregs[K + q] == vregs["next_k"][j][q]


def emit(name, order, what):
    print("// %s: %d instructions (%d s_nop)" % (what, len(order), sum(1 for o in order if o.text.startswith("s_nop"))))
    print("#define %s \\" % name)
    for i, o in enumerate(order):
        last = i == len(order) - 1
        print('    "%s%s"%s' % (o.text, "" if last else "\\n\\t", "" if last else " \\"))


def main():
    random.seed(11)
    print("// generated by tools/gen_row_round_asm.py -- do not edit.  Physical registers: state v[%d:%d] (in and out), seeds v[%d:%d] v[%d:%d]," %
          (S_LO, S_HI, SEED_A, SEED_A + 1, SEED_B, SEED_B + 1))
    print("// c0 v%d, column 0 v%d, zeros v%d v%d, lane-0 mask s[%d:%d]; v184 .. v255 and s%d .. s%d are clobbered." % (C0, COL0, ZA, ZB, MASK0, MASK0 + 1, SINK, FC + 1))
    inputs = {S_LO, S_HI, SEED_A, SEED_A + 1, SEED_B, SEED_B + 1}
    for name, build, partial, what in (("STARKHIP_ROW_FULL_ROUND_ASM", block_full, False, "full round: x^7 of every element, circulant layer, fold"),
                                       ("STARKHIP_ROW_PARTIAL_ROUND_ASM", block_partial, True, "partial round: x^7 of element 0 under the layer of the other eleven, fold")):
        order = schedule(build())
        check_hazards(order, inputs)
        test(order, partial)
        emit(name, order, what)
    order = schedule(block_triple())
    check_hazards(order, inputs | set(range(K1, K3 + 4)))
    test_triple(order)
    emit("STARKHIP_ROW_TRIPLE_ASM", order, "three partial rounds at once (poseidon_merged.h): three S-boxes, two dot products summed over the row, one dense layer")
    for name, r in (("N3K0", N3K), ("N3K1", N3K + 4), ("N3K2", N3K + 8), ("MISC0", R1), ("MISC1", N3C0)):
        print('#define STARKHIP_ROW_%s "{v[%d:%d]}"' % (name, r, r + 3))
    for name, r in (("K1", K1), ("K2", K2), ("K3", K3)):
        print('#define STARKHIP_ROW_%s "+{v[%d:%d]}"' % (name, r, r + 3))
    print('#define STARKHIP_ROW_MASKE "{s[%d:%d]}"' % (MASKE, MASKE + 1))
    print("#define STARKHIP_ROW_TRIPLE_CLOBBERS %s" % ", ".join(['"v%d"' % r for r in range(D1A, XS3 + 2)]))
    print('#define STARKHIP_ROW_STATE_OUT "={v[%d:%d]}"' % (S_LO, S_HI))
    print('#define STARKHIP_ROW_STATE_LO "{v%d}"' % S_LO)
    print('#define STARKHIP_ROW_STATE_HI "{v%d}"' % S_HI)
    print('#define STARKHIP_ROW_SEED_A "+{v[%d:%d]}"' % (SEED_A, SEED_A + 1))
    print('#define STARKHIP_ROW_SEED_B "+{v[%d:%d]}"' % (SEED_B, SEED_B + 1))
    print('#define STARKHIP_ROW_ADDR "{v%d}"' % ADDR)
    print('#define STARKHIP_ROW_C0 "{v%d}"' % C0)
    print('#define STARKHIP_ROW_COL0 "{v%d}"' % COL0)
    print('#define STARKHIP_ROW_ZA "{v%d}"' % ZA)
    print('#define STARKHIP_ROW_ZB "{v%d}"' % ZB)
    print('#define STARKHIP_ROW_MASK0 "{s[%d:%d]}"' % (MASK0, MASK0 + 1))
    bound = {S_LO, S_HI, SEED_A, SEED_A + 1, SEED_B, SEED_B + 1, C0, COL0, ZA, ZB}
    vs = [r for r in range(184, 256) if r not in bound]
    ss = list(range(SINK, FC + 2))
    print("#define STARKHIP_ROW_CLOBBERS %s" % ", ".join(['"v%d"' % r for r in vs] + ['"s%d"' % r for r in ss]))


if __name__ == "__main__":
    main()
