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
    op = {4: "ds_read_b128", 2: "ds_read_b64"}[count]
    rng = "v[%d:%d]" % (first, first + count - 1)
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
    cr = COEFR + 16 * (key[1] % 2) if key[0] == "row" else COEFR + 16 * key[1]
    for q in range(3):
        load(prog, cr + 4 * q, 4, A_COEF, coef_off + 16 * q, (key, q))
    if key[0] == "row":
        load(prog, cr + 12, 4, A_COEF, coef_off + 48, (key, 3))
    for j in range(12):
        madc(prog, A, T + 2 * j, ("v", cr + j), seed=seed_regs if j == 0 else None)
        madc(prog, B, T + 2 * j + 1, ("v", cr + j), seed=seed_regs + 2 if j == 0 else None)
    return cr


ROW_OFF, M0_OFF, N20_OFF = 0, 12 * 64, 12 * 64 + 48   # LaneTables: row[12][16], m0[12], n20[12] contiguous


def block_triple():
    prog = []
    sbox(prog, T, T, 0)                                       # x1 replaces element 0: T is u'
    load(prog, SEEDR, 4, A_K12, 0, ("k12", 0))
    load(prog, SEEDR + 4, 4, A_K12, 16, ("k12", 1))
    dot(prog, ACC, ACC + 2, M0_OFF, SEEDR, ("dot", 0))
    fold_to(prog, YY, ACC, ACC + 2, 0)
    sbox(prog, YY + 2, YY, 1)                                  # x2
    dot(prog, ACC + 4, ACC + 6, N20_OFF, SEEDR + 4, ("dot", 1))
    madc(prog, ACC + 4, YY + 2, 25)
    madc(prog, ACC + 6, YY + 3, 25)
    fold_to(prog, YY, ACC + 4, ACC + 6, 1)
    sbox(prog, YY + 4, YY, 0)                                  # x3
    for r in range(12):
        sd = SEEDR + 8 + 4 * (r % 2)
        load(prog, sd, 4, A_K3, 16 * r, ("k3", r))
        A, B = ACC + 4 * (r % 2), ACC + 4 * (r % 2) + 2
        cr = dot(prog, A, B, ROW_OFF + 64 * r, sd, ("row", r))
        madc(prog, A, YY + 2, ("v", cr + 12))
        madc(prog, B, YY + 3, ("v", cr + 12))
        madc(prog, A, YY + 4, ("v", cr + 13))
        madc(prog, B, YY + 5, ("v", cr + 13))
        fold_to(prog, O + 2 * r, A, B, r % 2)
    for e in range(12):
        prog.append(Ins("v_mov_b64 %s, %s" % (vp(T + 2 * e), vp(O + 2 * e)), [O + 2 * e, O + 2 * e + 1], [T + 2 * e, T + 2 * e + 1], sem=("mov64", T + 2 * e, O + 2 * e)))
    return prog


# ---------------------------------------------------------------- scheduling with load latency, counted waits
LOAD_LATENCY = 16


def schedule(prog):
    # a consumer of a loaded register is kept LOAD_LATENCY slots behind the load (there is other work); the waits are counted below
    for ins in prog:
        ins.min_after_load = LOAD_LATENCY
    order = G.schedule_with(prog, lambda producer, consumer, d: max(d, LOAD_LATENCY) if getattr(producer, "is_load", False) else d)
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


def check_hazards(order):
    real = [o for o in order]
    for i, ins in enumerate(real):
        for back in (1, 2):
            if i - back < 0:
                continue
            assert not (ins.sreads & real[i - back].swrites), ("W1", i, ins.text)
        if i >= 1 and not ins.text.startswith(("ds_read", "s_waitcnt")):
            assert not (ins.writes & (real[i - 1].reads - real[i - 1].writes)), ("W3", i, ins.text)


# ---------------------------------------------------------------- tests
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


def test_triple(order):
    for _ in range(30):
        state = [rnd() for _ in range(12)]
        c1, c2, c3 = [[random.getrandbits(64) % P for _ in range(12)] for _ in range(3)]
        M, N2, N3, k1, k2, k3 = G.merged_tables(c1, c2, c3)
        want = state
        for c in (c1, c2, c3):
            want = G.reference_round(want, c, True)
        vregs = fresh()
        set_state(vregs, state)
        mem = vregs["mem"]
        mem[("k12", 0)], mem[("k12", 1)] = pair4(k1), pair4(k2)
        for r in range(12):
            mem[("k3", r)] = pair4(k3[r])
            row = [N3[r][j] for j in range(12)] + [N2[r][0], M[r][0], 0, 0]
            for q in range(4):
                mem[(("row", r), q)] = [[x] for x in row[4 * q:4 * q + 4]]
        for q in range(3):
            mem[(("dot", 0), q)] = [[M[0][j]] for j in range(4 * q, 4 * q + 4)]
            mem[(("dot", 1), q)] = [[N2[0][j]] for j in range(4 * q, 4 * q + 4)]
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
            ("STARKHIP_LANE_TRIPLE_ASM", block_triple(), test_triple, "three partial rounds at once (poseidon_merged.h)")):
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
    ss = list(range(SINK, FCS[1] + 2))
    print("#define STARKHIP_LANE_CLOBBERS %s" % ", ".join(['"v%d"' % r for r in vs] + ['"s%d"' % r for r in ss]))


if __name__ == "__main__":
    main()
