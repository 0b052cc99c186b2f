#!/usr/bin/env python3
"""Generates the instruction schedule of `row_layer_circ` (poseidon_dev.h): the circulant MDS layer of the row form as ONE
inline-asm block, ordered so that no gfx950 wait state is left unfilled.

A lone wave (the row form exists for commitments that cannot fill the chip) issues one instruction per ~5 cycles whatever
it depends on, but hazards still cost whole slots: a DPP move may read a VGPR only two instructions after a VALU wrote it, and
hipcc, given the C++ form, reuses one temporary for every rotated operand, which puts an s_nop between each multiply-add and
the next move (31 s_nop in a 146-instruction round).  Here the moves of one group are issued while the multiply-adds of the
previous group retire, over rotating temporaries.

Rules checked below (every violation is an error, not a warning):
  R1  a DPP instruction reads a VGPR written by one of the two instructions before it;
  R2  an instruction writes a VGPR that the instruction just before it read.

    python tools/gen_row_layer_asm.py            # prints the C string for poseidon_dev.h
"""
CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]

# operands of the asm statement, in order
#   %0 A (v64, in/out)  %1 B (v64, in/out)  %2 lo (v32, in/out: mirrored in place)  %3 hi (same)
#   %4..%9 six temporaries t0..t5 (v32)  %10 z_lo %11 z_hi (v32)  %12 scratch sgpr pair (carry-out sink)  %13 c0 (v32: 17, + 8 on lane 0)
OPS = {"A": "%0", "B": "%1", "lo": "%2", "hi": "%3", "t0": "%4", "t1": "%5", "t2": "%6", "t3": "%7", "t4": "%8", "t5": "%9",
       "zl": "%10", "zh": "%11", "scr": "%12", "c0": "%13"}

prog = []  # (text, reads, writes, is_dpp)


def shl(dst, src, k):
    prog.append(("v_mov_b32_dpp {d}, {s} row_shl:%d row_mask:0xf bank_mask:0xf bound_ctrl:1" % k, [src], [dst], True, dst, src))


def mirror(reg):
    prog.append(("v_mov_b32_dpp {d}, {s} row_shr:12 row_mask:0xf bank_mask:0x8", [reg], [reg], True, reg, reg))


def nop():
    prog.append(("s_nop 0", [], [], False, "A", "A"))


def mad(acc, src, coef):
    c = OPS["c0"] if coef == "c0" else str(coef)
    reads = [src, acc] + (["c0"] if coef == "c0" else [])
    prog.append(("v_mad_u64_u32 {d}, %s, {s}, %s, {d}" % (OPS["scr"], c), reads, [acc], False, acc, src))


# ---- the schedule
# lo / hi were written by the caller's last instructions, which this block cannot see: nothing may read them through DPP before two
# instructions have passed -- the two k = 0 multiply-adds go first (the mirror only changes lanes 12 .. 15, whose results are unused)
mad("A", "lo", "c0"); mad("B", "hi", "c0")
mirror("lo"); mirror("hi")
nop()
# group x: rotations 1..4 of the mirrored state; rot 4 (= z) first so that it is old enough to be mirrored after the others
shl("zl", "lo", 4); shl("zh", "hi", 4)
shl("t0", "lo", 1); shl("t1", "hi", 1); shl("t2", "lo", 2); shl("t3", "hi", 2); shl("t4", "lo", 3); shl("t5", "hi", 3)
mad("A", "zl", CIRC[4]); mad("B", "zh", CIRC[4])          # z is read here before it is mirrored in place
mad("A", "t0", CIRC[1])
mirror("zl")
mad("B", "t1", CIRC[1])
mirror("zh")
mad("A", "t2", CIRC[2]); mad("B", "t3", CIRC[2])
# group z: lo / hi are free now: they take w = rot 8
shl("lo", "zl", 4); shl("hi", "zh", 4)
shl("t0", "zl", 1); shl("t1", "zh", 1)
mad("A", "t4", CIRC[3]); mad("B", "t5", CIRC[3])
shl("t2", "zl", 2); shl("t3", "zh", 2); shl("t4", "zl", 3); shl("t5", "zh", 3)
mad("A", "lo", CIRC[8]); mad("B", "hi", CIRC[8])
mad("A", "t0", CIRC[5])
mirror("lo")
mad("B", "t1", CIRC[5])
mirror("hi")
mad("A", "t2", CIRC[6]); mad("B", "t3", CIRC[6])
# group w
shl("t0", "lo", 1); shl("t1", "hi", 1)
mad("A", "t4", CIRC[7]); mad("B", "t5", CIRC[7])
shl("t2", "lo", 2); shl("t3", "hi", 2); shl("t4", "lo", 3); shl("t5", "hi", 3)
mad("A", "t0", CIRC[9]); mad("B", "t1", CIRC[9])
mad("A", "t2", CIRC[10]); mad("B", "t3", CIRC[10])
mad("A", "t4", CIRC[11]); mad("B", "t5", CIRC[11])

# ---- hazard check
for i, (text, reads, writes, is_dpp, d, s) in enumerate(prog):
    if is_dpp:
        for back in (1, 2):
            if i - back >= 0:
                for w in prog[i - back][2]:
                    assert w not in reads, "R1 at %d: %s reads %s written %d before" % (i, text, w, back)
    if i >= 1:
        for w in writes:
            prev_reads = prog[i - 1][1]
            if w in prev_reads and not (w in prog[i - 1][2] and not prog[i - 1][3] and not is_dpp):  # acc += ... ; acc += ... is the normal chain
                raise AssertionError("R2 at %d: %s writes %s read just before" % (i, text, w))

# ---- functional check: simulate on 16 lanes with integers
import random
x = [random.getrandbits(64) for _ in range(12)] + [0] * 4
reg = {"lo": [v & 0xFFFFFFFF for v in x], "hi": [v >> 32 for v in x], "A": [5] * 16, "B": [7] * 16,
       "c0": [17 + (8 if e == 0 else 0) for e in range(16)]}
for t in ("t0", "t1", "t2", "t3", "t4", "t5", "zl", "zh"):
    reg[t] = [0] * 16
for text, reads, writes, is_dpp, d, s in prog:
    if "row_shl" in text:
        k = int(text.split("row_shl:")[1].split()[0])
        reg[d] = [reg[s][i + k] if i + k < 16 else 0 for i in range(16)]
    elif "row_shr:12" in text:
        old = reg[d][:]
        reg[d] = [old[i - 12] if i >= 12 else old[i] for i in range(16)]
    elif text.startswith("s_nop"):
        pass
    else:
        coef = text.split(", ")[-2]
        cv = reg["c0"] if coef == OPS["c0"] else [int(coef)] * 16
        reg[d] = [reg[d][i] + reg[s][i] * cv[i] for i in range(16)]
for e in range(12):
    wantA = 5 + sum(CIRC[k] * (x[(e + k) % 12] & 0xFFFFFFFF) for k in range(12)) + (8 * (x[0] & 0xFFFFFFFF) if e == 0 else 0)
    wantB = 7 + sum(CIRC[k] * (x[(e + k) % 12] >> 32) for k in range(12)) + (8 * (x[0] >> 32) if e == 0 else 0)
    assert reg["A"][e] == wantA and reg["B"][e] == wantB, e

lines = [t.format(d=OPS[d], s=OPS[s]) for t, _, _, _, d, s in prog]
print("// generated by tools/gen_row_layer_asm.py (%d instructions: %d moves, %d multiply-adds; hazard rules R1, R2 checked there)" %
      (len(lines), sum(1 for p in prog if p[3]), sum(1 for p in prog if p[0].startswith("v_mad"))))
print("#define STARKHIP_ROW_LAYER_CIRC_ASM \\")
for i, l in enumerate(lines):
    print('    "%s%s"%s' % (l, "" if i == len(lines) - 1 else "\\n\\t", "" if i == len(lines) - 1 else " \\"))
