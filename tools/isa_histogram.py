#!/usr/bin/env python3
"""Opcode histograms of the hot loops, from the gfx950 assembly hipcc emits (--save-temps): the evidence behind the instruction
counts DESIGN.md quotes.  Runs where hipcc is (no GPU needed).

    python tools/isa_histogram.py > profiles/rNN_isa_histograms.txt

Per kernel: registers, then for every basic block of more than `--min` instructions its size, the vector / scalar / LDS / memory
split and the most frequent opcodes."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
KERNELS = [
    ("kernels_lde.hip", r"lde_columns_wave_kernelE", "LDE, 2^13 rows (wave-resident kernel): the column loop holds the inverse transform and the coset loop"),
    ("kernels_hash.hip", r"leaf_hash_kernelE", "leaf hash, quad form: the blocks are the round loops of one permutation"),
    ("kernels_hash.hip", r"leaf_hash_pair_kernelE", "leaf hash, pair form (a lone big commitment): the blocks are the generated rounds (csrc/pair_round_asm.inc) and the absorb loop around them"),
    ("kernels_quotient.hip", r"quotient_tiles_kernelILb0ELj0E", "tiled quotient evaluator: record steps (12 v_mad_u64_u32 each) and piece ends"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min", type=int, default=60)
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        for src, pat, what in KERNELS:
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "starky_bls12_381_amd", "csrc"),
                   "-I" + os.path.join(ROOT, "include"), "--save-temps", "-c", os.path.join(ROOT, "starky_bls12_381_amd", "csrc", src), "-o", "x.o"]
            subprocess.run(cmd, cwd=tmp, check=True, stderr=subprocess.DEVNULL)
            asm = open(os.path.join(tmp, src.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
            start = next(i for i, l in enumerate(asm) if re.match(r"^_ZN8starkhip\d+" + pat, l) and l.rstrip().endswith(":") is False and ":" in l)
            end = next(i for i in range(start, len(asm)) if asm[i].startswith(".Lfunc_end"))
            name = asm[start].split(":")[0]
            meta = {}
            for key in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "group_segment_fixed_size"):
                m = [l for l in asm if re.search(r"\." + key + r":", l)]
                names = [i for i, l in enumerate(asm) if ".name:" in l and name in l]
                if names:
                    seg = asm[names[0]:names[0] + 40]
                    mm = [l for l in seg if re.search(r"\." + key + r":", l)]
                    if mm:
                        meta[key] = mm[0].split(":")[1].strip()
            print(f"== {name}\n   {what}\n   {meta}")
            blocks, blk = collections.OrderedDict(), "entry"
            blocks[blk] = []
            for l in asm[start + 1:end]:
                m = re.match(r"^(\.LBB[0-9_]+):", l)
                if m:
                    blk = m.group(1)
                    blocks[blk] = []
                    continue
                t = l.strip()
                if not t or t.startswith(";") or t.startswith("."):
                    continue
                blocks[blk].append(t.split()[0])
                if t.startswith(("s_cbranch", "s_branch")):  # straight-line code after a loop's back edge is a block of its own
                    blk = blk.split("+")[0] + "+" + str(sum(1 for k in blocks if k.split("+")[0] == blk.split("+")[0]))
                    blocks[blk] = []
            total = collections.Counter(op for ops in blocks.values() for op in ops)
            print(f"   whole kernel: {sum(total.values())} instructions, {sum(v for k, v in total.items() if k.startswith('v_'))} vector")
            for b, ops in blocks.items():
                if len(ops) < args.min:
                    continue
                c = collections.Counter(ops)
                v = sum(x for k, x in c.items() if k.startswith("v_"))
                sc = sum(x for k, x in c.items() if k.startswith("s_") and k != "s_nop")
                nop = c.get("s_nop", 0)
                ds = sum(x for k, x in c.items() if k.startswith("ds_"))
                mem = sum(x for k, x in c.items() if k.startswith(("global_", "flat_", "buffer_", "scratch_")))
                top = ", ".join(f"{k} {x}" for k, x in c.most_common(10))
                print(f"   {b}: {len(ops)} = vector {v} + scalar {sc} + s_nop {nop} + LDS {ds} + memory {mem}\n      {top}")
            print()


if __name__ == "__main__":
    sys.exit(main())
