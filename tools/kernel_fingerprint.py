"""SHA-256 of the source files that define each heavy kernel: profiles/pmc_traffic_latest.json records it when the PMC passes are
taken (tools/pmc_traffic.py), bench.py recomputes it and prints `roofline.traffic` only when they agree -- a traffic figure
measured on an older kernel is reported as stale (null), not as current."""
import hashlib
import os

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CSRC = os.path.join(ROOT, "starky_bls12_381_amd", "csrc")
KERNEL_SOURCES = {
    "leaf_hash_kernel": ("kernels_hash.hip", "poseidon_dev.h", "poseidon_merged.h", "gl_dev.h", "row_layer_asm.inc", "row_round_asm.inc", "lane_round_asm.inc"),
    "leaf_hash_pair_kernel": ("kernels_hash.hip", "poseidon_dev.h", "poseidon_merged.h", "gl_dev.h", "pair_round_asm.inc", "lane_round_asm.inc"),
    "leaf_hash_lane_kernel": ("kernels_hash.hip", "poseidon_dev.h", "poseidon_merged.h", "gl_dev.h", "lane_round_asm.inc"),
    "quotient_tiles_kernel": ("kernels_quotient.hip", "quotient_plan.h", "gl_dev.h"),
    "lde_columns_v2_kernel": ("kernels_lde.hip", "gl_dev.h"),
    "lde_columns_wave_kernel": ("kernels_lde.hip", "gl_dev.h"),
}


def kernel_fingerprint(kernel):
    h = hashlib.sha256()
    for name in KERNEL_SOURCES[kernel]:
        h.update(open(os.path.join(CSRC, name), "rb").read())
    return h.hexdigest()
