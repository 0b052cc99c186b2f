"""Helpers shared by the AIR tests: limb packing and seeded synthetic inputs."""
import json
import os

import numpy as np

BLS_P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def limbs(v, n=12):
    return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def from_limbs(a):
    return sum(int(x) << (32 * i) for i, x in enumerate(a))


def fp_arr(*vals):
    out = []
    for v in vals:
        out += limbs(int(v))
    return np.array(out, dtype=np.uint32)


def splitmix64(seed):
    state = seed & 0xFFFFFFFFFFFFFFFF
    while True:
        state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        yield z ^ (z >> 31)


def random_fp(gen):
    while True:
        v = 0
        for i in range(6):
            v |= next(gen) << (64 * i)
        v &= (1 << 381) - 1
        if v < BLS_P:
            return v


def random_fp12(seed):
    g = splitmix64(seed)
    return fp_arr(*[random_fp(g) for _ in range(12)])


def native_vectors():
    return json.load(open(os.path.join(GOLDEN, "native_vectors.json")))


ONE_FP12 = fp_arr(1, *([0] * 11))
