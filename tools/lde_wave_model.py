#!/usr/bin/env python3
"""Executable model of the wave-resident LDE kernel for 2^13 rows (csrc/kernels_lde_wave.hip): every index map, LDS address
function and twiddle table of the kernel, in Python integers, checked against a plain NTT.

Why a model: the kernel keeps a column's 8192 elements as 512 threads x 16 registers and moves index BITS between registers,
lanes and waves.  A transform is three radix-16 passes in registers and one radix-2; between them the four index bits a pass
needs are brought into the registers by an exchange.  The plan (13 bits = 4 register + 6 lane + 3 wave):

  inverse (values in natural order -> coefficients)              forward (coefficients -> values in natural order)
    load   R = j12..9 | W = j8..6  | L = j5..0                      load   R = c12..9 | W = c3..1 | L = c8, c7..4, c0   (= the inverse's end)
    pass 0 -> k0..3                                                 pass 0 -> k0..3
    X exchange (through the whole image, two barriers)              w exchange (inside the wave's own LDS slice, no barrier)
           R = j8..5  | W = k3..1 | L = k0, j4..0                          R = c8..5 | W = c3..1 | L = c0, c4, k3..0
    pass 1 -> k4..7                                                 pass 1 -> k4..7
    w exchange                                                      X exchange
           R = j4..1  | W = k3..1 | L = j0, k7..4, k0                      R = c4..1 | W = k7..5 | L = c0, k4, k3..0
    pass 2 -> k8..11                                                pass 2 -> k8..11
    v_permlane32_swap_b32 on register pairs + radix 2 -> k12        v_permlane32_swap_b32 + radix 2 -> k12
    end    R = k12..9 | W = k3..1 | L = k8, k7..4, k0               store  two runs of 32 consecutive points per wave and register

The inverse ends in exactly the layout the forward transform starts from, so the coefficients never leave the thread that made
them: coset 0 is transformed from registers, the other cosets re-read the thread's own 16 words.  After an X exchange every wave
reads only its own eighth of the image, so the w exchanges of a wave touch nothing another wave reads: two barriers per transform.

`python tools/lde_wave_model.py` runs the checks (also tests/test_lde_wave_model_cpu.py): both transforms against a plain NTT, the
LDS address functions for bank conflicts under the ds_write_b64 / ds_read_b64 rules of MI355X_MICROARCH.md (LDS section), and
prints nothing but "ok".  The table builders below are the specification of the host-side builders in the kernel's source."""
import random

P = 0xFFFFFFFF00000001
GEN2 = 1753635133440165772  # GL_POWER_OF_TWO_GENERATOR (order 2^32), csrc/gl.h
LOG_N = 13
N = 1 << LOG_N
T = N // 16  # threads per column


def root_of_unity(k):
    r = GEN2
    for _ in range(k, 32):
        r = r * r % P
    return r


def ntt_plain(x, w):
    """X[k] = sum_j x[j] w^(j k), iterative radix-2 (reference)."""
    n = len(x)
    a = list(x)
    j = 0
    for i in range(1, n):  # bit reversal
        bit = n >> 1
        while j & bit:
            j ^= bit
            bit >>= 1
        j |= bit
        if i < j:
            a[i], a[j] = a[j], a[i]
    length = 2
    while length <= n:
        wl = pow(w, n // length, P)
        for s in range(0, n, length):
            cur = 1
            for k in range(length // 2):
                u, v = a[s + k], a[s + k + length // 2] * cur % P
                a[s + k], a[s + k + length // 2] = (u + v) % P, (u - v) % P
                cur = cur * wl % P
        length <<= 1
    return a


def dft16(v, w16):
    return [sum(v[j] * pow(w16, j * k, P) for j in range(16)) % P for k in range(16)]


# ------------------------------------------------------------------------------------------------ layouts (thread t = 64 w + l)
def inv_after_x(t):
    """inverse, after the X exchange: thread -> (k1, b): w = k1 >> 1, l = b | (k1 & 1) << 5"""
    w, l = t >> 6, t & 63
    return (w << 1) | (l >> 5), l & 31


def inv_after_w(t):
    """inverse, after the w exchange: thread -> (k1, k2, e): l = (k1 & 1) | k2 << 1 | e << 5"""
    w, l = t >> 6, t & 63
    return (w << 1) | (l & 1), (l >> 1) & 15, l >> 5


def coef_index(t, i):
    """coefficient a thread holds in register i at the end of the inverse / start of the forward transform:
    c12..9 = i, c3..1 = w, c0 = l0, c7..4 = l4..1, c8 = l5"""
    w, l = t >> 6, t & 63
    return (i << 9) | ((l >> 5) << 8) | (((l >> 1) & 15) << 4) | (w << 1) | (l & 1)


def fwd_after_w(t):
    """forward, after the w exchange: thread -> (k1, c4, c0, c3..1): l = k1 | c4 << 4 | c0 << 5, w = c3..1"""
    w, l = t >> 6, t & 63
    return l & 15, (l >> 4) & 1, l >> 5, w


def fwd_after_x(t):
    """forward, after the X exchange: thread -> (k1, k2, e): w = k2 >> 1, l = k1 | (k2 & 1) << 4 | e << 5"""
    w, l = t >> 6, t & 63
    return l & 15, (w << 1) | ((l >> 4) & 1), l >> 5


# ------------------------------------------------------------------------------------------------ LDS address functions (8-byte words)
def lds_inv_x(k1, j2):            # whole image; reader wave k1 >> 1 owns [1024 (k1 >> 1), + 1024)
    return k1 * 512 + j2


def lds_inv_w(k1, k2, b):         # inside the slice of wave k1 >> 1
    return (k1 >> 1) * 1024 + b * 32 + (((k2 << 1) | (k1 & 1)) ^ b)


def lds_fwd_w(w, k1, a, c4, c0):  # inside the slice of wave w (= c3..1); a = c8..5
    g = (c0 << 1) | ((a & 3) << 2)  # c0, c5, c6 spread the sixteen lanes of a store group over the sixteen 8-byte slots
    return w * 1024 + ((a << 1) | c0) * 32 + (((k1 << 1) | c4) ^ g)


def lds_fwd_x(k1, k2, b):         # whole image; reader wave k2 >> 1 owns [1024 (k2 >> 1), + 1024)
    return (k2 >> 1) * 1024 + b * 32 + (k2 & 1) * 16 + k1


def check_conflicts(write_addr, read_addr, name):
    """write_addr(t, reg) / read_addr(t, reg): word addresses.  ds_write_b64: four groups of 16 contiguous lanes, 32 banks of 4 B
    (two per word: conflict-free = 16 distinct words mod 16); ds_read_b64: two groups of 32 lanes, 64 banks (32 distinct words mod 32)."""
    for reg in range(16):
        for wave in range(8):
            for g in range(4):
                slots = {write_addr(wave * 64 + g * 16 + k, reg) % 16 for k in range(16)}
                assert len(slots) == 16, (name, "write", reg, wave, g)
            for h in range(2):
                slots = {read_addr(wave * 64 + h * 32 + k, reg) % 32 for k in range(32)}
                assert len(slots) == 32, (name, "read", reg, wave, h)


# ------------------------------------------------------------------------------------------------ tables (what the host uploads)
def tables(inverse):
    """Per-thread twiddles in the order the kernel reads them.  tw1[k1][t]: after pass 0; tw2[k2][b]: after pass 1 (b = the five
    low index bits of the 512-point sub-transform); tw3[k4]: the exponents e of 2^e for the radix-2 step (32nd roots of unity are
    powers of two).  The inverse transform's n^-1 is folded into the coset table, not into these."""
    w = root_of_unity(LOG_N)
    if inverse:
        w = pow(w, P - 2, P)
    tw1 = [[0] * T for _ in range(16)]
    for t in range(T):
        j2 = t if inverse else (coef_index(t, 0) & 511)
        for k1 in range(16):
            tw1[k1][t] = pow(w, j2 * k1, P)
    w512 = pow(w, 16, P)
    tw2 = [[pow(w512, b * k2, P) for b in range(32)] for k2 in range(16)]
    w32 = pow(w, 256, P)
    tw3 = [pow(w32, k4, P) for k4 in range(16)]
    return w, tw1, tw2, tw3


def pow2_exponent(x):
    """e with 2^e = x mod p (the 192 powers of two are the 192nd roots of unity)."""
    v = 1
    for e in range(192):
        if v == x:
            return e
        v = v * 2 % P
    raise ValueError("not a power of two")


def coset_table(rate_bits, s):
    """cs[i][t] = n^-1 (7 w_N^s)^c for the coefficient c a thread holds in register i (coef_index)."""
    wN = root_of_unity(LOG_N + rate_bits)
    shift = 7 * pow(wN, s, P) % P
    ninv = pow(N, P - 2, P)
    return [[ninv * pow(shift, coef_index(t, i), P) % P for t in range(T)] for i in range(16)]


# ------------------------------------------------------------------------------------------------ the two transforms
def inverse_transform(x, lds_log=None):
    """values x[0..n) -> regs[t][i] = n * coefficient coef_index(t, i) (n^-1 is in the coset table)."""
    w, tw1, tw2, tw3 = tables(True)
    w16 = pow(w, 512, P)
    regs = [[x[t + 512 * i] for i in range(16)] for t in range(T)]
    regs = [dft16(r, w16) for r in regs]                                            # pass 0: i = j12..9 -> k1
    regs = [[regs[t][k1] * tw1[k1][t] % P for k1 in range(16)] for t in range(T)]
    lds = [None] * N
    for t in range(T):                                                              # X exchange
        for k1 in range(16):
            lds[lds_inv_x(k1, t)] = regs[t][k1]
    new = []
    for t in range(T):
        k1, b = inv_after_x(t)
        new.append([lds[lds_inv_x(k1, a * 32 + b)] for a in range(16)])
    regs = [dft16(r, w16) for r in new]                                             # pass 1: a = j8..5 -> k2
    for t in range(T):
        k1, b = inv_after_x(t)
        regs[t] = [regs[t][k2] * tw2[k2][b] % P for k2 in range(16)]
    lds = [None] * N
    for t in range(T):                                                              # w exchange
        k1, b = inv_after_x(t)
        for k2 in range(16):
            a = lds_inv_w(k1, k2, b)
            assert a >> 10 == t >> 6 and lds[a] is None
            lds[a] = regs[t][k2]
    new = []
    for t in range(T):
        k1, k2, e = inv_after_w(t)
        new.append([lds[lds_inv_w(k1, k2, 2 * d + e)] for d in range(16)])
    regs = [dft16(r, w16) for r in new]                                             # pass 2: d = j4..1 -> k4
    for t in range(T):
        if inv_after_w(t)[2]:
            regs[t] = [regs[t][k4] * tw3[k4] % P for k4 in range(16)]
    out = [[0] * 16 for _ in range(T)]
    for t in range(T):                                                              # v_permlane32_swap(v[2m], v[2m+1]) + radix 2
        lo, hi = (t & ~32), (t | 32)
        for m in range(8):
            k4 = 2 * m + ((t >> 5) & 1)        # the lane's bit 5 is k4's low bit afterwards
            A, B = regs[lo][k4], regs[hi][k4]  # e = 0 / e = 1
            out[t][m] = (A + B) % P            # k5 = 0 -> register i = k12..9 = k5 * 8 + m
            out[t][8 + m] = (A - B) % P
    return out


def forward_transform(regs_in, s, rate_bits, prescaled=False):
    """regs_in[t][i] = n * coefficient coef_index(t, i) -> values on coset s in natural order."""
    w, tw1, tw2, tw3 = tables(False)
    w16 = pow(w, 512, P)
    cs = coset_table(rate_bits, s)
    regs = [[regs_in[t][i] * cs[i][t] % P for i in range(16)] for t in range(T)]
    regs = [dft16(r, w16) for r in regs]                                            # pass 0: i = c12..9 -> k1
    regs = [[regs[t][k1] * tw1[k1][t] % P for k1 in range(16)] for t in range(T)]
    lds = [None] * N
    for t in range(T):                                                              # w exchange
        c = coef_index(t, 0)
        a, c4, c0 = (c >> 5) & 15, (c >> 4) & 1, c & 1
        for k1 in range(16):
            ad = lds_fwd_w(t >> 6, k1, a, c4, c0)
            assert ad >> 10 == t >> 6 and lds[ad] is None
            lds[ad] = regs[t][k1]
    new = []
    for t in range(T):
        k1, c4, c0, w3 = fwd_after_w(t)
        new.append([lds[lds_fwd_w(w3, k1, a, c4, c0)] for a in range(16)])
    regs = [dft16(r, w16) for r in new]                                             # pass 1: a = c8..5 -> k2
    for t in range(T):
        k1, c4, c0, w3 = fwd_after_w(t)
        b = (c4 << 4) | (w3 << 1) | c0
        regs[t] = [regs[t][k2] * tw2[k2][b] % P for k2 in range(16)]
    lds = [None] * N
    for t in range(T):                                                              # X exchange
        k1, c4, c0, w3 = fwd_after_w(t)
        b = (c4 << 4) | (w3 << 1) | c0
        for k2 in range(16):
            ad = lds_fwd_x(k1, k2, b)
            assert lds[ad] is None
            lds[ad] = regs[t][k2]
    new = []
    for t in range(T):
        k1, k2, e = fwd_after_x(t)
        assert all(lds_fwd_x(k1, k2, 2 * d + e) >> 10 == t >> 6 for d in range(16))   # a wave reads its own slice only
        new.append([lds[lds_fwd_x(k1, k2, 2 * d + e)] for d in range(16)])
    regs = [dft16(r, w16) for r in new]                                             # pass 2: d = c4..1 -> k4
    for t in range(T):
        if fwd_after_x(t)[2]:
            regs[t] = [regs[t][k4] * tw3[k4] % P for k4 in range(16)]
    out = [0] * N
    for t in range(T):                                                              # v_permlane32_swap(v[r], v[r+8]) + radix 2
        lo, hi = (t & ~32), (t | 32)
        k1, k2, _ = fwd_after_x(t)
        for r in range(8):
            k4 = r + 8 * ((t >> 5) & 1)
            A, B = regs[lo][k4], regs[hi][k4]
            out[store_index(t, r)] = (A + B) % P
            out[store_index(t, r + 8)] = (A - B) % P
    return out


def store_index(t, reg):
    """natural point index of register `reg` of thread t after the last step of the forward transform:
    k3..0 = l3..0, k4 = l4, k7..5 = w, k10..8 = reg2..0, k11 = l5, k12 = reg3"""
    w, l = t >> 6, t & 63
    return (l & 15) | (((l >> 4) & 1) << 4) | (w << 5) | ((reg & 7) << 8) | ((l >> 5) << 11) | ((reg >> 3) << 12)


def run_checks(seed=1, rate_bits=2):
    rng = random.Random(seed)
    x = [rng.randrange(P) for _ in range(N)]
    w = root_of_unity(LOG_N)
    ninv = pow(N, P - 2, P)
    coeffs = [v * ninv % P for v in ntt_plain(x, pow(w, P - 2, P))]
    regs = inverse_transform(x)
    for t in range(T):
        for i in range(16):
            assert regs[t][i] * ninv % P == coeffs[coef_index(t, i)], (t, i)
    wN = root_of_unity(LOG_N + rate_bits)
    for s in (0, (1 << rate_bits) - 1):
        shift = 7 * pow(wN, s, P) % P
        want = ntt_plain([coeffs[c] * pow(shift, c, P) % P for c in range(N)], w)
        got = forward_transform(regs, s, rate_bits)
        assert got == want, s
    # the radix-2 twiddles are powers of two
    for inverse in (False, True):
        assert all(pow(2, pow2_exponent(v), P) == v for v in tables(inverse)[3])
    # bank conflicts of the four exchanges
    check_conflicts(lambda t, r: lds_inv_x(r, t), lambda t, r: lds_inv_x(inv_after_x(t)[0], r * 32 + inv_after_x(t)[1]), "inverse X")
    check_conflicts(lambda t, r: lds_inv_w(inv_after_x(t)[0], r, inv_after_x(t)[1]),
                    lambda t, r: lds_inv_w(inv_after_w(t)[0], inv_after_w(t)[1], 2 * r + inv_after_w(t)[2]), "inverse w")

    def fw_write(t, k1):
        c = coef_index(t, 0)
        return lds_fwd_w(t >> 6, k1, (c >> 5) & 15, (c >> 4) & 1, c & 1)

    def fw_read(t, a):
        k1, c4, c0, w3 = fwd_after_w(t)
        return lds_fwd_w(w3, k1, a, c4, c0)

    check_conflicts(fw_write, fw_read, "forward w")

    def fx_write(t, k2):
        k1, c4, c0, w3 = fwd_after_w(t)
        return lds_fwd_x(k1, k2, (c4 << 4) | (w3 << 1) | c0)

    def fx_read(t, d):
        k1, k2, e = fwd_after_x(t)
        return lds_fwd_x(k1, k2, 2 * d + e)

    check_conflicts(fx_write, fx_read, "forward X")
    # the stores of the forward transform: per register a wave writes two runs of 32 consecutive points
    for reg in range(16):
        for wave in range(8):
            idx = sorted(store_index(wave * 64 + l, reg) for l in range(64))
            assert idx[:32] == list(range(idx[0], idx[0] + 32)) and idx[32:] == list(range(idx[32], idx[32] + 32))
    assert sorted(store_index(t, r) for t in range(T) for r in range(16)) == list(range(N))


if __name__ == "__main__":
    run_checks()
    print("ok")
