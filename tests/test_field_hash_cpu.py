"""CPU tests: oracle vs known-answer vectors, product host code vs oracle, C-ABI exports."""
import ctypes
import os
import re

import numpy as np

import oracle_lib as O
import starky_bls12_381_amd as S

P = S.P
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_round_constants_spot_values():
    # constants quoted from plonky2's poseidon_goldilocks.rs
    rc = O.round_constants()
    assert [hex(int(x)) for x in rc[:4]] == ["0xb585f766f2144405", "0x7746a55f43921ad7", "0xb2fb0d31cee799b4", "0xf6760a4803427d7"]
    assert int(rc[12]) == 0x86287821F722C881 and int(rc[15]) == 0xA484C4C5EF6A0781
    assert int(rc[359]) == 0xBC8DFB627FE558FC
    assert all(int(x) < P for x in rc)


KAT_ZERO = [0x3C18A9786CB0B359, 0xC4055E3364A246C3, 0x7953DB0AB48808F4, 0xC71603F33A1144CA, 0xD7709673896996DC, 0x46A84E87642F44ED,
            0xD032648251EE0B3C, 0x1C687363B207DF62, 0xDF8565563E8045FE, 0x40F5B37FF4254DAE, 0xD070F637B431067C, 0x1792B1C4342109D7]
KAT_IOTA = [0xD64E1E3EFC5B8E9E, 0x53666633020AAA47, 0xD40285597C6A8825, 0x613A4F81E81231D2, 0x414754BFEBD051F0, 0xCB1F8980294A023F,
            0x6EB2A9E4D54A9D0F, 0x1902BC3AF467E056, 0xF045D5EAFDC6021F, 0xE4150F77CAAA3BE5, 0xC9BFD01D39B50CCE, 0x5C0A27FCB0E1459B]
KAT_NEG1_HEAD = [0xBE0085CFC57A8357, 0xD95AF71847D05C09, 0xCF55A13D33C1C953, 0x95803A74F4530E82]


def test_poseidon_known_answers_oracle_and_host():
    for perm in (O.poseidon_permute, S.poseidon_permute_host):
        assert [int(x) for x in perm(np.zeros(12, dtype=np.uint64))] == KAT_ZERO
        assert [int(x) for x in perm(np.arange(12, dtype=np.uint64))] == KAT_IOTA
        assert [int(x) for x in perm(np.full(12, P - 1, dtype=np.uint64))[:4]] == KAT_NEG1_HEAD


def test_host_permutation_matches_oracle_on_random_states():
    rng = np.random.default_rng(7)
    for _ in range(200):
        s = rng.integers(0, P, size=12, dtype=np.uint64)
        assert np.array_equal(O.poseidon_permute(s), S.poseidon_permute_host(s))
    # extreme limbs
    for v in (0, 1, P - 1, 0xFFFFFFFF, 0xFFFFFFFF00000000, 1 << 63):
        s = np.full(12, v, dtype=np.uint64)
        assert np.array_equal(O.poseidon_permute(s), S.poseidon_permute_host(s))


def test_oracle_fast_reduction_matches_plain_modulo():
    rng = np.random.default_rng(11)
    vals = [0, 1, P - 1, P - 2, 0xFFFFFFFF, 0xFFFFFFFF00000000, 1 << 32, (1 << 32) - 1] + [int(x) for x in rng.integers(0, P, size=64, dtype=np.uint64)]
    for a in vals:
        for b in vals[:16]:
            assert O.lib.oracle_mul(a, b) == O.lib.oracle_mul_slow(a, b) == (a * b) % P


def test_oracle_ntt_round_trip_and_definition():
    rng = np.random.default_rng(3)
    n, logn = 64, 6
    coeffs = rng.integers(0, P, size=n, dtype=np.uint64)
    v = coeffs.copy()
    O.lib.oracle_fft(v.ctypes.data_as(O._u64p), logn)
    w = pow(1753635133440165772, 1 << (32 - logn), P)
    for i in (0, 1, 5, 63):
        assert int(v[i]) == sum(int(c) * pow(w, i * k, P) for k, c in enumerate(coeffs)) % P
    O.lib.oracle_ifft(v.ctypes.data_as(O._u64p), logn)
    assert np.array_equal(v, coeffs)


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "starkhip.h")).read()
    names = set(re.findall(r"\b(starkhip_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 25
    lib = ctypes.CDLL(S.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def _top_level_args(text):
    depth, n, seen = 0, 0, False
    for ch in text:
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        elif ch == "," and depth == 0:
            n += 1
        if not ch.isspace():
            seen = True
    return n + 1 if seen and text.strip() != "void" else 0


def test_rust_binding_matches_the_header():
    """bindings/rust/starkhip-sys cannot be compiled here (no Rust toolchain): every `extern "C"` function it declares must
    exist in include/starkhip.h with the same number of parameters, and its struct mirrors must have the header's fields."""
    hdr = open(os.path.join(ROOT, "include", "starkhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    rs = open(os.path.join(ROOT, "bindings", "rust", "starkhip-sys", "src", "lib.rs")).read()
    rs_nc = re.sub(r"//[^\n]*", "", rs)
    c_fns = {m.group(1): _top_level_args(m.group(2)) for m in re.finditer(r"\b(starkhip_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S)}
    block = rs_nc[rs_nc.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    r_fns = {m.group(1): _top_level_args(m.group(2)) for m in re.finditer(r"pub fn (starkhip_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->[^;]*)?;", block, flags=re.S)}
    assert len(r_fns) >= 28
    for name, n in r_fns.items():
        assert name in c_fns, name
        assert c_fns[name] == n, (name, c_fns[name], n)
    for must in ("starkhip_prove", "starkhip_prove_compact", "starkhip_verify", "starkhip_proof_layout", "starkhip_trace_log_begin",
                 "starkhip_trace_set_threads", "starkhip_config_for_air", "starkhip_init", "starkhip_shutdown", "starkhip_free"):
        assert must in r_fns, must

    def c_fields(struct_end):
        body = hdr[:hdr.index(struct_end)]
        body = body[body.rindex("typedef struct {") + len("typedef struct {"):]
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(",")[0:1] + decl.split(",")[1:]:
                out.append(re.sub(r"\[.*?\]", "", part.split()[-1]))
        return out

    def r_fields(name):
        body = rs_nc[rs_nc.index("pub struct %s {" % name):]
        body = body[:body.index("}")]
        return re.findall(r"pub ([a-z0-9_]+):", body)

    assert r_fields("starkhip_config_t") == c_fields("} starkhip_config_t;")
    assert r_fields("starkhip_proof_layout_t") == c_fields("} starkhip_proof_layout_t;")
    assert r_fields("starkhip_pool_config_t") == c_fields("} starkhip_pool_config_t;")
    assert r_fields("starkhip_ticket_info_t") == c_fields("} starkhip_ticket_info_t;")
    for must in ("starkhip_pool_create", "starkhip_pool_submit_witness", "starkhip_pool_wait", "starkhip_pool_destroy"):
        assert must in r_fns, must
    # the field-by-field hand-off to the recursion stage reads every offset the layout struct offers
    handoff = rs[rs.index("pub mod handoff"):]
    for field in ("off_trace_cap", "off_quotient_cap", "off_local_values", "off_next_values", "off_quotient_openings", "off_fri_caps",
                  "off_query_rounds", "query_round_words", "off_final_poly", "off_pow_witness", "off_public_inputs", "q_trace_leaf",
                  "q_trace_siblings", "q_quotient_leaf", "q_quotient_siblings", "q_step_evals", "q_step_siblings", "step_sibling_count"):
        assert "l." + field in handoff, field
    c_codes = dict(re.findall(r"(STARKHIP_(?:OK|ERR_[A-Z_]+))\s*=\s*(-?\d+)", hdr))
    r_codes = dict(re.findall(r"pub const (STARKHIP_(?:OK|ERR_[A-Z_]+)): c_int = (-?\d+);", rs_nc))
    assert r_codes == c_codes
    c_airs = {k: v for k, v in re.findall(r"STARKHIP_AIR_([A-Z0-9_]+)\s*=\s*(\d+)", hdr) if not k.startswith("TEST")}
    r_airs = dict(re.findall(r"^\s*([A-Za-z0-9]+) = (\d+),", rs_nc[rs_nc.index("pub enum Air {"):rs_nc.index("pub const STARKHIP_OK")], flags=re.M))
    assert sorted(r_airs.values()) == sorted(c_airs.values()) and len(r_airs) == 5


def test_config_mirrors_standard_fast_config():
    cfg = S.StarkConfig.standard_fast_config()
    assert (cfg.security_bits, cfg.num_challenges, cfg.rate_bits, cfg.cap_height, cfg.proof_of_work_bits, cfg.arity_bits,
            cfg.final_poly_bits, cfg.num_query_rounds) == (100, 2, 1, 4, 16, 4, 5, 84)
    assert S.StarkConfig.for_air(S.AIR_FINAL_EXP).rate_bits == 2
    assert S.StarkConfig.for_air(S.AIR_PAIRING_PRECOMP).rate_bits == 2
    assert S.StarkConfig.for_air(S.AIR_MILLER_LOOP).rate_bits == 1
    assert S.StarkConfig.for_air(S.AIR_FP12_MUL).rate_bits == 1


def test_merged_partial_round_tables_reproduce_the_permutation():
    """The leaf-hash kernels take the 22 partial rounds three at a time (quad form: integer matrices M Mz Mz, per-lane coefficient
    views, folded constants) or four at a time (lane and pair forms: M Mz Mz Mz, whose rows must still sum to less than 2^32 for the
    64-bit accumulators) from host-built tables.  The library replays both formulations on the CPU, with exactly those tables, against
    the plain permutation."""
    import starky_bls12_381_amd as S
    assert S.lib.starkhip_selfcheck_hash_tables(200) == 0



def test_lde_launch_plan_never_overwrites_a_column_it_has_not_read():
    """A trace waits for its LDE inside the buffer the LDE is written to, as its last C n words (prover.hip, csrc/lde_ranges.h): parked
    column c' at words [(R - 1) C n + c' n, + n), the LDE block of column c at [R c n, R (c + 1) n).  Workgroups of one launch run in any
    order, so a launch over [a, b) may touch only parked columns an earlier launch has read -- except the last launch, which reads a
    COPY of its columns and may touch anything.  Checked on the plan the library actually launches from, for the five AIRs' shapes and
    random ones, with plain interval arithmetic (n = 1: the row count scales both sides alike)."""
    import starky_bls12_381_amd as S
    f = S.lib.starkhip_lde_launch_ranges
    f.restype = ctypes.c_size_t
    f.argtypes = [ctypes.c_size_t, ctypes.c_uint, ctypes.POINTER(ctypes.c_uint64), ctypes.c_size_t]
    rng = np.random.default_rng(0x1DE)
    shapes = [(73527, 2), (29376, 2), (97330, 1), (60285, 1), (2, 1), (3, 2), (1, 1), (64, 3), (65, 1), (4097, 3)]
    shapes += [(int(rng.integers(1, 200000)), int(rng.integers(1, 4))) for _ in range(300)]
    for C, r in shapes:
        buf = (ctypes.c_uint64 * (3 * 64))()
        k = f(C, r, buf, 64)
        assert 1 <= k <= 64
        plan = [(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]) for i in range(k)]
        R = 1 << r
        assert plan[0][0] == 0 and plan[-1][1] == C and all(plan[i][1] == plan[i + 1][0] for i in range(k - 1))
        assert all(a < b for a, b, _ in plan) and [c for _, _, c in plan] == [0] * (k - 1) + [1]
        assert plan[-1][1] - plan[-1][0] <= max(C // 32, 64)          # what the copy buffer holds (ctx_reserve)
        if C > 64:
            assert k <= 2 + int(np.ceil(np.log(32) / np.log(R)))     # a handful of launches, not log_R(C)
        for a, b, from_copy in plan:
            if from_copy:
                continue
            # words the launch writes: [R a, R b); parked columns still unread: [a, C) at words (R - 1) C + [a, C)
            assert R * b <= (R - 1) * C + a, (C, r, a, b)
