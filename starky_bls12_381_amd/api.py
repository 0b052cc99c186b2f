"""ctypes binding of libstarkhip.so (C ABI in include/starkhip.h).

Host-side mirror of the reference's call surface for the starky prove() path
(/root/reference/src/aggregate_proof.rs:23-179): `StarkConfig`, `prove`, `verify_stark_proof`,
per-AIR `generate_trace`.  Python is only the harness here; all work happens behind the C ABI.
There is NO CPU fallback: if the shared library is missing, importing this module raises.
"""
import ctypes as C
import os
import time
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# STARKHIP_LIBRARY: another build of the same C ABI (the sanitizer build of the host code, `make asan`); never a CPU fallback --
# that build answers NO_DEVICE on every device path
LIB_PATH = os.environ.get("STARKHIP_LIBRARY") or os.path.join(_HERE, "libstarkhip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make` (or __graft_entry__.build()); "
        "starky_bls12_381_amd has no CPU fallback for the prover")

lib = C.CDLL(LIB_PATH)

P = 0xFFFFFFFF00000001
POW_SEARCH = 0xFFFFFFFFFFFFFFFF
N_PHASES = 11
PHASE_NAMES = ["upload", "ifft_lde", "trace_merkle", "quotient", "quotient_commit", "openings", "fri_combine",
               "fri_commit", "pow", "queries", "total"]

AIR_FP12_MUL, AIR_PAIRING_PRECOMP, AIR_MILLER_LOOP, AIR_FINAL_EXP, AIR_ECC_AGGREGATE, AIR_TEST_FIBONACCI = 0, 1, 2, 3, 4, 100
AIR_NAMES = {AIR_FP12_MUL: "FP12MulStark", AIR_PAIRING_PRECOMP: "PairingPrecompStark", AIR_MILLER_LOOP: "MillerLoopStark",
             AIR_FINAL_EXP: "FinalExponentiateStark", AIR_ECC_AGGREGATE: "ECCAggStark", AIR_TEST_FIBONACCI: "TestFibonacci"}
ECC_NUM_POINTS = 512  # src/ecc_aggregate.rs:7

ERR_QUOTIENT_NOT_DIVISIBLE, ERR_ZETA_IN_SUBGROUP, ERR_BAD_SHAPE, ERR_HIP, ERR_OOM, ERR_NO_DEVICE, ERR_VERIFY, ERR_BAD_AIR = range(-1, -9, -1)


class StarkConfig(C.Structure):
    """Mirror of starky::config::StarkConfig + FriConfig (flattened)."""
    _fields_ = [(n, C.c_uint32) for n in ("security_bits", "num_challenges", "rate_bits", "cap_height", "proof_of_work_bits",
                                          "arity_bits", "final_poly_bits", "num_query_rounds")]

    @staticmethod
    def standard_fast_config():
        cfg = StarkConfig()
        lib.starkhip_config_standard_fast(C.byref(cfg))
        return cfg

    @staticmethod
    def for_air(air):
        cfg = StarkConfig()
        _chk(lib.starkhip_config_for_air(air, C.byref(cfg)))
        return cfg


class StarkhipError(RuntimeError):
    def __init__(self, code):
        self.code = code
        super().__init__(f"starkhip error {code}: {lib.starkhip_error_string(code).decode()}")


def _chk(rc):
    if rc != 0:
        raise StarkhipError(rc)


_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
lib.starkhip_error_string.restype = C.c_char_p
lib.starkhip_error_string.argtypes = [C.c_int]
lib.starkhip_config_standard_fast.argtypes = [C.POINTER(StarkConfig)]
lib.starkhip_config_for_air.argtypes = [C.c_int, C.POINTER(StarkConfig)]
for _f in ("columns", "public_inputs", "constraint_degree", "num_constraints", "default_rows"):
    getattr(lib, "starkhip_air_" + _f).argtypes = [C.c_int]
lib.starkhip_air_program.argtypes = [C.c_int, C.POINTER(_u64p), C.POINTER(C.c_size_t)]
lib.starkhip_init.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
lib.starkhip_shutdown.argtypes = [C.c_void_p]
lib.starkhip_shutdown.restype = None
lib.starkhip_prove.argtypes = [C.c_void_p, C.c_int, C.POINTER(StarkConfig), C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, _u64p, C.c_size_t,
                               C.c_uint64, C.POINTER(_u64p), C.POINTER(C.c_size_t)]
lib.starkhip_prove_columns.argtypes = [C.c_void_p, C.c_int, C.POINTER(StarkConfig), C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, _u64p, C.c_size_t,
                                       C.c_uint64, C.POINTER(_u64p), C.POINTER(C.c_size_t)]
lib.starkhip_last_timings.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
lib.starkhip_last_kernel_timings.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
lib.starkhip_last_host_timings.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
lib.starkhip_lde_batch.argtypes = [C.c_void_p, _u64p, C.c_size_t, C.c_uint, C.c_uint, _u64p, _u64p]
lib.starkhip_merkle_cap.argtypes = [C.c_void_p, _u64p, C.c_size_t, C.c_uint, C.c_uint, _u64p]
lib.starkhip_poseidon_permute_batch.argtypes = [C.c_void_p, _u64p, C.c_size_t]
lib.starkhip_field_ops_batch.argtypes = [C.c_void_p, C.c_int, _u64p, _u64p, _u64p, C.c_size_t]
lib.starkhip_poseidon_permute_host.argtypes = [_u64p]
lib.starkhip_poseidon_permute_host.restype = None
lib.starkhip_verify.argtypes = [C.c_int, C.POINTER(StarkConfig), _u64p, C.c_size_t]
lib.starkhip_free.argtypes = [C.c_void_p]
lib.starkhip_free.restype = None
lib.starkhip_host_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
lib.starkhip_trace_log_begin.argtypes = [C.POINTER(C.c_void_p)]
lib.starkhip_trace_log_end.argtypes = [C.c_void_p]
lib.starkhip_trace_set_threads.argtypes = [C.c_int]
lib.starkhip_trace_log_free.argtypes = [C.c_void_p]
lib.starkhip_trace_log_free.restype = None
lib.starkhip_trace_log_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_size_t)] * 4
lib.starkhip_trace_log_from_writes.argtypes = [C.c_size_t, C.c_size_t, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_void_p)]
lib.starkhip_lde_bench.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
lib.starkhip_trace_log_expand_device.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
lib.starkhip_trace_log_overwrites.argtypes = [C.c_void_p] + [C.POINTER(C.c_size_t)] * 2
lib.starkhip_trace_log_expand_host.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_size_t)]
lib.starkhip_prove_compact.argtypes = [C.c_void_p, C.c_int, C.POINTER(StarkConfig), C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t, C.c_uint64,
                                       C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_size_t)]
lib.starkhip_host_free.argtypes = [C.c_void_p]
lib.starkhip_host_free.restype = None
lib.starkhip_trace_fibonacci.argtypes = [C.c_uint64, C.c_uint64, _u64p, C.c_size_t, _u64p]
lib.starkhip_trace_fp12_mul.argtypes = [_u32p, _u32p, _u64p, C.c_size_t, _u64p]
lib.starkhip_trace_final_exp.argtypes = [_u32p, _u64p, C.c_size_t, _u64p]
lib.starkhip_trace_miller_loop.argtypes = [_u32p, _u32p, _u32p, _u32p, _u32p, _u64p, C.c_size_t, _u64p]
lib.starkhip_trace_pairing_precomp.argtypes = [_u32p, _u32p, _u32p, _u64p, C.c_size_t, _u64p]
lib.starkhip_native_fp12_mul.argtypes = [_u32p, _u32p, _u32p]
lib.starkhip_native_final_exponentiate.argtypes = [_u32p, _u32p]
lib.starkhip_native_miller_loop.argtypes = [_u32p, _u32p, _u32p, _u32p, _u32p, _u32p]
lib.starkhip_native_pairing_precomp.argtypes = [_u32p, _u32p, _u32p, _u32p]
_u8p = C.POINTER(C.c_uint8)
lib.starkhip_trace_ecc_aggregate.argtypes = [_u32p, _u8p, _u64p, C.c_size_t, _u64p]
lib.starkhip_native_g1_aggregate.argtypes = [_u32p, _u8p, _u32p]


def _p64(a):
    return a.ctypes.data_as(_u64p)


def _p32(a):
    return a.ctypes.data_as(_u32p)


def _limbs(x, n):
    a = np.ascontiguousarray(x, dtype=np.uint32).reshape(-1)
    assert a.size == n, (a.size, n)
    return a


# ----------------------------------------------------------------------------- AIR metadata
def air_columns(air):
    r = lib.starkhip_air_columns(air)
    if r < 0:
        raise StarkhipError(r)
    return r


def air_public_inputs(air):
    r = lib.starkhip_air_public_inputs(air)
    if r < 0:
        raise StarkhipError(r)
    return r


def air_constraint_degree(air):
    r = lib.starkhip_air_constraint_degree(air)
    if r < 0:
        raise StarkhipError(r)
    return r


def air_num_constraints(air):
    r = lib.starkhip_air_num_constraints(air)
    if r < 0:
        raise StarkhipError(r)
    return r


def air_default_rows(air):
    r = lib.starkhip_air_default_rows(air)
    if r < 0:
        raise StarkhipError(r)
    return r


def air_program(air):
    """Serialised constraint program (numpy uint64 copy)."""
    blob = _u64p()
    words = C.c_size_t()
    _chk(lib.starkhip_air_program(air, C.byref(blob), C.byref(words)))
    return np.ctypeslib.as_array(blob, shape=(words.value,)).copy()


lib.starkhip_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
lib.starkhip_quotient_plan_check.argtypes = [C.c_int, C.c_uint, C.c_uint64, _u64p]


class ProofLayout(C.Structure):
    """starkhip_proof_layout_t: word offsets of every field of a proof blob."""
    _fields_ = [(n, C.c_size_t) for n in (
        "n_columns", "n_quotient_polys", "degree_bits", "rate_bits", "cap_height", "n_fri_layers", "n_query_rounds", "final_poly_len",
        "n_public_inputs", "arity_bits", "off_trace_cap", "off_quotient_cap", "off_local_values", "off_next_values",
        "off_quotient_openings", "off_fri_caps", "off_query_rounds", "query_round_words", "off_final_poly", "off_pow_witness",
        "off_public_inputs", "total_words", "q_trace_leaf", "q_trace_siblings", "q_quotient_leaf", "q_quotient_siblings",
        "initial_sibling_count")] + [(n, C.c_size_t * 16) for n in ("q_step_evals", "q_step_siblings", "step_sibling_count")]


lib.starkhip_proof_layout.argtypes = [_u64p, C.c_size_t, C.POINTER(ProofLayout)]


def proof_layout(proof):
    """Offsets of the fields of a proof blob (starkhip_proof_layout)."""
    proof = np.ascontiguousarray(proof, dtype=np.uint64)
    out = ProofLayout()
    _chk(lib.starkhip_proof_layout(_p64(proof), proof.size, C.byref(out)))
    return out


def quotient_plan_check(air, want_chunks=4, seed=1):
    """CPU replay of the tiled constraint plan against the plain fold; returns the plan statistics (see starkhip.h)."""
    stats = np.zeros(8, dtype=np.uint64)
    _chk(lib.starkhip_quotient_plan_check(air, want_chunks, seed, _p64(stats)))
    return dict(zip(("chunks", "supergroups", "pieces", "records", "lds_cell_records", "direct_loads", "tiles", "contributions"),
                    (int(x) for x in stats)))


lib.starkhip_air_eval_frame.argtypes = [C.c_int, _u64p, _u64p, _u64p, _u64p, _u64p, C.c_int, _u64p]


def air_eval_frame(air, local, nxt, pis, masks, alphas):
    """The ConstraintConsumer accumulators after one `eval_packed_generic` on an arbitrary frame over the quadratic
    extension (host evaluator of the verifier).  local / nxt: [columns, 2], masks: [4, 2], alphas: [k, 2] -> [k, 2]."""
    local, nxt, masks, alphas = (np.ascontiguousarray(x, dtype=np.uint64) for x in (local, nxt, masks, alphas))
    pis = np.ascontiguousarray(pis, dtype=np.uint64)
    cols = air_columns(air)
    if local.shape != (cols, 2) or nxt.shape != (cols, 2) or masks.shape != (4, 2) or alphas.ndim != 2 or alphas.shape[1] != 2 \
            or pis.size != air_public_inputs(air):
        raise StarkhipError(ERR_BAD_SHAPE)
    out = np.zeros_like(alphas)
    _chk(lib.starkhip_air_eval_frame(air, _p64(local), _p64(nxt), _p64(pis), _p64(masks), _p64(alphas), alphas.shape[0], _p64(out)))
    return out


# ----------------------------------------------------------------------------- traces (generate_trace)
class CompactTrace:
    """A trace recorded as runs (starkhip_trace_log_*): the generator's limb vectors with the rows they
    repeat on, expanded on the device by `Prover.prove`.  `shape` is the shape of the dense trace it stands for."""

    def __init__(self, handle):
        self._h = handle
        v = [C.c_size_t() for _ in range(4)]
        _chk(lib.starkhip_trace_log_info(handle, *[C.byref(x) for x in v]))
        self.shape = (v[0].value, v[1].value)
        self.n_records, self.nbytes = v[2].value, 4 * (v[3].value + v[2].value)
        self._fin = weakref.finalize(self, lib.starkhip_trace_log_free, handle)

    def overwrites(self):
        """(records whose run a later clear took back to zero rows, cells cleared inside a longer run) -- the traces of the fillers'
        set-then-clear idiom (trace_log.h: TraceLog::set)."""
        e, z = C.c_size_t(), C.c_size_t()
        _chk(lib.starkhip_trace_log_overwrites(self._h, C.byref(e), C.byref(z)))
        return e.value, z.value

    def expand(self):
        """(dense row-major trace, number of cells two records disagree on) -- CPU replay, for tests."""
        out = np.empty(self.shape, dtype=np.uint64)
        bad = C.c_size_t()
        _chk(lib.starkhip_trace_log_expand_host(self._h, _p64(out), C.byref(bad)))
        return out, bad.value


class _Recording:
    """with _Recording() as r: <one starkhip_trace_* call with a null trace>;  r.trace is the CompactTrace."""

    def __enter__(self):
        self._h = C.c_void_p()
        _chk(lib.starkhip_trace_log_begin(C.byref(self._h)))
        self.trace = None
        return self

    def __exit__(self, et, ev, tb):
        rc = lib.starkhip_trace_log_end(self._h)
        if et is None and rc == 0:
            self.trace = CompactTrace(self._h)
        else:
            lib.starkhip_trace_log_free(self._h)
            if et is None:
                raise StarkhipError(rc)
        return False


def set_trace_threads(n):
    """Host threads one RECORDING generator call may use (starkhip_trace_set_threads; only trace_final_exp uses more than
    one).  Returns the previous setting.  The recorded trace does not depend on it."""
    return int(lib.starkhip_trace_set_threads(int(n)))


def _generate(air, n_rows, out, compact, call):
    """Run one generator: into a dense array (returned, or `out`), or recorded (compact=True)."""
    if compact:
        pis = np.zeros(air_public_inputs(air), dtype=np.uint64)
        with _Recording() as r:
            _chk(call(C.cast(None, _u64p), n_rows or air_default_rows(air), _p64(pis)))
        return r.trace, pis
    t, pis, n = _trace_alloc(air, n_rows, out)
    _chk(call(_p64(t), n, _p64(pis)))
    return t, pis


def _trace_alloc(air, n_rows, out=None):
    """`out`: optional C-contiguous uint64 [n_rows][columns] array to generate into (e.g. Prover.host_array); the
    generators clear it themselves."""
    n_rows = n_rows or air_default_rows(air)
    shape = (n_rows, air_columns(air))
    if out is None:
        out = np.zeros(shape, dtype=np.uint64)
    elif out.shape != shape or out.dtype != np.uint64 or not out.flags.c_contiguous:
        raise ValueError(f"trace buffer must be C-contiguous uint64 {shape}, got {out.dtype} {out.shape}")
    return out, np.zeros(air_public_inputs(air), dtype=np.uint64), n_rows


def trace_from_writes(n_rows, n_cols, writes):
    """A CompactTrace from explicit (row, col, value) writes applied in order through the recorder (tests of its corner cases)."""
    w = np.ascontiguousarray(writes, dtype=np.uint64).reshape(-1, 3)
    h = C.c_void_p()
    _chk(lib.starkhip_trace_log_from_writes(n_rows, n_cols, _p64(w), w.shape[0], C.byref(h)))
    return CompactTrace(h)


def trace_fibonacci(x0, x1, n_rows=None, out=None):
    t, pis, n = _trace_alloc(AIR_TEST_FIBONACCI, n_rows, out)
    _chk(lib.starkhip_trace_fibonacci(x0, x1, _p64(t), n, _p64(pis)))
    return t, pis


def trace_fp12_mul(x, y, n_rows=None, out=None, compact=False):
    """FP12MulStark::generate_trace + the public inputs of fp12_mul_main (src/aggregate_proof.rs:117-148).
    compact=True (every generator below): a CompactTrace instead of the dense rows."""
    xs, ys = _limbs(x, 144), _limbs(y, 144)
    return _generate(AIR_FP12_MUL, n_rows, out, compact, lambda t, n, p: lib.starkhip_trace_fp12_mul(_p32(xs), _p32(ys), t, n, p))


def trace_final_exp(x, n_rows=None, out=None, compact=False):
    xs = _limbs(x, 144)
    return _generate(AIR_FINAL_EXP, n_rows, out, compact, lambda t, n, p: lib.starkhip_trace_final_exp(_p32(xs), t, n, p))


def _ecc_inputs(points, bits):
    pts = np.ascontiguousarray(points, dtype=np.uint32).reshape(-1)
    b = np.ascontiguousarray(np.asarray(bits).astype(bool), dtype=np.uint8).reshape(-1)
    assert pts.size == 24 * ECC_NUM_POINTS and b.size == ECC_NUM_POINTS, (pts.size, b.size)
    return pts, b


def trace_ecc_aggregate(points, bits, n_rows=None, out=None, compact=False):
    """ECCAggStark::generate_trace + ec_aggregate_main's public inputs (src/ecc_aggregate.rs:39-84, src/aggregate_proof.rs:191-209).
    points: [512][24] u32 limbs (x then y); bits: 512 booleans.  The aggregate lands in the last 24 public inputs."""
    pts, b = _ecc_inputs(points, bits)
    return _generate(AIR_ECC_AGGREGATE, n_rows, out, compact,
                     lambda t, n, p: lib.starkhip_trace_ecc_aggregate(_p32(pts), b.ctypes.data_as(_u8p), t, n, p))


def native_g1_aggregate(points, bits):
    """Sum of the points whose bit is set, in the reference's order of operations; 24 limbs (x, y)."""
    pts, b = _ecc_inputs(points, bits)
    out = np.zeros(24, dtype=np.uint32)
    _chk(lib.starkhip_native_g1_aggregate(_p32(pts), b.ctypes.data_as(_u8p), _p32(out)))
    return out


def trace_miller_loop(px, py, qx, qy, qz, n_rows=None, out=None, compact=False):
    a = [_limbs(px, 12), _limbs(py, 12), _limbs(qx, 24), _limbs(qy, 24), _limbs(qz, 24)]
    return _generate(AIR_MILLER_LOOP, n_rows, out, compact, lambda t, n, p: lib.starkhip_trace_miller_loop(*[_p32(v) for v in a], t, n, p))


def trace_pairing_precomp(qx, qy, qz, n_rows=None, out=None, compact=False):
    a = [_limbs(qx, 24), _limbs(qy, 24), _limbs(qz, 24)]
    return _generate(AIR_PAIRING_PRECOMP, n_rows, out, compact, lambda t, n, p: lib.starkhip_trace_pairing_precomp(*[_p32(v) for v in a], t, n, p))


def native_fp12_mul(x, y):
    out = np.zeros(144, dtype=np.uint32)
    _chk(lib.starkhip_native_fp12_mul(_p32(_limbs(x, 144)), _p32(_limbs(y, 144)), _p32(out)))
    return out


def native_final_exponentiate(x):
    out = np.zeros(144, dtype=np.uint32)
    _chk(lib.starkhip_native_final_exponentiate(_p32(_limbs(x, 144)), _p32(out)))
    return out


def native_miller_loop(px, py, qx, qy, qz):
    out = np.zeros(144, dtype=np.uint32)
    _chk(lib.starkhip_native_miller_loop(_p32(_limbs(px, 12)), _p32(_limbs(py, 12)), _p32(_limbs(qx, 24)), _p32(_limbs(qy, 24)),
                                         _p32(_limbs(qz, 24)), _p32(out)))
    return out


def native_pairing_precomp(qx, qy, qz):
    out = np.zeros(68 * 72, dtype=np.uint32)
    _chk(lib.starkhip_native_pairing_precomp(_p32(_limbs(qx, 24)), _p32(_limbs(qy, 24)), _p32(_limbs(qz, 24)), _p32(out)))
    return out


def poseidon_permute_host(state):
    s = np.ascontiguousarray(state, dtype=np.uint64).copy()
    assert s.size == 12
    lib.starkhip_poseidon_permute_host(_p64(s))
    return s


def _column_table(columns):
    """(pointer table, the arrays it points into, n_rows) for starkhip_prove_columns / starkhip_pool_submit_columns."""
    keep = [np.ascontiguousarray(c, dtype=np.uint64) for c in columns]
    if not keep or any(c.ndim != 1 or c.size != keep[0].size for c in keep):
        raise StarkhipError(ERR_BAD_SHAPE)
    addrs = np.fromiter((c.ctypes.data for c in keep), dtype=np.uint64, count=len(keep))
    table = (C.c_void_p * len(keep)).from_buffer_copy(addrs.tobytes())
    return table, keep, keep[0].size


# ----------------------------------------------------------------------------- prover / verifier
class Prover:
    """One context per GPU (starkhip_init).  `prove` mirrors starky::prover::prove."""

    def __init__(self, device=0):
        self._ctx = C.c_void_p()
        _chk(lib.starkhip_init(device, C.byref(self._ctx)))

    def close(self):
        if self._ctx:
            lib.starkhip_shutdown(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def prove(self, air, config, trace, public_inputs, pow_witness=POW_SEARCH, layout=0):
        """trace: numpy uint64, row-major [n][C] (layout 0) or column-major [C][n] (layout 1), or a CompactTrace."""
        if isinstance(trace, CompactTrace):
            pis = np.ascontiguousarray(public_inputs, dtype=np.uint64)
            out = _u64p()
            words = C.c_size_t()
            t0 = time.perf_counter()
            rc = lib.starkhip_prove_compact(self._ctx, air, C.byref(config), trace._h, _p64(pis), pis.size, pow_witness, C.byref(out), C.byref(words))
            self.last_call_s = time.perf_counter() - t0  # the C call alone, without the copy into a numpy array below
            _chk(rc)
            proof = np.ctypeslib.as_array(out, shape=(words.value,)).copy()
            lib.starkhip_free(out)
            return proof
        trace = np.ascontiguousarray(trace, dtype=np.uint64)
        if trace.ndim != 2 or layout not in (0, 1):
            raise StarkhipError(ERR_BAD_SHAPE)
        n_rows, n_cols = trace.shape if layout == 0 else trace.shape[::-1]
        pis = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        out = _u64p()
        words = C.c_size_t()
        t0 = time.perf_counter()
        rc = lib.starkhip_prove(self._ctx, air, C.byref(config), trace.ctypes.data_as(C.c_void_p), n_rows, n_cols, layout, 0, _p64(pis), pis.size,
                                pow_witness, C.byref(out), C.byref(words))
        self.last_call_s = time.perf_counter() - t0
        _chk(rc)
        proof = np.ctypeslib.as_array(out, shape=(words.value,)).copy()
        lib.starkhip_free(out)
        return proof

    def prove_columns(self, air, config, columns, public_inputs, pow_witness=POW_SEARCH):
        """starky's literal `prove(stark, &config, trace_poly_values, ..)`: `columns` is a sequence of separately allocated 1-D uint64
        arrays, one per trace column (`Vec<PolynomialValues<F>>`, src/aggregate_proof.rs:168-175)."""
        table, keep, n_rows = _column_table(columns)
        pis = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        out = _u64p()
        words = C.c_size_t()
        _chk(lib.starkhip_prove_columns(self._ctx, air, C.byref(config), table, n_rows, len(keep), _p64(pis), pis.size, pow_witness, C.byref(out),
                                        C.byref(words)))
        proof = np.ctypeslib.as_array(out, shape=(words.value,)).copy()
        lib.starkhip_free(out)
        return proof

    def lde_bench(self, n_cols, log_n, rate_bits, reps=5, const_per_64=0, device_ptr=None, each=False):
        """starkhip_lde_bench: average milliseconds of one values -> LDE launch over `n_cols` synthetic columns."""
        ms = C.c_float()
        per = (C.c_float * 16)()
        _chk(lib.starkhip_lde_bench(self._ctx, n_cols, log_n, rate_bits, reps, const_per_64, C.c_void_p(device_ptr) if device_ptr else None, C.byref(ms), per))
        return [round(float(x), 2) for x in per[:max(1, min(reps, 16))]] if each else float(ms.value)

    def expand_trace(self, compact):
        """A CompactTrace through the device's expansion kernels; returns the dense row-major [n][C] matrix (tests)."""
        n, c = compact.shape
        out = np.empty((c, n), dtype=np.uint64)
        _chk(lib.starkhip_trace_log_expand_device(self._ctx, compact._h, _p64(out)))
        return np.ascontiguousarray(out.T)

    def set_option(self, name, value):
        """Tuning knob of this context (starkhip_set_option)."""
        _chk(lib.starkhip_set_option(self._ctx, name.encode(), int(value)))

    def prove_device(self, air, config, trace_ptr, n_rows, public_inputs, pow_witness=POW_SEARCH, layout=1, keep=True):
        """trace_ptr: integer device address of a uint64 trace (n_rows x air_columns(air) words) already resident in HBM
        (benchmark path)."""
        pis = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        out = _u64p()
        words = C.c_size_t()
        _chk(lib.starkhip_prove(self._ctx, air, C.byref(config), C.c_void_p(trace_ptr), n_rows, air_columns(air), layout, 1, _p64(pis), pis.size, pow_witness,
                                C.byref(out), C.byref(words)))
        proof = np.ctypeslib.as_array(out, shape=(words.value,)).copy() if keep else None
        lib.starkhip_free(out)
        return proof

    def host_array(self, shape, dtype=np.uint64):
        """Page-locked host array (starkhip_host_alloc) to generate traces into, again and again (`trace_*(…, out=buf)`):
        no 4.8 GB of first-touch page faults per FinalExp trace.  Freed when the array (and every view of it) is gone."""
        shape = tuple(int(s) for s in np.atleast_1d(shape))
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        _chk(lib.starkhip_host_alloc(self._ctx, nbytes, C.byref(p)))
        raw = (C.c_ubyte * nbytes).from_address(p.value)
        weakref.finalize(raw, lib.starkhip_host_free, p.value)
        return np.frombuffer(raw, dtype=dtype).reshape(shape)

    def last_timings(self):
        ms = (C.c_float * N_PHASES)()
        _chk(lib.starkhip_last_timings(self._ctx, ms))
        return dict(zip(PHASE_NAMES, [float(x) for x in ms]))

    def last_host_timings(self):
        ms = (C.c_float * 2)()
        _chk(lib.starkhip_last_host_timings(self._ctx, ms))
        return {"fiat_shamir": float(ms[0]), "other": float(ms[1])}

    def last_kernel_timings(self):
        ms = (C.c_float * 3)()
        _chk(lib.starkhip_last_kernel_timings(self._ctx, ms))
        return {"lde_columns": float(ms[0]), "leaf_hash": float(ms[1]), "quotient_eval": float(ms[2])}

    def lde_batch(self, values_colmajor, rate_bits):
        v = np.ascontiguousarray(values_colmajor, dtype=np.uint64)
        ncols, n = v.shape
        log_n = n.bit_length() - 1
        coeffs = np.zeros_like(v)
        lde = np.zeros((ncols, n << rate_bits), dtype=np.uint64)
        _chk(lib.starkhip_lde_batch(self._ctx, _p64(v), ncols, log_n, rate_bits, _p64(coeffs), _p64(lde)))
        return coeffs, lde

    def merkle_cap(self, lde_colmajor, cap_height):
        v = np.ascontiguousarray(lde_colmajor, dtype=np.uint64)
        ncols, N = v.shape
        cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
        _chk(lib.starkhip_merkle_cap(self._ctx, _p64(v), ncols, N.bit_length() - 1, cap_height, _p64(cap)))
        return cap

    def field_ops(self, op, a, b):
        """Device field arithmetic under test (kernels_selftest.hip): canonical(op(a[i], b[i]))."""
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        assert a.shape == b.shape and a.ndim == 1
        out = np.zeros_like(a)
        _chk(lib.starkhip_field_ops_batch(self._ctx, op, _p64(a), _p64(b), _p64(out), a.size))
        return out

    def poseidon_permute_batch(self, states):
        s = np.ascontiguousarray(states, dtype=np.uint64).copy()
        _chk(lib.starkhip_poseidon_permute_batch(self._ctx, _p64(s), s.shape[0]))
        return s


# ----------------------------------------------------------------------------- proof pool (submit / wait)
class PoolConfig(C.Structure):
    """starkhip_pool_config_t (0 = the library's default)."""
    _fields_ = [("device", C.c_int), ("big_contexts", C.c_uint), ("small_contexts", C.c_uint), ("generator_threads", C.c_uint),
                ("trace_threads", C.c_uint), ("commit_policy", C.c_uint), ("stream_priority", C.c_uint), ("warm_up", C.c_uint), ("gather_ms", C.c_float)]


class TicketInfo(C.Structure):
    _fields_ = [("phase_ms", C.c_float * N_PHASES), ("kernel_ms", C.c_float * 3), ("host_ms", C.c_float * 2), ("t_submit", C.c_double), ("t_generate_start", C.c_double),
                ("t_generate_end", C.c_double), ("t_prove_start", C.c_double), ("t_done", C.c_double), ("leaf_hash_form", C.c_int),
                ("leaf_hash_group", C.c_uint)]


class PoolStats(C.Structure):
    _fields_ = [(n, C.c_ulong) for n in ("big_commit_launches", "small_commit_launches", "small_commit_requests", "max_merged_commitments")]


lib.starkhip_pool_create.argtypes = [C.POINTER(PoolConfig), C.POINTER(C.c_void_p)]
lib.starkhip_pool_destroy.argtypes = [C.c_void_p]
lib.starkhip_pool_destroy.restype = None
lib.starkhip_pool_submit.argtypes = [C.c_void_p, C.c_int, C.POINTER(StarkConfig), C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, _u64p,
                                     C.c_size_t, C.c_uint64, C.POINTER(C.c_uint64)]
lib.starkhip_pool_submit_compact.argtypes = [C.c_void_p, C.c_int, C.POINTER(StarkConfig), C.c_void_p, _u64p, C.c_size_t, C.c_uint64,
                                             C.POINTER(C.c_uint64)]
lib.starkhip_pool_submit_witness.argtypes = [C.c_void_p, C.c_int, C.POINTER(StarkConfig), _u32p, C.c_size_t, C.c_uint64, C.POINTER(C.c_uint64)]
lib.starkhip_pool_wait.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(_u64p), C.POINTER(C.c_size_t), C.POINTER(TicketInfo)]
lib.starkhip_pool_stats.argtypes = [C.c_void_p, C.POINTER(PoolStats)]
lib.starkhip_pool_submit_columns.argtypes = [C.c_void_p, C.c_int, C.POINTER(StarkConfig), C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, _u64p, C.c_size_t,
                                             C.c_uint64, C.POINTER(C.c_uint64)]


class PoolReservation(C.Structure):
    _fields_ = [("device_bytes", C.c_uint64), ("pinned_host_bytes", C.c_uint64), ("big_context_device_bytes", C.c_uint64),
                ("small_context_device_bytes", C.c_uint64), ("big_contexts", C.c_uint), ("small_contexts", C.c_uint)]


lib.starkhip_pool_reservation.argtypes = [C.c_void_p, C.POINTER(PoolReservation)]
# a pool per device behind one handle (starkhip_multipool_*): the same submits with a `slot` argument after the handle
lib.starkhip_multipool_create.argtypes = [C.POINTER(C.c_int), C.c_size_t, C.POINTER(PoolConfig), C.POINTER(C.c_void_p)]
lib.starkhip_multipool_destroy.argtypes = [C.c_void_p]
lib.starkhip_multipool_destroy.restype = None
lib.starkhip_multipool_size.argtypes = [C.c_void_p]
lib.starkhip_multipool_size.restype = C.c_size_t
lib.starkhip_multipool_pool.argtypes = [C.c_void_p, C.c_size_t]
lib.starkhip_multipool_pool.restype = C.c_void_p
lib.starkhip_multipool_device.argtypes = [C.c_void_p, C.c_size_t]
lib.starkhip_multipool_submit.argtypes = [C.c_void_p, C.c_int] + lib.starkhip_pool_submit.argtypes[1:]
lib.starkhip_multipool_submit_compact.argtypes = [C.c_void_p, C.c_int] + lib.starkhip_pool_submit_compact.argtypes[1:]
lib.starkhip_multipool_submit_witness.argtypes = [C.c_void_p, C.c_int] + lib.starkhip_pool_submit_witness.argtypes[1:]
lib.starkhip_multipool_submit_columns.argtypes = [C.c_void_p, C.c_int] + lib.starkhip_pool_submit_columns.argtypes[1:]
lib.starkhip_multipool_submit_witness_batch.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(_u32p), C.POINTER(C.c_size_t), C.c_uint64,
                                                        C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
lib.starkhip_multipool_ticket_slot.argtypes = [C.c_void_p, C.c_uint64]
lib.starkhip_multipool_wait.argtypes = lib.starkhip_pool_wait.argtypes
lib.starkhip_plan_lpt.argtypes = [C.c_size_t, C.POINTER(C.c_int), C.c_size_t, C.POINTER(C.c_int)]
lib.starkhip_air_cost.argtypes = [C.c_int]
lib.starkhip_air_cost.restype = C.c_double
lib.starkhip_hw_queues_status.argtypes = []


def plan_lpt(airs, n_pools):
    """starkhip_plan_lpt: the slot each job of a batch gets on `n_pools` idle pools (longest processing time first)."""
    a = (C.c_int * len(airs))(*[int(x) for x in airs])
    out = (C.c_int * len(airs))()
    _chk(lib.starkhip_plan_lpt(len(airs), a, n_pools, out))
    return [int(x) for x in out]


def air_cost(air):
    return float(lib.starkhip_air_cost(int(air)))


def hw_queues_late():
    """True when the HIP runtime was up before the first pool could ask for its hardware queues (starkhip_hw_queues_status)."""
    return bool(lib.starkhip_hw_queues_status())


class PoolHostInfo(C.Structure):
    _fields_ = [(n, C.c_uint) for n in ("cpu_budget", "generator_threads", "trace_threads_big", "trace_threads_small", "prover_threads")] + [("device", C.c_int), ("pools_on_device", C.c_uint)]


lib.starkhip_pool_host_info.argtypes = [C.c_void_p, C.POINTER(PoolHostInfo)]
lib.starkhip_host_cpu_seconds.argtypes = [C.POINTER(C.c_double)]
lib.starkhip_host_cpu_seconds.restype = None
lib.starkhip_cpu_budget.argtypes = []
lib.starkhip_cpu_budget.restype = C.c_uint
lib.starkhip_proof_blob_stats.argtypes = [C.POINTER(C.c_uint64)]
lib.starkhip_proof_blob_stats.restype = None


def host_cpu_seconds():
    """starkhip_host_cpu_seconds: cumulative CPU seconds of the pools' host work -- {"recording", "proving"}."""
    out = (C.c_double * 3)()
    lib.starkhip_host_cpu_seconds(out)
    return {"recording": float(out[0]), "proving": float(out[1]), "of_proving_in_device_waits": float(out[2])}


def proof_blob_stats():
    """Process-wide counters of the recycled page-locked proof blobs (starkhip_proof_blob_stats)."""
    out = (C.c_uint64 * 5)()
    lib.starkhip_proof_blob_stats(out)
    return dict(zip(("blobs", "busy", "bytes", "taken", "missed"), [int(x) for x in out]))


def witness_operands(air, *args):
    """The packed u32 operand vector `starkhip_pool_submit_witness` takes for `air` from the arguments of its `trace_*`
    generator (starkhip.h has the layouts)."""
    if air == AIR_FP12_MUL:
        parts = [_limbs(args[0], 144), _limbs(args[1], 144)]
    elif air == AIR_FINAL_EXP:
        parts = [_limbs(args[0], 144)]
    elif air == AIR_MILLER_LOOP:
        parts = [_limbs(args[0], 12), _limbs(args[1], 12), _limbs(args[2], 24), _limbs(args[3], 24), _limbs(args[4], 24)]
    elif air == AIR_PAIRING_PRECOMP:
        parts = [_limbs(args[0], 24), _limbs(args[1], 24), _limbs(args[2], 24)]
    elif air == AIR_ECC_AGGREGATE:
        pts, b = _ecc_inputs(args[0], args[1])
        parts = [pts, b.astype(np.uint32)]
    elif air == AIR_TEST_FIBONACCI:
        x0, x1 = int(args[0]), int(args[1])
        parts = [np.array([x0 & 0xFFFFFFFF, x0 >> 32, x1 & 0xFFFFFFFF, x1 >> 32], dtype=np.uint32)]
    else:
        raise StarkhipError(ERR_BAD_AIR)
    return np.ascontiguousarray(np.concatenate(parts), dtype=np.uint32)


class ProofPool:
    """starkhip_pool_*: the library's own scheduler of many proofs on one GPU -- prover contexts with a host thread each,
    generator threads, merged trace commitments for the small AIRs.  The caller side of the reference's six-proof sequence
    (src/aggregate_proof.rs:304-370) only submits and waits:

        t = pool.submit_witness(AIR_MILLER_LOOP, px, py, qx, qy, qz)     # generate_trace + prove, both inside the pool
        proof, info = pool.wait(t)
    """

    def __init__(self, device=0, big_contexts=0, small_contexts=0, generator_threads=0, trace_threads=0, commit_policy=0, gather_ms=0.0,
                 stream_priority=0, warm_up=0, devices=None):
        """`devices` (a list of ordinals, one pool each; an ordinal may repeat): ONE process on several GPUs through
        starkhip_multipool_* -- the submits then take `slot` (-1: the library places the job, longest processing time first)."""
        cfg = PoolConfig(device, big_contexts, small_contexts, generator_threads, trace_threads, commit_policy, stream_priority, warm_up, gather_ms)
        self._h = C.c_void_p()
        self._multi = devices is not None
        if self._multi:
            devs = (C.c_int * len(devices))(*[int(d) for d in devices])
            _chk(lib.starkhip_multipool_create(devs, len(devices), C.byref(cfg), C.byref(self._h)))
        else:
            _chk(lib.starkhip_pool_create(C.byref(cfg), C.byref(self._h)))
        self._keep = {}  # ticket -> inputs that must outlive the proof

    def _call(self, name, slot, *args):
        if self._multi:
            return getattr(lib, "starkhip_multipool_" + name)(self._h, slot, *args)
        return getattr(lib, "starkhip_pool_" + name)(self._h, *args)

    @property
    def n_pools(self):
        return int(lib.starkhip_multipool_size(self._h)) if self._multi else 1

    def slot_of(self, ticket):
        return int(lib.starkhip_multipool_ticket_slot(self._h, ticket)) if self._multi else 0

    def _pools(self):
        return [C.c_void_p(lib.starkhip_multipool_pool(self._h, k)) for k in range(self.n_pools)] if self._multi else [self._h]

    def close(self):
        if self._h:
            (lib.starkhip_multipool_destroy if self._multi else lib.starkhip_pool_destroy)(self._h)
            self._h = C.c_void_p()
            self._keep.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def submit(self, air, config, trace, public_inputs, pow_witness=POW_SEARCH, layout=0, slot=-1):
        """As Prover.prove, asynchronously: `trace` a host array or a CompactTrace.  Returns the ticket."""
        pis = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        cfg = StarkConfig.from_buffer_copy(config)
        t = C.c_uint64()
        if isinstance(trace, CompactTrace):
            _chk(self._call("submit_compact", slot, air, C.byref(cfg), trace._h, _p64(pis), pis.size, pow_witness, C.byref(t)))
        else:
            trace = np.ascontiguousarray(trace, dtype=np.uint64)
            if trace.ndim != 2 or layout not in (0, 1):
                raise StarkhipError(ERR_BAD_SHAPE)
            n_rows, n_cols = trace.shape if layout == 0 else trace.shape[::-1]
            _chk(self._call("submit", slot, air, C.byref(cfg), trace.ctypes.data_as(C.c_void_p), n_rows, n_cols, layout, 0, _p64(pis), pis.size,
                            pow_witness, C.byref(t)))
        self._keep[t.value] = (trace, pis, cfg)
        return t.value

    def submit_device(self, air, config, trace_ptr, n_rows, public_inputs, pow_witness=POW_SEARCH, layout=1, slot=0):
        """`trace_ptr`: device address of a uint64 trace already resident in HBM (benchmark path); on several devices `slot` names
        the pool of the device the trace lives on."""
        pis = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        cfg = StarkConfig.from_buffer_copy(config)
        t = C.c_uint64()
        _chk(self._call("submit", slot, air, C.byref(cfg), C.c_void_p(trace_ptr), n_rows, air_columns(air), layout, 1, _p64(pis), pis.size,
                        pow_witness, C.byref(t)))
        self._keep[t.value] = (pis, cfg)
        return t.value

    def submit_columns(self, air, config, columns, public_inputs, pow_witness=POW_SEARCH, slot=-1):
        """As Prover.prove_columns, asynchronously: one separately allocated array per trace column."""
        table, keep, n_rows = _column_table(columns)
        pis = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        cfg = StarkConfig.from_buffer_copy(config)
        t = C.c_uint64()
        _chk(self._call("submit_columns", slot, air, C.byref(cfg), table, n_rows, len(keep), _p64(pis), pis.size, pow_witness, C.byref(t)))
        self._keep[t.value] = (keep, pis, cfg)  # the table itself was copied by the library
        return t.value

    def submit_witness(self, air, *generator_args, config=None, pow_witness=POW_SEARCH, slot=-1):
        """generate_trace + prove inside the pool, from the arguments of `trace_<air>`."""
        ops = witness_operands(air, *generator_args)
        t = C.c_uint64()
        cfgp = C.byref(StarkConfig.from_buffer_copy(config)) if config is not None else None
        _chk(self._call("submit_witness", slot, air, cfgp, _p32(ops), ops.size, pow_witness, C.byref(t)))
        return t.value

    def submit_witness_batch(self, jobs, pow_witness=POW_SEARCH):
        """jobs: [(air, generator args...)]; on several devices the batch is placed longest processing time first
        (starkhip_multipool_submit_witness_batch), on one pool in the given order.  Returns the tickets in the jobs' order."""
        ops = [witness_operands(j[0], *j[1:]) for j in jobs]
        if not self._multi:
            out = []
            for j, o in zip(jobs, ops):
                t = C.c_uint64()
                _chk(lib.starkhip_pool_submit_witness(self._h, j[0], None, _p32(o), o.size, pow_witness, C.byref(t)))
                out.append(t.value)
            return out
        n = len(jobs)
        airs = (C.c_int * n)(*[int(j[0]) for j in jobs])
        ptrs = (_u32p * n)(*[_p32(o) for o in ops])
        lens = (C.c_size_t * n)(*[o.size for o in ops])
        tickets = (C.c_uint64 * n)()
        rcs = (C.c_int * n)()
        rc = lib.starkhip_multipool_submit_witness_batch(self._h, n, airs, ptrs, lens, pow_witness, tickets, rcs)
        if rc != 0:
            # the call is not all-or-nothing: the jobs it accepted are running.  Wait for them and drop their proofs before the error
            # goes up (the Rust binding does the same by building its tickets first) -- a ticket nobody holds could never be freed
            for t in tickets:
                if int(t):
                    out, words, info = _u64p(), C.c_size_t(), TicketInfo()
                    if lib.starkhip_multipool_wait(self._h, int(t), C.byref(out), C.byref(words), C.byref(info)) == 0:
                        lib.starkhip_free(out)
            _chk(rc)
        return [int(t) for t in tickets]

    def wait(self, ticket, keep=True):
        """(proof, info) of `ticket`; raises StarkhipError with the proof's status if it failed.  info: phase_ms / kernel_ms
        dicts and the job's timeline in seconds since the pool was created."""
        out = _u64p()
        words = C.c_size_t()
        info = TicketInfo()
        rc = (lib.starkhip_multipool_wait if self._multi else lib.starkhip_pool_wait)(self._h, ticket, C.byref(out), C.byref(words), C.byref(info))
        self._keep.pop(ticket, None)
        _chk(rc)
        proof = np.ctypeslib.as_array(out, shape=(words.value,)).copy() if keep else None
        lib.starkhip_free(out)
        return proof, {"phase_ms": dict(zip(PHASE_NAMES, [float(x) for x in info.phase_ms])),
                       "kernel_ms": {"lde_columns": float(info.kernel_ms[0]), "leaf_hash": float(info.kernel_ms[1]), "quotient_eval": float(info.kernel_ms[2])},
                       "host_ms": {"fiat_shamir": float(info.host_ms[0]), "other": float(info.host_ms[1])},
                       "timeline_s": [info.t_submit, info.t_generate_start, info.t_generate_end, info.t_prove_start, info.t_done],
                       "leaf_hash_form": ("quad", "row", "merged", "lane", "host", "pair")[min(info.leaf_hash_form, 5)], "leaf_hash_group": int(info.leaf_hash_group)}

    def reservation(self):
        """starkhip_pool_reservation: what the pool's contexts hold (bytes; summed over the devices' pools, the per-context figures the largest)."""
        out = None
        for h in self._pools():
            r = PoolReservation()
            _chk(lib.starkhip_pool_reservation(h, C.byref(r)))
            one = {n: int(getattr(r, n)) for n, _ in PoolReservation._fields_}
            if out is None:
                out = one
            else:
                for n in one:
                    out[n] = max(out[n], one[n]) if n.endswith("context_device_bytes") else out[n] + one[n]
        return out

    def host_info(self):
        """starkhip_pool_host_info of every pool behind this handle: the CPUs it plans with and its host threads."""
        out = []
        for h in self._pools():
            r = PoolHostInfo()
            _chk(lib.starkhip_pool_host_info(h, C.byref(r)))
            out.append({n: int(getattr(r, n)) for n, _ in PoolHostInfo._fields_})
        return out

    def stats(self, per_pool=False):
        each = []
        for h in self._pools():
            s = PoolStats()
            _chk(lib.starkhip_pool_stats(h, C.byref(s)))
            each.append({n: int(getattr(s, n)) for n, _ in PoolStats._fields_})
        if per_pool:
            return each
        return {n: (max if n == "max_merged_commitments" else sum)(e[n] for e in each) for n in each[0]}


def verify_stark_proof(air, config, proof):
    """Mirror of starky::verifier::verify_stark_proof; raises StarkhipError on rejection."""
    p = np.ascontiguousarray(proof, dtype=np.uint64)
    _chk(lib.starkhip_verify(air, C.byref(config), _p64(p), p.size))


def trace_rows_to_poly_values(trace_rows):
    """starky::util::trace_rows_to_poly_values: row-major [n][C] -> column-major [C][n]."""
    return np.ascontiguousarray(np.asarray(trace_rows, dtype=np.uint64).T)
