"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.environ.get("STARKHIP_ORACLE_LIBRARY") or os.path.join(ORACLE_DIR, "liboracle.so")  # override: the sanitizer build (make asan)


def build():
    if os.environ.get("STARKHIP_ORACLE_LIBRARY"):
        return
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(ORACLE_DIR, "stark_oracle.c")):
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)


build()
lib = C.CDLL(LIB)
_u64p = C.POINTER(C.c_uint64)


class OracleConfig(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("security_bits", "num_challenges", "rate_bits", "cap_height", "proof_of_work_bits",
                                          "arity_bits", "final_poly_bits", "num_query_rounds")]


lib.oracle_poseidon_permute.argtypes = [_u64p]
lib.oracle_round_constants.argtypes = [_u64p]
lib.oracle_hash_no_pad.argtypes = [_u64p, C.c_size_t, _u64p]
lib.oracle_two_to_one.argtypes = [_u64p, _u64p, _u64p]
lib.oracle_fft.argtypes = [_u64p, C.c_uint]
lib.oracle_ifft.argtypes = [_u64p, C.c_uint]
lib.oracle_coset_fft.argtypes = [_u64p, C.c_uint, C.c_uint64]
lib.oracle_lde_rows.argtypes = [_u64p, C.c_size_t, C.c_uint, C.c_uint, _u64p, _u64p]
lib.oracle_merkle_cap.argtypes = [_u64p, C.c_size_t, C.c_uint, C.c_uint, _u64p]
lib.oracle_check_trace.argtypes = [_u64p, C.c_size_t, _u64p, C.c_size_t, _u64p, _u64p]
lib.oracle_check_trace.restype = C.c_long
lib.oracle_prove.argtypes = [_u64p, C.c_size_t, C.POINTER(OracleConfig), _u64p, C.c_uint32, _u64p, C.c_uint64, C.POINTER(_u64p),
                             C.POINTER(C.c_size_t)]
lib.oracle_free.argtypes = [C.c_void_p]
lib.oracle_set_zeta_on_coset.argtypes = [C.c_long]
lib.oracle_set_zeta_on_coset.restype = None
lib.oracle_mul.argtypes = [C.c_uint64, C.c_uint64]
lib.oracle_mul.restype = C.c_uint64
lib.oracle_mul_slow.argtypes = [C.c_uint64, C.c_uint64]
lib.oracle_mul_slow.restype = C.c_uint64


def _p(a):
    return a.ctypes.data_as(_u64p)


def poseidon_permute(state):
    s = np.ascontiguousarray(state, dtype=np.uint64).copy()
    lib.oracle_poseidon_permute(_p(s))
    return s


def round_constants():
    rc = np.zeros(360, dtype=np.uint64)
    lib.oracle_round_constants(_p(rc))
    return rc


def hash_no_pad(x):
    x = np.ascontiguousarray(x, dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    lib.oracle_hash_no_pad(_p(x), x.size, _p(out))
    return out


def lde_rows(cols, rate_bits):
    """cols column-major [C][n] -> (coeffs [C][n], lde row-major [N][C], natural point order)."""
    cols = np.ascontiguousarray(cols, dtype=np.uint64)
    C_, n = cols.shape
    coeffs = np.zeros_like(cols)
    lde = np.zeros((n << rate_bits, C_), dtype=np.uint64)
    lib.oracle_lde_rows(_p(cols), C_, n.bit_length() - 1, rate_bits, _p(coeffs), _p(lde))
    return coeffs, lde


def merkle_cap(rows_natural, cap_h):
    rows = np.ascontiguousarray(rows_natural, dtype=np.uint64)
    N, width = rows.shape
    cap = np.zeros((1 << cap_h, 4), dtype=np.uint64)
    lib.oracle_merkle_cap(_p(rows), width, N.bit_length() - 1, cap_h, _p(cap))
    return cap


def check_trace(air_blob, trace_rows, pis):
    """Returns (violations, (constraint, row, value) of the lowest-index violation)."""
    blob = np.ascontiguousarray(air_blob, dtype=np.uint64)
    t = np.ascontiguousarray(trace_rows, dtype=np.uint64)
    p = np.ascontiguousarray(pis, dtype=np.uint64)
    out3 = np.zeros(3, dtype=np.uint64)
    bad = lib.oracle_check_trace(_p(blob), blob.size, _p(t), t.shape[0], _p(p), _p(out3))
    return bad, tuple(int(x) for x in out3)


lib.oracle_eval_frame.argtypes = [_u64p, C.c_size_t, _u64p, _u64p, _u64p, _u64p, _u64p]


def eval_frame(air_blob, local, nxt, pis, masks):
    """mask(kind) * c_k of every constraint on one arbitrary frame."""
    blob = np.ascontiguousarray(air_blob, dtype=np.uint64)
    a = [np.ascontiguousarray(x, dtype=np.uint64) for x in (local, nxt, pis, masks)]
    out = np.zeros(int(blob[4]), dtype=np.uint64)
    assert lib.oracle_eval_frame(_p(blob), blob.size, *[_p(x) for x in a], _p(out)) == 0
    return out


def prove(air_blob, config, trace_cols, pis, pow_witness=0xFFFFFFFFFFFFFFFF):
    """trace_cols column-major [C][n].  `config` any object with the StarkConfig field names."""
    blob = np.ascontiguousarray(air_blob, dtype=np.uint64)
    t = np.ascontiguousarray(trace_cols, dtype=np.uint64)
    p = np.ascontiguousarray(pis, dtype=np.uint64)
    cfg = OracleConfig(*[getattr(config, n) for n, _ in OracleConfig._fields_])
    out = _u64p()
    words = C.c_size_t()
    rc = lib.oracle_prove(_p(blob), blob.size, C.byref(cfg), _p(t), t.shape[1], _p(p), pow_witness, C.byref(out), C.byref(words))
    if rc != 0:
        raise RuntimeError(f"oracle_prove failed: {rc}")
    proof = np.ctypeslib.as_array(out, shape=(words.value,)).copy()
    lib.oracle_free(out)
    return proof


lib.oracle_bench_quotient.argtypes = [_u64p, C.c_size_t, _u64p, C.c_size_t, _u64p]
lib.oracle_bench_quotient.restype = C.c_uint64


def bench_quotient(air_blob, rows, pis):
    """Evaluate every constraint (two alphas) at rows.shape[0] - 1 points; returns a checksum."""
    blob = np.ascontiguousarray(air_blob, dtype=np.uint64)
    r = np.ascontiguousarray(rows, dtype=np.uint64)
    p = np.ascontiguousarray(pis, dtype=np.uint64)
    return lib.oracle_bench_quotient(_p(blob), blob.size, _p(r), r.shape[0] - 1, _p(p))
