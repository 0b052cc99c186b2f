"""Compact traces: a generator that records its writes as runs must stand for exactly the dense matrix
it would have filled, with no cell that two records disagree on (the device expands records in parallel).
FinalExp (4.8 GB dense) is covered in test_airs_cpu.py next to the dense trace it already builds."""
import ctypes as C

import numpy as np
import pytest

import starky_bls12_381_amd as S
from starky_bls12_381_amd import aggregate as A
from bls_util import random_fp12
from test_aggregate_cpu import _bls_points


def _cases():
    _, pk, hm, sig = _bls_points()
    jobs, _ = A.signature_jobs(pk, hm, sig)
    yield "fp12_mul", S.trace_fp12_mul, (random_fp12(0x5EED4000), random_fp12(0x5EED4001))
    yield "pairing_precomp", S.trace_pairing_precomp, jobs["pp2"][1]
    yield "miller_loop", S.trace_miller_loop, jobs["ml2"][1]


@pytest.mark.parametrize("name,fn,args", list(_cases()), ids=lambda v: v if isinstance(v, str) else "")
def test_recorded_trace_expands_to_the_dense_one(name, fn, args):
    dense, pis = fn(*args)
    compact, cpis = fn(*args, compact=True)
    assert compact.shape == dense.shape and np.array_equal(cpis, pis)
    expanded, conflicts = compact.expand()
    assert conflicts == 0
    assert np.array_equal(expanded, dense)
    assert compact.nbytes * 5 < dense.nbytes  # FP12Mul (16 rows) has the least repetition: 10x; the 1024-row AIRs 7-10x


@pytest.mark.parametrize("job,fn", [("ml1", S.trace_miller_loop), ("pp2", S.trace_pairing_precomp)])
def test_recorded_on_several_threads(job, fn):
    """starkhip_trace_set_threads: the 12-row blocks of the Miller loop / of the pairing precomputation are filled as tasks once
    the running value at every block start is known from a native pass; the parts are taken over in task order, so the
    recording is the same for any thread count > 1 and stands for the dense matrix."""
    _, pk, hm, sig = _bls_points()
    jobs, _ = A.signature_jobs(pk, hm, sig)
    dense, pis = fn(*jobs[job][1])
    seen = []
    try:
        for threads in (2, 5):
            S.set_trace_threads(threads)
            compact, cpis = fn(*jobs[job][1], compact=True)
            expanded, conflicts = compact.expand()
            assert conflicts == 0 and np.array_equal(expanded, dense) and np.array_equal(cpis, pis)
            seen.append((compact.n_records, compact.nbytes))
    finally:
        S.set_trace_threads(1)
    assert seen[0] == seen[1]
    assert S.set_trace_threads(0) == 1 and S.set_trace_threads(1) == 1  # clamped to >= 1


def test_ecc_aggregate_recorded_trace():
    from test_ecc_aggregate_cpu import pack, reference_vector
    pts, bits, _ = reference_vector()
    arr, b = pack(pts, bits)
    dense, pis = S.trace_ecc_aggregate(arr, b)
    compact, cpis = S.trace_ecc_aggregate(arr, b, compact=True)
    expanded, conflicts = compact.expand()
    assert conflicts == 0 and np.array_equal(expanded, dense) and np.array_equal(cpis, pis)


def test_recording_is_one_generator_call_per_log_and_per_thread():
    x, y = random_fp12(0x5EED4002), random_fp12(0x5EED4003)
    h, h2 = C.c_void_p(), C.c_void_p()
    assert S.lib.starkhip_trace_log_begin(C.byref(h)) == 0
    try:
        assert S.lib.starkhip_trace_log_begin(C.byref(h2)) == S.ERR_BAD_SHAPE  # this thread is already recording
        assert S.lib.starkhip_trace_log_end(C.c_void_p(1)) == S.ERR_BAD_SHAPE   # not the armed log
        pis = np.zeros(S.air_public_inputs(S.AIR_FP12_MUL), dtype=np.uint64)
        xs, ys = np.ascontiguousarray(x, dtype=np.uint32), np.ascontiguousarray(y, dtype=np.uint32)
        call = lambda: S.lib.starkhip_trace_fp12_mul(S.api._p32(xs), S.api._p32(ys), None, 16, S.api._p64(pis))  # noqa: E731
        assert call() == 0
        assert call() == S.ERR_BAD_SHAPE  # a second generator call would mix two traces in one log
        assert S.lib.starkhip_trace_log_end(h) == 0
    finally:
        S.lib.starkhip_trace_log_free(h)
    # not recording any more: a null dense buffer is refused, not written through
    assert S.lib.starkhip_trace_fp12_mul(S.api._p32(xs), S.api._p32(ys), None, 16, S.api._p64(pis)) == S.ERR_BAD_SHAPE
    # an empty recording is an error, and the prover entry refuses a log without a context and the raw layout code
    assert S.lib.starkhip_trace_log_begin(C.byref(h)) == 0
    assert S.lib.starkhip_trace_log_end(h) == S.ERR_BAD_SHAPE
    S.lib.starkhip_trace_log_free(h)
    c, cpis = S.trace_fp12_mul(x, y, compact=True)
    cfg = S.StarkConfig.for_air(S.AIR_FP12_MUL)
    out, words = C.POINTER(C.c_uint64)(), C.c_size_t()
    assert S.lib.starkhip_prove_compact(None, S.AIR_FP12_MUL, C.byref(cfg), c._h, S.api._p64(cpis), cpis.size, S.POW_SEARCH, C.byref(out),
                                        C.byref(words)) == S.ERR_NO_DEVICE


def test_a_one_row_run_that_is_cleared_again_leaves_an_empty_record():
    """TraceLog::set takes the row back from the run that just wrote it ("selector = 1 on rows a..b", then "selector(b) = 0"): a ONE-row
    run ends up with zero rows and its record stays in the log.  Such a record stands for no cell -- the host replay and the device
    expansion (kernels_trace.hip: the lane -> (limb, row) mapping divides by the run length) must skip it."""
    rows, cols = 8, 3
    writes = [(r, 0, 1 + r) for r in range(rows)]
    writes += [(2, 1, 1), (2, 1, 0)]                 # one-row run, cleared again: empty record
    writes += [(r, 2, 1) for r in range(3, 7)] + [(6, 2, 0)]   # "rows 3..6", then the last one cleared: the run shrinks to 3..5
    writes += [(r, 1, 7) for r in range(4, 8)] + [(5, 1, 0)]   # a clear inside the latest run: a late zero
    log = S.trace_from_writes(rows, cols, writes)
    assert log.overwrites() == (1, 1)
    dense = np.zeros((rows, cols), dtype=np.uint64)
    for r, c, v in writes:
        dense[r, c] = v
    got, conflicts = log.expand()
    assert conflicts == 0 and np.array_equal(got, dense)
    with pytest.raises(S.StarkhipError):
        S.trace_from_writes(rows, cols, [(rows, 0, 1)])       # outside the trace
    with pytest.raises(S.StarkhipError):
        S.trace_from_writes(rows, cols, [(0, 0, 1 << 32)])    # a cell of more than 32 bits
