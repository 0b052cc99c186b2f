"""Host side of the end-to-end signature driver (starky_bls12_381_amd/signature.py) without a GPU: synthetic valid
signatures, the batch plan, and the generate -> prove pipeline with a stand-in prover (real compact trace generators)."""
import numpy as np
import pytest

import starky_bls12_381_amd as S
from starky_bls12_381_amd import aggregate as A
from starky_bls12_381_amd import parallel
from starky_bls12_381_amd import signature as G
from bls_util import native_vectors


def _vector():
    return native_vectors()["bls_signature"]


def test_synthetic_signatures_are_valid_and_different():
    sigs = G.synthetic_signatures(4, _vector(), seed=5)
    seen = set()
    for pk, hm, sig in sigs:
        _, natives = A.signature_jobs(pk, hm, sig)
        assert A.signature_is_valid(natives)   # e(pk, H) * e(-G, sig) == 1 with the product's natives
        seen.add(bytes(pk[0]) + bytes(sig[0]))
    assert len(seen) == 4
    # a signature under another key does not check
    _, natives = A.signature_jobs(sigs[0][0], sigs[1][1], sigs[1][2])
    assert not A.signature_is_valid(natives)
    words = G.pack_operands(sigs)
    assert words.shape == (4, G.OPERAND_WORDS)
    back = G.unpack_operands(words)
    assert all(np.array_equal(a, b) for x, y in zip(sigs, back) for p, q in zip(x, y) for a, b in zip(p, q))


def test_batch_plan_for_eight_signatures_on_eight_ranks():
    plan = G.plan_batch(8, 8)
    flat = sorted(j for r in plan for j in r)
    assert flat == sorted((i, n) for i in range(8) for n in A.JOB_ORDER)
    assert all(sum(1 for _, n in r if n == "final_exp") == 1 for r in plan)   # one FinalExp per GPU (BASELINE configs[4])


def _fake_prove(pv, air, cfg, trace, pis):
    assert isinstance(trace, S.CompactTrace) and trace.shape[1] == S.air_columns(air)
    return np.concatenate([np.zeros(3, dtype=np.uint64), np.asarray(pis, dtype=np.uint64)])


def test_pipeline_generates_on_threads_and_links_hold_on_what_was_proven():
    sigs = G.synthetic_signatures(1, _vector(), seed=9)
    mine = G.plan_batch(1, 1)[0]
    args, natives = G.job_arguments(sigs, mine)
    results, stats = G.run_jobs([object(), object(), object()], mine, args, gen_threads=6, prove=_fake_prove)
    assert sorted(results) == sorted(mine)
    six = G.signature_proofs(results, 0)
    assert A.check_links(six)
    assert A.check_statement(six, sigs[0][0], sigs[0][1], sigs[0][2])
    assert A.signature_is_valid(natives[0], six)
    assert stats["generate_s"] > stats["wall_s"] * 0.5 or stats["wall_s"] < 2.0   # generation ran on several threads
    # the statement check is about the POINTS: proofs of another signature verify and link, but are not this statement
    other = G.synthetic_signatures(2, _vector(), seed=9)[1]
    assert not A.check_statement(six, sigs[0][0], other[1], other[2])
    # ... and about the KEY: six proofs made for another key pk' that also satisfies e(pk', H(m)) e(-G, sig) = 1 would verify and
    # link; what makes them a different statement is ml1's G1 operand (src/aggregate_proof.rs:540-545)
    assert not A.check_statement(six, other[0], sigs[0][1], sigs[0][2])
    import pytest
    with pytest.raises(ValueError):
        A.check_statement(six, None, sigs[0][1], sigs[0][2])   # no key and no ECCAggStark proof that publishes it
    # ... and a final_exp proof that attests to something else than 1 is not a valid signature
    bad = dict(six)
    air, blob, cfg = bad["final_exp"]
    blob = blob.copy()
    blob[-1] ^= np.uint64(1)
    bad["final_exp"] = (air, blob, cfg)
    assert not A.check_statement(bad, sigs[0][0], sigs[0][1], sigs[0][2])


def test_pipeline_reports_a_failing_job():
    import pytest

    def boom(pv, air, cfg, trace, pis):
        raise RuntimeError("device lost")
    sigs = G.synthetic_signatures(1, _vector(), seed=9)
    mine = [(0, "fp12_mul")]
    args, _ = G.job_arguments(sigs, mine)
    with pytest.raises(RuntimeError):
        G.run_jobs([object()], mine, args, gen_threads=1, prove=boom)


def test_by_type_schedule_starts_final_exp_after_the_small_proofs():
    """`big_after_small`: with two pools the FinalExp contexts wait for the last small proof (measured slower on one GPU and off by
    default, kept as an option); the timeline and the restored process-wide settings are part of the contract."""
    import time
    jobs = [(i, n) for i in range(2) for n in A.JOB_ORDER]
    args = {j: () for j in jobs}
    order = []

    def generate(name, *a):
        return name, np.zeros(1, dtype=np.uint64)

    def prove(pv, air, cfg, trace, pis):
        time.sleep(0.02 if trace != "final_exp" else 0.0)
        order.append(trace)
        return np.zeros(1, dtype=np.uint64)

    before = S.set_trace_threads(1)
    res, st = G.run_jobs({"big": [object()], "small": [object(), object()]}, jobs, args, gen_threads=4, prove=prove, generate=generate,
                         big_after_small=True)
    assert len(res) == 12 and set(st["timeline"]) == set(jobs)
    first_big = order.index("final_exp")
    assert first_big == 10 and order[10:] == ["final_exp", "final_exp"]   # every small proof came first
    for (i, n), (g0, g1, p0, p1) in st["timeline"].items():
        assert g0 <= g1 <= p0 <= p1
    assert S.set_trace_threads(before) == 1  # injected generators: the setting was not touched
    order.clear()
    res, _ = G.run_jobs({"big": [object()], "small": [object(), object()]}, jobs, args, gen_threads=4, prove=prove, generate=generate)
    assert len(res) == 12 and order.index("final_exp") < 10  # both pools at once: FinalExp does not wait


class _FakePool(S.ProofPool):
    """Stands in for the library's proof pool on a box without a GPU: `submit_witness` unpacks the operand vector exactly as the
    pool's generator thread does (starkhip.h's layouts), runs the REAL recording generator and returns a blob that ends in the
    public inputs, as every proof does."""

    def __init__(self):   # no starkhip_pool_create
        self._jobs = {}
        self._next = 1

    def close(self):
        pass

    def submit_witness(self, air, *generator_args, config=None, pow_witness=S.POW_SEARCH):
        w = S.witness_operands(air, *generator_args)
        if air == S.AIR_FP12_MUL:
            assert w.size == 288
            _, pis = S.trace_fp12_mul(w[:144], w[144:], compact=True)
        elif air == S.AIR_FINAL_EXP:
            assert w.size == 144
            pis = np.concatenate([w.astype(np.uint64), S.native_final_exponentiate(w).astype(np.uint64)])   # public inputs only: input, output
        elif air == S.AIR_MILLER_LOOP:
            assert w.size == 96
            _, pis = S.trace_miller_loop(w[:12], w[12:24], w[24:48], w[48:72], w[72:96], compact=True)
        elif air == S.AIR_PAIRING_PRECOMP:
            assert w.size == 72
            _, pis = S.trace_pairing_precomp(w[:24], w[24:48], w[48:72], compact=True)
        else:
            raise AssertionError(air)
        t = self._next
        self._next += 1
        self._jobs[t] = np.concatenate([np.full(5, 7 + air, dtype=np.uint64), np.asarray(pis, dtype=np.uint64)])
        return t

    def wait(self, ticket, keep=True):
        return self._jobs.pop(ticket), {"timeline_s": [0.0, 0.0, 0.01, 0.01, 0.02], "phase_ms": {}, "kernel_ms": {}, "host_ms": {}}

    def stats(self):
        return {}


def test_pool_driver_submits_every_job_with_the_operands_its_generator_takes():
    """signature.run_jobs_pool / one_step on a pool: the four jobs that need only the operands go first, the two that need the
    native Miller-loop values follow; what comes back links, binds to the statement and attests to a valid signature."""
    sigs = G.synthetic_signatures(2, _vector(), seed=11)
    mine = G.plan_batch(2, 1)[0]
    elapsed, results, stats, got_sigs, natives = G.one_step(None, 2, _FakePool(), mine, sigs)
    assert sorted(results) == sorted(mine) and sorted(natives) == [0, 1]
    verdicts = G.check_signatures(results, got_sigs, natives, 2)
    assert verdicts == {0: True, 1: True}
    # the FinalExp public inputs the fake pool derived from the packed operands are the natives' product and its exponentiation
    fe = results[(1, "final_exp")][1][-288:]
    assert np.array_equal(fe[:144], natives[1]["product"].astype(np.uint64)) and np.array_equal(fe[144:], natives[1]["final"].astype(np.uint64))
    assert stats["generate_s"] >= 0 and elapsed > 0


def test_compiled_placement_rule_is_the_python_plan():
    """starkhip_plan_lpt (what starkhip_multipool_submit_witness_batch applies inside ONE process) places a batch exactly as
    signature.plan_batch / parallel.assign_jobs place it across torchrun ranks: same cost table, longest first, ties to the lowest index.
    BASELINE configs[3]: one signature on six devices -- one proof each; configs[4]: 48 proofs on eight -- a FinalExp proof per device
    first, then two MillerLoop, two PairingPrecomp and an FP12Mul each."""
    for air, cost in parallel.AIR_COST.items():
        assert S.api.air_cost(air) == cost
    for batch, world in ((1, 6), (8, 8), (8, 6), (3, 2), (5, 8)):
        jobs = [(i, name) for i in range(batch) for name in A.JOB_ORDER]
        airs = [A.JOB_AIR[name] for _, name in jobs]
        slots = S.api.plan_lpt(airs, world)
        want = [None] * len(jobs)
        for rank, mine in enumerate(G.plan_batch(batch, world)):
            for j in mine:
                want[jobs.index(j)] = rank
        assert slots == want, (batch, world)
    one = S.api.plan_lpt([A.JOB_AIR[n] for n in A.JOB_ORDER], 6)
    assert sorted(one) == list(range(6)) and one[A.JOB_ORDER.index("final_exp")] == 0
    eight = S.api.plan_lpt([A.JOB_AIR[n] for _ in range(8) for n in A.JOB_ORDER], 8)
    per_dev = [sorted(A.JOB_AIR[A.JOB_ORDER[k % 6]] for k, s in enumerate(eight) if s == d) for d in range(8)]
    assert all(p == sorted(A.JOB_AIR[n] for n in A.JOB_ORDER) for p in per_dev)   # every device: a FinalExp, two MillerLoop, two PairingPrecomp, an FP12Mul


def test_multi_device_handle_without_a_gpu_fails_loudly():
    """No CPU fallback: without a device the handle cannot be created, and a null handle is refused by every entry point."""
    import ctypes as C
    with pytest.raises(S.StarkhipError) as e:
        S.ProofPool(devices=[0, 1])
    assert e.value.code == S.ERR_NO_DEVICE
    t = C.c_uint64()
    ops = (C.c_uint32 * 144)()
    assert S.lib.starkhip_multipool_submit_witness(None, -1, S.AIR_FINAL_EXP, None, ops, 144, S.POW_SEARCH, C.byref(t)) == S.ERR_NO_DEVICE
    assert S.lib.starkhip_multipool_wait(None, 1, None, None, None) == S.ERR_NO_DEVICE
    assert S.lib.starkhip_multipool_size(None) == 0 and S.lib.starkhip_multipool_ticket_slot(None, 1 << 48) == -1
    assert S.api.hw_queues_late() is False   # nothing in this process has initialised HIP behind the library's back
    cpu = S.api.host_cpu_seconds()
    assert set(cpu) == {"recording", "proving", "of_proving_in_device_waits"} and all(v >= 0 for v in cpu.values())
