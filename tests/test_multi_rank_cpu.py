"""N > 1 path on CPU: two gloo ranks run the job assignment, the public-input broadcast, the barrier/max-reduce that
bench.py uses around the timed region, and (host side only) trace generation for their share of the proofs."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import starky_bls12_381_amd as S
    from starky_bls12_381_amd import parallel as P
    from bls_util import random_fp12
    dist = P.init_distributed("gloo")
    assert dist is not None and dist.get_world_size() == world
    # the six proofs of one signature verification: 2 x precomp, 2 x miller, fp12_mul, final_exp
    airs = [S.AIR_PAIRING_PRECOMP, S.AIR_MILLER_LOOP, S.AIR_PAIRING_PRECOMP, S.AIR_MILLER_LOOP, S.AIR_FP12_MUL, S.AIR_FINAL_EXP]
    plan = P.assign_jobs([P.AIR_COST[a] for a in airs], world)
    mine = plan[rank]
    # rank 0 owns the input; everybody needs the FP12Mul public inputs of job 4
    x, y = random_fp12(0x5EED2000), random_fp12(0x5EED2001)
    if rank == 0:
        _, pis = S.trace_fp12_mul(x, y)
    else:
        pis = np.zeros(S.air_public_inputs(S.AIR_FP12_MUL), dtype=np.uint64)
    pis = P.broadcast_u64(dist, pis, src=0)
    digest = int(np.bitwise_xor.reduce(pis))
    dist.barrier()
    elapsed = 1.0 + rank  # pretend rank 1 is slower
    slowest = P.max_over_ranks(dist, elapsed)
    done = P.sum_over_ranks(dist, len(mine))
    q.put((rank, mine, digest, slowest, done))
    dist.barrier()
    dist.destroy_process_group()


def test_two_gloo_ranks_share_six_proofs():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    jobs = sorted(j for _, mine, _, _, _ in results for j in mine)
    assert jobs == list(range(6))                       # every proof exactly once
    assert [5] == results[0][1]                          # LPT: FinalExp alone on one rank, everything else on the other
    assert results[0][2] == results[1][2] != 0           # broadcast public inputs identical on both ranks
    assert all(r[3] == 2.0 for r in results)             # max over ranks
    assert all(r[4] == 6.0 for r in results)


def test_lpt_assignment_for_a_batch_of_eight_signatures():
    from starky_bls12_381_amd import parallel as P
    airs = [1, 2, 1, 2, 0, 3] * 8
    plan = P.assign_jobs([P.AIR_COST[a] for a in airs], 8)
    assert sorted(j for r in plan for j in r) == list(range(48))
    loads = [sum(P.AIR_COST[airs[j]] for j in r) for r in plan]
    assert max(loads) - min(loads) < 25.0                # within one MillerLoop of each other
    assert all(sum(1 for j in r if airs[j] == 3) == 1 for r in plan)  # one FinalExp per GPU


def _collect_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import starky_bls12_381_amd as S
    from starky_bls12_381_amd import aggregate as A
    from starky_bls12_381_amd import parallel as P
    dist = P.init_distributed("gloo")
    # stand-in blobs (no GPU here): what matters is who produced which proof and that the public-input tails survive
    mine = {}
    for name in A.signature_plan(world)[rank]:
        air = A.JOB_AIR[name]
        n_pis = S.air_public_inputs(air)
        blob = np.arange(n_pis + 16, dtype=np.uint64) * np.uint64(1 + A.JOB_ORDER.index(name))
        mine[name] = (air, blob, S.StarkConfig.for_air(air))
    merged = A.collect_proofs(dist, mine)
    summary = {name: (int(air), int(np.bitwise_xor.reduce(proof)), int(cfg.rate_bits)) for name, (air, proof, cfg) in merged.items()}
    q.put((rank, sorted(mine), summary))
    dist.barrier()
    dist.destroy_process_group()


def test_two_gloo_ranks_collect_all_six_proofs_everywhere():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_collect_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from starky_bls12_381_amd import aggregate as A
    assert sorted(results[0][1] + results[1][1]) == sorted(A.JOB_ORDER)   # shares are disjoint and cover the plan
    assert results[0][2] == results[1][2]                                  # both ranks end up with the same six proofs
    assert sorted(results[0][2]) == sorted(A.JOB_ORDER)
    assert results[0][2]["final_exp"][2] == 2 and results[0][2]["fp12_mul"][2] == 1   # configs rebuilt per AIR


def test_collect_rejects_a_proof_produced_twice():
    import pytest
    from starky_bls12_381_amd import aggregate as A

    class TwoRanksBothRanFp12Mul:
        def get_world_size(self):
            return 2

        def get_rank(self):
            return 0

        def broadcast(self, buf, src=0):
            pass

        def all_gather_object(self, out, obj):
            out[0] = obj
            out[1] = obj
    with pytest.raises(ValueError):
        A.collect_proofs(TwoRanksBothRanFp12Mul(), {"fp12_mul": (A.JOB_AIR["fp12_mul"], np.zeros(4, dtype=np.uint64), None)})


# ---- the end-to-end signature driver (starky_bls12_381_amd/signature.py, tools/bench_signature.py) over gloo ranks
def _fake_prove(pv, air, cfg, trace, pis):
    """Stand-in for Prover.prove on a box without a GPU: a blob that ends in the public inputs, as every proof does."""
    return np.concatenate([np.full(5, 7 + air, dtype=np.uint64), np.asarray(pis, dtype=np.uint64)])


def _signature_worker(rank, world, port, batch, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from starky_bls12_381_amd import aggregate as A
    from starky_bls12_381_amd import parallel as P
    from starky_bls12_381_amd import signature as G
    from bls_util import native_vectors
    dist = P.init_distributed("gloo")
    # the step tools/bench_signature.py times (signature.one_step), with an injected prove
    signatures = G.synthetic_signatures(batch, native_vectors()["bls_signature"], 0x2000) if rank == 0 else None
    mine = G.plan_batch(batch, world)[rank]
    elapsed, results, stats, sigs, natives = G.one_step(dist, batch, [object(), object()], mine, signatures, gen_threads=3, prove=_fake_prove)
    slowest = elapsed + 1e-9
    merged = G.collect_results(dist, results)
    checked = G.check_signatures(merged, sigs, natives, batch)
    verdicts = [checked.get(i, False) for i in range(batch)]
    digest = int(np.bitwise_xor.reduce(G.pack_operands(sigs).reshape(-1)))
    q.put((rank, sorted(mine), verdicts, digest, slowest >= stats["wall_s"]))
    dist.barrier()
    dist.destroy_process_group()


def _run_signature_ranks(world, batch):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_signature_worker, args=(r, world, port, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return results


def test_signature_driver_on_two_gloo_ranks():
    res = _run_signature_ranks(2, 1)
    assert res[0][1] == [(0, "final_exp")]                                    # FinalExp alone on one rank
    assert sorted(res[0][1] + res[1][1]) == sorted((0, n) for n in ("pp1", "ml1", "pp2", "ml2", "fp12_mul", "final_exp"))
    assert res[0][3] == res[1][3] != 0                                          # the operand broadcast reached both ranks
    assert all(r[2] == [True] and r[4] for r in res)                            # links + statement hold on every rank after collection


def test_signature_driver_on_six_gloo_ranks_one_proof_per_rank():
    """BASELINE configs[3]: the six proofs of one signature check, one per rank."""
    res = _run_signature_ranks(6, 1)
    assert sorted(len(r[1]) for r in res) == [1] * 6
    assert len({r[3] for r in res}) == 1
    assert all(r[2] == [True] for r in res)
