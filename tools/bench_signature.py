#!/usr/bin/env python3
"""Signatures/s END TO END on N MI355X: operands in -> (trace generation + the six STARK proofs per signature) -> proofs out.

BASELINE.json configs[3] (six proofs of one signature sharded over GPUs) and configs[4] (a batch of 8 signatures = 48 proofs):

    python tools/bench_signature.py --batch 8                                   # one GPU
    python tools/bench_signature.py --gpus N --batch 8                          # starts itself under torch.distributed.run (a child process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/bench_signature.py --gpus N --batch 8                             # one process per GPU, RCCL

Per step: rank 0 owns the B (different, valid, synthetic) signatures and broadcasts their operands with one collective;
every rank derives the same longest-first plan over the 6 B jobs, computes the natives its jobs need, records the traces of
its jobs on host threads (compact runs, expanded on the device) and proves them on `--inflight` contexts -- trace generation
is inside the timed region and overlaps proving.  The timed region is bracketed by barrier + synchronize; the time is the
max over ranks.  Afterwards (untimed) every proof is verified by the product's CPU verifier and, for the signatures whose six
proofs are on this rank (always at N = 1; with --collect at N > 1), the links and the statement are checked.

Prints one JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

# one hardware queue per in-flight proof: with the HIP default of 4, more streams share queues and their kernels serialise
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1, help="signatures per step (6 proofs each)")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--inflight", type=int, default=6, help="prover contexts per GPU when --small-inflight is 0 (one pool)")
    ap.add_argument("--big-inflight", type=int, default=6, help="contexts for FinalExp proofs (two pools); five or more: lane-form commitment groups")
    ap.add_argument("--priority", type=int, default=1, help="pool: 1 = FinalExp-class contexts on high-priority streams, 0 = all alike")
    ap.add_argument("--small-inflight", type=int, default=0, help="contexts for the 1024-row AIRs; 0 = one pool of --inflight contexts; "
                                                                   "default for --batch > 1: 16")
    ap.add_argument("--gen-threads", type=int, default=0, help="recordings under way at once per GPU (0 = the pool's default: a quarter of the CPU budget; "
                                                               "the python driver uses 12 when 0)")
    ap.add_argument("--trace-threads", type=int, default=0, help="host threads one recording generator call may use; 0 = automatic "
                                                                  "(2 x gen-threads / jobs on the rank, at most 8)")
    ap.add_argument("--by-type", type=int, default=0, help="1: the FinalExp contexts start when the small proofs are done (two pools); "
                                                           "0 (default): both pools at once -- measured for a batch of 8: 2.61 against "
                                                           "2.26 signatures/s by type")
    ap.add_argument("--driver", choices=("pool", "python"), default="pool",
                    help="pool (default): the library's proof pool (starkhip_pool_*: generator threads, contexts, merged commitments inside "
                         "libstarkhip.so); python: the round-2 driver (signature.run_jobs: Python threads over plain contexts)")
    ap.add_argument("--policy", type=int, default=0, help="pool: commit policy (0: merged small commitments, 1: also keep the two classes apart, 2: no scheduling)")
    ap.add_argument("--gather-ms", type=float, default=0.0, help="pool: how long a merged commitment waits for stragglers (0 = default)")
    ap.add_argument("--collect", action="store_true", help="N > 1: gather every proof on every rank afterwards (raw buffers) and check all signatures")
    ap.add_argument("--no-verify", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python tools/bench_signature.py --gpus N` by itself: one child `python -m torch.distributed.run ...` (never an exec, and
        # before torch or the GPU is touched in this process); its output is relayed, its exit code returned
        from bench import self_launch
        raise SystemExit(self_launch(sys.argv[1:], args.gpus, script=__file__))

    import torch
    import starky_bls12_381_amd as S
    from starky_bls12_381_amd import aggregate as A
    from starky_bls12_381_amd import parallel
    from starky_bls12_381_amd import signature as G
    from bls_util import native_vectors

    G.tune_host_allocator()
    rank, local_rank, world = parallel.rank_info()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    rehearse = os.environ.get("STARKHIP_BENCH_REHEARSE") == "1"  # all ranks on cuda:0 over gloo: control-flow rehearsal, not a measurement
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)  # the launcher exec'd us before anything touched the GPU
    dist = parallel.init_distributed("gloo" if rehearse else "nccl") if world > 1 else None
    dev = "cpu" if rehearse else f"cuda:{local_rank}"

    if args.small_inflight == 0 and args.batch > 1:
        args.small_inflight = 12
    if args.driver == "pool":
        big = max(1, args.big_inflight if args.batch > 1 else 1)
        small = args.small_inflight if args.small_inflight > 0 else 5
        provers = S.ProofPool(local_rank, big_contexts=big, small_contexts=small, generator_threads=args.gen_threads,
                              trace_threads=args.trace_threads, commit_policy=args.policy, gather_ms=args.gather_ms, stream_priority=args.priority, warm_up=1)
        all_provers = [provers]
    elif args.small_inflight > 0:
        provers = {"big": [S.Prover(local_rank) for _ in range(max(1, args.big_inflight))],
                   "small": [S.Prover(local_rank) for _ in range(args.small_inflight)]}
        all_provers = provers["big"] + provers["small"]
    else:
        provers = [S.Prover(local_rank) for _ in range(max(1, args.inflight))]
        all_provers = provers
    plan = G.plan_batch(args.batch, world)
    mine = plan[rank]
    results, stats = {}, {}

    def one_step(seed):
        signatures = G.synthetic_signatures(args.batch, native_vectors()["bls_signature"], seed) if rank == 0 else None
        return G.one_step(dist, args.batch, provers, mine, signatures, device=dev, sync=torch.cuda.synchronize, gen_threads=args.gen_threads or 12,
                          trace_threads=args.trace_threads or None, big_after_small=bool(args.by_type))

    for w in range(args.warmup):
        one_step(0x1000 + w)
    total = 0.0
    for k in range(args.steps):
        el, results, stats, sigs, natives = one_step(0x2000 + k)
        total += el
    el = total / max(1, args.steps)

    # ---- untimed: what the reference does right after each prove (verify_stark_proof) and what its recursion enforces
    verified = 0
    if not args.no_verify:
        for (_, name), (air, proof, cfg) in results.items():
            S.verify_stark_proof(air, cfg, proof)
            verified += 1
    merged = G.collect_results(dist, results, device=dev) if (dist is not None and args.collect) else results
    verdicts = G.check_signatures(merged, sigs, natives, args.batch)
    checked, valid = len(verdicts), sum(verdicts.values())
    n_verified = int(parallel.sum_over_ranks(dist, verified, device=dev))
    if rank == 0:
        per_air = {}
        for (_, name), (air, proof, _) in results.items():
            per_air.setdefault(S.AIR_NAMES[air], []).append(int(proof.size) * 8)
        out = {
            "metric": "BLS signature checks/s, end to end (operands -> trace generation -> 6 STARK proofs each)",
            "value": args.batch / el, "unit": "signatures/s", "n_gpus": world, "batch": args.batch, "steps": args.steps,
            "ms_per_step": el * 1e3, "ms_per_signature": el * 1e3 / args.batch,
            "proofs_per_step": 6 * args.batch, "driver": args.driver, "contexts_per_gpu": ({"big": big, "small": small} if args.driver == "pool" else {k: len(v) for k, v in provers.items()} if isinstance(provers, dict) else len(provers)), "pool_commit_stats": stats.get("pool"), "generator_threads_per_gpu": args.gen_threads, "threads_per_generator_call": stats.get("trace_threads"), "final_exp_after_small_proofs": bool(args.by_type),
            "rank0": {"jobs": len(mine), "generate_s_sum": stats.get("generate_s"), "prove_s_sum": stats.get("prove_s"), "wall_s": stats.get("wall_s")},
            "timeline_ms_rank0": ({f"{i}:{n}": [round(1e3 * t, 1) for t in v] for (i, n), v in sorted(stats.get("timeline", {}).items())}
                                  if args.batch == 1 else None),  # per job: generation start, end, proof start, end
            "proofs_verified_after_timing": n_verified, "signatures_checked_on_rank0": checked, "signatures_valid_on_rank0": valid,
            "proof_bytes_on_rank0": {k: v[0] for k, v in per_air.items()},
            "data": "synthetic valid signatures derived from the reference vector src/native.rs:1480-1498 (a different set each step)"
                    + (" -- REHEARSAL over gloo on one GPU, not a measurement" if rehearse else ""),
            "plan": {str(r): [f"{i}:{n}" for i, n in jobs] for r, jobs in enumerate(plan)} if args.batch <= 2 else f"{6 * args.batch} jobs, longest first over {world} rank(s)",
        }
        print(json.dumps(out), flush=True)
    for pv in all_provers:
        pv.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
