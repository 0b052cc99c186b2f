#!/usr/bin/env python3
"""Benchmark: starky proofs/s for FinalExponentiateStark (73527 columns x 8192 rows) on N MI355X.

A "step" is one full prove() of one FinalExp trace whose column-major u64[C][n] values are already
resident in HBM (BASELINE.json configs[2]).  Each rank proves its own independent proof (the six proofs
of a signature verification shard at proof granularity): weak scaling, no data-path
collective; torch.distributed is used only for the barrier and the max-over-ranks time.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_GINSTR = 256 * 4 * 2.4 / 4  # wave-instructions/ns: one 4-cycle integer VALU instruction per SIMD (measured, tools/valu_rate_bench.hip)
POSEIDON_QUAD_INSTRS = 8 * 261 + 7 * 309 + 152  # VALU instructions of one permutation in the 4-lane form (ISA of poseidon_dev.h: 8 full rounds, 7 merged triples of partial rounds, 1 single partial round)


def synthetic_final_exp_input(seed):
    """12 Fp coordinates below p from splitmix64(seed); any invertible Fp12 is provable."""
    from bls_util import random_fp12
    return random_fp12(seed)


def cpu_baseline_sample(S, blob, n_cols, log_n, rate_bits, budget_cols=1024, budget_points=256):
    """Time the CPU oracle on a bounded slice of the same workload and scale to one whole proof.

    LDE + Merkle leaf hashing run on `budget_cols` of the C columns (cost linear in C); the constraint
    evaluation runs on `budget_points` of the N coset points with all C columns (cost linear in points).
    Openings / FRI are < 5 % of the CPU time and are left out, which flatters the CPU."""
    import numpy as np
    import oracle_lib as O
    rng = np.random.default_rng(1)
    n = 1 << log_n
    N = n << rate_bits
    cols = rng.integers(0, S.P, size=(budget_cols, n), dtype=np.uint64)
    t0 = time.time()
    _, lde_rows = O.lde_rows(cols, rate_bits)
    t_lde = time.time() - t0
    t0 = time.time()
    O.merkle_cap(lde_rows, 4)
    t_hash = time.time() - t0
    rows = rng.integers(0, S.P, size=(budget_points + 1, n_cols), dtype=np.uint64)
    pis = np.zeros(S.air_public_inputs(S.AIR_FINAL_EXP), dtype=np.uint64)
    t0 = time.time()
    O.bench_quotient(blob, rows, pis)  # every constraint, both alphas, at budget_points points
    t_q = time.time() - t0
    scale_c = n_cols / budget_cols
    total = t_lde * scale_c + t_hash * scale_c + t_q * (N / budget_points)
    return {
        "value": 1.0 / total, "unit": "proofs/s", "cores": int(O.lib.oracle_num_threads()), "kind": "port",
        "sample": (f"CPU oracle (OpenMP C restatement, not the reference's Rust): LDE+leaf-hash on {budget_cols}/{n_cols} columns x {n} rows, "
                   f"constraint evaluation on {budget_points}/{N} coset points; scaled linearly to one proof "
                   f"(lde {t_lde * scale_c:.1f}s + hash {t_hash * scale_c:.1f}s + quotient {t_q * N / budget_points:.1f}s); openings/FRI omitted"),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=4,
                    help="proofs in flight per GPU (independent contexts on separate host threads and HIP streams); "
                         "1 = one proof at a time (latency); the default hides the host-side Fiat-Shamir hashing and the launch gaps of "
                         "each proof behind the kernels of the others (measured: 1: 3.7, 2: 4.5, 3: 5.0, 4: 5.1 proofs/s)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import starky_bls12_381_amd as S

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # STARKHIP_BENCH_REHEARSE=1: every rank on cuda:0 with the gloo backend -- a rehearsal of the N > 1 control flow on a
    # one-GPU box (the numbers it prints are not a measurement; RCCL refuses two ranks on one device)
    rehearse = os.environ.get("STARKHIP_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    from starky_bls12_381_amd import parallel
    dist = parallel.init_distributed("gloo" if rehearse else "nccl") if world > 1 else None
    reduce_device = "cpu" if rehearse else f"cuda:{local_rank}"

    air = S.AIR_FINAL_EXP
    cfg = S.StarkConfig.for_air(air)
    C, n = S.air_columns(air), S.air_default_rows(air)
    log_n = n.bit_length() - 1
    N = n << cfg.rate_bits

    # synthetic input, different per rank; trace generated on the host (the reference's generate_trace side),
    # moved to HBM as column-major u64 (as int64 bit patterns) before the timed region
    x = synthetic_final_exp_input(0x5EED0000 + 1 + rank)
    trace, pis = S.trace_final_exp(x)
    d_rows = torch.from_numpy(trace.view(np.int64)).to(f"cuda:{local_rank}")
    del trace
    d_cols = d_rows.t().contiguous()  # trace_rows_to_poly_values
    del d_rows
    torch.cuda.synchronize()
    import threading
    inflight = max(1, args.inflight)
    provers = [S.Prover(local_rank) for _ in range(inflight)]
    prover = provers[0]

    def step(pv, keep=False):
        return pv.prove_device(air, cfg, d_cols.data_ptr(), n, pis, layout=1, keep=keep)

    proof = None
    for w in range(args.warmup):
        for i, pv in enumerate(provers):
            pr = step(pv, keep=(w == 0 and i == 0))
            proof = pr if pr is not None else proof
    if proof is not None and rank == 0:
        S.verify_stark_proof(air, cfg, proof)  # untimed: the product's CPU verifier accepts what we time
    phase_ms = {k: 0.0 for k in S.PHASE_NAMES}
    kern_ms = {"lde_columns": 0.0, "leaf_hash": 0.0, "quotient_eval": 0.0}
    lock = threading.Lock()
    todo = list(range(args.steps))

    def worker(pv):
        while True:
            with lock:
                if not todo:
                    return
                todo.pop()
            step(pv)
            tm, km = pv.last_timings(), pv.last_kernel_timings()
            with lock:
                for k, v in tm.items():
                    phase_ms[k] += v
                for k, v in km.items():
                    kern_ms[k] += v

    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    if inflight == 1:
        worker(prover)
    else:
        threads = [threading.Thread(target=worker, args=(pv,)) for pv in provers]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = parallel.max_over_ranks(dist, elapsed, device=reduce_device)

    # untimed: the same kernels with the GPU to themselves (one proof in flight), for the uncontended roofline numbers
    solo_ms = {"lde_columns": 0.0, "leaf_hash": 0.0, "quotient_eval": 0.0}
    solo_phase = {k: 0.0 for k in S.PHASE_NAMES}
    n_solo = 2 if (rank == 0 and inflight > 1) else 0
    for _ in range(n_solo):
        step(prover)
        for k, v in prover.last_kernel_timings().items():
            solo_ms[k] += v / n_solo
        for k, v in prover.last_timings().items():
            solo_phase[k] += v / n_solo

    if rank == 0:
        steps = max(1, args.steps)
        phase_ms = {k: v / steps for k, v in phase_ms.items()}
        kern_ms = {k: v / steps for k, v in kern_ms.items()}
        # algorithmic bytes per launch: u64 cells, dense, minimum traffic of the decomposition
        alg = {"lde_columns": 8.0 * C * (n + n + N),  # read values, write coeffs + LDE (IFFT and LDE fused in one kernel)
               "leaf_hash": 8.0 * C * N,              # read the LDE once
               "quotient_eval": 8.0 * C * N}          # read the LDE on the quotient coset once
        dominant = max(kern_ms, key=kern_ms.get)
        # HBM-side bytes per launch from the committed PMC passes (bench.py cannot collect counters itself)
        traffic, traffic_src = None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")))
            key = {"lde_columns": "lde_columns_v2_kernel", "leaf_hash": "leaf_hash_kernel", "quotient_eval": "quotient_eval_kernel"}[dominant]
            traffic, traffic_src = pmc[key]["traffic_bytes"], pmc["_source"]
        except Exception:
            pass
        gbs = {k: alg[k] / (kern_ms[k] * 1e-3) / 1e9 if kern_ms[k] > 0 else 0.0 for k in alg}
        out = {
            "metric": "starky proofs/sec (FinalExponentiateStark 73527x8192)",
            "value": world * args.steps / elapsed,
            "unit": "proofs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 (Goldilocks field)",
            "data": "synthetic" + (" -- REHEARSAL: all ranks on one GPU over gloo, not a measurement" if rehearse else ""),
            "config": {"workload": "FinalExponentiateStark 73527 cols x 8192 rows, rate_bits 2, 360800 constraints, "
                                   "standard_fast_config (84 queries, 16 pow bits); one independent proof per GPU",
                       "parallelism": f"proof-parallel x{world}", "proofs_in_flight_per_gpu": inflight},
            "roofline": {"bound": "hbm", "kernel": dominant + "_kernel", "achieved": gbs[dominant], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs[dominant] / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg[dominant], "avg_launch_ms": kern_ms[dominant]},
            "kernels": {k: {"avg_ms": kern_ms[k], "algorithmic_GBps": gbs[k], "hbm_frac": gbs[k] / HBM_PEAK_GBS} for k in alg},
            "phase_ms": phase_ms,
            "note": ("kernel and phase times above are HIP-event durations inside the timed region; with more than one proof in flight "
                     "they include the time a kernel shares the CUs with the other proof's kernels. 'solo' repeats them with one proof in flight."),
            "reference_published": {"value": 1 / 92.0, "unit": "proofs/s", "hardware": "AWS r6a.8xlarge, 32-core EPYC 7R13 (reference README.md:39)"},
        }
        if n_solo:
            out["solo"] = {"kernels": {k: {"avg_ms": solo_ms[k], "algorithmic_GBps": alg[k] / (solo_ms[k] * 1e-3) / 1e9,
                                           "hbm_frac": alg[k] / (solo_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS} for k in alg if solo_ms[k] > 0},
                           "phase_ms": solo_phase}
        # the leaf hash is VALU-issue bound, not HBM bound: static instruction count of one quad permutation x permutations / 16 quads per wave
        perms = (C + 7) // 8 * N
        wave_instr = perms * POSEIDON_QUAD_INSTRS / 16.0
        lh_ms = (solo_ms["leaf_hash"] if n_solo else kern_ms["leaf_hash"])
        if lh_ms > 0:
            out["leaf_hash_valu"] = {"wave_instructions": wave_instr, "achieved_Ginstr_per_s": wave_instr / (lh_ms * 1e-3) / 1e9,
                                     "peak_Ginstr_per_s": VALU_PEAK_GINSTR, "frac": wave_instr / (lh_ms * 1e-3) / 1e9 / VALU_PEAK_GINSTR,
                                     "basis": "8 full rounds x 261 + 7 merged triples of partial rounds x 309 + 1 partial round x 152 VALU instructions per 4-lane permutation (ISA count); "
                                              "peak = 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per integer VALU instruction (tools/valu_rate_bench.hip)"}
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only: other ranks would sit in the teardown barrier meanwhile
            try:
                out["cpu_baseline"] = cpu_baseline_sample(S, S.air_program(air), C, log_n, cfg.rate_bits)
            except Exception as e:  # the oracle is a checker, never a dependency of the measured path
                out["cpu_baseline"] = {"value": None, "unit": "proofs/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    for pv in provers:
        pv.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
