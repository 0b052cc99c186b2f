#!/usr/bin/env python3
"""Benchmark: starky proofs/s for FinalExponentiateStark (73527 columns x 8192 rows) on N MI355X.

A "step" is one full proof of one FinalExp statement AT THE REFERENCE'S OWN BOUNDARY: the timed region starts from the
driver's operand (one Fp12, 144 u32 limbs on the host) and ends with the proof blob in host memory -- generate_trace, the
upload, trace_rows_to_poly_values, prove() and the read-back are all inside, exactly what
/root/reference/src/aggregate_proof.rs:158-176 brackets with `Instant::now()` (proofs/s including H2D and
D2H).  The proofs go through the library's own scheduler (`starkhip_pool_submit_witness` / `starkhip_pool_wait`): `--inflight`
FinalExp-class contexts per GPU, generator threads that record the traces, lane-form commitment groups.  Each rank proves its
own independent proofs (the six proofs of a signature verification shard at proof granularity): weak
scaling, no data-path collective; torch.distributed is used only for the barrier and the max-over-ranks time.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts itself under torch.distributed.run as a CHILD
process (before torch or the GPU is touched) and relays its output; launched by torchrun it just runs as a rank.
`python bench.py --devices-in-process N` measures the same metric the way the reference's own caller would use a node: ONE process,
a pool per device behind `starkhip_multipool_*` (no process group, no collective), N x steps proofs placed by the library; it prints
the same JSON line (`config.parallelism` says which form ran).

Prints ONE JSON line on rank 0.  What is measured where (DESIGN.md section 6 has every field):

  value / ms_per_step     the timed region: operands -> proof bytes, `--inflight` contexts per GPU
  roofline                the kernel with the most device time INSIDE the timed region (with five or more in flight:
                          leaf_hash_lane_kernel in groups), HIP events on the stream it is launched on, taken in the timed region
  kernels                 rank 0, untimed, ONE proof in flight: uncontended HIP-event durations of the three heavy kernels (the
                          figures the committed rocprof summaries under profiles/ must agree with); PMC traffic per launch from
                          profiles/pmc_traffic_latest.json
  value_host_rows         rank 0, untimed leg on the SAME pool: page-locked host ROWS -> H2D -> transpose -> proof -> D2H
                          (BASELINE.md section 4's literal hand-over); value_compact: a recorded trace; value_device_resident: the
                          column-major trace already in HBM (rounds 1-3's headline)
  cpu_baseline            rank 0 at N = 1: the CPU oracle on a bounded sample of the same workload, measured in this run
  value_steady_state      proofs/s over the middle of the timed region only: completions after the first and before the last `inflight`
                          per pool (the pool's start and tail dropped); null when the run is shorter than three waves
  host                    what the host side had: CPUs granted, the pool's CPU budget and threads, process CPU-seconds per proof in the
                          timed region -- so that a scaling curve that bends can be attributed to host or device
  per_rank                --gpus N > 1: every rank's own proofs/s, CPU budget, device ordinal and PCI address (value stays all ranks' proofs /
                          the slowest rank's time); process_group / rccl_ranks: the backend and the ranks torch.distributed counts;
                          devices_distinct: N ranks on N devices (false in a rehearsal on one card)
"""
import argparse
import json
import os
import subprocess
import sys
import time

# One hardware queue per in-flight proof and commitment stream.  The HIP runtime reads this once, when it is first used in the process:
# starkhip_pool_create sets it for a process that has not touched HIP yet, but here torch does first (torch.cuda.set_device), and with
# HIP's default of 4 queues the pool's streams share queues and kernels of different proofs wait behind each other (measured on one
# box, 48 proofs: 6.61 / 6.76 proofs/s without it, 6.97 / 7.12 with it; 24 queues: 6.27 -- profiles/r04_ab_experiments.txt 17).
# A host that embeds the pool beside other HIP users sets it the same way.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# wave-instructions per ns: one integer VALU instruction per SIMD every 4 cycles (measured: tools/valu_rate_bench.hip,
# profiles/r02_valu_rates.txt); 256 CUs x 4 SIMDs x 2.4 GHz / 4
VALU_PEAK_GINSTR = 256 * 4 * 2.4 / 4
# VALU instructions of one permutation in the 4-lane form (ISA of poseidon_dev.h: 8 full rounds, 7 merged triples of
# partial rounds, 1 single partial round)
POSEIDON_QUAD_INSTRS = 7 * 261 + 204 + 7 * 309 + 152


def poseidon_pair_slots():
    """issue slots per wave and 32 permutations of the pair form (two lanes per permutation), as tools/gen_pair_round_asm.py writes them
    into the header of csrc/pair_round_asm.inc"""
    import re
    try:
        text = open(os.path.join(ROOT, "starky_bls12_381_amd", "csrc", "pair_round_asm.inc")).read(2000)
        return int(re.search(r"= (\d+) issue slots", text).group(1))
    except (OSError, AttributeError):
        return None


def poseidon_lane_slots():
    """issue slots of one permutation in the lane form (one permutation per lane), from the block sizes tools/gen_lane_round_asm.py writes
    into csrc/lane_round_asm.inc: 7 full rounds with the circulant layer on the matrix pipe, the capacity-only last round, 5 merges of four
    partial rounds, the two plain partial rounds"""
    import re
    sizes = {}
    for line in open(os.path.join(ROOT, "starky_bls12_381_amd", "csrc", "lane_round_asm.inc")):
        m = re.match(r"// (.*?): (\d+) instructions", line)
        if m:
            sizes[m.group(1)] = int(m.group(2))
    try:
        return (7 * sizes["full round, circulant layer on the matrix pipe"] + sizes["last full round before an absorb: the capacity outputs only"]
                + 5 * sizes["four partial rounds at once"] + 2 * sizes["partial round, circulant layer on the matrix pipe"])
    except KeyError:
        return None

KERNELS = ("lde_columns", "leaf_hash", "quotient_eval")
PMC_NAMES = {"lde_columns": ("lde_columns_wave_kernel", "lde_columns_v2_kernel"), "leaf_hash": ("leaf_hash_pair_kernel", "leaf_hash_kernel"), "leaf_hash_lane": ("leaf_hash_lane_kernel",),
             "quotient_eval": ("quotient_tiles_kernel", "quotient_eval_kernel")}
FORM_KERNEL = {"quad": "leaf_hash_kernel", "lane": "leaf_hash_lane_kernel", "row": "leaf_hash_row_kernel", "merged": "leaf_hash_multi_kernel",
               "host": "host threads", "pair": "leaf_hash_pair_kernel"}


def self_launch(argv, n_gpus, script=None):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start `python -m torch.distributed.run ... bench.py <args>` as a child,
    relay its stdout line by line, return its exit code.  Nothing in THIS process has imported torch or touched the GPU.
    (tools/bench_signature.py starts itself the same way, with `script` = its own path.)"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(script or __file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, bufsize=1)
    try:
        for line in child.stdout:
            sys.stdout.write(line)
            sys.stdout.flush()
    finally:
        rc = child.wait()
    return rc


def cpu_quota():
    """CPUs the process may use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, round(int(q) / int(p))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, round(q / p)))
        except (OSError, ValueError):
            pass
    return n


def synthetic_final_exp_input(seed):
    """12 Fp coordinates below p from splitmix64(seed); any invertible Fp12 is provable."""
    from bls_util import random_fp12
    return random_fp12(seed)


def cpu_baseline_sample(S, blob, n_cols, log_n, rate_bits, budget_cols=4096, budget_points=1024):
    """Time the CPU oracle on a bounded slice of the same workload IN THIS RUN and scale to one whole proof.

    LDE + Merkle leaf hashing run on `budget_cols` of the C columns (cost linear in C); the constraint
    evaluation runs on `budget_points` of the N coset points with all C columns (cost linear in points).
    Openings / FRI are < 5 % of the CPU time and are left out, which flatters the CPU."""
    import numpy as np
    # the oracle's OpenMP team = the CPUs this process may really use (a GPU box shows 256 hardware threads, its cgroup grants 16:
    # more threads than the quota are throttled, not faster -- profiles/r03_cpu_share.txt), so that `cores` is what was used
    import oracle_lib as O
    if "OMP_NUM_THREADS" not in os.environ:
        O.lib.oracle_set_threads(cpu_quota())  # (the OpenMP runtime is already loaded -- by torch -- so the environment is not read again)
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
    del lde_rows
    rows = rng.integers(0, S.P, size=(budget_points + 1, n_cols), dtype=np.uint64)
    pis = np.zeros(S.air_public_inputs(S.AIR_FINAL_EXP), dtype=np.uint64)
    t0 = time.time()
    O.bench_quotient(blob, rows, pis)  # every constraint, both alphas, at budget_points points
    t_q = time.time() - t0
    scale_c = n_cols / budget_cols
    total = t_lde * scale_c + t_hash * scale_c + t_q * (N / budget_points)
    out = {"value": 1.0 / total, "unit": "proofs/s", "cores": int(O.lib.oracle_num_threads()), "kind": "port",
           "sample": (f"measured in this run: CPU oracle (OpenMP C restatement, not the reference's Rust), {t_lde + t_hash + t_q:.1f} s of work: LDE + leaf hash on "
                      f"{budget_cols}/{n_cols} columns x {n} rows, constraint evaluation on {budget_points}/{N} coset points; scaled linearly "
                      f"to one proof (lde {t_lde * scale_c:.1f} s + hash {t_hash * scale_c:.1f} s + quotient {t_q * N / budget_points:.1f} s = {total:.1f} s; "
                      f"openings / FRI omitted); `cores` = the OpenMP threads used = the CPUs the box's cgroup grants")}
    # The oracle has also proven a WHOLE FinalExp trace once on a GPU box's host (tests/make_final_exp_golden.py): a constant from
    # another run, reported apart from this run's measurement.
    for name in ("r03_oracle_full_final_exp.txt", "r01_oracle_full_final_exp.txt"):
        try:
            line = open(os.path.join(ROOT, "profiles", name)).readline()
            full_s = float(line.split("prove:")[1].split("s")[0])
            out["recorded_full_proof"] = {"value": 1.0 / full_s, "unit": "proofs/s", "seconds": full_s, "threads": int(line.split(" on ")[1].split()[0]),
                                          "source": "profiles/" + name + " (an earlier run on another box; not measured now): " + line.strip()}
            break
        except (OSError, IndexError, ValueError):
            continue
    return out


def thread_cpu_by_name():
    """CPU seconds of this process's LIVE threads by thread name (/proc/self/task/*/schedstat): which threads the host time of the timed
    region went to -- the pool's own (starkhip-gen / -rec / -ctx / -hash), this script's, or the HIP runtime's.  Threads that have exited
    (the helpers of a recording) are missing; their time is in `recording`."""
    out = {}
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                name = open(f"/proc/self/task/{tid}/comm").read().strip()
                if int(tid) == os.getpid():
                    name = "main thread (this script)"
                elif not name.startswith("starkhip"):
                    name += f" (unnamed: HIP runtime, torch) tid {tid}" if os.environ.get("STARKHIP_BENCH_THREAD_IDS") else " (unnamed: HIP runtime, torch)"
                ns = int(open(f"/proc/self/task/{tid}/schedstat").read().split()[0])
            except (OSError, ValueError, IndexError):
                continue
            out[name] = out.get(name, 0.0) + ns * 1e-9
    except OSError:
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-boundary", action="store_true", help="skip the untimed legs (host rows, recorded trace, device-resident trace)")
    ap.add_argument("--no-solo", action="store_true", help="skip the untimed one-proof-in-flight pass (`kernels`, latency); for profiling runs")
    ap.add_argument("--input", choices=("witness", "device"), default="witness",
                    help="what the TIMED region starts from: witness = the driver's operand on the host (generate_trace, upload and read-back inside; "
                         "the reference's boundary, the default); device = a column-major trace already in HBM (rounds 1-3's headline, for A/B runs)")
    ap.add_argument("--devices-in-process", type=int, default=0,
                    help="N > 0: ONE process drives N devices through starkhip_multipool_* (a pool per device, jobs placed by the library), the form the "
                         "reference's single-process caller would use; N x steps proofs; no torch.distributed.  With STARKHIP_BENCH_REHEARSE=1 the N "
                         "pools share device 0 (a rehearsal on a one-GPU box, not a measurement)")
    ap.add_argument("--inflight", type=int, default=8,
                    help="proofs in flight per GPU (independent contexts on separate host threads and HIP streams); "
                         "1 = one proof at a time (latency); several hide the host-side Fiat-Shamir hashing and the launch gaps of "
                         "each proof behind the kernels of the others, and from five on the pool sends the trace commitments out in groups "
                         "of four in the lane form of the leaf hash (19.6 GB of HBM per context)")
    args = ap.parse_args()

    n_dev = max(0, args.devices_in_process)
    if n_dev and (args.gpus > 1 or args.input != "witness"):
        raise SystemExit("--devices-in-process is its own launch form: without --gpus, from operands")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))

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
    # the LDE kernel these options and this shape take (csrc/prover.hip run_lde_trace): 8192 rows go through the wave-resident kernel
    # unless "lde_impl" = 1 is set (bench.py never sets it); every other power of two from 2^8 on through lde_columns_v2_kernel
    LDE_KERNEL = "lde_columns_wave_kernel" if n == 8192 else "lde_columns_v2_kernel"
    log_n = n.bit_length() - 1
    N = n << cfg.rate_bits

    inflight = max(1, args.inflight)
    if rehearse and world > 1:
        inflight = max(1, min(inflight, 8 // world))  # the rehearsing ranks share ONE card's memory
    # the in-flight proofs go through the library's own scheduler, as a caller of the C ABI would drive them: `inflight` FinalExp-class
    # contexts, one host thread each inside libstarkhip.so, generator threads that record the traces (a pool reserves every buffer of
    # its contexts when it is created -- 19.6 GB per FinalExp-class context; should eight ever not fit, fewer proofs in flight are
    # still a valid measurement of the same metric)
    pool = None
    if n_dev and rehearse:
        inflight = max(1, min(inflight, 8 // n_dev))  # the rehearsing pools share ONE card's memory
    in_process_devices = ([0] * n_dev if rehearse else list(range(n_dev))) if n_dev else None
    n_pools = max(1, n_dev)
    for k in sorted({inflight, min(inflight, 6), min(inflight, 4)}, reverse=True):
        try:
            pool = S.ProofPool(local_rank, big_contexts=k, small_contexts=1, warm_up=1, devices=in_process_devices)
            inflight = k
            break
        except S.StarkhipError as e:
            print(f"bench.py: a pool of {k} FinalExp-class contexts could not be created ({e}); trying fewer", file=sys.stderr)
            torch.cuda.empty_cache()
    if pool is None:
        raise SystemExit("bench.py: no proof pool could be created")
    reservation = pool.reservation()

    # synthetic statements, a different one per rank AND per context
    n_inputs = inflight * n_pools
    seeds = [0x5EED0000 + 1 + rank * n_inputs + i for i in range(n_inputs)]
    inputs = [synthetic_final_exp_input(s) for s in seeds]
    helper = S.Prover(local_rank)  # page-locked staging for the host-rows leg / device traces
    host_rows = None
    device_work = []

    def make_device_traces():
        # traces generated on the host, moved to HBM as column-major u64 (as int64 bit patterns)
        nonlocal host_rows
        if device_work:
            return
        if host_rows is None:
            host_rows = helper.host_array((n, C))  # page-locked, reused for every generated trace
        for i in range(inflight):
            _, pis = S.trace_final_exp(inputs[i], out=host_rows)
            d_rows = torch.from_numpy(host_rows.view(np.int64)).to(f"cuda:{local_rank}")
            device_work.append((d_rows.t().contiguous(), pis))  # trace_rows_to_poly_values
            del d_rows
            torch.cuda.empty_cache()  # the row-major copy goes back to the device, not into torch's cache: the library allocates beside torch
        torch.cuda.synchronize()

    if args.input == "device":
        make_device_traces()

    def submit(i):
        if args.input == "device":
            d_cols, pis = device_work[i % inflight]
            return pool.submit_device(air, cfg, d_cols.data_ptr(), n, pis, layout=1)
        return pool.submit_witness(air, inputs[i % n_inputs])

    # warm-up: every context proves once (buffers, tables, plans); one proof in flight at a time gives the reference bytes of
    # each input, which the proofs of the timed region are compared with below
    solo_proofs = {}
    for i in range(n_inputs):
        pr, _ = pool.wait(submit(i))
        solo_proofs[i] = pr
    for w in range(max(1, args.warmup)):
        for t in [submit(i) for i in range(n_inputs)]:  # ... and all contexts at once
            pool.wait(t, keep=False)
    phase_ms = {k: 0.0 for k in S.PHASE_NAMES}
    total_steps = args.steps * n_pools  # per process: `steps` per device

    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    cpu0 = time.process_time()
    host_cpu0 = S.api.host_cpu_seconds()
    threads0 = thread_cpu_by_name()
    t0 = time.perf_counter()
    tickets = [submit(k) for k in range(total_steps)]
    timed_last = {}
    done_by_slot = {}  # pool slot -> completion times of the timed proofs (seconds since that pool was created)
    timed_kernel_ms = {}  # kernel name -> launch durations of the trace-commitment / LDE / quotient kernels inside the timed region
    timed_groups = []
    gen_ms = []
    for k, t in enumerate(tickets):
        keep = k >= total_steps - n_inputs  # the LAST proof of every input made inside the timed region is kept and checked below
        slot = pool.slot_of(t)
        pr, info = pool.wait(t, keep=keep)
        done_by_slot.setdefault(slot, []).append(info["timeline_s"][4])
        if keep:
            timed_last[k % n_inputs] = pr
        for name, v in info["phase_ms"].items():
            phase_ms[name] += v
        hk = FORM_KERNEL[info["leaf_hash_form"]]
        timed_kernel_ms.setdefault(hk, []).append(info["kernel_ms"]["leaf_hash"])
        timed_kernel_ms.setdefault(LDE_KERNEL, []).append(info["kernel_ms"]["lde_columns"])
        timed_kernel_ms.setdefault("quotient_tiles_kernel", []).append(info["kernel_ms"]["quotient_eval"])
        if info["leaf_hash_form"] == "lane":
            timed_groups.append(info["leaf_hash_group"])
        tl = info["timeline_s"]
        gen_ms.append((tl[2] - tl[1]) * 1e3)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed_own = time.perf_counter() - t0
    cpu_s_per_proof = (time.process_time() - cpu0) / max(1, total_steps)
    host_cpu1 = S.api.host_cpu_seconds()
    cpu_split = {k: (host_cpu1[k] - host_cpu0[k]) / max(1, total_steps) for k in host_cpu1}
    cpu_split["runtime_and_caller"] = cpu_s_per_proof - cpu_split["recording"] - cpu_split["proving"]
    threads1 = thread_cpu_by_name()
    cpu_by_thread = {k: round((v - threads0.get(k, 0.0)) / max(1, total_steps), 4) for k, v in threads1.items()}
    cpu_by_thread = dict(sorted(((k, v) for k, v in cpu_by_thread.items() if v >= 0.0005), key=lambda kv: -kv[1])[:10])
    elapsed = parallel.max_over_ranks(dist, elapsed_own, device=reduce_device)
    # every rank's own rate and CPU budget, for rank 0's line (N > 1: a bent curve must be attributable to a rank and to host or device)
    host_info = pool.host_info()
    # ... and the device it ran on: ordinal and PCI address (domain << 16 | bus << 8 | device), so that the line shows N ranks on N devices
    dev_ordinal = torch.cuda.current_device()
    props = torch.cuda.get_device_properties(dev_ordinal)
    pci = (int(getattr(props, "pci_domain_id", 0)) << 16) | (int(getattr(props, "pci_bus_id", 0)) << 8) | int(getattr(props, "pci_device_id", 0))
    per_rank = parallel.gather_over_ranks(dist, [total_steps / elapsed_own, float(host_info[0]["cpu_budget"]), cpu_s_per_proof, float(dev_ordinal), float(pci)],
                                          device=reduce_device)
    group = ({"backend": str(dist.get_backend()), "ranks": int(dist.get_world_size()),
              "collectives_in_timed_region": "the two barriers that bracket it; no data-path collective (proof-parallel)"} if dist is not None else None)
    # steady state: per pool, the completions after its first `inflight` and up to its last `inflight` -- start-up and tail dropped
    steady, steady_n = 0.0, 0
    for times in done_by_slot.values():
        times = sorted(times)
        if len(times) >= 3 * inflight:
            a, b = times[inflight - 1], times[len(times) - inflight - 1]
            if b > a:
                steady += (len(times) - 2 * inflight) / (b - a)
                steady_n += 1
    steady = parallel.sum_over_ranks(dist, steady if steady_n == len(done_by_slot) else float("nan"), device=reduce_device)

    # ---- untimed: what was timed is checked -- every kept proof of the timed region is accepted by the verifier and equals,
    # byte for byte, the proof of the same input made with nothing else in flight; rank 0's first input also has an oracle digest
    timed_verified = 0
    for i, pr in sorted(timed_last.items()):
        S.verify_stark_proof(air, cfg, pr)
        if not np.array_equal(pr, solo_proofs[i]):
            raise SystemExit(f"proof of input {seeds[i]:#x} made with {inflight} in flight per device differs from the one made alone")
        timed_verified += 1
    oracle_match = None
    if rank == 0 and 0 in timed_last:
        try:
            import hashlib
            want = open(os.path.join(ROOT, "tests", "golden", "final_exp_seed_%x_proof.sha256" % seeds[0])).read().split()[0]
            oracle_match = hashlib.sha256(timed_last[0].tobytes()).hexdigest() == want
        except OSError:
            pass
        if oracle_match is False:
            raise SystemExit("timed proof differs from the CPU oracle's digest")

    if rank == 0:
        steps = max(1, args.steps)
        alg = {"lde_columns": 8.0 * C * (n + N),  # read values, write the LDE (IFFT and LDE fused in one kernel; round 4: no coefficients kept)
               "leaf_hash": 8.0 * C * N,              # read the LDE once
               "quotient_eval": 8.0 * C * N}          # read the LDE on the quotient coset once
        alg_by_kernel = {LDE_KERNEL: alg["lde_columns"], "quotient_tiles_kernel": alg["quotient_eval"]}
        for kname in FORM_KERNEL.values():
            alg_by_kernel[kname] = alg["leaf_hash"]
        perms = (C + 7) // 8 * N
        # HBM-side bytes per launch from the committed PMC passes (bench.py cannot collect counters itself); a figure taken on another
        # version of the kernel's sources is stale: not reported as this run's traffic
        pmc, pmc_src, pmc_stale = {}, None, []
        try:
            from tools.kernel_fingerprint import kernel_fingerprint
            raw = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")))
            pmc_src = raw.get("_source")
            for k, names in PMC_NAMES.items():
                for nm in names:
                    if nm in raw:
                        if raw[nm].get("source_sha256") == kernel_fingerprint(nm):
                            pmc[k] = raw[nm]["traffic_bytes"]
                        else:
                            pmc_stale.append(nm)
                        break
        except Exception:
            pass

        # ---- the roofline block: the kernel with the most device time INSIDE the timed region, durations from HIP events recorded on
        # the stream the kernel was launched on (the pool's scheduler records them around each commitment launch)
        dom = max(timed_kernel_ms, key=lambda k: sum(timed_kernel_ms[k]))
        dom_ms = sum(timed_kernel_ms[dom]) / len(timed_kernel_ms[dom])
        side = (sum(timed_groups) / len(timed_groups)) if (dom == "leaf_hash_lane_kernel" and timed_groups) else 1.0
        per_launch_gbs = alg_by_kernel[dom] / (dom_ms * 1e-3) / 1e9
        pmc_key = {"leaf_hash_lane_kernel": "leaf_hash_lane", "leaf_hash_kernel": "leaf_hash", "leaf_hash_pair_kernel": "leaf_hash", LDE_KERNEL: "lde_columns",
                   "quotient_tiles_kernel": "quotient_eval"}.get(dom)
        # `bound`: what limits the kernel.  Every heavy kernel of this prover is limited by the SIMDs' vector instruction issue (64-bit modular
        # arithmetic out of 32-bit multiply-adds), not by HBM or the matrix pipe; achieved / peak / frac / traffic stay on the HBM axis, as the
        # benchmark contract prescribes for a byte-moving path, and the `valu` block gives the fraction of the issue peak
        roofline = {"bound": "valu-issue", "bound_axis_of_achieved_peak_frac": "hbm", "kernel": dom, "achieved": per_launch_gbs * side, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": per_launch_gbs * side / HBM_PEAK_GBS, "traffic": pmc.get(pmc_key),
                    "traffic_source": pmc_src, "traffic_stale_for": pmc_stale or None,
                    "algorithmic_bytes_per_launch": alg_by_kernel[dom], "avg_launch_ms": dom_ms, "launches_timed": len(timed_kernel_ms[dom]),
                    "launches_side_by_side": side, "achieved_per_launch": per_launch_gbs,
                    "group_sizes": ({str(k): timed_groups.count(k) for k in sorted(set(timed_groups))} if timed_groups else None),
                    "achieved_is": ("algorithmic bytes per launch / average launch duration x the launches of a group that run side by side (a lane-form "
                                    "launch holds a quarter of the chip's registers; `achieved_per_launch` is the plain quotient)"
                                    if side > 1 else "algorithmic bytes per launch / average launch duration"),
                    "durations": (f"HIP events on the launch stream, the {len(timed_kernel_ms[dom])} launches of the timed region ({inflight} proofs in flight per device); "
                                  "the other proofs' kernels run beside these launches, so a launch is longer than the same launches with the chip to "
                                  "themselves while the proofs/s are higher"),
                    "bound_note": "achieved / peak / frac are algorithmic bytes against the 8 TB/s of HBM (the contract's axis); the kernel's limiter is `bound`",
                    "share_of_timed_kernel_time": {k: sum(v) for k, v in timed_kernel_ms.items()},
                    "limiter": "integer VALU issue"}
        POSEIDON_LANE_SLOTS = poseidon_lane_slots()
        if dom == "leaf_hash_lane_kernel" and POSEIDON_LANE_SLOTS:
            slots = perms / 64.0 * POSEIDON_LANE_SLOTS
            roofline["valu"] = {"issue_slots_per_launch": slots, "achieved_Gslots_per_s": slots * side / (dom_ms * 1e-3) / 1e9, "peak_Ginstr_per_s": VALU_PEAK_GINSTR,
                                "frac": slots * side / (dom_ms * 1e-3) / 1e9 / VALU_PEAK_GINSTR,
                                "basis": f"{POSEIDON_LANE_SLOTS} issue slots per permutation and lane (block sizes in csrc/lane_round_asm.inc; the eight MFMAs of a matrix-pipe round "
                                         f"counted as one slot each), {perms} permutations / 64 lanes; peak = 256 CUs x 4 "
                                         "SIMDs x 2.4 GHz / 4 cycles per integer VALU instruction (tools/valu_rate_bench.hip); some slots are 2-cycle instructions"}

        # ---- untimed: the same proof with the GPU to itself (one in flight): uncontended kernel and phase durations
        solo_ms = {k: 0.0 for k in KERNELS}
        solo_phase = {k: 0.0 for k in S.PHASE_NAMES}
        solo_host = {"fiat_shamir": 0.0, "other": 0.0}
        t_solo, solo_gen = None, None
        kernels = {}
        if not args.no_solo:
            n_solo = 3
            t_solo = time.perf_counter()
            solo_gen = 0.0
            for _ in range(n_solo):
                _, info = pool.wait(submit(0), keep=False)
                for k, v in info["kernel_ms"].items():
                    solo_ms[k] += v / n_solo
                for k, v in info["phase_ms"].items():
                    solo_phase[k] += v / n_solo
                for k, v in info["host_ms"].items():
                    solo_host[k] += v / n_solo
                solo_gen += (info["timeline_s"][2] - info["timeline_s"][1]) * 1e3 / n_solo
            t_solo = (time.perf_counter() - t_solo) / n_solo
            for k in KERNELS:
                ms = solo_ms[k]
                gbs = alg[k] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
                kernels[k] = {"avg_ms": ms, "algorithmic_bytes": alg[k], "algorithmic_GBps": gbs, "hbm_frac": gbs / HBM_PEAK_GBS,
                              "traffic_bytes": pmc.get(k), "traffic_over_algorithmic": (pmc[k] / alg[k]) if k in pmc else None}
            # the leaf hash alone: a lone FinalExp commitment goes out in the pair form (32 permutations per wave, the generator's slot count);
            # a library that still sends it through the quad form: static instruction count of one quad permutation, 16 quads per wave
            pair_slots = poseidon_pair_slots()
            solo_form = info["leaf_hash_form"]
            wave_instr = perms * pair_slots / 32.0 if (solo_form == "pair" and pair_slots) else perms * POSEIDON_QUAD_INSTRS / 16.0
            lh_ms = solo_ms["leaf_hash"]
            kernels["leaf_hash"]["valu"] = {"kernel": FORM_KERNEL.get(solo_form, solo_form), "wave_instructions": wave_instr,
                                            "note": "one wave per SIMD: a lone wave issues a 64-bit-encoded instruction every 4.7 - 5.2 cycles (profiles/r03_k_valu_rates.txt), the peak counts 4",
                                            "achieved_Ginstr_per_s": wave_instr / (lh_ms * 1e-3) / 1e9 if lh_ms > 0 else 0.0, "peak_Ginstr_per_s": VALU_PEAK_GINSTR,
                                            "frac": (wave_instr / (lh_ms * 1e-3) / 1e9 / VALU_PEAK_GINSTR) if lh_ms > 0 else 0.0}
        if "leaf_hash_lane_kernel" in timed_kernel_ms:
            v = timed_kernel_ms["leaf_hash_lane_kernel"]
            ms = sum(v) / len(v)
            g = sum(timed_groups) / len(timed_groups)
            kernels["leaf_hash_lane_in_groups"] = {"avg_ms": ms, "launches": len(v), "launches_side_by_side": g, "ms_per_commitment": ms / g,
                                                   "algorithmic_bytes": alg["leaf_hash"], "algorithmic_GBps": alg["leaf_hash"] * g / (ms * 1e-3) / 1e9,
                                                   "hbm_frac": alg["leaf_hash"] * g / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic_bytes": pmc.get("leaf_hash_lane"),
                                                   "traffic_over_algorithmic": (pmc["leaf_hash_lane"] / alg["leaf_hash"]) if "leaf_hash_lane" in pmc else None,
                                                   "measured": "inside the timed region"}
        lh_ms = solo_ms["leaf_hash"]
        out = {
            "metric": "starky proofs/sec (FinalExponentiateStark 73527x8192)",
            "value": world * total_steps / elapsed,
            "unit": "proofs/s",
            "n_gpus": world * n_pools, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "value_steady_state": None if steady != steady else steady,
            "value_steady_state_is": (f"proofs/s over the middle of the timed region: per pool, the completions after its first {inflight} and up to its last "
                                      f"{inflight} (start-up and tail of the pool dropped), summed over pools and ranks; null when a pool made fewer than "
                                      f"{3 * inflight} proofs in the timed region"),
            "host": {"cpus_granted": cpu_quota(), "cpu_budget_process": int(S.lib.starkhip_cpu_budget()), "pools": host_info,
                     "cpu_seconds_per_proof": cpu_s_per_proof, "cpu_seconds_per_proof_by_role": cpu_split, "cpu_seconds_per_proof_by_thread_name": cpu_by_thread, "hw_queues_late": S.api.hw_queues_late(),
                     "note": ("cpus_granted: cgroup quota / affinity mask of this process; cpu_budget_process: the library's figure (the same, divided by "
                              "LOCAL_WORLD_SIZE under torch.distributed.run); pools[].cpu_budget: what each pool plans with (divided again by the pools of an "
                              "in-process multi-device handle), its generator threads and the threads one FinalExp recording may use; cpu_seconds_per_proof: "
                              "process CPU time over the timed region / proofs; by_role: recording = inside starkhip_trace_* on the generator threads and their helpers, "
                              "proving = the context threads inside prove() (launches, Fiat-Shamir sponge, upload gather), runtime_and_caller = the rest (the HIP "
                              "runtime's own threads, this script)")},
            "process_group": group,  # --gpus N > 1: the backend torch.distributed runs on ("nccl" = RCCL) and the ranks IT counts
            "rccl_ranks": (group["ranks"] if group and group["backend"] == "nccl" else None),
            "devices_distinct": (len({int(v[4]) for v in per_rank}) == len(per_rank) if world > 1 else None),  # False in a rehearsal on one card
            "per_rank": ([{"rank": r, "proofs_per_s": v[0], "cpu_budget": int(v[1]), "cpu_seconds_per_proof": v[2], "device_ordinal": int(v[3]),
                           "pci": "%04x:%02x:%02x" % (int(v[4]) >> 16, (int(v[4]) >> 8) & 0xFF, int(v[4]) & 0xFF)} for r, v in enumerate(per_rank)]
                         if world > 1 else None),
            "per_rank_min_max": ([min(v[0] for v in per_rank), max(v[0] for v in per_rank)] if world > 1 else None),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 (Goldilocks field)",
            "data": "synthetic" + (" -- REHEARSAL: all ranks / pools on one GPU, not a measurement" if (rehearse and (world > 1 or n_dev > 1)) else ""),
            "config": {"workload": "FinalExponentiateStark 73527 cols x 8192 rows, rate_bits 2, 360800 constraints, "
                                   "standard_fast_config (84 queries, 16 pow bits); independent proofs, a different input per context and rank",
                       "parallelism": (f"proof-parallel x{n_pools}, ONE process: a pool per device behind starkhip_multipool_* (longest-job-first placement by the library, "
                                       "no process group, no collective)" + (" -- REHEARSAL: the pools share device 0" if (n_dev and rehearse) else "")
                                       if n_dev else f"proof-parallel x{world}"), "proofs_in_flight_per_gpu": inflight,
                       "timed_region": ("operand (one Fp12 on the host) -> generate_trace (recorded, 153 MB) -> upload -> expansion on the device -> prove -> proof bytes "
                                        "on the host: the reference's own boundary (src/aggregate_proof.rs:158-176)" if args.input == "witness"
                                        else "column-major trace resident in HBM -> prove -> proof bytes on the host (--input device)"),
                       "driver": (("starkhip_multipool_submit_witness / starkhip_multipool_wait" if n_dev else
                                   "starkhip_pool_submit_witness / starkhip_pool_wait" if args.input == "witness" else "starkhip_pool_submit (device pointer) / starkhip_pool_wait")
                                  + " (in-flight scheduling inside libstarkhip.so)"),
                       "pool_reservation_GB": {"device": reservation["device_bytes"] / 1e9, "per_final_exp_context": reservation["big_context_device_bytes"] / 1e9,
                                               "page_locked_host": reservation["pinned_host_bytes"] / 1e9}},
            "roofline": roofline,
            "kernels": kernels,
            # the two rates the proof is governed by, from the uncontended launches
            "poseidon_perms_per_s": perms / (lh_ms * 1e-3) if lh_ms > 0 else None,
            "constraint_evals_per_s": (S.air_num_constraints(air) * float(N) / (solo_ms["quotient_eval"] * 1e-3)) if solo_ms["quotient_eval"] > 0 else None,
            "timed_proofs_verified": timed_verified,
            "timed_proofs_check": (f"the last proof of each of the {inflight} inputs made INSIDE the timed region: accepted by the verifier and byte-identical to "
                                   "the proof of the same input made with one in flight"
                                   + ("; input 0x5eed0001 also matches the CPU oracle's digest (tests/golden/final_exp_seed_5eed0001_proof.sha256)" if oracle_match else "")),
            "oracle_digest_match": oracle_match,
            "generate_trace_ms_timed_region": (sum(gen_ms) / len(gen_ms)) if (gen_ms and args.input == "witness") else None,
            "latency_ms_one_in_flight": t_solo * 1e3 if t_solo else None,
            "generate_trace_ms_one_in_flight": solo_gen if args.input == "witness" else None,
            "phase_ms_one_in_flight": solo_phase if t_solo else None,
            "host_ms_one_in_flight": dict(solo_host, note="wall time of host work inside prove(): the challenger's sequential Poseidon sponge (inside the "
                                                            "device phases fri_combine / fri_commit), and the FRI batches' divisions by X - z") if t_solo else None,
            "phase_ms_timed_region": {k: v / steps for k, v in phase_ms.items()},
            "note": ("roofline: the timed region's dominant kernel, durations taken inside the timed region; kernels: durations with ONE proof in flight "
                     "(+ the lane-form groups of the timed region); phase_ms_timed_region: HIP-event phase durations inside the timed "
                     "region, which include the time a kernel shares the CUs with the other contexts' kernels"),
            "reference_published": {"value": 1 / 92.0, "unit": "proofs/s", "hardware": "AWS r6a.8xlarge, 32-core EPYC 7R13 (reference README.md:39)"},
        }
        if not args.no_boundary and world == 1 and not n_dev:  # per-GPU figures, taken at N = 1 (other ranks would wait in the teardown meanwhile)
            # ---- untimed legs on the SAME pool, `inflight` in flight: the other hand-over forms of the same boundary
            try:
                def leg(submit_one, reps):
                    for t in [submit_one(i) for i in range(inflight)]:  # warm-up
                        pool.wait(t, keep=False)
                    t0 = time.perf_counter()
                    for t in [submit_one(i) for i in range(reps)]:
                        pool.wait(t, keep=False)
                    return reps / (time.perf_counter() - t0)
                reps = 3 * inflight  # three waves of the pool per leg
                if host_rows is None:
                    host_rows = helper.host_array((n, C))
                _, pis0 = S.trace_final_exp(inputs[0], out=host_rows)
                out["value_host_rows"] = {"value": leg(lambda i: pool.submit(air, cfg, host_rows, pis0), reps), "unit": "proofs/s per GPU", "in_flight": inflight,
                                          "what": "page-locked host ROWS (4.8 GB, what generate_trace returns) -> H2D -> transpose -> proof -> D2H; generation not included"}
                compact, cpis = S.trace_final_exp(inputs[0], compact=True)
                out["value_compact"] = {"value": leg(lambda i: pool.submit(air, cfg, compact, cpis), reps), "unit": "proofs/s per GPU", "in_flight": inflight,
                                        "what": "recorded trace (153 MB of runs) -> upload -> expansion on the device -> proof -> D2H; generation not included"}
                if args.input == "witness":
                    make_device_traces()
                    out["value_device_resident"] = {"value": leg(lambda i: pool.submit_device(air, cfg, device_work[i % inflight][0].data_ptr(), n,
                                                                                              device_work[i % inflight][1], layout=1), reps),
                                                    "unit": "proofs/s per GPU", "in_flight": inflight,
                                                    "what": "column-major trace already in HBM -> proof -> D2H (rounds 1-3's headline; no generation, no upload)"}
                else:
                    out["value_witness"] = {"value": leg(lambda i: pool.submit_witness(air, inputs[i % inflight]), reps), "unit": "proofs/s per GPU",
                                            "in_flight": inflight, "what": "operand -> generate_trace -> upload -> proof -> D2H"}
            except Exception as e:  # never lose the main line to an auxiliary leg
                out["value_host_rows"] = {"value": None, "error": str(e)}
        if not args.no_cpu_baseline and world == 1 and n_dev <= 1:  # rank 0 at N = 1 only: other ranks would sit in the teardown barrier meanwhile
            try:
                out["cpu_baseline"] = cpu_baseline_sample(S, S.air_program(air), C, log_n, cfg.rate_bits)
            except Exception as e:  # the oracle is a checker, never a dependency of the measured path
                out["cpu_baseline"] = {"value": None, "unit": "proofs/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    pool.close()
    helper.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
