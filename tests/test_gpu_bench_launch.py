"""`python bench.py --gpus 2` started WITHOUT torchrun on the GPU box: bench.py launches `python -m torch.distributed.run` itself as a
child process (never an exec; before torch or the GPU is touched in the parent) and relays rank 0's JSON line.  Two ranks rehearse on
the one card over gloo (STARKHIP_BENCH_REHEARSE=1: control flow, not a measurement); on an 8-GPU node the same command line with
`--gpus 8` needs nothing else (BASELINE.json configs[3], [4]; /root/reference/src/aggregate_proof.rs:304-370 is what shards)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("ranks", [2, 4])
def test_bench_gpus_n_starts_itself_and_prints_one_line_for_n_ranks(ranks):
    """Four ranks is what one card allows beside the test process (the box admits six GPU processes); the eight-rank control flow -- launcher
    environment, process group, the line's reductions, teardown -- runs on the CPU in tests/test_bench_launch_cpu.py."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["STARKHIP_BENCH_REHEARSE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--inflight", "2",
                        "--no-cpu-baseline", "--no-boundary", "--no-solo"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["steps"] == 2 and d["value"] > 0
    assert d["timed_proofs_verified"] == 2 and "REHEARSAL" in d["data"]
    assert d["config"]["parallelism"] == f"proof-parallel x{ranks}"
    assert d["per_rank"] and len(d["per_rank"]) == ranks and all(r["proofs_per_s"] > 0 and r["cpu_budget"] >= 1 for r in d["per_rank"])
    lo, hi = d["per_rank_min_max"]
    assert lo <= hi and lo > 0
    # what the driver reads to see that the process group saw N ranks on N devices: here gloo, every rank on the one card -- and the line says so
    assert d["process_group"]["backend"] == "gloo" and d["process_group"]["ranks"] == ranks and d["rccl_ranks"] is None
    assert d["devices_distinct"] is False and {r["device_ordinal"] for r in d["per_rank"]} == {0} and len({r["pci"] for r in d["per_rank"]}) == 1


def _bench(args, env_extra=None, prefix=()):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env.update(env_extra or {})
    r = subprocess.run(list(prefix) + [sys.executable, os.path.join(ROOT, "bench.py")] + [str(a) for a in args], capture_output=True, text=True, timeout=900,
                       env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_devices_in_process_prints_the_same_line_from_one_process():
    """`bench.py --devices-in-process 2`: the reference's single-process caller shape -- one process, a pool per device behind
    starkhip_multipool_* -- rehearsed with both pools on the one card.  Same JSON line, both pools prove, every timed proof verifies and
    equals the proof made alone."""
    d = _bench(["--devices-in-process", 2, "--steps", 4, "--warmup", 1, "--inflight", 2, "--no-cpu-baseline", "--no-solo"], {"STARKHIP_BENCH_REHEARSE": "1"})
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0 and "REHEARSAL" in d["data"]
    assert "ONE process" in d["config"]["parallelism"] and "starkhip_multipool" in d["config"]["driver"]
    assert d["timed_proofs_verified"] == 4                      # 2 in flight x 2 pools: the last proof of each of the four inputs
    assert len(d["host"]["pools"]) == 2 and all(p["cpu_budget"] >= 1 for p in d["host"]["pools"])
    assert all(p["pools_on_device"] == 2 for p in d["host"]["pools"])   # both pools on the one card: a handle that says so (and once on stderr)
    assert d["host"]["pools"][0]["cpu_budget"] <= max(1, d["host"]["cpu_budget_process"] // 2)   # the process's CPUs are split between its pools
    assert d["roofline"]["frac"] > 0 and d["per_rank"] is None


def test_bench_under_a_two_cpu_mask_reports_the_budget_and_still_verifies():
    """A host with two CPUs for the process (plain `taskset` on the command -- NOT under rocprofv3): the line must say so (cpus_granted, the
    pool's budget and threads, CPU-seconds per proof) and the proofs must still verify and match.  Two CPUs now feed the GPU: round 6 brought
    the host cost from 0.45 to 0.16 - 0.20 CPU-seconds per proof (recordings 0.27 -> 0.12, the runtime thread that polled during cross-stream
    waits gone), and the pool keeps 7.4 proofs/s on two CPUs where round 5 kept 5.4 (profiles/r06_cpu_sensitivity.txt; 24 steps).  The
    bounds below leave room for a slower box and for the ramp of a 24-proof run."""
    cpus = sorted(os.sched_getaffinity(0))
    if len(cpus) < 2:
        pytest.skip("fewer than two CPUs")
    mask = ",".join(str(c) for c in cpus[:2])
    d = _bench(["--steps", 24, "--warmup", 1, "--no-cpu-baseline", "--no-boundary", "--no-solo"], prefix=("taskset", "-c", mask))
    h = d["host"]
    assert h["cpus_granted"] == 2 and h["cpu_budget_process"] == 2 and h["pools"][0]["cpu_budget"] == 2
    assert h["pools"][0]["generator_threads"] >= 1 and h["pools"][0]["trace_threads_big"] == 1     # 3/4 of two CPUs: one thread per FinalExp recording
    assert 0.05 < h["cpu_seconds_per_proof"] < 0.30
    assert d["timed_proofs_verified"] == 8 and d["oracle_digest_match"] is True
    assert d["value"] >= 6.5, d["value"]    # measured 7.0 - 7.4 with two CPUs (round 5: 5.4)
