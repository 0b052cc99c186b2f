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


def test_bench_gpus_2_starts_itself_and_prints_one_line_for_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["STARKHIP_BENCH_REHEARSE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--inflight", "2",
                        "--no-cpu-baseline", "--no-boundary", "--no-solo"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0
    assert d["timed_proofs_verified"] == 2 and "REHEARSAL" in d["data"]
    assert d["config"]["parallelism"] == "proof-parallel x2"
