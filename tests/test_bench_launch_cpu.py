"""`python bench.py --gpus N` (N > 1) outside torchrun starts itself under torch.distributed.run as a CHILD process -- never an
exec, and before torch or the GPU is touched -- relays the child's output and returns its exit code (bench.self_launch; the same for
tools/bench_signature.py).  Here on the CPU with a stand-in script: the ranks come up with the launcher's environment."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, body, n=2, args=()):
    script = tmp_path / "ranks.py"
    script.write_text(textwrap.dedent(body))
    code = f"import sys; sys.path.insert(0, {ROOT!r}); import bench; raise SystemExit(bench.self_launch({list(args)!r}, {n}, script={str(script)!r}))"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)


def test_self_launch_starts_one_rank_per_gpu_and_relays_output(tmp_path):
    r = _run(tmp_path, """
        import os, sys
        print("rank", os.environ["RANK"], "of", os.environ["WORLD_SIZE"], "local", os.environ["LOCAL_RANK"], os.environ["MASTER_ADDR"], sys.argv[1:], flush=True)
    """, args=("--gpus", "2", "--steps", "3"))
    assert r.returncode == 0, r.stderr
    lines = sorted(l for l in r.stdout.splitlines() if l.startswith("rank "))
    assert lines == ["rank 0 of 2 local 0 127.0.0.1 ['--gpus', '2', '--steps', '3']", "rank 1 of 2 local 1 127.0.0.1 ['--gpus', '2', '--steps', '3']"]


def test_self_launch_returns_the_childs_exit_code(tmp_path):
    r = _run(tmp_path, """
        import os, sys
        sys.exit(7 if os.environ["RANK"] == "1" else 0)
    """)
    assert r.returncode != 0


def test_importing_bench_does_not_import_torch():
    """self_launch must run before anything initialises the GPU: importing bench.py pulls in neither torch nor the library."""
    code = f"import sys; sys.path.insert(0, {ROOT!r}); import bench; assert 'torch' not in sys.modules and 'starky_bls12_381_amd' not in sys.modules"
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120).returncode == 0
