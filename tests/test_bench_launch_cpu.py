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


def test_eight_ranks_come_up_reduce_and_tear_down_over_gloo(tmp_path):
    """The control flow of `bench.py --gpus 8` with eight ranks, on the CPU: the launcher's environment, the process group (gloo here, RCCL on
    the node), the three reductions the line is made of (max of the times, the per-rank gather, the sum) and the teardown barrier.  No GPU is
    touched; what a rank would prove is a sleep of its own length."""
    r = _run(tmp_path, f"""
        import os, sys, time
        sys.path.insert(0, {ROOT!r})
        from starky_bls12_381_amd import parallel
        rank, local, world = parallel.rank_info()
        dist = parallel.init_distributed("gloo")
        assert dist.get_world_size() == 8 and dist.get_backend() == "gloo"
        dist.barrier()
        t0 = time.perf_counter()
        time.sleep(0.05 * (rank + 1))
        own = time.perf_counter() - t0
        dist.barrier()
        slowest = parallel.max_over_ranks(dist, own)
        rows = parallel.gather_over_ranks(dist, [rank, local, own])
        total = parallel.sum_over_ranks(dist, 1.0)
        assert slowest >= 0.4 and total == 8.0
        assert [int(v[0]) for v in rows] == list(range(8)) and [int(v[1]) for v in rows] == list(range(8))
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0:
            print("eight ranks done", flush=True)
    """, n=8, args=("--gpus", "8"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "eight ranks done" in r.stdout


def test_the_library_has_no_peer_to_peer_call():
    """Multi-GPU is proof-parallel: a pool per device, operands through host memory, nothing copied between devices.
    The shared library must not even import a peer-access or peer-copy entry point of the HIP runtime."""
    lib = os.path.join(ROOT, "starky_bls12_381_amd", "libstarkhip.so")
    out = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    peer = [l for l in out.stdout.splitlines() if "Peer" in l or "hipIpc" in l]
    assert peer == [], peer
