"""The proof pool's host logic (csrc/scheduler.cpp: generator threads, context workers, commitment scheduler, shutdown) under
ThreadSanitizer on the CPU: tests/tsan_pool_main.cpp against the stand-in device of csrc/host_only_stubs.cc (no GPU, no proofs:
contexts exist, prove() sleeps, asks the scheduler for its commitment and returns a blob).  All three commit policies, jobs that
fail before and after their commitment, concurrent submit / wait, destroy with work queued."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.slow
def test_pool_threads_are_race_free_under_thread_sanitizer(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "tsan_pool")
    srcs = sorted(glob.glob(os.path.join(ROOT, "starky_bls12_381_amd", "csrc", "*.cpp"))) + [os.path.join(ROOT, "starky_bls12_381_amd", "csrc", "host_only_stubs.cc")]
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "include"),
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", exe, os.path.join(ROOT, "tests", "tsan_pool_main.cpp")] + srcs + ["-lpthread"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    if b.returncode != 0 and "tsan" in b.stderr.lower() and "cannot find" in b.stderr.lower():
        pytest.skip("libtsan not installed")
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "ThreadSanitizer" not in r.stderr
    for policy in (0, 1, 2):
        assert f"policy {policy}: ok" in r.stdout
