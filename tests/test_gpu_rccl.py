"""RCCL on the one GPU a test box has: a ONE-rank process group with backend "nccl" (= RCCL on ROCm) runs the path's
collectives on device tensors -- the operand broadcast (signature.broadcast_operands -> parallel.broadcast_u64: u64 words as int64
on cuda:0), the max / sum reductions of the benchmark's timing, and the raw-buffer proof collection (aggregate.collect_proofs)
on real proofs.  The multi-rank control flow is covered over gloo in test_multi_rank_cpu.py; this test is about the RCCL
code path itself executing on an MI355X.  It runs in a child process: the rendezvous environment and the process group are
that process's own."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, socket, faulthandler
faulthandler.dump_traceback_later(120, exit=True)   # a stuck collective shows where, instead of hanging the suite
def mark(what): print("[rccl-child]", what, file=sys.stderr, flush=True)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "tests")]
import numpy as np, torch, torch.distributed as dist
import starky_bls12_381_amd as S
from starky_bls12_381_amd import aggregate as A, parallel as P, signature as G
from bls_util import native_vectors, random_fp12
torch.cuda.set_device(0)
mark("init_process_group")
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))   # world of one: parallel.init_distributed only joins worlds > 1
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
dev = "cuda:0"
mark("broadcast")
# 1. the operand broadcast: u64 words travel as int64 on the device and come back bit for bit (values above 2^63 included)
sigs = G.synthetic_signatures(2, native_vectors()["bls_signature"], 0x77)
got = G.broadcast_operands(dist, sigs, 2, device=dev)
assert all(np.array_equal(a, b) for x, y in zip(sigs, got) for p, q in zip(x, y) for a, b in zip(p, q))
edge = np.array([0, 1, 2**63, 2**64 - 1, S.P - 1], dtype=np.uint64)
assert np.array_equal(P.broadcast_u64(dist, edge, device=dev), edge)
mark("reductions")
# 2. the timing reductions on a cuda tensor
assert P.max_over_ranks(dist, 1.25, device=dev) == 1.25 and P.sum_over_ranks(dist, 3.0, device=dev) == 3.0
mark("proofs")
# 3. proof collection: raw u64 buffers of REAL proofs through dist.broadcast on the device
pv = S.Prover(0)
mine = {}
for name, air, gen in (("fp12_mul", S.AIR_FP12_MUL, lambda: S.trace_fp12_mul(random_fp12(0x5EED4000), random_fp12(0x5EED4001), compact=True)),
                       ("toy", S.AIR_TEST_FIBONACCI, lambda: S.trace_fibonacci(3, 5, 256))):
    t, pis = gen()
    cfg = S.StarkConfig.for_air(air)
    mine[name] = (air, pv.prove(air, cfg, t, pis), cfg)
pv.close()
mark("collect")
merged = A.collect_proofs(dist, mine, device=dev)
assert sorted(merged) == sorted(mine)
for name, (air, proof, cfg) in merged.items():
    assert np.array_equal(proof, mine[name][1])
    S.verify_stark_proof(air, cfg, proof)
mark("barrier")
dist.barrier()
dist.destroy_process_group()
print("rccl one-rank ok")
"""


def test_one_rank_rccl_group_runs_the_paths_collectives_on_the_gpu():
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl one-rank ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
