"""End-to-end driver of BLS signature checks: operands -> traces -> the six STARK proofs per signature, on one or many GPUs.

What the reference does in `generate_aggregate_proof` between the milagro calls and the plonky2 recursion
(/root/reference/src/aggregate_proof.rs:304-370): for each signature, natively compute the two Miller-loop values, then
run the six drivers (generate_trace + prove + verify each, :23-179) one after the other on one thread.  Here:

  * a BATCH of signatures becomes 6 x B independent jobs, dealt to the ranks (one process per GPU) longest first
    (`plan_batch`; FinalExp is ~3/4 of a signature's work, so every GPU gets whole FinalExp proofs first);
  * rank 0 owns the input and broadcasts the operands -- 168 u32 limbs per signature: pk (x, y), H(m) (x, y), signature
    (x, y) -- with ONE collective (`broadcast_operands`: RCCL over xGMI on GPUs, gloo in the CPU tests); nothing else
    crosses ranks on the data path;
  * on a rank, generator threads record the traces of its jobs as compact runs (starkhip_trace_log_*: recording is
    per-thread, 0.3 s for FinalExp) while prover contexts consume them, so trace generation is INSIDE the timed region and
    overlapped with proving (`run_jobs`);
  * afterwards the proofs can be collected (`aggregate.collect_proofs`), verified and checked against the statement
    (`aggregate.check_links`, `check_statement`).

`synthetic_signatures` makes B different valid signatures from the reference's vector (src/native.rs:1480-1498): with secret
scalars s_i, t_i:  H_i = t_i * H(m),  sig_i = s_i * H_i,  pk_i = s_i * G1  =>  e(pk_i, H_i) * e(-G1, sig_i) = 1.
"""
import queue
import threading
import time

import numpy as np

from . import aggregate as A
from . import api as S
from . import parallel
from .eth_input import P as BLS_P
from .eth_input import ec2_mul

OPERAND_WORDS = 24 + 48 + 48  # pk (x, y: 12 limbs each), H(m) (x, y: 24 each), signature (x, y: 24 each); z = (1, 0) is implied


# ------------------------------------------------------------------------------------------------ inputs
def _g1_add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    (x1, y1), (x2, y2) = p1, p2
    if x1 == x2:
        if (y1 + y2) % BLS_P == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, BLS_P) % BLS_P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, BLS_P) % BLS_P
    x3 = (lam * lam - x1 - x2) % BLS_P
    return x3, (lam * (x1 - x3) - y1) % BLS_P


def _g1_mul(p, k):
    r = None
    while k:
        if k & 1:
            r = _g1_add(r, p)
        p = _g1_add(p, p)
        k >>= 1
    return r


def synthetic_signatures(count, vector, seed=0x51607A7E):
    """`count` different VALID signatures derived from the reference's test vector `vector` (the dict of decimal strings in
    tests/golden/native_vectors.json["bls_signature"]: generator gx, gy and the message point hm_*).
    Returns a list of (pk, hm, sig) limb tuples as `aggregate.signature_jobs` takes them (hm, sig with z = (1, 0))."""
    b = {k: int(v) for k, v in vector.items()}
    g1 = (b["gx"], b["gy"])
    hm0 = ((b["hm_x1"], b["hm_x2"]), (b["hm_y1"], b["hm_y2"]))
    one = A.fp2_limbs(1, 0)
    state = seed & 0xFFFFFFFFFFFFFFFF

    def nxt():
        nonlocal state  # splitmix64
        state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return (z ^ (z >> 31)) | 1

    out = []
    for _ in range(count):
        s, t = nxt(), nxt()
        h = ec2_mul(hm0, t)
        sig = ec2_mul(h, s)
        pk = _g1_mul(g1, s)
        out.append(((A.fp_limbs(pk[0]), A.fp_limbs(pk[1])),
                    (A.fp2_limbs(*h[0]), A.fp2_limbs(*h[1]), one.copy()),
                    (A.fp2_limbs(*sig[0]), A.fp2_limbs(*sig[1]), one.copy())))
    return out


def pack_operands(signatures):
    """[(pk, hm, sig)] -> uint64 [B, OPERAND_WORDS] (one u32 limb per word: what travels in the broadcast)."""
    out = np.zeros((len(signatures), OPERAND_WORDS), dtype=np.uint64)
    for i, (pk, hm, sig) in enumerate(signatures):
        out[i] = np.concatenate([pk[0], pk[1], hm[0], hm[1], sig[0], sig[1]]).astype(np.uint64)
    return out


def unpack_operands(words):
    one = A.fp2_limbs(1, 0)
    sigs = []
    for row in np.asarray(words, dtype=np.uint64).reshape(-1, OPERAND_WORDS):
        r = row.astype(np.uint32)
        sigs.append(((r[0:12].copy(), r[12:24].copy()), (r[24:48].copy(), r[48:72].copy(), one.copy()), (r[72:96].copy(), r[96:120].copy(), one.copy())))
    return sigs


def broadcast_operands(dist, signatures, batch, device="cpu", src=0):
    """Rank `src` holds `signatures` (others pass None); every rank returns the same list.  One broadcast of
    batch x 120 words (the "RCCL broadcast of public inputs over xGMI" of the north star: everything a rank needs to derive
    the public inputs of its proofs)."""
    if dist is None:
        return list(signatures)
    rank = dist.get_rank()
    words = pack_operands(signatures) if rank == src else np.zeros((batch, OPERAND_WORDS), dtype=np.uint64)
    return unpack_operands(parallel.broadcast_u64(dist, words.reshape(-1), src=src, device=device))


def tune_host_allocator():
    """For a DRIVER process that records traces all the time: keep glibc from handing every large log buffer back to the kernel
    (mmap threshold / trim threshold up, 256 MB of top pad), so a recording does not start with page faults on 150 MB of fresh
    memory each time: -15..20 % generator time measured.  Process-wide, so a driver's choice, not the library's."""
    import ctypes
    try:
        libc = ctypes.CDLL("libc.so.6")
        ok = libc.mallopt(-3, 1 << 30)   # M_MMAP_THRESHOLD
        ok &= libc.mallopt(-1, 1 << 30)  # M_TRIM_THRESHOLD
        ok &= libc.mallopt(-2, 1 << 28)  # M_TOP_PAD
        return bool(ok)
    except OSError:
        return False


# ------------------------------------------------------------------------------------------------ plan
def plan_batch(batch, world):
    """Per-rank job lists [(signature index, job name)], longest-processing-time-first over the 6 x batch jobs."""
    jobs = [(i, name) for i in range(batch) for name in A.JOB_ORDER]
    costs = [parallel.AIR_COST[A.JOB_AIR[name]] for _, name in jobs]
    return [[jobs[j] for j in idxs] for idxs in parallel.assign_jobs(costs, world)]


GENERATORS = {"pp1": S.trace_pairing_precomp, "pp2": S.trace_pairing_precomp, "ml1": S.trace_miller_loop, "ml2": S.trace_miller_loop,
              "fp12_mul": S.trace_fp12_mul, "final_exp": S.trace_final_exp}


def job_arguments(signatures, my_jobs):
    """Arguments of the trace generators for this rank's jobs; the natives (two Miller loops and their product per signature)
    are computed only for the signatures that need them here (fp12_mul and final_exp jobs), as the reference does at :352-363."""
    cache = {}
    out = {}
    for i, name in my_jobs:
        if i not in cache:
            cache[i] = A.signature_jobs(*signatures[i])
        out[(i, name)] = cache[i][0][name][1]
    return out, {i: v[1] for i, v in cache.items()}


# ------------------------------------------------------------------------------------------------ execution on one rank
BIG = ("final_exp",)  # jobs that get their own contexts: the FinalExp leaf hash is a one-shot grid of two waves per SIMD


def run_jobs(provers, my_jobs, args, gen_threads=6, prove=None, generate=None, queue_depth=None, trace_threads=None, big_after_small=False):
    """Generate and prove `my_jobs` on this rank: `gen_threads` host threads record compact traces into bounded queues, one
    host thread per prover context takes them and proves.  Returns ({(i, name): (air, proof, cfg)},
    {"generate_s": sum of generator time, "prove_s": sum of prover time, "wall_s": wall time}).

    `provers`: a list of contexts (one pool), or {"big": [...], "small": [...]}: FinalExp proofs (4.8 GB traces, 25 GB of
    buffers each) go to the few `big` contexts, the 1024-row AIRs -- latency chains that only fill the chip side by side --
    to the many `small` ones; both pools run at once.  Half of the generator threads start on the FinalExp traces (0.3 s
    each), the other half on the small ones, so neither pool waits for the other's traces.

    `trace_threads`: host threads ONE recording generator call may use (starkhip_trace_set_threads: the FinalExp and
    MillerLoop generators fill their gadget blocks in parallel once the native chain is known).  Default: with few jobs on
    this rank (one signature) 16, for a batch 1.

    `big_after_small` (two pools only): the FinalExp contexts start when the last small proof is done -- scheduling by type, so
    that a FinalExp commitment (two 176-register waves on every SIMD) and the small proofs' kernels do not wait for each other's
    register space.  Measured for a batch of 8 on one GPU: 2.26 signatures/s against 2.61 with both
    pools at once (the small proofs are latency chains that do not fill the chip by themselves), so it is off by default.

    `prove(prover, air, cfg, trace, pis)` and `generate(name, *args)` are injectable (CPU tests run the control flow without
    a GPU); defaults: Prover.prove and the compact trace generators."""
    if prove is None:
        def prove(pv, air, cfg, trace, pis):
            return pv.prove(air, cfg, trace, pis)
    if generate is None:
        def generate(name, *a):
            return GENERATORS[name](*a, compact=True)
        if trace_threads is None:
            # few jobs on this rank (one signature): the recording itself is the latency -- FinalExp 0.21 s on one thread,
            # 0.025 s on 16 (EPYC 9575F, profiles/r02_c_trace_threads.txt); a batch keeps one thread per call
            trace_threads = 16 if len(my_jobs) <= 6 else max(1, min(8, 2 * gen_threads // len(my_jobs)))
        restore_threads = S.set_trace_threads(trace_threads)
    else:
        restore_threads = None
    pools = provers if isinstance(provers, dict) else {"big": list(provers), "small": None}
    shared = pools.get("small") is None  # one pool takes everything
    order = sorted(my_jobs, key=lambda j: -parallel.AIR_COST[A.JOB_AIR[j[1]]])
    todo = {"big": [j for j in order if j[1] in BIG], "small": [j for j in order if j[1] not in BIG]}
    lock = threading.Lock()
    n_ctx = {"big": len(pools["big"]), "small": len(pools["big"]) if shared else len(pools["small"])}
    ready = {k: queue.Queue(maxsize=queue_depth or max(2, 2 * n_ctx[k])) for k in ("big", "small")}
    if big_after_small:
        # every FinalExp trace may wait recorded (150 MB each).  Whatever `queue_depth` says: a "big"-first generator that blocked
        # on a full queue here would never record the small traces the FinalExp provers are waiting for
        ready["big"] = queue.Queue(maxsize=max(2, len(todo["big"])))
    if shared:
        ready["small"] = ready["big"]
    results, errors = {}, []
    t_gen, t_prove = [0.0], [0.0]
    timeline = {}  # job -> [generation start, end, proof start, end] in seconds from the start of run_jobs

    def generator(first):
        second = "small" if first == "big" else "big"
        while True:
            with lock:
                if errors:
                    return
                kind = first if todo[first] else second
                if not todo[kind]:
                    return
                job = todo[kind].pop(0)
            try:
                t0 = time.perf_counter()
                trace, pis = generate(job[1], *args[job])
                with lock:
                    t_gen[0] += time.perf_counter() - t0
                    timeline[job] = [t0 - t_begin, time.perf_counter() - t_begin, None, None]
                ready[kind].put((job, trace, pis))
            except Exception as e:  # noqa: BLE001 -- reported to the caller below
                with lock:
                    errors.append(e)
                small_done.set()
                return

    n_small = len(todo["small"])
    small_left = [n_small]
    small_done = threading.Event()
    if not (big_after_small and not shared and n_small):
        small_done.set()

    def prover_loop(pv, kind):
        if kind == "big":
            small_done.wait()
        while True:
            item = ready[kind].get()
            if item is None:
                return
            job, trace, pis = item
            try:
                air = A.JOB_AIR[job[1]]
                cfg = S.StarkConfig.for_air(air)
                t0 = time.perf_counter()
                proof = prove(pv, air, cfg, trace, pis)
                with lock:
                    t_prove[0] += time.perf_counter() - t0
                    results[job] = (air, proof, cfg)
                    timeline[job][2:] = [t0 - t_begin, time.perf_counter() - t_begin]
            except Exception as e:  # noqa: BLE001
                with lock:
                    errors.append(e)
                small_done.set()  # never leave the other pool waiting
            if job[1] not in BIG:
                with lock:
                    small_left[0] -= 1
                    if small_left[0] <= 0:
                        small_done.set()

    t0 = t_begin = time.perf_counter()
    n_gen = max(1, min(gen_threads, len(order) or 1))
    n_big_gen = min(len(todo["big"]), max(1, n_gen // 2)) if todo["big"] else 0
    gens = [threading.Thread(target=generator, args=("big" if g < n_big_gen else "small",)) for g in range(n_gen)]
    pros = [threading.Thread(target=prover_loop, args=(pv, "big")) for pv in pools["big"]]
    if not shared:
        pros += [threading.Thread(target=prover_loop, args=(pv, "small")) for pv in pools["small"]]
    for t in gens + pros:
        t.start()
    for t in gens:
        t.join()
    for pv in pools["big"]:
        ready["big"].put(None)
    if not shared:
        for pv in pools["small"]:
            ready["small"].put(None)
    for t in pros:
        t.join()
    wall = time.perf_counter() - t0
    if restore_threads is not None:
        S.set_trace_threads(restore_threads)  # process-wide setting: leave it as it was found
    if errors:
        raise errors[0]
    return results, {"generate_s": t_gen[0], "prove_s": t_prove[0], "wall_s": wall, "timeline": timeline, "trace_threads": trace_threads}


def signature_proofs(results, index):
    """The six proofs of signature `index` out of `run_jobs` / collected results, keyed by job name."""
    return {name: results[(index, name)] for name in A.JOB_ORDER if (index, name) in results}


# ------------------------------------------------------------------------------------------------ execution on the library's pool
NEEDS_NATIVES = ("fp12_mul", "final_exp")


def run_jobs_pool(pool, my_jobs, signatures):
    """`my_jobs` of this rank on a `ProofPool` (starkhip_pool_*): trace generation, the in-flight scheduling and the merged
    commitments all happen inside the library; here the jobs are only submitted and waited for.  Jobs that need no natives
    (the precomputations and Miller loops) are submitted at once; per signature with an fp12_mul / final_exp job here, the natives
    are computed on a host thread (ctypes releases the GIL) and those jobs follow, FinalExp first.
    Returns (results, stats, natives) as `run_jobs` + `job_arguments` do."""
    from concurrent.futures import ThreadPoolExecutor
    t_begin = time.perf_counter()
    tickets, natives = {}, {}
    lock = threading.Lock()
    mine = set(my_jobs)
    need = sorted({i for i, name in mine if name in NEEDS_NATIVES})

    def with_natives(i):
        nat = A.signature_natives(*signatures[i])
        with lock:
            natives[i] = nat
        for name in ("final_exp", "fp12_mul"):
            if (i, name) in mine:
                t = pool.submit_witness(A.JOB_AIR[name], *A.job_operands(name, *signatures[i], natives=nat))
                with lock:
                    tickets[(i, name)] = t

    ex = ThreadPoolExecutor(max_workers=max(1, min(16, len(need)))) if need else None
    futs = [ex.submit(with_natives, i) for i in need] if ex else []
    for i, name in sorted(mine, key=lambda j: -parallel.AIR_COST[A.JOB_AIR[j[1]]]):
        if name not in NEEDS_NATIVES:
            t = pool.submit_witness(A.JOB_AIR[name], *A.job_operands(name, *signatures[i]))
            with lock:
                tickets[(i, name)] = t
    for f in futs:
        f.result()
    if ex:
        ex.shutdown()
    results, timeline = {}, {}
    t_gen = t_prove = 0.0
    for job, t in tickets.items():
        air = A.JOB_AIR[job[1]]
        proof, info = pool.wait(t)
        results[job] = (air, proof, S.StarkConfig.for_air(air))
        tl = info["timeline_s"]
        timeline[job] = tl
        t_gen += tl[2] - tl[1]
        t_prove += tl[4] - tl[3]
    if timeline:  # relative to the first submit of this step
        t0 = min(v[0] for v in timeline.values())
        timeline = {k: [x - t0 for x in v[1:]] for k, v in timeline.items()}
    return results, {"generate_s": t_gen, "prove_s": t_prove, "wall_s": time.perf_counter() - t_begin, "timeline": timeline,
                     "trace_threads": "automatic (pool)", "pool": pool.stats()}, natives


# ------------------------------------------------------------------------------------------------ one step, all ranks
def one_step(dist, batch, provers, mine, signatures, device="cpu", sync=None, **run_kw):
    """ONE step of the signature pipeline on this rank, exactly what `tools/bench_signature.py` times and what the gloo tests
    drive: barrier, [t0] operand broadcast from rank 0 (`signatures` is None on the other ranks), natives for this rank's jobs,
    `run_jobs` (trace generation overlapped with proving), barrier, [t1], max over ranks of t1 - t0.
    `sync()` (e.g. torch.cuda.synchronize) runs right before each barrier so that the region brackets finished GPU work.
    `provers`: a `ProofPool` (the product path: scheduling inside the library, `run_jobs_pool`) or contexts for the Python
    driver `run_jobs` (kept for the CPU tests, which inject prove / generate, and for A/B measurements).
    Returns (elapsed_s, results, stats, sigs, natives); `run_kw` goes to `run_jobs`."""
    if sync is not None:
        sync()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    sigs = broadcast_operands(dist, signatures, batch, device=device)
    if isinstance(provers, S.ProofPool):
        results, stats, natives = run_jobs_pool(provers, mine, sigs)
    else:
        job_args, natives = job_arguments(sigs, mine)
        results, stats = run_jobs(provers, mine, job_args, **run_kw)
    if sync is not None:
        sync()
    if dist is not None:
        dist.barrier()
    elapsed = parallel.max_over_ranks(dist, time.perf_counter() - t0, device=device)
    return elapsed, results, stats, sigs, natives


def collect_results(dist, results, device="cpu"):
    """`run_jobs` results {(i, name): (air, proof, cfg)} of every rank merged on every rank (`aggregate.collect_proofs`)."""
    if dist is None:
        return dict(results)
    flat = {f"{i}:{name}": v for (i, name), v in results.items()}
    return {(int(k.split(":")[0]), k.split(":")[1]): v for k, v in A.collect_proofs(dist, flat, device=device).items()}


def check_signatures(merged, sigs, natives, batch):
    """Per signature whose six proofs are in `merged`: the links between them, the statement (key, H(m), signature, -G, output 1)
    and, where this rank computed the natives, that the proven output is the native one.  Returns {index: bool}."""
    verdicts = {}
    for i in range(batch):
        six = signature_proofs(merged, i)
        if len(six) != 6:
            continue
        ok = A.check_links(six) and A.check_statement(six, sigs[i][0], sigs[i][1], sigs[i][2])
        if i in natives:
            ok = ok and A.signature_is_valid(natives[i], six)
        verdicts[i] = bool(ok)
    return verdicts
