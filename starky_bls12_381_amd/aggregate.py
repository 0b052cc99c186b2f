"""Host drivers around the hot path: the STARK half of the reference's `generate_aggregate_proof`.

Mirrors /root/reference/src/aggregate_proof.rs:
  calc_pairing_precomp    :23-72    PairingPrecompStark(1024), rate_bits 2
  miller_loop_main        :74-118   MillerLoopStark(1024)
  fp12_mul_main           :120-151  FP12MulStark(16)
  final_exponentiate_main :153-179  FinalExponentiateStark(8192), rate_bits 2
each = build config, generate trace + public inputs (recorded as runs and expanded on the device, SURVEY §8f-2: the
4.8 GB of FinalExp rows never exist on the host), prove, verify_stark_proof, return (air, proof, config);
and the six-proof plan of one BLS signature check (:304-370): pp1, ml1 on (aggregate pk, H(m)), pp2, ml2 on
(-G1 generator, signature), fp12_mul on the two Miller-loop values, final_exponentiate on their product.
  ec_aggregate_main       :181-221  ECCAggStark(8192), rate_bits 2 (the aggregate public key the pairing proofs take as pk)
Out of scope here (SURVEY.md §8f / DESIGN.md §8): hash-to-curve, key decompression, the plonky2 recursion.

Points are passed as u32 limb arrays (Fp = 12 limbs, Fp2 = 24: c0 then c1), exactly what the reference's
`get_u32_slice()` yields.  `prover` is an `api.Prover` bound to one GPU; nothing here touches the oracle.
"""
import numpy as np

from . import api as S
from . import parallel

# -G1 generator, the fixed first argument of the second pairing (src/aggregate_proof.rs:336-337)
NEG_G1_X = 3685416753713387016781088315183077757961620795782546409894578378688607592378376318836054947676345821548104185464507
NEG_G1_Y = 2662903010277190920397318445793982934971948944000658264905514399707520226534504357969962973775649129045502516118218

JOB_ORDER = ("pp1", "ml1", "pp2", "ml2", "fp12_mul", "final_exp")  # the order the reference proves them in
JOB_AIR = {"pp1": S.AIR_PAIRING_PRECOMP, "pp2": S.AIR_PAIRING_PRECOMP, "ml1": S.AIR_MILLER_LOOP, "ml2": S.AIR_MILLER_LOOP,
           "fp12_mul": S.AIR_FP12_MUL, "final_exp": S.AIR_FINAL_EXP}


def fp_limbs(v):
    """One Fp as 12 little-endian u32 limbs (src/native.rs Fp([u32; 12]))."""
    v = int(v)
    return np.array([(v >> (32 * i)) & 0xFFFFFFFF for i in range(12)], dtype=np.uint32)


def fp2_limbs(c0, c1):
    return np.concatenate([fp_limbs(c0), fp_limbs(c1)])


def _prove_and_verify(prover, air, trace, pis):
    cfg = S.StarkConfig.for_air(air)
    proof = prover.prove(air, cfg, trace, pis)
    S.verify_stark_proof(air, cfg, proof)  # src/aggregate_proof.rs:67,113,146,177
    return air, proof, cfg


def calc_pairing_precomp(prover, x, y, z):
    """src/aggregate_proof.rs:23-72.  x, y, z: Fp2 limb arrays of the G2 point."""
    trace, pis = S.trace_pairing_precomp(x, y, z, compact=True)
    return _prove_and_verify(prover, S.AIR_PAIRING_PRECOMP, trace, pis)


def miller_loop_main(prover, x, y, q_x, q_y, q_z):
    """src/aggregate_proof.rs:74-118.  (x, y): G1 point (Fp limbs); (q_x, q_y, q_z): G2 point (Fp2 limbs)."""
    trace, pis = S.trace_miller_loop(x, y, q_x, q_y, q_z, compact=True)
    return _prove_and_verify(prover, S.AIR_MILLER_LOOP, trace, pis)


def fp12_mul_main(prover, x, y):
    """src/aggregate_proof.rs:120-151."""
    trace, pis = S.trace_fp12_mul(x, y, compact=True)
    return _prove_and_verify(prover, S.AIR_FP12_MUL, trace, pis)


def final_exponentiate_main(prover, x):
    """src/aggregate_proof.rs:153-179."""
    trace, pis = S.trace_final_exp(x, compact=True)
    return _prove_and_verify(prover, S.AIR_FINAL_EXP, trace, pis)


def ec_aggregate_main(prover, points, bits):
    """src/aggregate_proof.rs:181-221: ECCAggStark over the 512 sync-committee keys and their participation bits.
    points: [512][24] u32 limbs (x, y); the aggregate (the `pk` of the pairing proofs) is the last 24 public inputs."""
    trace, pis = S.trace_ecc_aggregate(points, bits, compact=True)
    return _prove_and_verify(prover, S.AIR_ECC_AGGREGATE, trace, pis)


def signature_natives(pk, hm, sig):
    """The native values that link the six proofs (src/aggregate_proof.rs:352-363): the two Miller-loop values, their product
    and its final exponentiation, as Fp12 limbs."""
    neg_g = (fp_limbs(NEG_G1_X), fp_limbs(NEG_G1_Y))
    ml1 = S.native_miller_loop(pk[0], pk[1], hm[0], hm[1], hm[2])
    ml2 = S.native_miller_loop(neg_g[0], neg_g[1], sig[0], sig[1], sig[2])
    product = S.native_fp12_mul(ml1, ml2)
    return {"ml1": ml1, "ml2": ml2, "product": product, "final": S.native_final_exponentiate(product)}


def job_operands(name, pk, hm, sig, natives=None):
    """Generator arguments of job `name` of one signature (the operands `ProofPool.submit_witness` packs); the fp12_mul and
    final_exp jobs need `natives`."""
    neg_g = (fp_limbs(NEG_G1_X), fp_limbs(NEG_G1_Y))
    if name == "pp1":
        return (hm[0], hm[1], hm[2])
    if name == "pp2":
        return (sig[0], sig[1], sig[2])
    if name == "ml1":
        return (pk[0], pk[1], hm[0], hm[1], hm[2])
    if name == "ml2":
        return (neg_g[0], neg_g[1], sig[0], sig[1], sig[2])
    if name == "fp12_mul":
        return (natives["ml1"], natives["ml2"])
    if name == "final_exp":
        return (natives["product"],)
    raise KeyError(name)


def signature_jobs(pk, hm, sig):
    """The six proof jobs of one signature check, in the reference's order, with the natives that link them.

    pk = (x, y) Fp limbs of the aggregate public key; hm, sig = (x, y, z) Fp2 limbs of H(m) and the signature.
    Returns (jobs, natives): jobs[name] = (driver function, args); natives = {"ml1", "ml2", "product", "final"} Fp12 limbs.
    The Miller-loop values are computed natively first, as the reference does at :352-353, so that all six jobs are
    independent and can be proven on different GPUs."""
    neg_g = (fp_limbs(NEG_G1_X), fp_limbs(NEG_G1_Y))
    natives = signature_natives(pk, hm, sig)
    ml1, ml2, product = natives["ml1"], natives["ml2"], natives["product"]
    jobs = {
        "pp1": (calc_pairing_precomp, (hm[0], hm[1], hm[2])),
        "ml1": (miller_loop_main, (pk[0], pk[1], hm[0], hm[1], hm[2])),
        "pp2": (calc_pairing_precomp, (sig[0], sig[1], sig[2])),
        "ml2": (miller_loop_main, (neg_g[0], neg_g[1], sig[0], sig[1], sig[2])),
        "fp12_mul": (fp12_mul_main, (ml1, ml2)),
        "final_exp": (final_exponentiate_main, (product,)),
    }
    return jobs, natives


def signature_is_valid(natives, proofs=None):
    """e(pk, H(m)) * e(-G, sig) == 1  <=>  final_exponentiate(ml1 * ml2) == 1 (src/native.rs:1522-1526).

    With `proofs` (the six finished proofs) the verdict is read from what was PROVEN -- the output public inputs of the
    final_exp proof, as the reference's recursion does (src/aggregate_proof.rs:590-598) -- and the host natives must agree
    with it; without, it is the host natives' value only (a scheduling aid before anything is proven)."""
    one = np.zeros(144, dtype=np.uint32)
    one[0] = 1
    native_ok = bool(np.array_equal(np.asarray(natives["final"], dtype=np.uint32), one))
    if proofs is None:
        return native_ok
    fe = _pis(proofs, "final_exp")
    if not np.array_equal(fe[144:288], np.asarray(natives["final"], dtype=np.uint64)):
        raise ValueError("final_exp proof attests to a different value than the host natives computed")
    return bool(np.array_equal(fe[144:288], one.astype(np.uint64)))


def _pis(proofs, name):
    air, proof, _ = proofs[name]
    n = S.air_public_inputs(air)
    return np.asarray(proof[-n:], dtype=np.uint64)


def check_statement(proofs, pk, hm, sig):
    """The bindings the reference's recursive aggregation puts on the public inputs, besides the cross-proof links of
    `check_links` (src/aggregate_proof.rs:507-520, 540-545, 552-568, 576-581, 590-598): the first precompute ran on H(m) with
    Z = (1, 0); the second on the signature point with Z = (1, 0); the FIRST Miller loop's G1 operand is the aggregate public
    key -- the reference ties ml1's (PX, PY) to the key through the ECCAggStark proof's last 24 public inputs (:540-545); here
    `pk` = (x, y) Fp limb arrays is that key, or None when an "ec" proof is in `proofs` and carries it -- the second Miller
    loop's G1 operand is -G; the final exponentiation's output is 1.  hm, sig: (x, y[, z]) Fp2 limb arrays of the points the
    statement is about.  Proofs that verify and link but attest to another key or other points make this False: without the
    key binding six proofs made for any pk' with e(pk', H(m)) * e(-G, sig) = 1 would pass for (hm, sig)."""
    one = np.zeros(24, dtype=np.uint64)
    one[0] = 1
    pp1, pp2, ml1, ml2, fe = _pis(proofs, "pp1"), _pis(proofs, "pp2"), _pis(proofs, "ml1"), _pis(proofs, "ml2"), _pis(proofs, "final_exp")
    if pk is None:
        if "ec" not in proofs:
            raise ValueError("check_statement needs the public key: pass pk, or include the 'ec' (ECCAggStark) proof that publishes it")
        key = _pis(proofs, "ec")[-24:]
    else:
        key = np.concatenate([np.asarray(pk[0], dtype=np.uint64), np.asarray(pk[1], dtype=np.uint64)])
        if "ec" in proofs and not np.array_equal(_pis(proofs, "ec")[-24:], key):
            return False
    if not np.array_equal(ml1[0:24], key):
        return False
    ok = bool(np.array_equal(pp1[0:24], np.asarray(hm[0], dtype=np.uint64)) and np.array_equal(pp1[24:48], np.asarray(hm[1], dtype=np.uint64)))
    ok &= bool(np.array_equal(pp1[48:72], one))
    ok &= bool(np.array_equal(pp2[0:24], np.asarray(sig[0], dtype=np.uint64)) and np.array_equal(pp2[24:48], np.asarray(sig[1], dtype=np.uint64)))
    ok &= bool(np.array_equal(pp2[48:72], one))
    ok &= bool(np.array_equal(ml2[0:12], fp_limbs(NEG_G1_X).astype(np.uint64)) and np.array_equal(ml2[12:24], fp_limbs(NEG_G1_Y).astype(np.uint64)))
    fe_one = np.zeros(144, dtype=np.uint64)
    fe_one[0] = 1
    ok &= bool(np.array_equal(fe[144:288], fe_one))
    return ok


def check_links(proofs):
    """The cross-proof equalities the reference's recursive aggregation enforces on public inputs:
    ell_coeffs produced by pp_i are the ones ml_i consumed; ml outputs are the fp12_mul inputs; its output is the
    final_exp input.  `proofs[name]` = (air, proof, cfg); public inputs are the tail of each proof blob."""
    def pis(name):
        return _pis(proofs, name)
    ok = True
    for i in ("1", "2"):
        pp, ml = pis("pp" + i), pis("ml" + i)
        ell = pp[72:]                 # after x, y, z (3 Fp2 = 72 limbs): 68 x 3 Fp2 coefficients
        ok &= bool(np.array_equal(ell, ml[24:24 + ell.size]))   # after px, py (2 Fp = 24 limbs)
    if "ec" in proofs:  # the aggregate ECCAggStark publishes is the G1 operand of the first Miller loop
        ok &= bool(np.array_equal(pis("ec")[-24:], pis("ml1")[0:24]))
    mul = pis("fp12_mul")
    ok &= bool(np.array_equal(pis("ml1")[-144:], mul[0:144]))
    ok &= bool(np.array_equal(pis("ml2")[-144:], mul[144:288]))
    ok &= bool(np.array_equal(mul[288:432], pis("final_exp")[0:144]))
    return ok


def signature_plan(world, names=JOB_ORDER):
    """Per-rank job names: longest-processing-time-first on the per-AIR cost weights (parallel.AIR_COST)."""
    plan = parallel.assign_jobs([parallel.AIR_COST[JOB_AIR[n]] for n in names], world)
    return [[names[i] for i in idxs] for idxs in plan]


def prove_signature(prover, pk, hm, sig, dist=None, names=JOB_ORDER):
    """Prove this rank's share of the six proofs of one signature check.

    Jobs are assigned with longest-processing-time scheduling on the per-AIR cost weights (parallel.AIR_COST):
    one process per GPU, no collective on the data path; every rank derives the same plan from the same inputs.
    Returns {name: (air, proof, cfg)} for the jobs this rank ran, and the natives."""
    rank, _, world = parallel.rank_info() if dist is not None else (0, 0, 1)
    jobs, natives = signature_jobs(pk, hm, sig)
    out = {}
    for name in signature_plan(world, names)[rank]:
        fn, args = jobs[name]
        out[name] = fn(prover, *args)
    return out, natives


def collect_proofs(dist, mine, device="cpu"):
    """Every rank's share {name: (air, proof, cfg)} merged into one dict on every rank, so that any of them can run
    `check_links` / `check_statement` and hand the six proofs to the recursion stage.  Proofs are tens of MB each (FinalExp
    52 MB, MillerLoop 68 MB): the ranks first exchange a small table (name, AIR, words) and then every proof travels ONCE as a
    raw u64 buffer, broadcast by the rank that made it -- never pickled.  This is result collection after the timed data
    path, not a data-path collective."""
    if dist is None:
        return dict(mine)
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    table = [(name, int(air), int(np.asarray(proof).size)) for name, (air, proof, _) in sorted(mine.items())]
    tables = [None] * world
    dist.all_gather_object(tables, table)  # a few dozen bytes per proof
    merged = {}
    for src, part in enumerate(tables):
        for name, air, words in part:
            if name in merged:
                raise ValueError(f"proof {name!r} was produced by more than one rank")
            if src == rank:
                buf = torch.from_numpy(np.ascontiguousarray(mine[name][1], dtype=np.uint64).view(np.int64).copy()).to(device)
            else:
                buf = torch.empty(words, dtype=torch.int64, device=device)
            dist.broadcast(buf, src=src)
            merged[name] = (air, buf.cpu().numpy().view(np.uint64), S.StarkConfig.for_air(air))
    return merged
