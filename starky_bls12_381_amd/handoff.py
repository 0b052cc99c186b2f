"""Proof hand-off format: the proof blob of include/starkhip.h as the nested value
`StarkProofWithPublicInputs<GoldilocksField, PoseidonGoldilocksConfig, 2>` has under serde -- what a Rust shim would
`serde_json::from_str` and pass to `starky::recursive_verifier::*` (src/aggregate_proof.rs:435-439) without re-proving.

Shape (field names and nesting of starky 0.1.x / plonky2 0.1.x, restated from memory of the crates, SURVEY App. A.9;
they are NOT under /root/reference and there is no Rust toolchain here, so this mapping is unpinned):
  {"proof": {"trace_cap": [HashOut..], "permutation_zs_cap": null, "quotient_polys_cap": [HashOut..],
             "openings": {"local_values": [Ext..], "next_values": [Ext..], "permutation_zs": null,
                          "permutation_zs_next": null, "quotient_polys": [Ext..]},
             "opening_proof": {"commit_phase_merkle_caps": [[HashOut..]..],
                               "query_round_proofs": [{"initial_trees_proof": {"evals_proofs": [[[F..], {"siblings": [HashOut..]}], ..]},
                                                       "steps": [{"evals": [Ext..], "merkle_proof": {"siblings": [HashOut..]}}, ..]}, ..],
                               "final_poly": {"coeffs": [Ext..]}, "pow_witness": F}},
   "public_inputs": [F..]}
with F = canonical u64 (GoldilocksField is a transparent newtype), Ext = [a0, a1] (QuadraticExtension is a newtype over
[F; 2]), HashOut = {"elements": [F; 4]}, MerkleCap a newtype over Vec<HashOut>.  Python ints keep all 64 bits;
`dumps` writes them as JSON numbers, as serde_json does for u64.
"""
import json

import numpy as np

MAGIC = 0x3130304652505353


class _Layout:
    def __init__(self, blob):
        h = [int(x) for x in blob[:16]]
        if len(blob) < 16 or h[0] != MAGIC:
            raise ValueError("not a starkhip proof blob")
        (self.C, self.Q, self.log_n, self.rate_bits, self.cap_h, self.L, self.n_queries, self.final_len, self.n_pis, self.arity_bits,
         self.n_challenges) = h[1:12]
        self.ncap = 1 << self.cap_h
        self.log_N = self.log_n + self.rate_bits
        self.depth0 = self.log_N - self.cap_h
        lg = self.log_N
        self.layer_depth = []
        for _ in range(self.L):
            lg -= self.arity_bits
            self.layer_depth.append(lg - self.cap_h)

    def header(self):
        return [MAGIC, self.C, self.Q, self.log_n, self.rate_bits, self.cap_h, self.L, self.n_queries, self.final_len, self.n_pis, self.arity_bits,
                self.n_challenges, 0, 0, 0, 0]


def _hashes(words):
    return [{"elements": [int(x) for x in words[i:i + 4]]} for i in range(0, len(words), 4)]


def _exts(words):
    return [[int(words[i]), int(words[i + 1])] for i in range(0, len(words), 2)]


def proof_to_value(blob):
    """Proof blob (numpy uint64) -> nested dict / list value in serde's shape."""
    blob = np.asarray(blob, dtype=np.uint64)
    lay = _Layout(blob)
    pos = 16

    def take(n):
        nonlocal pos
        out = blob[pos:pos + n]
        if len(out) != n:
            raise ValueError("truncated proof blob")
        pos += n
        return out
    trace_cap = _hashes(take(4 * lay.ncap))
    quot_cap = _hashes(take(4 * lay.ncap))
    local_values = _exts(take(2 * lay.C))
    next_values = _exts(take(2 * lay.C))
    quotient_polys = _exts(take(2 * lay.Q))
    caps = [_hashes(take(4 * lay.ncap)) for _ in range(lay.L)]
    rounds = []
    for _ in range(lay.n_queries):
        evals_proofs = []
        for width in (lay.C, lay.Q):  # oracle 0 = trace, oracle 1 = quotient polys (no permutation oracle)
            leaf = [int(x) for x in take(width)]
            evals_proofs.append([leaf, {"siblings": _hashes(take(4 * lay.depth0))}])
        steps = []
        for d in lay.layer_depth:
            evals = _exts(take(2 << lay.arity_bits))
            steps.append({"evals": evals, "merkle_proof": {"siblings": _hashes(take(4 * d))}})
        rounds.append({"initial_trees_proof": {"evals_proofs": evals_proofs}, "steps": steps})
    final_poly = {"coeffs": _exts(take(2 * lay.final_len))}
    pow_witness = int(take(1)[0])
    public_inputs = [int(x) for x in take(lay.n_pis)]
    if pos != len(blob):
        raise ValueError("trailing words in proof blob")
    return {
        "proof": {
            "trace_cap": trace_cap,
            "permutation_zs_cap": None,
            "quotient_polys_cap": quot_cap,
            "openings": {"local_values": local_values, "next_values": next_values, "permutation_zs": None, "permutation_zs_next": None,
                         "quotient_polys": quotient_polys},
            "opening_proof": {"commit_phase_merkle_caps": caps, "query_round_proofs": rounds, "final_poly": final_poly,
                              "pow_witness": pow_witness},
        },
        "public_inputs": public_inputs,
    }


def value_to_proof(value, degree_bits, rate_bits, arity_bits=4, num_challenges=2):
    """Inverse of proof_to_value; the blob header needs the two sizes the nested value does not carry."""
    p = value["proof"]
    op, fri = p["openings"], p["opening_proof"]
    C, Q = len(op["local_values"]), len(op["quotient_polys"])
    cap_h = (len(p["trace_cap"]) - 1).bit_length()
    words = [MAGIC, C, Q, degree_bits, rate_bits, cap_h, len(fri["commit_phase_merkle_caps"]), len(fri["query_round_proofs"]),
             len(fri["final_poly"]["coeffs"]), len(value["public_inputs"]), arity_bits, num_challenges, 0, 0, 0, 0]

    def put_hashes(hs):
        for h in hs:
            words.extend(h["elements"])

    def put_exts(es):
        for e in es:
            words.extend(e)
    put_hashes(p["trace_cap"])
    put_hashes(p["quotient_polys_cap"])
    put_exts(op["local_values"])
    put_exts(op["next_values"])
    put_exts(op["quotient_polys"])
    for cap in fri["commit_phase_merkle_caps"]:
        put_hashes(cap)
    for r in fri["query_round_proofs"]:
        for leaf, mp in r["initial_trees_proof"]["evals_proofs"]:
            words.extend(leaf)
            put_hashes(mp["siblings"])
        for st in r["steps"]:
            put_exts(st["evals"])
            put_hashes(st["merkle_proof"]["siblings"])
    put_exts(fri["final_poly"]["coeffs"])
    words.append(fri["pow_witness"])
    words.extend(value["public_inputs"])
    return np.array(words, dtype=np.uint64)


def dumps(blob):
    """Proof blob -> JSON text in serde_json's form of StarkProofWithPublicInputs."""
    return json.dumps(proof_to_value(blob), separators=(",", ":"))


def loads(text, degree_bits, rate_bits, arity_bits=4, num_challenges=2):
    return value_to_proof(json.loads(text), degree_bits, rate_bits, arity_bits, num_challenges)
