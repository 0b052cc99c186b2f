"""Decoder of the product's flat constraint program (csrc/air_ir.h) into plain Python values, and the canonical
polynomial form shared with tools/rust_subset.py (the extractor that reads the reference's Rust source).  Test code."""
import numpy as np

P = 0xFFFFFFFF00000001
KINDS = ("plain", "transition", "first", "last")
REF_NEXT, REF_COMPL, COL_MASK = 1 << 30, 1 << 31, 0xFFFFFF
PI_FLAG = 1 << 31


def parse_blob(blob):
    b = np.asarray(blob, dtype=np.uint64)
    assert int(b[0]) == 0x3152495F52494153, "not an AIR program blob"
    n_cols, n_pis, degree, n_constraints, n_consts, n_code, n_groups = (int(x) for x in b[1:8])
    consts = [int(x) for x in b[8:8 + n_consts]]
    code = b[8 + n_consts:8 + n_consts + (n_code + 1) // 2].view(np.uint32)[:n_code].tolist()
    return dict(n_cols=n_cols, n_pis=n_pis, degree=degree, n_constraints=n_constraints, consts=consts, code=code, n_groups=n_groups)


def constraints(prog):
    """Yield (kind index, gates [cellref...], terms [(coef, pi index or -1, [cellref...])...]) per constraint, in order."""
    code, consts = prog["code"], prog["consts"]
    i = 0
    while True:
        gw = code[i]
        if gw == 0:
            return
        assert gw & 15 == 1
        kind, ng, m = (gw >> 4) & 3, (gw >> 8) & 255, gw >> 16
        gates = code[i + 1:i + 1 + ng]
        i += 1 + ng
        for _ in range(m):
            terms = []
            while True:
                tw = code[i]
                nf, ck, last, idx = tw & 3, (tw >> 2) & 7, tw & 32, tw >> 6
                cells = code[i + 1:i + 1 + nf]
                i += 1 + nf
                if ck == 0:
                    terms.append((1, -1, cells))
                elif ck == 1:
                    terms.append((P - 1, -1, cells))
                elif ck == 2:
                    terms.append((consts[idx], -1, cells))
                elif ck == 3:
                    terms.append((1, idx, cells))
                elif ck == 4:
                    terms.append((P - 1, idx, cells))
                else:
                    raise ValueError("bad coefficient kind")
                if last:
                    break
            yield kind, gates, terms


def _code(ref):
    return (ref & COL_MASK) | (ref & REF_NEXT)


def expand(gates, terms):
    """The constraint as an expanded polynomial {sorted tuple of variable codes: coefficient}."""
    poly = {}
    for coef, pi, cells in terms:
        if coef == 0:
            continue
        m = [_code(c) for c in cells]
        if pi >= 0:
            m.append(PI_FLAG | pi)
        m = tuple(sorted(m))
        v = (poly.get(m, 0) + coef) % P
        if v:
            poly[m] = v
        else:
            poly.pop(m, None)
    for g in gates:
        gc = _code(g)
        out = {}
        if g & REF_COMPL:   # (1 - g) * poly
            out = dict(poly)
            for m, c in poly.items():
                m2 = tuple(sorted(m + (gc,)))
                v = (out.get(m2, 0) - c) % P
                if v:
                    out[m2] = v
                else:
                    out.pop(m2, None)
        else:
            for m, c in poly.items():
                m2 = tuple(sorted(m + (gc,)))
                v = (out.get(m2, 0) + c) % P
                if v:
                    out[m2] = v
                else:
                    out.pop(m2, None)
        poly = out
    return poly


def canonical(poly):
    out = bytearray()
    for m in sorted(poly):
        out += len(m).to_bytes(4, "little")
        for v in m:
            out += v.to_bytes(4, "little")
        out += poly[m].to_bytes(8, "little")
    return bytes(out)


def evaluate(poly, local, nxt, pis):
    """Value of an expanded polynomial on a frame (Python ints mod P)."""
    acc = 0
    for m, c in poly.items():
        t = c
        for v in m:
            if v & PI_FLAG:
                t = t * int(pis[v & 0x7FFFFFFF]) % P
            elif v & REF_NEXT:
                t = t * int(nxt[v & COL_MASK]) % P
            else:
                t = t * int(local[v]) % P
        acc = (acc + t) % P
    return acc
