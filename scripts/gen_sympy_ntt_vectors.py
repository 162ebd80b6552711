#!/usr/bin/env python3
"""Golden vectors for the subgroup NTT over Goldilocks from an INDEPENDENT implementation: sympy.discrete.transforms.ntt.

sympy's `ntt(seq, prime)` computes X[k] = sum_j x[j] w^(j k) mod p with w = g^((p - 1) / n), g = primitive_root(p) = 7 for
p = 2^64 - 2^32 + 1: the root arkworks derives (GENERATOR = 7, TWO_ADIC_ROOT_OF_UNITY = 7^((p-1)/2^32), root of a size-n
domain = its 2^32/n-th power = 7^((p-1)/n); gpu/src/plan.rs:386-398 takes group_gen from that domain).  It is not the reference
(the reference stores no FFT vectors and cannot be built here), but it shares no code and no author with
oracle/pyref and oracle/c.

    python scripts/gen_sympy_ntt_vectors.py       # needs sympy (build container); writes tests/golden/sympy_ntt_goldilocks.json

Inputs come from a 64-bit LCG restated in the fixture, so tests regenerate them without sympy; outputs are stored whole up to
2^8 points and as SHA-256 of the little-endian words above (2^10, 2^12), forward and inverse."""
import hashlib
import json
import os
import struct

from sympy import primitive_root
from sympy.discrete.transforms import intt, ntt

P = (1 << 64) - (1 << 32) + 1
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sympy_ntt_goldilocks.json")
LCG_A, LCG_C = 6364136223846793005, 1442695040888963407


def inputs(n, seed):
    s, out = seed, []
    for _ in range(n):
        s = (s * LCG_A + LCG_C) % (1 << 64)
        out.append(s % P)
    return out


def digest(v):
    return hashlib.sha256(b"".join(struct.pack("<Q", x) for x in v)).hexdigest()


def main():
    assert primitive_root(P) == 7
    cases = []
    for log_n in (4, 6, 8, 10, 12):
        n = 1 << log_n
        x = inputs(n, 0x6d696e69 + log_n)
        fwd = [int(v) for v in ntt(x, P)]
        inv = [int(v) for v in intt(x, P)]
        assert [int(v) for v in intt(fwd, P)] == x
        case = {"log_n": log_n, "seed": 0x6d696e69 + log_n, "forward_sha256": digest(fwd), "inverse_sha256": digest(inv)}
        if log_n <= 8:
            case["forward"] = [format(v, "016x") for v in fwd]
            case["inverse"] = [format(v, "016x") for v in inv]
        cases.append(case)
    doc = {"source": "sympy %s sympy.discrete.transforms.ntt / intt, prime 2^64 - 2^32 + 1, primitive root 7" % __import__("sympy").__version__,
           "inputs": "s <- (s * %d + %d) mod 2^64 from the seed; x = s mod p (canonical integers, NOT Montgomery words)" % (LCG_A, LCG_C),
           "lcg": [LCG_A, LCG_C], "cases": cases}
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")
    print("wrote", OUT, [c["log_n"] for c in cases])


if __name__ == "__main__":
    main()
