"""Pins the oracle (oracle/pyref + oracle/c) against every literal
known-answer the reference carries in-tree for this path, and
the two oracle implementations against each other on the reference's own
test shapes (gpu/tests/shaders.rs:17-117: n = 2048, 4096, 65536; subgroup and
coset offset = GENERATOR)."""
import hashlib

import numpy as np
import pytest

from oracle import cref
from oracle.pyref import fields as F
from oracle.pyref import fri as pyfri
from oracle.pyref import merkle as pymerkle
from oracle.pyref import ntt as pyntt

GL = F.GL
P = F.GL_P


# ---- literal constants in the reference tree ---------------------------------
def test_bit_reverse_kat():
    # gpu/src/utils.rs:227-236
    assert F.bit_reverse(list(range(16))) == [0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15]
    a = np.arange(16, dtype=np.uint64)
    assert cref.bit_reverse(a, 4).tolist() == [0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15]


def test_goldilocks_constants():
    assert P == 18446744069414584321                       # felt_u64.h.metal:124
    assert GL.R == 4294967295                              # felt_u64.h.metal:118
    assert F.GL_R2 == 18446744065119617025                 # felt_u64.h.metal:127
    assert GL.to_mont(2) == 8589934590                     # gpu/src/fields.rs:82
    assert GL.from_mont(2305843009213693952) == 16140901060737761281   # fields.rs:85-88
    assert (P ** 3 - 1) // 2 ** 32 == 1461501636310055817916238417282618014431694553085  # fields.rs:75-76
    assert F.GL_TWO_ADIC_ROOT == 1753635133440165772
    assert pow(F.GL_TWO_ADIC_ROOT, 1 << 32, P) == 1 and pow(F.GL_TWO_ADIC_ROOT, 1 << 31, P) == P - 1
    L = cref.lib()
    assert L.oracle_gl_to_mont(2) == 8589934590
    assert L.oracle_gl_to_mont(1) == 4294967295
    assert L.oracle_gl_from_mont(2305843009213693952) == 16140901060737761281


def _u256(hi, a, b, lo):
    return (hi << 192) | (a << 128) | (b << 64) | lo


def test_fp252_constants():
    # felt_u256.h.metal:101-108 (u256(high .. low) limb order)
    N = _u256(576460752303423505, 0, 0, 1)
    assert N == F.F252_P
    assert _u256(576460752303422960, 18446744073709551615, 18446744073709551615, 18446744073709551585) == F.F252_R
    assert _u256(576413109808302096, 18446744073700081664, 5151653887, 18446741271209837569) == F.F252_R2
    nprime = _u256(576460752303423504, 18446744073709551615, 18446744073709551615, 18446744073709551615)
    assert (nprime * N) % (1 << 256) == (1 << 256) - 1     # N' = -N^-1 mod R


def test_fq3_nonresidue_is_cubic_nonresidue():
    # x^3 - 2 irreducible  <=>  2 is not a cube:  2^((p-1)/3) != 1
    assert pow(2, (P - 1) // 3, P) != 1
    a = (123456789, 987654321, 555)
    assert F.FQ3.mul(a, F.FQ3.inv(a)) == (1, 0, 0)


def test_sha256_matches_hashlib():
    rng = np.random.default_rng(1)
    for ln in [0, 1, 55, 56, 63, 64, 65, 119, 120, 128, 208, 256, 1000]:
        msg = rng.integers(0, 256, size=ln, dtype=np.uint8)
        out = np.zeros(32, dtype=np.uint8)
        cref.lib().oracle_sha256(cref._p8(msg) if ln else cref._p8(np.zeros(1, np.uint8)), ln, cref._p8(out))
        assert out.tobytes() == hashlib.sha256(msg.tobytes()).digest()


# ---- the two oracle implementations agree, and agree with the definition ------
def test_pyref_ntt_is_the_dft():
    rng = np.random.default_rng(2)
    for n in [1, 2, 4, 8, 32, 64]:
        v = [int(x) for x in rng.integers(0, P, size=n, dtype=np.uint64)]
        w = GL.root_of_unity(n)
        assert pyntt.ntt(GL, v, w) == pyntt.dft_naive(GL, v, w)
    v3 = [tuple(int(x) for x in rng.integers(0, P, size=3, dtype=np.uint64)) for _ in range(16)]
    w = GL.root_of_unity(16)
    assert pyntt.ntt(GL, v3, w) == pyntt.dft_naive(GL, v3, w)


def test_pyref_fft_is_polynomial_evaluation():
    rng = np.random.default_rng(3)
    n = 64
    coeffs = [int(x) for x in rng.integers(0, P, size=n, dtype=np.uint64)]
    for off in (1, 7):
        d = pyntt.Domain(GL, n, off)
        ev = pyntt.fft(d, coeffs)
        for i in (0, 1, 17, 63):
            assert ev[i] == pyntt.horner(GL, coeffs, d.element(i))
        assert pyntt.ifft(d, ev) == coeffs


def _mont_list(a):
    return [GL.to_mont(int(x)) for x in a]


@pytest.mark.parametrize("n,offset", [(2048, 1), (4096, 1), (65536, 1), (2048, 7), (4096, 7)])
def test_c_oracle_fft_matches_pyref_fp(n, offset):
    # shapes of gpu/tests/shaders.rs:17-40
    rng = np.random.default_rng(n + offset)
    canon = rng.integers(0, P, size=n, dtype=np.uint64)
    d = pyntt.Domain(GL, n, offset)
    want = pyntt.fft(d, [int(x) for x in canon])
    got = cref.ntt(np.array(_mont_list(canon), dtype=np.uint64), n.bit_length() - 1, 1, False, offset)
    assert [GL.from_mont(int(x)) for x in got] == want


@pytest.mark.parametrize("n,offset", [(2048, 1), (4096, 7)])
def test_c_oracle_ifft_matches_pyref_fp(n, offset):
    # gpu/tests/shaders.rs:94-117
    rng = np.random.default_rng(n * 3 + offset)
    canon = rng.integers(0, P, size=n, dtype=np.uint64)
    d = pyntt.Domain(GL, n, offset)
    want = pyntt.ifft(d, [int(x) for x in canon])
    got = cref.ntt(np.array(_mont_list(canon), dtype=np.uint64), n.bit_length() - 1, 1, True, offset)
    assert [GL.from_mont(int(x)) for x in got] == want


@pytest.mark.parametrize("n,offset", [(2048, 1), (2048, 7)])
def test_c_oracle_fft_matches_pyref_fq3(n, offset):
    # gpu/tests/shaders.rs:43-66
    rng = np.random.default_rng(n + 11 * offset)
    canon = rng.integers(0, P, size=3 * n, dtype=np.uint64)
    elems = [tuple(int(x) for x in canon[3 * i:3 * i + 3]) for i in range(n)]
    d = pyntt.Domain(GL, n, offset)
    want = pyntt.fft(d, elems)
    got = cref.ntt(np.array(_mont_list(canon), dtype=np.uint64), n.bit_length() - 1, 3, False, offset)
    got = [GL.from_mont(int(x)) for x in got]
    assert [tuple(got[3 * i:3 * i + 3]) for i in range(n)] == want


def test_c_oracle_lde_matches_pyref():
    rng = np.random.default_rng(5)
    n, blow = 256, 8
    canon = rng.integers(0, P, size=n, dtype=np.uint64)
    want = pyntt.lde_bit_reversed(GL, [int(x) for x in canon], blow, 7)
    got = cref.lde(np.array(_mont_list(canon), dtype=np.uint64), 8, 3, 1, 7, True)
    assert [GL.from_mont(int(x)) for x in got] == want


def test_c_oracle_merkle_matches_pyref():
    rng = np.random.default_rng(6)
    nrows, ncols = 64, 5
    canon = rng.integers(0, P, size=(ncols, nrows), dtype=np.uint64)
    cols_m = [np.array(_mont_list(c), dtype=np.uint64) for c in canon]
    leaves = cref.sha256_rows(cols_m, 1)
    want_leaves = pymerkle.hash_rows(GL, [[int(x) for x in c] for c in canon])
    assert [bytes(l) for l in leaves] == want_leaves
    nodes = cref.sha256_merkle(leaves)
    want_nodes = pymerkle.build_merkle_nodes(want_leaves)
    assert [bytes(x) for x in nodes] == want_nodes
    # Fq3 rows: c0||c1||c2 per element
    canon3 = rng.integers(0, P, size=(2, nrows * 3), dtype=np.uint64)
    cols3 = [np.array(_mont_list(c), dtype=np.uint64) for c in canon3]
    leaves3 = cref.sha256_rows(cols3, 3)
    elems = [[tuple(int(x) for x in c[3 * i:3 * i + 3]) for i in range(nrows)] for c in canon3]
    assert [bytes(l) for l in leaves3] == pymerkle.hash_rows(F.FQ3, elems)


@pytest.mark.parametrize("ff", [2, 4, 8, 16])
def test_c_oracle_fri_fold_matches_pyref(ff):
    rng = np.random.default_rng(7 + ff)
    n = 256
    canon = rng.integers(0, P, size=n, dtype=np.uint64)
    alpha = int(rng.integers(0, P, dtype=np.uint64))
    want = pyfri.apply_drp(GL, None, [int(x) for x in canon], 1, alpha, ff)
    got = cref.fri_fold(np.array(_mont_list(canon), dtype=np.uint64), 8, 1, ff,
                        np.array([GL.to_mont(alpha)], dtype=np.uint64), 1)
    assert [GL.from_mont(int(x)) for x in got] == want


def test_c_oracle_fri_fold_fq3_matches_pyref():
    rng = np.random.default_rng(17)
    n, ff = 128, 4
    canon = rng.integers(0, P, size=3 * n, dtype=np.uint64)
    alpha = tuple(int(x) for x in rng.integers(0, P, size=3, dtype=np.uint64))
    elems = [tuple(int(x) for x in canon[3 * i:3 * i + 3]) for i in range(n)]
    want = pyfri.apply_drp(GL, F.FQ3, elems, 1, alpha, ff)
    got = cref.fri_fold(np.array(_mont_list(canon), dtype=np.uint64), 7, 3, ff,
                        np.array(_mont_list(alpha), dtype=np.uint64), 1)
    got = [GL.from_mont(int(x)) for x in got]
    assert [tuple(got[3 * i:3 * i + 3]) for i in range(n // ff)] == want


def test_c_oracle_field_ops_match_pyref():
    rng = np.random.default_rng(8)
    L = cref.lib()
    edge = [0, 1, 2, P - 1, P - 2, (1 << 32) - 1, 1 << 32, (1 << 63), GL.R, F.GL_R2]
    vals = edge + [int(x) for x in rng.integers(0, P, size=200, dtype=np.uint64)]
    for a in vals[:40]:
        for b in vals[:40]:
            am, bm = GL.to_mont(a), GL.to_mont(b)
            assert GL.from_mont(L.oracle_gl_mul(am, bm)) == GL.mul(a, b)
            assert GL.from_mont(L.oracle_gl_add(am, bm)) == GL.add(a, b)
            assert GL.from_mont(L.oracle_gl_sub(am, bm)) == GL.sub(a, b)
    for a in vals[1:60]:
        assert GL.from_mont(L.oracle_gl_inv(GL.to_mont(a))) == GL.inv(a)
    # Fq3 mul / inv
    for _ in range(50):
        a = tuple(int(x) for x in rng.integers(0, P, size=3, dtype=np.uint64))
        b = tuple(int(x) for x in rng.integers(0, P, size=3, dtype=np.uint64))
        am = np.array(_mont_list(a), dtype=np.uint64)
        bm = np.array(_mont_list(b), dtype=np.uint64)
        out = np.zeros(3, dtype=np.uint64)
        L.oracle_fq3_mul(cref._p(am), cref._p(bm), cref._p(out))
        assert tuple(GL.from_mont(int(x)) for x in out) == F.FQ3.mul(a, b)
        L.oracle_fq3_inv(cref._p(am), cref._p(out))
        assert tuple(GL.from_mont(int(x)) for x in out) == F.FQ3.inv(a)


# ---- an independent implementation as a second pin (round 3): sympy's number-theoretic transform, tests/golden/sympy_ntt_goldilocks.json
def _sympy_cases():
    import json
    import os
    doc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sympy_ntt_goldilocks.json")))
    a, c = doc["lcg"]
    for case in doc["cases"]:
        s, x = case["seed"], []
        for _ in range(1 << case["log_n"]):
            s = (s * a + c) % (1 << 64)
            x.append(s % F.GL.p)
        yield case, x


def _le_digest(v):
    import struct
    return hashlib.sha256(b"".join(struct.pack("<Q", int(e)) for e in v)).hexdigest()


def test_both_oracles_match_sympys_ntt():
    """oracle/pyref (big integers) and oracle/c (Montgomery words) against vectors produced by sympy.discrete.transforms.ntt / intt
    (scripts/gen_sympy_ntt_vectors.py): same prime, same root 7^((p-1)/n) as arkworks' domain, subgroup transforms 2^4 .. 2^12,
    forward and inverse; plus the coset transform as the subgroup transform of x_i 7^i."""
    to_mont = lambda v: np.array([cref.lib().oracle_gl_to_mont(int(e)) for e in v], dtype=np.uint64)
    from_mont = lambda a: [F.GL.from_mont(int(e)) for e in a]
    for case, x in _sympy_cases():
        log_n = case["log_n"]
        n = 1 << log_n
        dom = pyntt.Domain(F.GL, n)
        fwd_py = pyntt.fft(dom, x) if log_n <= 10 else None
        inv_py = pyntt.ifft(dom, x) if log_n <= 10 else None
        fwd_c = from_mont(cref.ntt(to_mont(x), log_n, 1, False, 1))
        inv_c = from_mont(cref.ntt(to_mont(x), log_n, 1, True, 1))
        assert _le_digest(fwd_c) == case["forward_sha256"] and _le_digest(inv_c) == case["inverse_sha256"]
        if fwd_py is not None:
            assert fwd_py == fwd_c and inv_py == inv_c
        if "forward" in case:
            assert [format(v, "016x") for v in fwd_c] == case["forward"] and [format(v, "016x") for v in inv_c] == case["inverse"]
    # live re-computation when sympy is importable (it is in the build container; the fixture is what travels)
    try:
        from sympy.discrete.transforms import ntt as sympy_ntt
    except ImportError:
        return
    case, x = next(_sympy_cases())
    n = len(x)
    assert _le_digest([int(v) for v in sympy_ntt(x, F.GL.p)]) == case["forward_sha256"]
    coset_in = [(v * pow(7, i, F.GL.p)) % F.GL.p for i, v in enumerate(x)]
    want = [int(v) for v in sympy_ntt(coset_in, F.GL.p)]
    assert from_mont(cref.ntt(to_mont(x), case["log_n"], 1, False, 7)) == want


def test_cache_blocked_orders_of_the_c_oracle_equal_the_plain_loops():
    """oracle/c (round 5): the transform takes its butterflies block by block and the bit reversal works tile by tile -- the same
    pairs, twiddles and swaps in another order.  The plain stage-by-stage loop is kept as a test hook; both must give the same words
    (sizes around the switch points: 2^12 / 2^13 for the stages, 2^15 / 2^16 for the tiled reversal; Fp and Fq3 element widths)."""
    import ctypes
    from oracle import cref
    L = cref.lib()
    u64p = ctypes.POINTER(ctypes.c_uint64)
    L.oracle_gl_root_of_unity.restype = ctypes.c_uint64
    L.oracle_gl_root_of_unity.argtypes = [ctypes.c_uint]
    L.oracle_ntt_stages_plain.argtypes = [u64p, ctypes.c_uint, ctypes.c_uint, u64p, ctypes.c_uint64]
    L.oracle_ntt_stages_blocked.argtypes = [u64p, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint64]
    L.oracle_ntt_stages_plain.restype = L.oracle_ntt_stages_blocked.restype = None
    for log_n in (12, 13, 14, 16, 18):
        for V in (1, 3):
            x = cref.random_elements((1 << log_n) * V, 40 + log_n + V)
            a, b = x.copy(), x.copy()
            root = L.oracle_gl_root_of_unity(log_n)
            L.oracle_ntt_stages_plain(cref._p(a), log_n, V, None, root)
            L.oracle_ntt_stages_blocked(cref._p(b), log_n, V, root)
            assert np.array_equal(a, b), (log_n, V)
    for log_n in (4, 15, 16, 17, 19):
        n = 1 << log_n
        idx = np.arange(n, dtype=np.uint64)
        rev = np.zeros(n, dtype=np.uint64)
        for bit in range(log_n):
            rev |= ((idx >> np.uint64(bit)) & np.uint64(1)) << np.uint64(log_n - 1 - bit)
        for V in (1, 3, 4):
            x = cref.random_elements(n * V, 90 + log_n + V)
            assert np.array_equal(cref.bit_reverse(x, log_n, V), x.reshape(n, V)[rev.astype(np.int64)].ravel()), (log_n, V)
