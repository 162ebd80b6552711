"""The 252-bit StarkWare field (a3): NTT / iNTT / stages / row hashing vs the pure-Python big-int
oracle, bit-exact.  Shapes follow gpu/tests/shaders.rs:69-91 (2048, 4096; subgroup and coset with
offset = GENERATOR = 3) and src/eval_gpu.rs:1054-1082 (constants on Fp252)."""
import hashlib

import numpy as np
import pytest

from oracle.pyref import fields as F
from oracle.pyref import ntt as pyntt
from tests import backends
from ministark_amd import (STARK252_FP, F252_GENERATOR, GpuFft, GpuIfft, GpuVec, Matrix, Radix2EvaluationDomain,
                           f252_from_mont_limbs, f252_to_mont_limbs)
from ministark_amd import stages as S

P = F.F252_P
KINDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


def _rand_canon(n, seed):
    rng = np.random.default_rng(seed)
    vals = [int.from_bytes(rng.bytes(32), "little") % P for _ in range(n)]
    vals[:3] = [0, 1, P - 1][: min(3, n)]
    return vals


def _to_dev(pl, vals):
    return GpuVec.from_numpy(pl, np.concatenate([f252_to_mont_limbs(v) for v in vals]) if vals else np.zeros(0, np.uint64), STARK252_FP)


def _from_dev(v):
    a = v.to_numpy().reshape(-1, 4)
    return [f252_from_mont_limbs(r) for r in a]


def test_field_constants():
    # felt_u256.h.metal:101-108
    assert list(f252_to_mont_limbs(1)) == [18446744073709551585, 18446744073709551615, 18446744073709551615, 576460752303422960]
    assert pow(F252_GENERATOR, (P - 1) // 2, P) == P - 1          # 3 is a non-residue: generates the 2-Sylow part


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("log_n,offset,inverse", [(11, 1, False), (12, 1, False), (11, 3, False), (12, 3, False), (11, 1, True), (12, 3, True), (5, 3, False), (9, 3, True)])
def test_fft_with_256_bit_field(kind, log_n, offset, inverse):       # gpu/tests/shaders.rs:69-91
    if kind == "emu" and log_n > 11:
        pytest.skip("kept short under the simulator")
    pl = backends.planner(kind)
    n = 1 << log_n
    vals = _rand_canon(n, log_n * 7 + offset)
    dom = Radix2EvaluationDomain(n, offset, STARK252_FP)
    v = _to_dev(pl, vals)
    plan = (GpuIfft if inverse else GpuFft)(dom, STARK252_FP, pl)
    plan.encode(v)
    plan.execute()
    d = pyntt.Domain(F.F252, n, offset)
    want = pyntt.ifft(d, vals) if inverse else pyntt.fft(d, vals)
    assert _from_dev(v) == want


@pytest.mark.parametrize("kind", KINDS)
def test_stages_252(kind):
    pl = backends.planner(kind)
    n = 64
    a, b = _rand_canon(n, 1), _rand_canon(n, 2)
    A, B, D = _to_dev(pl, a), _to_dev(pl, b), GpuVec(pl, n, STARK252_FP)
    S.MulIntoStage(pl, n, STARK252_FP).encode(D, A, B, 3)
    assert _from_dev(D) == [(a[i] * b[(i + 3) % n]) % P for i in range(n)]
    S.AddIntoStage(pl, n, STARK252_FP).encode(D, A, B, -1)
    assert _from_dev(D) == [(a[i] + b[(i - 1) % n]) % P for i in range(n)]
    S.NegIntoStage(pl, n, STARK252_FP).encode(D, A)
    assert _from_dev(D) == [(-x) % P for x in a]
    S.InverseIntoStage(pl, n, STARK252_FP).encode(D, A)
    assert _from_dev(D) == [0 if x == 0 else pow(x, -1, P) for x in a]
    S.ExpIntoStage(pl, n, STARK252_FP).encode(D, A, 13)
    assert _from_dev(D) == [pow(x, 13, P) for x in a]
    c = 123456789123456789123456789 % P
    S.MulIntoConstStage(pl, n, STARK252_FP).encode(D, A, f252_to_mont_limbs(c))
    assert _from_dev(D) == [(x * c) % P for x in a]
    S.AddAssignConstStage(pl, n, STARK252_FP).encode(D, f252_to_mont_limbs(5))
    assert _from_dev(D) == [(x * c + 5) % P for x in a]
    L = _to_dev(pl, a)
    S.MulPowStage(pl, n, STARK252_FP).encode(L, B, 3, 1)
    assert _from_dev(L) == [(a[i] * pow(b[(i + 1) % n], 3, P)) % P for i in range(n)]
    S.FillBuffStage(pl, n, STARK252_FP).encode(D, f252_to_mont_limbs(77))
    assert _from_dev(D) == [77] * n
    m = Matrix([_to_dev(pl, a), _to_dev(pl, b), _to_dev(pl, a)])
    assert _from_dev(S.sum_columns(m)) == [(2 * a[i] + b[i]) % P for i in range(n)]
    m.bit_reverse_rows()
    assert _from_dev(m.columns[1]) == F.bit_reverse(b)


@pytest.mark.parametrize("kind", KINDS)
def test_inverse_stage_on_a_long_column_252(kind):
    # from 4096 elements: Montgomery's trick (k_batch_inverse), zeros included, into another buffer and in place
    pl = backends.planner(kind)
    n = 4096
    a = _rand_canon(n, 7)
    a[5] = a[n - 1] = a[1000] = 0
    want = [0 if x == 0 else pow(x, -1, P) for x in a]
    A, D = _to_dev(pl, a), GpuVec(pl, n, STARK252_FP)
    S.InverseIntoStage(pl, n, STARK252_FP).encode(D, A)
    assert _from_dev(D) == want and _from_dev(A) == a
    S.InverseInPlaceStage(pl, n, STARK252_FP).encode(A)
    assert _from_dev(A) == want


@pytest.mark.parametrize("kind", KINDS)
def test_hash_rows_252(kind):
    pl = backends.planner(kind)
    n = 16
    a, b = _rand_canon(n, 3), _rand_canon(n, 4)
    leaves = Matrix([_to_dev(pl, a), _to_dev(pl, b)]).hash_rows().to_numpy().reshape(n, 32)
    for r in range(n):
        assert leaves[r].tobytes() == hashlib.sha256(a[r].to_bytes(32, "little") + b[r].to_bytes(32, "little")).digest()


def _mont_words(vals):
    return np.concatenate([f252_to_mont_limbs(v) for v in vals])


def test_c_oracle_ntt_252_matches_bigint():
    """oracle/c's Fp252 NTT / LDE (the checker of the large sizes below) against the big-integer restatement."""
    from oracle import cref
    for log_n, offset, inverse in [(0, 1, False), (1, 3, False), (7, 1, False), (9, 3, False), (9, 3, True), (10, 5, True)]:
        n = 1 << log_n
        vals = _rand_canon(n, 40 + log_n)
        d = pyntt.Domain(F.F252, n, offset)
        want = pyntt.ifft(d, vals) if inverse else pyntt.fft(d, vals)
        got = cref.ntt252(_mont_words(vals), log_n, inverse, f252_to_mont_limbs(offset)).reshape(-1, 4)
        assert [f252_from_mont_limbs(r) for r in got] == want
    vals = _rand_canon(64, 3)
    for br in (True, False):
        got = cref.lde252(_mont_words(vals), 6, 3, f252_to_mont_limbs(3), br).reshape(-1, 4)
        want = pyntt.lde_bit_reversed(F.F252, vals, 8, 3)
        assert [f252_from_mont_limbs(r) for r in got] == (want if br else F.bit_reverse(want))


def _ntt_case(pl, log_n, offset, inverse, seed):
    from oracle import cref
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 1 << 59, size=4 << log_n, dtype=np.uint64)      # limbs < 2^59 => canonical (< p)
    x[:4] = [0, 0, 0, 0]
    x[4:8] = f252_to_mont_limbs(P - 1)
    v = GpuVec.from_numpy(pl, x, STARK252_FP)
    plan = (GpuIfft if inverse else GpuFft)(Radix2EvaluationDomain(1 << log_n, offset, STARK252_FP), STARK252_FP, pl)
    plan.encode(v)
    plan.execute()
    assert np.array_equal(v.to_numpy(), cref.ntt252(x, log_n, inverse, f252_to_mont_limbs(offset)))


@pytest.mark.parametrize("log_n,offset,inverse", [(12, 3, False), (13, 1, True), (14, 3, True), (15, 5, False)])
def test_tiled_ntt_252_emu(log_n, offset, inverse):
    """Two tiled passes (fp252_ntt_kernels.h) with even and odd radices, against the C oracle."""
    _ntt_case(backends.planner("emu"), log_n, offset, inverse, log_n)


def test_tiled_ntt_252_three_passes_emu(monkeypatch):
    monkeypatch.setenv("MS_NTT252_PASSES", "3")
    pl = backends.planner("emu")
    _ntt_case(pl, 17, 3, False, 1)
    _ntt_case(pl, 17, 3, True, 2)


def _lde_case(pl, log_n, log_b, offset, seed):
    from oracle import cref
    rng = np.random.default_rng(seed)
    cols = [rng.integers(0, 1 << 59, size=4 << log_n, dtype=np.uint64) for _ in range(3)]
    m = Matrix([GpuVec.from_numpy(pl, c, STARK252_FP) for c in cols])
    off = f252_to_mont_limbs(offset)
    out = m.lde(1 << log_b, offset, True)
    for c in range(3):
        assert np.array_equal(out.columns[c].to_numpy(), cref.lde252(cols[c], log_n, log_b, off, True))
    nat = m.lde(1 << log_b, offset, False)
    assert np.array_equal(nat.columns[1].to_numpy(), cref.lde252(cols[1], log_n, log_b, off, False))
    # into_evaluations on a shorter coefficient column (src/matrix.rs:193-251)
    ev = m.bit_reversed_evaluate(Radix2EvaluationDomain(1 << (log_n + log_b), offset, STARK252_FP))
    padded = np.concatenate([cols[2], np.zeros((4 << (log_n + log_b)) - cols[2].size, dtype=np.uint64)])
    want = cref.bit_reverse(cref.ntt252(padded, log_n + log_b, False, off), log_n + log_b, 4)
    assert np.array_equal(ev.columns[2].to_numpy(), want)


@pytest.mark.parametrize("log_n,log_b", [(9, 2), (10, 3), (8, 4), (11, 1)])
def test_tiled_lde_252_emu(log_n, log_b):
    """Zero-extended input read from the short column + bit-reversed store fused into the last pass."""
    _lde_case(backends.planner("emu"), log_n, log_b, 3, log_n)


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,offset,inverse", [(11, 3, False), (16, 3, True), (19, 1, False), (20, 3, False), (20, 3, True), (21, 5, False), (22, 3, True)])
def test_tiled_ntt_252_hip(log_n, offset, inverse):
    """2^20 (two passes of radix 1024) and the three-pass sizes against the C oracle, every element."""
    _ntt_case(backends.planner("hip"), log_n, offset, inverse, log_n)


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,log_b", [(16, 2), (18, 2), (17, 4), (20, 1)])
def test_tiled_lde_252_hip(log_n, log_b):
    _lde_case(backends.planner("hip"), log_n, log_b, 3, log_n)


@pytest.mark.gpu
def test_roundtrip_2_16_252_hip():
    pl = backends.planner("hip")
    n = 1 << 16
    rng = np.random.default_rng(5)
    x = rng.integers(0, 1 << 59, size=4 * n, dtype=np.uint64)      # limbs < 2^59 => canonical (< p)
    v = GpuVec.from_numpy(pl, x, STARK252_FP)
    dom = Radix2EvaluationDomain(n, 3, STARK252_FP)
    f = GpuFft(dom, STARK252_FP, pl); f.encode(v); f.execute()
    assert not np.array_equal(v.to_numpy(), x)
    g = GpuIfft(dom, STARK252_FP, pl); g.encode(v); g.execute()
    assert np.array_equal(v.to_numpy(), x)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("log_n,blowup", [(4, 2), (6, 8), (11, 4)])
def test_lde_252(kind, log_n, blowup):                  # interpolate + bit_reversed_evaluate, src/prover.rs:50-51
    pl = backends.planner(kind)
    n = 1 << log_n
    cols = [_rand_canon(n, 10 + c) for c in range(2)]
    m = Matrix([_to_dev(pl, c) for c in cols])
    out = m.lde(blowup, 3, True)
    for c in range(2):
        assert _from_dev(out.columns[c]) == pyntt.lde_bit_reversed(F.F252, cols[c], blowup, 3)
    nat = m.lde(blowup, 5, False)
    assert _from_dev(nat.columns[0]) == F.bit_reverse(pyntt.lde_bit_reversed(F.F252, cols[0], blowup, 5))


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("ff", [2, 4, 8, 16])
@pytest.mark.parametrize("offset", [1, 3])
def test_fri_fold_252(kind, ff, offset):                # apply_drp, src/fri.rs:526-567
    from oracle.pyref import fri as ofri
    from ministark_amd import apply_drp
    pl = backends.planner(kind)
    n = 256
    evals = _rand_canon(n, 20 + ff)
    alpha = _rand_canon(1, 99)[0]
    got = _from_dev(apply_drp(_to_dev(pl, evals), f252_to_mont_limbs(alpha), ff, offset))
    assert got == ofri.apply_drp(F.F252, None, evals, offset, alpha, ff)


@pytest.mark.parametrize("kind", KINDS)
def test_mul_edge_values_252(kind):
    # the product works on nine 28-bit digits with lazy 64-bit columns (csrc/fp252.h): digit boundaries, all-ones
    # digits, values next to p and sparse limbs are where such a scheme would break
    pl = backends.planner(kind)
    n = 1 << (10 if kind == "emu" else 14)
    rng = np.random.default_rng(11)
    M = (1 << 28) - 1
    edge = [0, 1, 2, P - 1, P - 2, P - (1 << 28), (1 << 28) - 1, 1 << 28, (1 << 252) - 1 - (1 << 200), (1 << 251), (1 << 251) + 17 * (1 << 192),
            sum(M << (28 * k) for k in range(0, 9, 2)) % P, sum(M << (28 * k) for k in range(1, 9, 2)) % P, (1 << 224) - 1, 1 << 224, (1 << 192) * 17, P >> 1]
    def draw():
        out = []
        for _ in range(n):
            r = rng.random()
            if r < 0.35:
                out.append(edge[int(rng.integers(0, len(edge)))] % P)
            elif r < 0.55:
                v = 0
                for k in range(9):
                    v |= int(rng.choice([0, M, 1, M - 1, int(rng.integers(0, M + 1))])) << (28 * k)
                out.append(v % P)
            else:
                out.append(int.from_bytes(rng.bytes(32), "little") % P)
        return out
    a, b = draw(), draw()              # these are the words the kernel sees (Montgomery representatives), patterns intact
    limbs = lambda vals: np.array([(v >> (64 * k)) & ((1 << 64) - 1) for v in vals for k in range(4)], dtype=np.uint64)
    A, B, D = GpuVec.from_numpy(pl, limbs(a), STARK252_FP), GpuVec.from_numpy(pl, limbs(b), STARK252_FP), GpuVec(pl, n, STARK252_FP)
    S.MulIntoStage(pl, n, STARK252_FP).encode(D, A, B, 1)
    got = D.to_numpy().reshape(n, 4)
    rinv = pow(1 << 256, -1, P)
    for i in range(n):
        v = int(got[i][0]) | int(got[i][1]) << 64 | int(got[i][2]) << 128 | int(got[i][3]) << 192
        assert v == a[i] * b[(i + 1) % n] * rinv % P, f"element {i}"
