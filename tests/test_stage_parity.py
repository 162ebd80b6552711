"""Element-wise stages vs the oracle, bit-exact.  Cases follow gpu/tests/fields.rs:18-117
(MulPow fp*fp e=1, fq3*fp e=1, fq3*fq3 e=3 at n=2048) and the stage list of
gpu/src/stage.rs, with rotations (positive, negative, > n) and edge values (0, 1, p-1)."""
import numpy as np
import pytest

from oracle import cref
from tests import backends
from ministark_amd import GOLDILOCKS_FP as FP, GOLDILOCKS_FQ3 as FQ3, GpuVec, Matrix
from ministark_amd import stages as S

P = cref.GL_P
V = {FP: 1, FQ3: 3}
KINDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
PAIRS = [(FP, FP), (FQ3, FQ3), (FQ3, FP)]


def _vals(n_words, seed):
    a = cref.random_elements(n_words, seed)
    a[:4] = [0, 4294967295, P - 1, 1][: min(4, n_words)]     # 0, mont(1), p-1, raw 1
    return a


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("lf,rf", PAIRS)
@pytest.mark.parametrize("shift", [0, 1, -1, 5000])
def test_binary_stages(kind, lf, rf, shift):
    pl = backends.planner(kind)
    n = 2048 if kind == "hip" else 256
    a, b = _vals(n * V[lf], 1), _vals(n * V[rf], 2)
    for op, Into, Assign in ((S.ADD, S.AddIntoStage, S.AddAssignStage), (S.MUL, S.MulIntoStage, S.MulAssignStage)):
        want = cref.binary(op, V[lf], V[rf], a, b, shift)
        l, r, d = GpuVec.from_numpy(pl, a, lf), GpuVec.from_numpy(pl, b, rf), GpuVec(pl, n, lf)
        Into(pl, n, lf, rf).encode(d, l, r, shift)
        assert np.array_equal(d.to_numpy(), want)
        Assign(pl, n, lf, rf).encode(l, r, shift)
        assert np.array_equal(l.to_numpy(), want)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("lf,rf", PAIRS)
def test_const_stages(kind, lf, rf):
    pl = backends.planner(kind)
    n = 2048 if kind == "hip" else 128
    a, c = _vals(n * V[lf], 3), cref.random_elements(V[rf], 9)
    for op, Into, Assign in ((S.ADD, S.AddIntoConstStage, S.AddAssignConstStage), (S.MUL, S.MulIntoConstStage, S.MulAssignConstStage)):
        want = cref.binary_const(op, V[lf], V[rf], a, c)
        l, d = GpuVec.from_numpy(pl, a, lf), GpuVec(pl, n, lf)
        Into(pl, n, lf, rf).encode(d, l, c)
        assert np.array_equal(d.to_numpy(), want)
        Assign(pl, n, lf, rf).encode(l, c)
        assert np.array_equal(l.to_numpy(), want)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("lf,rf,e", [(FP, FP, 1), (FQ3, FP, 1), (FQ3, FQ3, 3), (FP, FP, 0), (FQ3, FQ3, 77)])
def test_mul_pow_stage(kind, lf, rf, e):                 # gpu/tests/fields.rs:18-117
    pl = backends.planner(kind)
    n = 2048 if kind == "hip" else 128
    a, b = _vals(n * V[lf], 4), _vals(n * V[rf], 5)
    for shift in (0, 3):
        l, r = GpuVec.from_numpy(pl, a, lf), GpuVec.from_numpy(pl, b, rf)
        S.MulPowStage(pl, n, lf, rf).encode(l, r, e, shift)
        assert np.array_equal(l.to_numpy(), cref.mul_pow(V[lf], V[rf], a, b, e, shift))


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("field", [FP, FQ3])
def test_unary_stages(kind, field):
    pl = backends.planner(kind)
    n = 1024 if kind == "hip" else 64
    a = _vals(n * V[field], 6)
    for op, Into, InPlace, e in ((S.NEG, S.NegIntoStage, S.NegInPlaceStage, None), (S.INV, S.InverseIntoStage, S.InverseInPlaceStage, None),
                                 (S.EXP, S.ExpIntoStage, S.ExpInPlaceStage, 11)):
        want = cref.unary(op, V[field], a, e or 0)
        s, d = GpuVec.from_numpy(pl, a, field), GpuVec(pl, n, field)
        args = (e,) if e is not None else ()
        Into(pl, n, field).encode(d, s, *args)
        assert np.array_equal(d.to_numpy(), want)
        InPlace(pl, n, field).encode(s, *args)
        assert np.array_equal(s.to_numpy(), want)
    # x * x^-1 == 1 wherever x != 0
    s = GpuVec.from_numpy(pl, a, field)
    inv = GpuVec(pl, n, field)
    S.InverseIntoStage(pl, n, field).encode(inv, s)
    S.MulAssignStage(pl, n, field, field).encode(inv, s)
    got = inv.to_numpy().reshape(n, V[field])
    nz = a.reshape(n, V[field]).any(axis=1)
    one = np.array([4294967295] + [0] * (V[field] - 1), dtype=np.uint64)
    assert np.array_equal(got[nz], np.tile(one, (int(nz.sum()), 1)))


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("field", [FP, FQ3])
def test_inverse_stages_on_long_columns(kind, field):
    # from 4096 elements the inverse stages run Montgomery's trick (k_batch_inverse): zeros sprinkled in (0^-1 = 0),
    # into another buffer and in place -- the same words as the per-element Fermat inverse
    pl = backends.planner(kind)
    n = 1 << 16 if kind == "hip" else 8192
    a = _vals(n * V[field], 16).reshape(n, V[field])
    a[::97] = 0
    a[-1] = 0
    a = np.ascontiguousarray(a.reshape(-1))
    want = cref.unary(S.INV, V[field], a, 0)
    s, d = GpuVec.from_numpy(pl, a, field), GpuVec(pl, n, field)
    S.InverseIntoStage(pl, n, field).encode(d, s)
    assert np.array_equal(d.to_numpy(), want)
    assert np.array_equal(s.to_numpy(), a)
    S.InverseInPlaceStage(pl, n, field).encode(s)
    assert np.array_equal(s.to_numpy(), want)


@pytest.mark.parametrize("kind", KINDS)
def test_convert_fill_sum(kind):
    pl = backends.planner(kind)
    n = 512
    a = _vals(n, 7)
    src, dst = GpuVec.from_numpy(pl, a, FP), GpuVec(pl, n, FQ3)
    S.ConvertIntoStage(pl, n, FQ3, FP).encode(dst, src)
    got = dst.to_numpy().reshape(n, 3)
    assert np.array_equal(got[:, 0], a) and not got[:, 1:].any()
    c = cref.random_elements(3, 8)
    S.FillBuffStage(pl, n, FQ3).encode(dst, c)
    assert np.array_equal(dst.to_numpy().reshape(n, 3), np.tile(c, (n, 1)))
    for field in (FP, FQ3):
        cols = [_vals(n * V[field], 20 + i) for i in range(5)]
        m = Matrix.from_numpy(pl, cols, field)
        assert np.array_equal(S.sum_columns(m).to_numpy(), cref.sum_columns(cols, V[field]))
