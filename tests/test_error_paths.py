"""Error behaviour at the boundary: the reference panics on bad input (assert! / unwrap, e.g. gpu/src/plan.rs:248,255,360;
src/matrix.rs:32-38; src/merkle.rs:149-160 returns Error::LeafIndexOutOfBounds); here the C ABI returns an error code with a
message (never a crash, never a silent wrong result) and the Python mirror raises.  Runs on the simulator build of the library."""
import ctypes

import numpy as np
import pytest

from tests import backends
from ministark_amd import (GOLDILOCKS_FP as FP, GpuFft, GpuVec, Matrix, MerkleTree, Radix2EvaluationDomain, apply_drp, expr as E,
                           grind_proof_of_work)
from ministark_amd._lib import MsError


@pytest.fixture(scope="module")
def pl():
    return backends.planner("emu")


def _vec(pl, n=16):
    return GpuVec.from_numpy(pl, np.arange(n, dtype=np.uint64), FP)


def test_transform_of_a_column_of_the_wrong_length(pl):                 # gpu/src/plan.rs:255 assert_eq!(self.n, buffer.len())
    plan = GpuFft(Radix2EvaluationDomain(32, 7), FP, pl)
    with pytest.raises(ValueError, match="16 elements, domain 32"):
        plan.encode(_vec(pl))


def test_domains(pl):
    with pytest.raises(ValueError, match="power of two"):
        Radix2EvaluationDomain(24)
    with pytest.raises(ValueError, match="two-adicity"):
        GpuFft(Radix2EvaluationDomain(1 << 33), FP, pl)


def test_unknown_field_and_null_arguments(pl):
    L, v = pl.lib, _vec(pl)
    with pytest.raises(MsError, match="unknown field id 7"):
        L.check(L.ms_bit_reverse(pl.handle, 7, 4, (ctypes.c_void_p * 1)(v.ptr), 1))
    with pytest.raises(MsError, match="null context"):
        L.check(L.ms_sync(None))
    with pytest.raises(MsError, match="null argument"):
        L.check(L.ms_ntt_encode(None, v.ptr))


@pytest.mark.parametrize("factor", [3, 32])
def test_fri_folding_factor(pl, factor):                                # src/fri.rs:186-192: 2, 4, 8, 16 only
    with pytest.raises(MsError, match=f"folding factor {factor} not supported"):
        apply_drp(_vec(pl, 64), np.array([5], dtype=np.uint64), factor)


def test_matrix_shapes(pl):
    with pytest.raises(ValueError, match="same length"):               # src/matrix.rs:32-38
        Matrix([_vec(pl, 16), _vec(pl, 8)])
    m = Matrix([_vec(pl), _vec(pl)])
    with pytest.raises(ValueError, match="powers of two"):
        m.lde(3)
    with pytest.raises(ValueError, match="do not split into 3 columns"):
        Matrix.from_chunks(_vec(pl), 3)
    with pytest.raises(MsError, match="row 99 out of range"):
        m.get_rows([99])
    with pytest.raises(IndexError, match="leaf index 16 out of bounds"):       # Error::LeafIndexOutOfBounds, src/merkle.rs:154-158
        MerkleTree.from_matrix(m).prove([16])


def test_constraint_programs_are_validated_before_they_run(pl):
    v = _vec(pl)
    none = np.zeros((0, 1), dtype=np.uint64)
    prog = E.compile_expr(E.Trace(0, 0) * E.Trace(5, 0), 8, False)      # column 5 of a one-column trace
    with pytest.raises(MsError, match="invalid instruction"):
        E.eval(prog, pl, none, none, 1, 7, 16, [v])
    L = pl.lib
    code = np.array([[99, 0, 0, 0]], dtype=np.uint32)
    out, off = GpuVec(pl, 16, FP), np.array([7], dtype=np.uint64)
    VP = ctypes.c_void_p
    with pytest.raises(MsError, match="invalid instruction 0 .op 99"):
        L.check(L.ms_eval_program_ex(pl.handle, code.ctypes.data, 1, None, 0, 4, 1, off.ctypes.data, None, (VP * 1)(v.ptr), 1, (VP * 1)(), 0,
                                     (VP * 1)(), (ctypes.c_uint * 1)(), 0, FP, out.ptr, 0))


def test_proof_of_work_bits(pl):
    with pytest.raises(MsError, match="<= 64"):
        grind_proof_of_work(pl, bytes(32), 70)


def test_deep_point_on_the_evaluation_coset_is_refused(pl):
    """ms_deep_compose evaluates the quotients on the coset 7<w_n> with one shared inversion per lane: an out-of-domain point ON that
    coset (Fq = Fp, probability ~ n / p) has no quotient there.  It is an error with a message, not a column of silently zeroed
    factors; the reference's synthetic division has no such point, the caller re-draws z."""
    from ministark_amd.composer import DeepCompositionCoeffs, DeepPolyComposer
    P = (1 << 64) - (1 << 32) + 1
    n = 64
    rng = np.random.default_rng(5)
    mk = lambda k: Matrix([GpuVec.from_numpy(pl, np.array([pow(2, 64, P) * int(x) % P for x in rng.integers(0, P, size=n, dtype=np.uint64)], dtype=np.uint64), FP) for _ in range(k)])
    g = Radix2EvaluationDomain(n).group_gen
    z = 7 * pow(g, 5, P) % P
    composer = DeepPolyComposer([(0, 0), (0, 1)], n, z, mk(1), None, mk(1))
    composer.get_ood_evals()
    with pytest.raises(MsError, match="lies on the evaluation coset"):
        composer.into_deep_poly(DeepCompositionCoeffs([3, 4], [5], (1, 2)))
    ok = DeepPolyComposer([(0, 0), (0, 1)], n, z + 1, mk(1), None, mk(1))          # next to it: fine
    ok.get_ood_evals()
    assert len(ok.into_deep_poly(DeepCompositionCoeffs([3, 4], [5], (1, 2)))) == n


def test_get_rows_of_no_positions(pl):
    m = Matrix([_vec(pl), _vec(pl)])
    assert m.get_rows([]).shape == (0, 2)
    assert m.get_rows([3, 1]).shape == (2, 2)
