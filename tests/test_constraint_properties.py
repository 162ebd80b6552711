"""End-to-end properties of the constraint pipeline, as the reference's tests/constraint.rs checks them
(evaluate_fibonacci_constraint :172-197, evaluate_binary_constraint :199-220, symbolic_evaluation_with_challenges
:88-113, evaluate_permutation_constraint :222-291, evaluate_zerofier_constraint :293-330):

    valid trace -> interpolate -> evaluate on the LDE coset -> constraint evaluation at every LDE point
    -> interpolate the result -> the polynomial vanishes on the trace domain except the last row
       (assert_valid_over_transition_domain, tests/constraint.rs:332-350)

No oracle is involved: the property holds only if NTT, LDE, the evaluator, the stage kernels and the
extension-column scan are all exact.  Every step runs through the library."""
import numpy as np
import pytest

from oracle.pyref.fields import GL
from tests import backends
from ministark_amd import GOLDILOCKS_FP as FP, GpuVec, Matrix, Radix2EvaluationDomain, running_product
from ministark_amd import expr as E
from ministark_amd import stages as S

P = GL.p
KINDS = [pytest.param("emu", 256, id="emu"), pytest.param("hip", 2048, id="hip", marks=pytest.mark.gpu),
         pytest.param("hip", 1 << 16, id="hip-2^16", marks=pytest.mark.gpu)]


def _mont(vals):
    return np.array([GL.to_mont(v % P) for v in vals], dtype=np.uint64)


def _constraint_polys_on_trace_domain(pl, columns, constraints, n, blowup, challenges=()):
    """-> for every constraint the values of its (interpolated) evaluation polynomial on the trace domain."""
    trace_dom, lde_dom = Radix2EvaluationDomain(n), Radix2EvaluationDomain(n * blowup, 7)
    lde = Matrix([GpuVec.from_numpy(pl, _mont(c), FP) for c in columns]).interpolate(trace_dom).evaluate(lde_dom)
    ch = _mont(challenges).reshape(-1, 1) if len(challenges) else np.zeros((1, 1), dtype=np.uint64)
    out = []
    for c in constraints:
        prog = E.compile_expr(c, len(columns), fq_is_ext=False)
        evals = E.eval(prog, pl, ch, ch[:1], blowup, 7, n * blowup, lde.columns)
        poly = Matrix([evals]).into_polynomials(lde_dom)
        # the trace domain is the subgroup of index `blowup` in the size-N subgroup
        on_subgroup = poly.evaluate(Radix2EvaluationDomain(n * blowup, 1)).to_numpy()[0]
        out.append(on_subgroup[::blowup])
    return out


def _assert_valid_over_transition_domain(values, expect_last_nonzero=False):
    assert not values[:-1].any(), f"constraint polynomial is non-zero at row {int(np.flatnonzero(values[:-1])[0])}"
    if expect_last_nonzero:
        assert values[-1] != 0


@pytest.mark.parametrize("kind,n", KINDS)
def test_fibonacci_constraint(kind, n):                 # tests/constraint.rs:172-197, gen_fib_matrix src/utils.rs:617-631
    pl = backends.planner(kind)
    c0, c1 = [1], [1]
    for _ in range(1, n):
        n0 = (c0[-1] + c1[-1]) % P
        c0.append(n0); c1.append((n0 + c1[-1]) % P)
    t = lambda c, o=0: E.Trace(c, o)
    cons = [t(0, 1) - (t(0) + t(1)), t(1, 1) - (t(0, 1) + t(1))]
    for v in _constraint_polys_on_trace_domain(pl, [c0, c1], cons, n, 1):
        _assert_valid_over_transition_domain(v, expect_last_nonzero=True)      # the wrap-around row breaks the recurrence


@pytest.mark.parametrize("kind,n", KINDS)
def test_binary_constraint_and_challenges(kind, n):     # tests/constraint.rs:199-220 and :88-113
    pl = backends.planner(kind)
    rng = np.random.default_rng(4)
    bits = rng.integers(0, 2, size=n)
    v = _constraint_polys_on_trace_domain(pl, [[int(b) for b in bits]], [E.Trace(0) * (E.Trace(0) - 1)], n, 2)[0]
    assert not v.any()                                  # holds on every row, the last one included
    alpha, beta = 3, 7
    col = [alpha if b else beta for b in bits]
    c = (E.Trace(0) - E.Challenge(0)) * (E.Trace(0) - E.Challenge(1))
    assert not _constraint_polys_on_trace_domain(pl, [col], [c], n, 2, challenges=(alpha, beta))[0].any()
    bad = list(col); bad[n // 3] = 5                    # one invalid row must show up exactly there
    v = _constraint_polys_on_trace_domain(pl, [bad], [c], n, 2, challenges=(alpha, beta))[0]
    assert list(np.flatnonzero(v)) == [n // 3]


@pytest.mark.parametrize("kind,n", KINDS)
def test_permutation_constraint(kind, n):               # tests/constraint.rs:222-291
    pl = backends.planner(kind)
    rng = np.random.default_rng(5)
    original = [int(x) for x in rng.integers(0, 1 << 62, size=n)]
    shuffled = list(original)
    rng.shuffle(shuffled)
    challenge = 0x1f2e3d4c5b6a7988 % P
    one = _mont([1])
    prods = []
    for col in (original, shuffled):
        # factors challenge - v on the device, then the running product scan (the reference's .scan(), :229-246)
        f = GpuVec.from_numpy(pl, _mont(col), FP)
        S.NegInPlaceStage(pl, n, FP).encode(f)
        S.AddAssignConstStage(pl, n, FP).encode(f, _mont([challenge]))
        prods.append(running_product(f, one))
    pcols = [[GL.from_mont(int(x)) for x in p.to_numpy()] for p in prods]
    final = [(pcols[k][-1] * (challenge - (original, shuffled)[k][-1])) % P for k in range(2)]
    assert final[0] == final[1]                          # same multiset -> same grand product (:282-283)
    t = lambda c, o=0: E.Trace(c, o)
    cons = [t(2) * (E.Challenge(0) - t(0)) - t(2, 1), t(3) * (E.Challenge(0) - t(1)) - t(3, 1)]
    for v in _constraint_polys_on_trace_domain(pl, [original, shuffled, pcols[0], pcols[1]], cons, n, 2, challenges=(challenge,)):
        _assert_valid_over_transition_domain(v)


@pytest.mark.parametrize("kind,n", KINDS[:2])
def test_zerofier_constraint(kind, n):                  # tests/constraint.rs:293-330 (blow-up 16)
    pl = backends.planner(kind)
    ch = (999, 43)
    instr = ord("+")
    curr_instr = [instr] * n
    f = GpuVec.from_numpy(pl, _mont([(ch[0] - ch[1] * instr) % P] * n), FP)
    permutation = [GL.from_mont(int(x)) for x in running_product(f, _mont([1])).to_numpy()]
    t = lambda c, o=0: E.Trace(c, o)
    c = t(0) * (t(1) * (E.Challenge(0) - E.Challenge(1) * t(0)) - t(1, 1)) + (t(0) - instr) * (t(1) - t(1, 1))
    _assert_valid_over_transition_domain(_constraint_polys_on_trace_domain(pl, [curr_instr, permutation], [c], n, 16, challenges=ch)[0])
