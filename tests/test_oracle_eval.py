"""The C restatement of eval_cpu::eval (oracle_eval_expr: 512-point chunks, batch inversion; src/eval_cpu.rs:33-150)
and of the DEEP composition pieces (src/utils.rs:124-175, src/composer.rs:100-188) pinned to the independent
big-integer Python restatements (per-point DAG recursion, literal synthetic division).  No GPU involved."""
import numpy as np
import pytest

from oracle import cref
from oracle.pyref import deep as odeep
from oracle.pyref import evalexpr
from oracle.pyref.fields import F252, GL
from ministark_amd import expr as E          # the DAG node classes only (duck-typed by both oracles)

fm = lambda v: int(GL.from_mont(int(v)))


def _cols(n, nb, ne):
    base = [cref.random_elements(n, 10 + k) for k in range(nb)]
    ext = [cref.random_elements(n * 3, 20 + k) for k in range(ne)]
    return base, ext


@pytest.mark.parametrize("log_n,lde_step", [(6, 4), (10, 2), (11, 1)])
def test_chunked_evaluator_fq3(log_n, lde_step):
    n = 1 << log_n
    x = E.X()
    b = [lambda o=0, k=k: E.Trace(k, o) for k in range(3)]
    e = [lambda o=0, k=k: E.Trace(3 + k, o) for k in range(2)]
    expr = ((e[0](1) - e[0]() * (E.Challenge(0) - b[0]() * E.Challenge(1) - b[2](1))) * (x - 1) / (x ** 16 - 1)
            + (b[1]() * b[0](-1) + E.Constant(5)) ** 3 * E.Hint(0) + E.Periodic([1, 2, 3, 4]) * e[1]() / (b[0]() - b[0]())
            + E.Constant((1, 2, 3)) * x / (e[1](2) - E.Constant(7)))
    base, ext = _cols(n, 3, 2)
    ch = cref.random_elements(6, 1).reshape(2, 3)
    hi = cref.random_elements(3, 2).reshape(1, 3)
    got = cref.eval_expr(expr, log_n, lde_step, 7, base, ext, ch, hi, True).reshape(n, 3)
    bc = [[fm(v) for v in c] for c in base]
    ec = [[tuple(fm(v) for v in c[3 * i:3 * i + 3]) for i in range(n)] for c in ext]
    want = evalexpr.eval_points(expr, range(n), n, lde_step, 7, bc, ec, [tuple(fm(v) for v in r) for r in ch],
                                [tuple(fm(v) for v in r) for r in hi], True)
    assert [tuple(fm(v) for v in got[i]) for i in range(n)] == want


def test_chunked_evaluator_fq_equals_fp():
    log_n, lde_step = 10, 4
    n = 1 << log_n
    x = E.X()
    b = [lambda o=0, k=k: E.Trace(k, o) for k in range(3)]
    expr = (b[0](1) - b[1]() * b[2]()) * (x - E.Constant(3)) / (x ** 16 - 1) * (E.Challenge(0) * x ** 3 + E.Challenge(1))
    base, _ = _cols(n, 3, 0)
    ch = cref.random_elements(2, 5).reshape(2, 1)
    got = cref.eval_expr(expr, log_n, lde_step, 7, base, [], ch, ch[:1], False)
    want = evalexpr.eval_points(expr, range(n), n, lde_step, 7, [[fm(v) for v in c] for c in base], [],
                                [fm(v) for v in ch.ravel()], [fm(ch[0, 0])], False)
    assert [fm(v) for v in got] == want


def test_chunked_evaluator_252_bit_field():
    log_n, lde_step = 9, 2
    n = 1 << log_n
    x = E.X()
    c = [lambda o=0, k=k: E.Trace(k, o) for k in range(3)]
    expr = (c[0](1) - c[1]() * c[2]()) * (x - E.Constant(3)) / (x ** 8 - 1) * (E.Challenge(0) * x ** 3 + E.Challenge(1)) + c[2](-1) ** 5
    rng = np.random.default_rng(3)
    cols = [rng.integers(0, 1 << 63, size=4 * n, dtype=np.uint64) for _ in range(3)]
    for col in cols:
        col[3::4] >>= np.uint64(4)
    ch = rng.integers(0, 1 << 59, size=(2, 4), dtype=np.uint64)
    got = cref.eval_expr(expr, log_n, lde_step, 3, cols, [], ch, ch[:1], False, field="f252").reshape(n, 4)
    un = lambda w: F252.from_mont(sum(int(v) << (64 * i) for i, v in enumerate(w)))
    want = evalexpr.eval_points(expr, range(n), n, lde_step, 3, [[un(col[4 * i:4 * i + 4]) for i in range(n)] for col in cols], [],
                                [un(r) for r in ch], [un(ch[0])], False, field=F252)
    assert [un(got[i]) for i in range(n)] == want


@pytest.mark.parametrize("ext", [True, False])
def test_deep_composition_pieces(ext):
    """oracle_horner_eval / oracle_divide_out_points_acc / oracle_degree_adjust against oracle/pyref/deep.py."""
    n, PW = 64, (3 if ext else 1)
    base = [cref.random_elements(n, 50 + k) for k in range(2)]
    extp = [cref.random_elements(n * 3, 60)] if ext else []
    polys = base + extp
    Vs = [1, 1] + ([3] if ext else [])
    q = (lambda w: tuple(fm(v) for v in w)) if ext else (lambda w: fm(w[0]))
    rng = np.random.default_rng(5)
    zs = [cref.random_elements(PW * 2, 70 + k) for k in range(len(polys))]
    cs = [cref.random_elements(PW * 2, 80 + k) for k in range(len(polys))]
    degree = (cref.random_elements(PW, 90), cref.random_elements(PW, 91))
    canon_poly = lambda p, V: [fm(v) for v in p] if V == 1 else [tuple(fm(v) for v in p[3 * i:3 * i + 3]) for i in range(n)]
    # Horner
    for p, V in zip(polys, Vs):
        got = cref.horner_eval(p, V, zs[0][:PW])
        assert q(got) == odeep.horner_evaluate(canon_poly(p, V), q(zs[0][:PW]))
    # quotient sum + degree adjustment
    got = cref.deep_compose(polys, Vs, list(zip(zs, cs)), n, PW, degree).reshape(n, PW)
    acc = None
    for p, V, z2, c2 in zip(polys, Vs, zs, cs):
        col = odeep.divide_out_points_into(canon_poly(p, V), [q(z2[:PW]), q(z2[PW:])], [q(c2[:PW]), q(c2[PW:])])
        acc = col if acc is None else [odeep._add(a, b) for a, b in zip(acc, col)]
    da, db = q(degree[0]), q(degree[1])
    want, last = [], odeep._zero(ext)
    for cval in acc:
        want.append(odeep._add(odeep._mul(cval, da), odeep._mul(last, db)))
        last = cval
    assert [q(got[i]) for i in range(n)] == want
