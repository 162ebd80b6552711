"""DEEP composition on device vs the oracle's literal restatement of src/composer.rs (synthetic
division per column, sum, degree adjustment), bit-exact; Horner OOD evaluations likewise."""
import numpy as np
import pytest

from oracle.pyref import deep as pydeep
from oracle.pyref.fields import GL
from tests import backends
from ministark_amd import GOLDILOCKS_FP as FP, GOLDILOCKS_FQ3 as FQ3, Matrix, Radix2EvaluationDomain
from ministark_amd.composer import DeepCompositionCoeffs, DeepPolyComposer

P = GL.p
KINDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


def _rq(rng, ext):
    v = tuple(int(x) for x in rng.integers(0, P, size=3, dtype=np.uint64))
    return v if ext else v[0]


def _mat(pl, cols, field):
    arrs = []
    for c in cols:
        flat = [w for e in c for w in (e if isinstance(e, tuple) else (e,))]
        arrs.append(np.array([GL.to_mont(x) for x in flat], dtype=np.uint64))
    return Matrix.from_numpy(pl, arrs, field)


def _canon_col(vec, ext):
    a = [GL.from_mont(int(x)) for x in vec.to_numpy()]
    return [tuple(a[3 * i:3 * i + 3]) for i in range(len(a) // 3)] if ext else a


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("log_n,ext,beta_zero", [(6, True, False), (8, True, True), (7, False, False), (13, True, False)])
def test_deep_matches_composer(kind, log_n, ext, beta_zero):
    if kind == "emu" and log_n > 8:
        pytest.skip("kept short under the simulator")
    pl = backends.planner(kind)
    rng = np.random.default_rng(log_n + ext)
    n = 1 << log_n
    nbase, next_, ncomp = 3, (2 if ext else 0), 2
    base = [[int(x) for x in rng.integers(0, P, size=n, dtype=np.uint64)] for _ in range(nbase)]
    extp = [[_rq(rng, True) for _ in range(n)] for _ in range(next_)]
    comp = [[_rq(rng, ext) for _ in range(n)] for _ in range(ncomp)]
    args = [(0, 0), (0, 1), (1, 0), (2, 1), (2, -1)] + ([(3, 0), (4, 1), (3, 1)] if ext else [])
    z = _rq(rng, ext)
    d = Radix2EvaluationDomain(n)
    g, g_inv = d.group_gen, d.group_gen_inv
    comp_field = FQ3 if ext else FP
    composer = DeepPolyComposer(args, n, z, _mat(pl, base, FP), _mat(pl, extp, FQ3) if ext else None, _mat(pl, comp, comp_field))
    got_exec, got_comp = composer.get_ood_evals()
    want_exec, want_comp = pydeep.get_ood_evals(z, g, g_inv, args, base, extp, comp)
    assert got_exec == want_exec and got_comp == want_comp
    ea = [_rq(rng, ext) for _ in args]
    ca = [_rq(rng, ext) for _ in range(ncomp)]
    zero = (0, 0, 0) if ext else 0
    degree = (_rq(rng, ext), zero if beta_zero else _rq(rng, ext))
    out = composer.into_deep_poly(DeepCompositionCoeffs(ea, ca, degree))
    want = pydeep.into_deep_poly(z, g, g_inv, args, base, extp, comp, ea, ca, degree)
    assert _canon_col(out, ext) == want


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("ext", [False, True])
def test_deep_points_share_one_inversion_across_points(kind, ext):
    # n >= 4096 with at most three (Fp) / four (Fq3) distinct out-of-domain points: deep_points takes 4 / 2 points per lane, whose
    # denominators share one inversion, and multiplies each point's quotient factor into the sum of its terms
    pl = backends.planner(kind)
    rng = np.random.default_rng(40 + ext)
    n = 1 << 12
    nbase, next_, ncomp = 2, (1 if ext else 0), 1
    base = [[int(x) for x in rng.integers(0, P, size=n, dtype=np.uint64)] for _ in range(nbase)]
    extp = [[_rq(rng, True) for _ in range(n)] for _ in range(next_)]
    comp = [[_rq(rng, ext) for _ in range(n)] for _ in range(ncomp)]
    args = [(0, 0), (0, 1), (1, 0), (1, 1)] + ([(2, 0), (2, 1)] if ext else [])          # points z, z g and z^1: three
    z = _rq(rng, ext)
    d = Radix2EvaluationDomain(n)
    g, g_inv = d.group_gen, d.group_gen_inv
    composer = DeepPolyComposer(args, n, z, _mat(pl, base, FP), _mat(pl, extp, FQ3) if ext else None, _mat(pl, comp, FQ3 if ext else FP))
    composer.get_ood_evals()
    ea, ca = [_rq(rng, ext) for _ in args], [_rq(rng, ext) for _ in range(ncomp)]
    degree = (_rq(rng, ext), _rq(rng, ext))
    out = composer.into_deep_poly(DeepCompositionCoeffs(ea, ca, degree))
    assert _canon_col(out, ext) == pydeep.into_deep_poly(z, g, g_inv, args, base, extp, comp, ea, ca, degree)


@pytest.mark.parametrize("kind", KINDS)
def test_horner_large(kind):
    # 2^20 coefficients: blocks of 16384 coefficients (64 per lane), multi-block reduction + host combine
    pl = backends.planner(kind)
    rng = np.random.default_rng(1)
    n = 1 << 20
    col = rng.integers(0, P, size=n, dtype=np.uint64)
    m = Matrix.from_numpy(pl, [col], FP)
    z = _rq(rng, True)
    cq = rng.integers(0, P, size=3 * 4096, dtype=np.uint64)
    comp = DeepPolyComposer([(0, 0), (0, 1)], n, z, m, None, Matrix.from_numpy(pl, [cq], FQ3))
    ex, cv = comp.get_ood_evals()
    cc = [GL.from_mont(int(x)) for x in cq]
    assert cv[0] == pydeep.horner_evaluate([tuple(cc[3 * i:3 * i + 3]) for i in range(4096)], z)
    canon = [GL.from_mont(int(x)) for x in col]
    d = Radix2EvaluationDomain(n)
    assert ex[0] == pydeep.horner_evaluate(canon, z)
    assert ex[1] == pydeep.horner_evaluate(canon, pydeep.point_for(z, d.group_gen, d.group_gen_inv, 1))


@pytest.mark.parametrize("kind", KINDS)
def test_deep_252(kind):
    # DeepPolyComposer over the 252-bit field (Fq = Fp): out-of-domain values against Python Horner, and the DEEP
    # polynomial against its definition  Q(X) = (a + b X) * sum_t alpha_t (P_t(X) - P_t(z_t)) / (X - z_t)  at random points
    from oracle.pyref.fields import F252
    from ministark_amd import STARK252_FP, f252_to_mont_limbs, f252_from_mont_limbs
    pl = backends.planner(kind)
    p = F252.p
    rng = np.random.default_rng(17)
    log_n = 7 if kind == "emu" else 12
    n = 1 << log_n
    rnd = lambda: int.from_bytes(rng.bytes(32), "little") % p
    nbase, ncomp = 3, 2
    polys = [[rnd() for _ in range(n)] for _ in range(nbase + ncomp)]
    mat = lambda cols: Matrix.from_numpy(pl, [np.concatenate([f252_to_mont_limbs(v) for v in c]) for c in cols], STARK252_FP)
    args = [(0, 0), (0, 1), (1, 0), (2, 1), (2, -1)]
    z = rnd()
    composer = DeepPolyComposer(args, n, z, mat(polys[:nbase]), None, mat(polys[nbase:]))
    got_exec, got_comp = composer.get_ood_evals()
    horner = lambda c, x: __import__("functools").reduce(lambda acc, v: (acc * x + v) % p, reversed(c), 0)
    g = F252.root_of_unity(n)
    point = lambda off: z * pow(g, off % n, p) % p
    z_n = pow(z, ncomp, p)
    assert got_exec == [horner(polys[c], point(o)) for c, o in args]
    assert got_comp == [horner(polys[nbase + c], z_n) for c in range(ncomp)]
    ea, ca, degree = [rnd() for _ in args], [rnd() for _ in range(ncomp)], (rnd(), rnd())
    q = composer.into_deep_poly(DeepCompositionCoeffs(ea, ca, degree)).to_numpy().reshape(n, 4)
    qc = [f252_from_mont_limbs(r) for r in q]
    terms = [(polys[nbase + c], z_n, ca[c]) for c in range(ncomp)] + [(polys[c], point(o), a) for (c, o), a in zip(args, ea)]
    for _ in range(3):
        r = rnd()
        want = sum(a * (horner(c, r) - horner(c, zt)) * pow(r - zt, -1, p) for c, zt, a in terms) % p
        want = want * (degree[0] + degree[1] * r) % p
        assert horner(qc, r) == want
    # degree bound: deg Q <= n - 2 before the adjustment, so the adjusted polynomial has n coefficients and no more
    assert len(qc) == n


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("ext", [False, True])
def test_horner_sums_of_unreduced_products_at_the_edges(kind, ext):
    # Fp coefficient columns: a lane adds sixteen unreduced 128-bit products c * y^k and reduces once (deep_kernels.h: Acc132).
    # Columns whose Montgomery words are all p - 1 (the largest factor), all 0, and mixed edge values, at points whose powers are
    # edge values themselves (z = 1, z = p - 1) and random ones; ragged length (n not a multiple of a block).
    pl = backends.planner(kind)
    rng = np.random.default_rng(77 + ext)
    n = 1 << 13
    words = [np.full(n, P - 1, dtype=np.uint64), np.zeros(n, dtype=np.uint64),
             rng.choice(np.array([0, 1, P - 1, P - 2, 0xFFFFFFFF, 1 << 32, 0xFFFFFFFF00000000], dtype=np.uint64), size=n)]
    m = Matrix.from_numpy(pl, words, FP)
    canon = [[GL.from_mont(int(x)) for x in w] for w in words]
    comp = [[_rq(rng, ext) for _ in range(8)]]
    for z in ((1, 0, 0) if ext else 1, (P - 1, 0, 0) if ext else P - 1, _rq(rng, ext), (0, 1, 0) if ext else 0):
        args = [(0, 0), (0, 1), (1, 0), (2, 0), (2, 1)]
        composer = DeepPolyComposer(args, n, z, m, None, _mat(pl, comp, FQ3 if ext else FP))
        got, _ = composer.get_ood_evals()
        d = Radix2EvaluationDomain(n)
        want = [pydeep.horner_evaluate(canon[c], pydeep.point_for(z, d.group_gen, d.group_gen_inv, off)) for c, off in args]
        assert got == want, z


@pytest.mark.parametrize("kind", KINDS)
def test_horner_three_levels_sparse(kind):
    # 2^25 coefficients: blocks of 4096 -> 8192 block values -> 2 -> 1, every level on the device (ms_horner_eval).  The column is
    # zero except at a few positions (first / last of a block, of a second-level block, the very last), so that the oracle is a
    # handful of powers: sum_i c_i z^i.
    pl = backends.planner(kind)
    rng = np.random.default_rng(9)
    n = 1 << 25
    pos = sorted(set([0, 1, 4095, 4096, (1 << 24) - 1, 1 << 24, (1 << 24) + 4097, n - 4096, n - 1] + [int(x) for x in rng.integers(0, n, size=12)]))
    vals = [int(x) for x in rng.integers(1, P, size=len(pos), dtype=np.uint64)]
    col = np.zeros(n, dtype=np.uint64)
    for i, v in zip(pos, vals):
        col[i] = GL.to_mont(v)
    m = Matrix.from_numpy(pl, [col], FP)
    z = _rq(rng, False)
    d = Radix2EvaluationDomain(n)
    composer = DeepPolyComposer([(0, 0), (0, 1)], n, z, m, None, _mat(pl, [[_rq(rng, False) for _ in range(4)]], FP))
    got, _ = composer.get_ood_evals()
    for off, g in zip((0, 1), got):
        x = pydeep.point_for(z, d.group_gen, d.group_gen_inv, off)
        assert g == sum(v * pow(x, i, P) for i, v in zip(pos, vals)) % P


@pytest.mark.parametrize("kind", KINDS)
def test_deep_rows_two_points_twenty_columns(kind):
    """The prover's own shape -- Fq = Fp, every column opened at z and g z, ONE composition column (its point z^1 = z): two points --
    against into_deep_poly + into_bit_reversed_evaluations, whole domain and a row shard; 20 columns per point: more than one 16-term
    window of the limb accumulators.  (Round 6 tried walking the terms by COLUMN, one load per column for both points: HBM fetches halve,
    2.23 -> 1.21 GB per 2^24 rows, and the kernel gets slower, 460 -> 490 us: one accumulator set per point costs the occupancy that hides its
    loads.  Not shipped; this test is what checked it.)"""
    from ministark_amd import Matrix
    pl = backends.planner(kind)
    n, blow = (4096, 2) if kind == "emu" else (1 << 14, 4)
    rng = np.random.default_rng(99)
    nbase = 20
    base = [[int(x) for x in rng.integers(0, P, size=n, dtype=np.uint64)] for _ in range(nbase)]
    comp = [[_rq(rng, False) for _ in range(n)]]
    args = [(c, o) for c in range(nbase) for o in (0, 1)][:-1]          # the last column at one point only
    z = _rq(rng, False)
    bm, cm = _mat(pl, base, FP), _mat(pl, comp, FP)
    composer = DeepPolyComposer(args, n, z, bm, None, cm)
    composer.get_ood_evals()
    co = DeepCompositionCoeffs([_rq(rng, False) for _ in args], [_rq(rng, False)], (_rq(rng, False), _rq(rng, False)))
    N = n * blow
    dom = Radix2EvaluationDomain(N, 7)
    want = Matrix([composer.into_deep_poly(co)]).into_bit_reversed_evaluations(dom).columns[0].to_numpy()
    bl, cl = bm.bit_reversed_evaluate(dom), cm.bit_reversed_evaluate(dom)
    assert np.array_equal(composer.into_deep_evaluations(co, bl, None, cl, N).to_numpy(), want)
    f, c = N // 2, N // 2                                                # the second half as a row shard
    sl = lambda m: Matrix.from_numpy(pl, [col[f:f + c] for col in m.to_numpy()], m.field)
    assert np.array_equal(composer.into_deep_evaluations(co, sl(bl), None, sl(cl), N, first=f).to_numpy(), want[f:f + c])


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("ext", [False, True])
def test_deep_evaluations_on_the_committed_ldes(kind, ext):
    """ms_deep_rows: the DEEP composition polynomial's bit-reversed LDE computed pointwise from the rows of the committed LDEs equals
    into_deep_poly + into_bit_reversed_evaluations (src/prover.rs:149-152) -- on the whole domain and on a row shard of it."""
    from ministark_amd import Matrix
    pl = backends.planner(kind)
    for n, blow in ((64, 4), (4096, 2)) if kind == "emu" else ((1 << 12, 8), (1 << 16, 4)):
        rng = np.random.default_rng(7 + ext)
        nbase, next_, ncomp = 2, (1 if ext else 0), 2
        base = [[int(x) for x in rng.integers(0, P, size=n, dtype=np.uint64)] for _ in range(nbase)]
        extp = [[_rq(rng, True) for _ in range(n)] for _ in range(next_)]
        comp = [[_rq(rng, ext) for _ in range(n)] for _ in range(ncomp)]
        args = [(0, 0), (0, 1), (1, 0), (1, -1)] + ([(2, 0), (2, 1)] if ext else [])
        z = _rq(rng, ext)
        bm, em, cm = _mat(pl, base, FP), (_mat(pl, extp, FQ3) if ext else None), _mat(pl, comp, FQ3 if ext else FP)
        composer = DeepPolyComposer(args, n, z, bm, em, cm)
        composer.get_ood_evals()
        co = DeepCompositionCoeffs([_rq(rng, ext) for _ in args], [_rq(rng, ext) for _ in range(ncomp)], (_rq(rng, ext), _rq(rng, ext)))
        N = n * blow
        dom = Radix2EvaluationDomain(N, 7)
        want = Matrix([composer.into_deep_poly(co)]).into_bit_reversed_evaluations(dom).columns[0].to_numpy()
        bl, cl = bm.bit_reversed_evaluate(dom), cm.bit_reversed_evaluate(dom)
        el = em.bit_reversed_evaluate(dom) if ext else None
        assert np.array_equal(composer.into_deep_evaluations(co, bl, el, cl, N).to_numpy(), want)
        f, c, V = 3 * N // 8, N // 8, (3 if ext else 1)                                   # rank 3 of 8
        sl = lambda m, words: Matrix.from_numpy(pl, [col[f * words:(f + c) * words] for col in m.to_numpy()], m.field)
        got = composer.into_deep_evaluations(co, sl(bl, 1), sl(el, 3) if ext else None, sl(cl, V), N, first=f).to_numpy()
        assert np.array_equal(got, want[f * V:(f + c) * V])
