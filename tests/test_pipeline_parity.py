"""The whole data-parallel prover pipeline against the oracle, bit for bit, on the shape of BASELINE config C1
(examples/brainfuck: 17 base Fp + 9 extension Fq3 columns, blow-up 16, FRI folding factor 16,
examples/brainfuck/main.rs:92-105; 128 rows here so that the big-integer oracle finishes in seconds).
Every phase of default_prove (src/prover.rs:25-174) that touches column data runs through the library;
the channel's draws are replaced by fixed values on both sides:

    base / extension trace: interpolate -> bit-reversed LDE -> Merkle root          prover.rs:50-79
    constraint evaluation on the LDE coset                                          prover.rs:98-107
    composition polynomial: iNTT -> chunks -> bit-reversed LDE -> Merkle root       prover.rs:110-124
    DEEP: out-of-domain evaluations, composition polynomial, its LDE                prover.rs:136-153
    FRI: commit + fold layers down to the remainder                                 fri.rs:179-249
    queries: rows and batched openings of the three trees                           trace.rs:113-157
"""
import numpy as np
import pytest

from oracle.pyref import deep as odeep
from oracle.pyref import evalexpr, fri as ofri, merkle as omerkle
from oracle.pyref import ntt as ontt
from oracle.pyref.fields import FQ3, GL, bit_reverse
from tests import backends
from ministark_amd import GOLDILOCKS_FP as FP, GOLDILOCKS_FQ3 as FQ3F, GpuVec, Matrix, MerkleTree, Queries, Radix2EvaluationDomain, apply_drp
from ministark_amd import expr as E
from ministark_amd.composer import DeepCompositionCoeffs, DeepPolyComposer

P = GL.p
KINDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
NBASE, NEXT, BLOWUP, FOLD, OFFSET = 17, 9, 16, 16, 7


def _mont(vals, ext):
    flat = [c for v in vals for c in v] if ext else vals
    return np.array([GL.to_mont(c) for c in flat], dtype=np.uint64)


def _canon(arr, ext):
    a = [GL.from_mont(int(x)) for x in arr]
    return [tuple(a[3 * i:3 * i + 3]) for i in range(len(a) // 3)] if ext else a


def _mat(pl, cols, ext):
    return Matrix([GpuVec.from_numpy(pl, _mont(c, ext), FQ3F if ext else FP) for c in cols])


@pytest.mark.parametrize("kind", KINDS)
def test_prover_pipeline_c1_shape(kind):
    pl = backends.planner(kind)
    rng = np.random.default_rng(2024)
    n = 128
    N = n * BLOWUP
    rq = lambda: tuple(int(x) for x in rng.integers(0, 1 << 62, size=3))
    base = [[int(x) for x in rng.integers(0, 1 << 62, size=n)] for _ in range(NBASE)]
    ext = [[rq() for _ in range(n)] for _ in range(NEXT)]
    challenges = [rq() for _ in range(4)]
    trace_dom, lde_dom = Radix2EvaluationDomain(n), Radix2EvaluationDomain(N, OFFSET)
    o_trace, o_lde = ontt.Domain(GL, n), ontt.Domain(GL, N, OFFSET)

    # ---- base / extension trace commitments
    base_polys = _mat(pl, base, False).interpolate(trace_dom)
    ext_polys = _mat(pl, ext, True).interpolate(trace_dom)
    base_lde, ext_lde = base_polys.bit_reversed_evaluate(lde_dom), ext_polys.bit_reversed_evaluate(lde_dom)
    base_tree, ext_tree = MerkleTree.from_matrix(base_lde), MerkleTree.from_matrix(ext_lde)
    o_base_polys = [ontt.ifft(o_trace, c) for c in base]
    o_ext_polys = [ontt.ifft(o_trace, c) for c in ext]
    o_base_nat = [ontt.fft(o_lde, c) for c in o_base_polys]
    o_ext_nat = [ontt.fft(o_lde, c) for c in o_ext_polys]
    o_base_lde, o_ext_lde = [bit_reverse(c) for c in o_base_nat], [bit_reverse(c) for c in o_ext_nat]
    o_base_leaves, o_ext_leaves = omerkle.hash_rows(GL, o_base_lde), omerkle.hash_rows(FQ3, o_ext_lde)
    o_base_nodes, o_ext_nodes = omerkle.build_merkle_nodes(o_base_leaves), omerkle.build_merkle_nodes(o_ext_leaves)
    assert base_tree.root() == o_base_nodes[1] and ext_tree.root() == o_ext_nodes[1]

    # ---- constraint evaluation over the LDE coset (natural order), 9 permutation-style constraints over a zerofier
    x = E.X()
    b = lambda c, o=0: E.Trace(c, o)
    e = lambda c, o=0: E.Trace(NBASE + c, o)
    expr = None
    for k in range(NEXT):
        t = (e(k, 1) - e(k) * (E.Challenge(k % 4) - b(k) * E.Challenge((k + 1) % 4) - b(k + 8, 1))) * (x - 1) / (x ** n - 1)
        expr = t if expr is None else expr + t * E.Challenge(k % 4)
    prog = E.compile_expr(expr, NBASE, True)
    ch = _mont(challenges, True).reshape(-1, 3)
    base_nat, ext_nat = base_lde.clone().bit_reverse_rows(), ext_lde.clone().bit_reverse_rows()
    comp_evals = E.eval(prog, pl, ch, ch[:1], BLOWUP, OFFSET, N, base_nat.columns, ext_nat.columns)
    o_comp_evals = evalexpr.eval_points(expr, list(range(N)), N, BLOWUP, OFFSET, o_base_nat, o_ext_nat, challenges, challenges[:1], True)
    assert _canon(comp_evals.to_numpy(), True) == o_comp_evals

    # ---- composition trace: coefficients, chunks(ce_blowup), LDE, commitment
    comp_poly = Matrix([comp_evals]).into_polynomials(lde_dom).columns[0]
    comp_polys = Matrix.from_chunks(comp_poly, BLOWUP)
    comp_lde = comp_polys.bit_reversed_evaluate(lde_dom)
    comp_tree = MerkleTree.from_matrix(comp_lde)
    o_comp_poly = ontt.ifft(o_lde, o_comp_evals)
    o_comp_polys = [o_comp_poly[c::BLOWUP] for c in range(BLOWUP)]
    o_comp_lde = [bit_reverse(ontt.fft(o_lde, c)) for c in o_comp_polys]
    o_comp_leaves = omerkle.hash_rows(FQ3, o_comp_lde)
    o_comp_nodes = omerkle.build_merkle_nodes(o_comp_leaves)
    assert comp_tree.root() == o_comp_nodes[1]

    # ---- DEEP composition
    z = rq()
    args = [(c, o) for c in range(NBASE + NEXT) for o in (0, 1)]
    composer = DeepPolyComposer(args, n, z, base_polys, ext_polys, comp_polys)
    got_exec, got_comp = composer.get_ood_evals()
    g, g_inv = trace_dom.group_gen, trace_dom.group_gen_inv
    want_exec, want_comp = odeep.get_ood_evals(z, g, g_inv, args, o_base_polys, o_ext_polys, o_comp_polys)
    assert got_exec == want_exec and got_comp == want_comp
    ea, ca, degree = [rq() for _ in args], [rq() for _ in range(BLOWUP)], (rq(), rq())
    deep_poly = composer.into_deep_poly(DeepCompositionCoeffs(ea, ca, degree))
    o_deep_poly = odeep.into_deep_poly(z, g, g_inv, args, o_base_polys, o_ext_polys, o_comp_polys, ea, ca, degree)
    assert _canon(deep_poly.to_numpy(), True) == o_deep_poly
    layer = Matrix([deep_poly]).into_bit_reversed_evaluations(lde_dom).columns[0]
    o_layer = bit_reverse(ontt.fft(o_lde, o_deep_poly))
    assert _canon(layer.to_numpy(), True) == o_layer

    # ---- FRI layers (fri.rs:179-231): commit the layer, draw alpha, fold by 16, until the remainder is small
    size, alphas = N, [rq() for _ in range(4)]
    k = 0
    while size > 64:
        tree = MerkleTree.from_fri_layer(layer, FOLD)
        rows = [[o_layer[r * FOLD + j] for r in range(size // FOLD)] for j in range(FOLD)]       # Matrix::from_arrays(as_chunks)
        assert tree.root() == omerkle.merkle_root(omerkle.hash_rows(FQ3, rows))
        layer = apply_drp(layer, _mont([alphas[k]], True), FOLD, 1)
        o_layer = ofri.apply_drp(GL, FQ3, o_layer, 1, alphas[k], FOLD)
        assert _canon(layer.to_numpy(), True) == o_layer
        size //= FOLD
        k += 1
    assert k == 2 and size == 8

    # ---- queries
    positions = [int(p) for p in rng.integers(0, N, size=12)] + [0, N - 1]
    q = Queries(base_lde, ext_lde, comp_lde, base_tree, ext_tree, comp_tree, positions)
    for rows, o_cols, V in ((q.base_trace_values, o_base_lde, 1), (q.extension_trace_values, o_ext_lde, 3), (q.composition_trace_values, o_comp_lde, 3)):
        for r, p in zip(rows, positions):
            want = [c[p] for c in o_cols]
            assert _canon(r, V == 3) == want
    for proof, leaves, nodes in ((q.base_trace_proof, o_base_leaves, o_base_nodes), (q.extension_trace_proof, o_ext_leaves, o_ext_nodes),
                                 (q.composition_trace_proof, o_comp_leaves, o_comp_nodes)):
        assert proof == omerkle.prove(leaves, nodes, positions)
        assert omerkle.verify(nodes[1], proof, positions)


# ---- the C5 shape (fib-like trace, ProofOptions::new(32, 4, 8, 8, 64)) against the C oracle, at size on the GPU ------------
from oracle.prover_chain import c5_oracle_chain as _c5_oracle_chain   # noqa: E402  (the chain lives with the oracle: bench.py times it too)


def _run_c5(kind, log_t, seed, air="fib"):
    """air = "fib": the reference's FibAirConfig (examples/fib/main.rs:73-140: 17 constraints, ce_blowup_factor 1);
    "additive": the cheaper stand-in of rounds 1-2 on a constraint-evaluation domain as large as the LDE domain."""
    from oracle import cref
    from ministark_amd import pipeline
    pl = backends.planner(kind)
    blowup, folding, ncols = 4, 8, 8
    n_t = 1 << log_t
    cols = [cref.random_elements(n_t, seed + c) for c in range(ncols)]
    comp, ce, nch = pipeline.fib_constraints(n_t, ncols) if air == "fib" else pipeline.additive_constraints(n_t, ncols, blowup)
    nlayers = pipeline.fri_num_layers(n_t * blowup, blowup, folding, 64)
    draws = pipeline.Draws(seed, ncols, nch, ce, 32, n_t * blowup, nlayers)
    got = pipeline.prove_phases(pl, Matrix.from_numpy(pl, cols, FP), comp, draws, blowup, folding, 64, 8, keep=True, ce_blowup=ce)
    want = _c5_oracle_chain(cols, log_t, blowup, folding, draws, comp, ce)
    assert got["base_root"] == want["base_root"]
    assert np.array_equal(got["comp_evals"].to_numpy(), want["comp_evals_br"])
    assert all(np.array_equal(g.to_numpy(), w) for g, w in zip(got["comp_polys"].columns, want["comp_polys"]))
    assert got["composition_root"] == want["composition_root"]
    assert ([int(v) for v in got["ood"][0]], [int(v) for v in got["ood"][1]]) == want["ood"]
    assert np.array_equal(got["deep_poly"].to_numpy(), want["deep_poly"])
    assert len(got["fri_roots"]) == nlayers and got["fri_roots"] == want["fri_roots"]
    assert np.array_equal(got["remainder"].to_numpy(), want["remainder"])
    assert got["nonce"] == want["nonce"]
    q = got["queries"]
    pos = draws.positions
    assert np.array_equal(q.base_trace_values, np.stack([c[pos] for c in want["lde_br"]], axis=1))
    assert np.array_equal(q.composition_trace_values, np.stack([c[pos] for c in want["comp_lde"]], axis=1))
    assert np.array_equal(got["remainder_coeffs"], want["remainder_coeffs"])
    # FRI layer openings (fri.rs:148-165): rows of `folding` consecutive evaluations at the folded positions + the Merkle views
    from oracle.pyref import merkle as omerkle
    fp = sorted(set(pos))
    assert len(got["fri_openings"]) == nlayers
    for opening, layer in zip(got["fri_openings"], want["fri_layers"]):
        fp = pipeline.fold_positions(fp, folding)
        assert opening["positions"] == fp
        rows = layer.reshape(-1, folding)
        assert np.array_equal(opening["rows"], rows[fp])
        cols = [np.ascontiguousarray(rows[:, k]) for k in range(folding)]
        leaves_np = cref.sha256_rows(cols, 1)
        leaves = [bytes(x) for x in leaves_np]
        nodes = [bytes(x) for x in cref.sha256_merkle(leaves_np)]
        assert opening["proof"] == omerkle.prove(leaves, nodes, fp)


def test_prover_pipeline_c5_shape_emu():
    _run_c5("emu", 7, 4242)


def test_prover_pipeline_additive_air_emu():
    _run_c5("emu", 7, 4243, air="additive")


@pytest.mark.gpu
def test_prover_pipeline_c5_at_size_hip():
    """BASELINE configs[4] on one GPU: 2^22 rows x 8 columns, blow-up 4 -- every commitment, evaluation, polynomial,
    FRI layer root, the remainder, the nonce and the queried rows against the CPU chain."""
    _run_c5("hip", 22, 0xC5)


@pytest.mark.gpu
def test_prover_pipeline_additive_air_hip():
    """The second shape: 8 additive transitions evaluated on the whole LDE domain (ce_blowup = 4, four composition columns)."""
    _run_c5("hip", 20, 0xADD, air="additive")


# ---- C1 at its REAL shape (BASELINE configs[0]; examples/brainfuck/main.rs:92-105, air.rs:26-27): 512 rows, 17 Fp + 9 Fq3 columns,
# blow-up 16, FRI folding factor 16 -- against oracle/c (the big-integer oracle above needs minutes at this size) ----------------
def _mw(v):
    """canonical Fq3 tuple / int -> Montgomery words"""
    return np.array([GL.to_mont(c) for c in (v if isinstance(v, tuple) else (v,))], dtype=np.uint64)


@pytest.mark.parametrize("kind", KINDS)
def test_prover_pipeline_c1_real_shape(kind):
    from oracle import cref
    from oracle.pyref import deep as od
    pl = backends.planner(kind)
    rng = np.random.default_rng(512)
    log_n, log_b = 9, 4
    n, N, log_N = 1 << log_n, 1 << (log_n + log_b), log_n + log_b
    rq = lambda: tuple(int(x) for x in rng.integers(0, 1 << 62, size=3))
    base = [cref.random_elements(n, 900 + c) for c in range(NBASE)]
    ext = [cref.random_elements(n, 950 + c, 3) for c in range(NEXT)]
    challenges = [rq() for _ in range(4)]
    trace_dom, lde_dom = Radix2EvaluationDomain(n), Radix2EvaluationDomain(N, OFFSET)
    mat = lambda cols, f: Matrix([GpuVec.from_numpy(pl, c, f) for c in cols])
    eq = lambda vec, want: np.array_equal(vec.to_numpy(), want)

    # ---- base / extension trace: interpolate, bit-reversed LDE, commitments (prover.rs:50-79)
    base_polys, ext_polys = mat(base, FP).interpolate(trace_dom), mat(ext, FQ3F).interpolate(trace_dom)
    base_lde, ext_lde = base_polys.bit_reversed_evaluate(lde_dom), ext_polys.bit_reversed_evaluate(lde_dom)
    base_tree, ext_tree = MerkleTree.from_matrix(base_lde), MerkleTree.from_matrix(ext_lde)
    o_base_polys = [cref.ntt(c, log_n, 1, True, 1) for c in base]
    o_ext_polys = [cref.ntt(c, log_n, 3, True, 1) for c in ext]
    assert all(eq(g, w) for g, w in zip(base_polys.columns, o_base_polys)) and all(eq(g, w) for g, w in zip(ext_polys.columns, o_ext_polys))
    o_base_nat = [cref.lde(c, log_n, log_b, 1, OFFSET, False) for c in base]
    o_ext_nat = [cref.lde(c, log_n, log_b, 3, OFFSET, False) for c in ext]
    o_base_lde = [cref.bit_reverse(c, log_N, 1) for c in o_base_nat]
    o_ext_lde = [cref.bit_reverse(c, log_N, 3) for c in o_ext_nat]
    assert all(eq(g, w) for g, w in zip(base_lde.columns, o_base_lde)) and all(eq(g, w) for g, w in zip(ext_lde.columns, o_ext_lde))
    o_base_nodes = cref.sha256_merkle(cref.sha256_rows(o_base_lde, 1))
    o_ext_nodes = cref.sha256_merkle(cref.sha256_rows(o_ext_lde, 3))
    assert base_tree.root() == o_base_nodes[1].tobytes() and ext_tree.root() == o_ext_nodes[1].tobytes()

    # ---- constraint evaluation over the LDE coset: 9 permutation-style constraints over a zerofier (prover.rs:98-107)
    x = E.X()
    b = lambda c, o=0: E.Trace(c, o)
    e = lambda c, o=0: E.Trace(NBASE + c, o)
    expr = None
    for k in range(NEXT):
        t = (e(k, 1) - e(k) * (E.Challenge(k % 4) - b(k) * E.Challenge((k + 1) % 4) - b(k + 8, 1))) * (x - 1) / (x ** n - 1)
        expr = t if expr is None else expr + t * E.Challenge(k % 4)
    prog = E.compile_expr(expr, NBASE, True)
    ch = np.array([_mw(c) for c in challenges], dtype=np.uint64)
    base_nat, ext_nat = base_lde.clone().bit_reverse_rows(), ext_lde.clone().bit_reverse_rows()
    comp_evals = E.eval(prog, pl, ch, ch[:1], BLOWUP, OFFSET, N, base_nat.columns, ext_nat.columns)
    o_comp_evals = cref.eval_expr(expr, log_N, BLOWUP, OFFSET, o_base_nat, o_ext_nat, ch, ch[:1], True)
    assert eq(comp_evals, o_comp_evals)

    # ---- composition trace: coefficients, chunks(ce_blowup), LDE, commitment (prover.rs:110-124)
    comp_poly = Matrix([comp_evals]).into_polynomials(lde_dom).columns[0]
    comp_polys = Matrix.from_chunks(comp_poly, BLOWUP)
    comp_lde = comp_polys.bit_reversed_evaluate(lde_dom)
    comp_tree = MerkleTree.from_matrix(comp_lde)
    o_comp_poly = cref.ntt(o_comp_evals, log_N, 3, True, OFFSET)
    o_comp_polys = [np.ascontiguousarray(o_comp_poly.reshape(-1, 3)[c::BLOWUP]).ravel() for c in range(BLOWUP)]
    assert all(eq(g, w) for g, w in zip(comp_polys.columns, o_comp_polys))

    def evaluate_br(coeffs):
        a = np.zeros(3 * N, dtype=np.uint64)
        a[:len(coeffs)] = coeffs
        return cref.bit_reverse(cref.ntt(a, log_N, 3, False, OFFSET), log_N, 3)
    o_comp_lde = [evaluate_br(p) for p in o_comp_polys]
    o_comp_nodes = cref.sha256_merkle(cref.sha256_rows(o_comp_lde, 3))
    assert comp_tree.root() == o_comp_nodes[1].tobytes()

    # ---- DEEP composition (composer.rs:43-188)
    z = rq()
    args = [(c, o) for c in range(NBASE + NEXT) for o in (0, 1)]
    composer = DeepPolyComposer(args, n, z, base_polys, ext_polys, comp_polys)
    got_exec, got_comp = composer.get_ood_evals()
    g, g_inv = trace_dom.group_gen, trace_dom.group_gen_inv
    z_n = od.qpow(z, BLOWUP)
    o_polys, o_Vs = o_base_polys + o_ext_polys, [1] * NBASE + [3] * NEXT
    want_exec = [tuple(int(v) for v in cref.from_mont(cref.horner_eval(o_polys[c], o_Vs[c], _mw(od.point_for(z, g, g_inv, o))))) for c, o in args]
    want_comp = [tuple(int(v) for v in cref.from_mont(cref.horner_eval(p, 3, _mw(z_n)))) for p in o_comp_polys]
    assert [tuple(v) for v in got_exec] == want_exec and [tuple(v) for v in got_comp] == want_comp
    ea, ca, degree = [rq() for _ in args], [rq() for _ in range(BLOWUP)], (rq(), rq())
    deep_poly = composer.into_deep_poly(DeepCompositionCoeffs(ea, ca, degree))
    terms = []
    for c in range(NBASE + NEXT):
        zs = [_mw(od.point_for(z, g, g_inv, o)) for (cc, o) in args if cc == c]
        al = [_mw(a) for (cc, o), a in zip(args, ea) if cc == c]
        terms.append((np.concatenate(zs), np.concatenate(al)))
    for c in range(BLOWUP):
        terms.append((_mw(z_n), _mw(ca[c])))
    o_deep_poly = cref.deep_compose(o_polys + o_comp_polys, o_Vs + [3] * BLOWUP, terms, n, 3, (_mw(degree[0]), _mw(degree[1])))
    assert eq(deep_poly, o_deep_poly)
    layer = Matrix([deep_poly]).into_bit_reversed_evaluations(lde_dom).columns[0]
    o_layer = evaluate_br(o_deep_poly)
    assert eq(layer, o_layer)

    # ---- FRI layers (fri.rs:179-231): commit, fold by 16 until the remainder is at most 64 evaluations
    size, alphas, k = N, [rq() for _ in range(4)], 0
    while size > 64:
        tree = MerkleTree.from_fri_layer(layer, FOLD)
        rows = [np.ascontiguousarray(o_layer.reshape(-1, 3)[j::FOLD]).ravel() for j in range(FOLD)]     # Matrix::from_arrays(as_chunks)
        assert tree.root() == cref.sha256_merkle(cref.sha256_rows(rows, 3))[1].tobytes()
        layer = apply_drp(layer, _mw(alphas[k]), FOLD, 1)
        o_layer = cref.fri_fold(o_layer, size.bit_length() - 1, 3, FOLD, _mw(alphas[k]), 1)
        assert eq(layer, o_layer)
        size //= FOLD
        k += 1
    assert k == 2 and size == 32

    # ---- queries (trace.rs:113-157): rows and batched openings of the three trees
    positions = [int(p) for p in rng.integers(0, N, size=30)] + [0, N - 1]
    q = Queries(base_lde, ext_lde, comp_lde, base_tree, ext_tree, comp_tree, positions)
    for rows, o_cols, V in ((q.base_trace_values, o_base_lde, 1), (q.extension_trace_values, o_ext_lde, 3), (q.composition_trace_values, o_comp_lde, 3)):
        for r, p in zip(rows, positions):
            assert np.array_equal(np.asarray(r, dtype=np.uint64).ravel(), np.concatenate([c[V * p:V * p + V] for c in o_cols]))
    for proof, nodes, cols, V in ((q.base_trace_proof, o_base_nodes, o_base_lde, 1), (q.extension_trace_proof, o_ext_nodes, o_ext_lde, 3),
                                  (q.composition_trace_proof, o_comp_nodes, o_comp_lde, 3)):
        leaves = [bytes(l) for l in cref.sha256_rows(cols, V)]
        node_list = [bytes(x) for x in nodes]
        assert proof == omerkle.prove(leaves, node_list, positions)
        assert omerkle.verify(node_list[1], proof, positions)
