"""The data-parallel phases of `default_prove` (src/prover.rs:25-174) for an Fq = Fp AIR, device-resident,
sequenced over the mirror in api.py / expr.py / composer.py -- what BASELINE.json's "end-to-end prove time"
measures on this backend (configs[4], the C5 shape: fib-like trace, ProofOptions::new(32, 4, 8, 8, 64),
examples/fib/main.rs:225).

The Fiat-Shamir channel (src/channel.rs: SHA-256 over a few digests, host work in the reference too) is
replaced by draws the caller fixes in advance, so that the CPU oracle can follow the same transcript and every
intermediate commitment can be compared (tests/test_pipeline_parity.py):
    interpolate + LDE + commit            prover.rs:50-55
    constraint evaluation                 prover.rs:88-107   (on the committed bit-reversed layout)
    composition trace                     prover.rs:111-124  (iNTT, split into ce_blowup columns, LDE, commit)
    DEEP composition + its LDE            prover.rs:137-152  (composer.rs:43-188)
    FRI layers: commit + fold             fri.rs:179-231
    FRI remainder                         fri.rs:232-248     (bit_reverse, iNTT on the subgroup; the first n / blowup coefficients)
    proof of work                         prover.rs:160, channel.rs:76-93
    trace / composition query openings    prover.rs:163-173  (trace.rs:113-157)
    FRI layer openings                    prover.rs:161, fri.rs:148-165, 615-622 (fold_positions, rows + Merkle views per layer)
What differs from a real run, on purpose: (i) the draws are fixed, so the proof-of-work seed is the last FRI root instead
of the channel's running digest (the same SHA-256 search either way); (ii) traces are random columns, not valid executions
(the data-parallel work does not depend on validity), so fri.rs:244's assertion that the remainder's high coefficients
vanish is not made; (iii) the composition constraint is lowered to its register program once per expression object, not once per proof
(an AIR's constraints are fixed; `_lowered`).  Proof serialisation and the channel's hashing of a few digests stay on the host in the reference too.
"""
import time

import numpy as np

from . import expr as E
from .api import (GL_P, GOLDILOCKS_FP, Matrix, MerkleTree, Queries, Radix2EvaluationDomain, apply_drp, gl_to_mont,
                  grind_proof_of_work)
from .composer import DeepCompositionCoeffs, DeepPolyComposer


def _degree(e, trace_degree):
    """`Constraint::degree` (src/constraints.rs:32-41, 152-158, 407-455): an upper bound (numerator, denominator) on the
    degree in X, with the reference's own (loose) arithmetic."""
    k, a = e.kind, e.args
    if k in ("const", "challenge", "hint"):
        return (0, 0)
    if k == "trace":
        return (trace_degree, 0)
    if k == "x":
        return (1, 0)
    if k == "periodic":                                   # PeriodicColumn::degree (src/constraints.rs:135-141)
        coeffs, interval = a[0], a[1]
        return ((len(coeffs) - 1) * ((trace_degree + 1) // interval), 0)
    if k == "neg":
        return _degree(a[0], trace_degree)
    if k == "pow":
        n, d = _degree(a[0], trace_degree)
        return (n * a[1], d * a[1])
    (an, ad), (bn, bd) = _degree(a[0], trace_degree), _degree(a[1], trace_degree)
    if k == "add":
        return (max(an + bd, bn + ad), ad + bd)
    if k == "mul":
        return (an + bn, ad + bd)
    if k == "div":
        return (an + bd, ad + bn)
    raise ValueError(k)


def _ceil_power_of_two(v):                                # src/utils.rs:76-82
    return v if v and (v & (v - 1)) == 0 else 1 << max(v, 1).bit_length() if v else 1


def constraint_blowup_factor(c, trace_len):
    """`Constraint::blowup_factor` (src/constraints.rs:162-166, 340-347) -- note the division by trace_len - 1."""
    n, d = _degree(c, trace_len - 1)
    return _ceil_power_of_two(max(n - d, 0)) // (trace_len - 1)


def composition_constraint(trace_len, constraints):
    """`AirConfig::composition_constraint` (src/air.rs:50-82): sum_i c_i (X^adj_i alpha_i + beta_i) with
    adj_i = (trace_len ce_blowup - 1) - (deg num_i - deg den_i).  CompositionCoeff(i) is Challenge(i) here (the AIRs this
    module drives have no other challenges; eval_constraint substitutes them as constants, src/air.rs:96-101).
    Returns (expression, ce_blowup_factor, number of composition coefficients)."""
    ce = max(constraint_blowup_factor(c, trace_len) for c in constraints)
    composition_degree = trace_len * ce - 1
    x = E.X()
    comp = None
    for i, c in enumerate(constraints):
        n, d = _degree(c, trace_len - 1)
        assert n - d <= composition_degree
        adj = composition_degree - (n - d)
        term = c * (x ** adj * E.Challenge(2 * i) + E.Challenge(2 * i + 1))
        comp = term if comp is None else comp + term
    return comp, ce, 2 * len(constraints)


def fib_air_constraints(trace_len, field=GOLDILOCKS_FP):
    """`FibAirConfig::constraints` (examples/fib/main.rs:73-140), in its order: 8 boundary constraints divided by (X - 1),
    the terminal constraint divided by (X - g^-1), 8 multiplicative transition constraints times (X - g^-1) / (X^n - 1).
    Hint(0) is the claimed n-th value.  field: the base field the trace domain lives in (the example's is Goldilocks;
    src/eval_gpu.rs:1054-1082 runs its evaluator over the 252-bit field as well)."""
    x = E.X()
    dom = Radix2EvaluationDomain(trace_len, 1, field)
    first_x, last_x = 1, pow(dom.group_gen, trace_len - 1, dom.p)
    curr, nxt = (lambda k: E.Trace(k, 0)), (lambda k: E.Trace(k, 1))
    v = [1, 2, 2]
    for k in range(3, 8):
        v.append(v[k - 2] * v[k - 1] % dom.p)                      # 4, 8, 32, 256, 8192
    boundary = [(curr(k) - E.Constant(v[k])) / (x - E.Constant(first_x)) for k in range(8)]
    terminal = [(curr(7) - E.Hint(0)) / (x - E.Constant(last_x))]
    tr = [nxt(0) - curr(6) * curr(7), nxt(1) - curr(7) * nxt(0)] + [nxt(k) - nxt(k - 2) * nxt(k - 1) for k in range(2, 8)]
    zer = (x - E.Constant(last_x)) / (x ** trace_len - E.Constant(1))
    return boundary + terminal + [t * zer for t in tr]


def fib_constraints(n_trace, ncols=8, field=GOLDILOCKS_FP):
    """The reference's fib AIR as `default_prove` sees it: (composition constraint, ce_blowup_factor, number of composition
    coefficients).  ce_blowup_factor is 1 for this AIR (every constraint has evaluation degree <= n - 1), i.e. the
    composition polynomial has n coefficients and one column (src/prover.rs:111-124)."""
    assert ncols == 8, "examples/fib has 8 columns"
    return composition_constraint(n_trace, fib_air_constraints(n_trace, field))


def additive_constraints(n_trace, ncols=8, ce_blowup=4):
    """A second, cheaper shape (the round-1/2 stand-in, kept as an extra case): additive transitions c_k = c_(k-2) + c_(k-1),
    each times (X - 3) / (X^n - 1) and (alpha_k X^3 + beta_k), evaluated on a constraint-evaluation domain of `ce_blowup` n points."""
    x = E.X()
    c = [lambda o=0, k=k: E.Trace(k, o) for k in range(ncols)]
    cons = [c[0](1) - (c[ncols - 2]() + c[ncols - 1]()), c[1](1) - (c[ncols - 1]() + c[0](1))]
    cons += [c[k]() - (c[k - 2]() + c[k - 1]()) for k in range(2, ncols)]
    zer = (x - E.Constant(3)) / (x ** n_trace - 1)
    comp = None
    for k, cn in enumerate(cons):
        term = cn * zer * (E.Challenge(2 * k) * x ** 3 + E.Challenge(2 * k + 1))
        comp = term if comp is None else comp + term
    return comp, ce_blowup, 2 * len(cons)


def mixed_air_constraints():
    """A 17 Fp + 9 Fq3-column composition in the shape of the brainfuck AIR (examples/brainfuck/air.rs:26-27, 68-125:
    running-product style extension columns driven by base columns and challenges, transition zerofier (X - 1) / (X^64 - 1),
    one boundary-style term divided by (X - 3)); 4 Fq3 challenges.  -> (expression, number of challenges)."""
    x = E.X()
    b = [lambda o=0, k=k: E.Trace(k, o) for k in range(17)]
    e = [lambda o=0, k=k: E.Trace(17 + k, o) for k in range(9)]
    expr = None
    for k in range(9):
        t = (e[k](1) - e[k]() * (E.Challenge(k % 4) - b[k]() * E.Challenge((k + 1) % 4) - b[k + 8](1))) * (x - 1) / (x ** 64 - 1)
        expr = t if expr is None else expr + t * E.Challenge(k % 4)
    expr = expr + (b[16]() ** 2 - b[16]()) * e[0]() / (x - E.Constant(3))
    return expr, 4


class Draws:
    """What the verifier's coin would supply, fixed up front (canonical integers of Fp)."""

    def __init__(self, seed, ncols, nchallenges, ce_blowup, nqueries, n_lde, nlayers):
        rng = np.random.default_rng(seed)
        r = lambda k: [int(v) for v in rng.integers(1, GL_P, size=k, dtype=np.uint64)]
        self.challenges = r(nchallenges)                                        # the composition coefficients (alpha_i, beta_i)
        self.hints = r(1)                                                       # FibHint::ClaimedNthFibNum
        self.z = r(1)[0]
        self.trace_args = [(c, o) for c in range(ncols) for o in (0, 1)]        # every column at the current and the next row
        self.deep = DeepCompositionCoeffs(r(len(self.trace_args)), r(ce_blowup), (r(1)[0], r(1)[0]))
        self.fri_alphas = r(nlayers)
        self.positions = [int(p) for p in rng.integers(0, n_lde, size=nqueries)]


def fold_positions(positions, folding_factor):
    """`fold_positions` (src/fri.rs:615-622): strictly increasing positions -> their cosets, deduplicated."""
    assert all(a < b for a, b in zip(positions, positions[1:]))
    out = []
    for p in positions:
        if not out or out[-1] != p // folding_factor:
            out.append(p // folding_factor)
    return out


def fri_layer_rows_launch(layer, folding_factor, positions, batch=None):
    """Rows `positions` of `Matrix::from_arrays(evaluations.as_chunks::<N>())` (src/fri.rs:213-215): N consecutive
    evaluations each, gathered on the device (32-byte records of ms_gather_digests).  Returns a function that downloads
    them as numpy [len(positions), N * words]."""
    from .api import FIELD_WORDS, DeviceBytes
    pl = layer.planner
    words = folding_factor * FIELD_WORDS[layer.field]
    if words % 4:
        return lambda: layer.to_numpy().reshape(-1, words)[positions]      # rows shorter than a 32-byte record: tiny layers only
    per = words // 4
    ids = (np.asarray(positions, dtype=np.uint64)[:, None] * np.uint64(per) + np.arange(per, dtype=np.uint64)).ravel()
    from .api import _gather_slot
    ptr, read, keep = _gather_slot(pl, 32 * len(ids), batch)
    nrec = len(layer) * FIELD_WORDS[layer.field] // 4
    if keep is batch and batch is not None:                    # a slice of the batch: joins its one launch (GatherBatch.flush)
        if ids.size and int(ids.max()) >= nrec:
            raise IndexError(f"row {int(ids.max()) // per} out of range")
        batch.defer_digests(layer.ptr, nrec, ids, ptr)
    else:
        pl.lib.check(pl.lib.ms_gather_digests(pl.handle, nrec, layer.ptr, ids.ctypes.data, len(ids), ptr))
    return lambda _keep=keep: np.array(read()[: 32 * len(ids)]).view(np.uint64).reshape(len(positions), words)


def fri_layer_rows(layer, folding_factor, positions):
    return fri_layer_rows_launch(layer, folding_factor, positions)()


def fri_num_layers(n_lde, blowup, folding, max_remainder_coeffs):
    """FriOptions::num_layers (src/fri.rs:49-56)."""
    layers, n = 0, n_lde
    while n > max_remainder_coeffs * blowup:
        n //= folding
        layers += 1
    return layers


_LOWERED = {}


def _lowered(comp_expr, ncols):
    """The register program of an AIR's composition constraint, lowered once per expression object (an AIR's constraints are fixed; the
    C++ example compiles its program outside the proof loop as well).  Keyed by identity: the expression is kept alive by the entry."""
    key = (id(comp_expr), ncols)
    hit = _LOWERED.get(key)
    if hit is None or hit[0] is not comp_expr:
        if len(_LOWERED) > 16:
            _LOWERED.clear()
        hit = (comp_expr, E.compile_expr(comp_expr, ncols, False))
        _LOWERED[key] = hit
    return hit[1]


def prove_phases(planner, trace, comp_expr, draws, blowup=4, folding=8, max_remainder_coeffs=64, grinding_bits=8, hash="sha256",
                 keep=False, ce_blowup=None, time_phases=True):
    """trace: Matrix of Fp columns (2^k rows).  ce_blowup: the AIR's ce_blowup_factor (src/air.rs:55-59; the constraint
    evaluation domain has trace_len * ce_blowup points, the composition trace ce_blowup columns); None = the LDE blow-up.
    Returns dict(roots=..., fri_roots=[...], remainder=GpuVec, nonce=int, queries=Queries, phases_ms={...}); with keep=True
    also the intermediate device objects (for parity tests).  time_phases=False: no device wait at the phase boundaries (two of the six
    are waits the proof itself does not need: after the evaluation and after DEEP) and no `phases_ms`."""
    pl = planner
    n_t = trace.num_rows()
    n_lde = n_t * blowup
    ce_blowup = blowup if ce_blowup is None else ce_blowup
    assert ce_blowup <= blowup                                                 # src/air.rs:149
    n_ce = n_t * ce_blowup
    trace_dom, lde_dom, ce_dom = Radix2EvaluationDomain(n_t), Radix2EvaluationDomain(n_lde, 7), Radix2EvaluationDomain(n_ce, 7)
    prog = _lowered(comp_expr, trace.num_cols())
    ch = np.array([gl_to_mont(c) for c in draws.challenges], dtype=np.uint64).reshape(-1, 1)
    hints = np.array([gl_to_mont(c) for c in draws.hints], dtype=np.uint64).reshape(-1, 1)
    out, phase = {}, {}
    t = time.perf_counter()

    def lap(name):
        nonlocal t
        if not time_phases:
            return
        pl.sync()
        now = time.perf_counter()
        phase[name] = (now - t) * 1e3
        t = now

    base_polys = trace.interpolate(trace_dom)                                  # prover.rs:50
    lde_t = base_polys.bit_reversed_evaluate(lde_dom)                          # prover.rs:51
    tree_t = MerkleTree.from_matrix(lde_t, hash)                               # prover.rs:52-55
    out["base_root"] = tree_t.root()
    lap("base trace: interpolate + LDE + commit")
    # the first n_ce rows of the committed (bit-reversed) LDE are the constraint-evaluation coset in its own bit-reversed order:
    # the evaluator works on them where they lie (the reference re-orders them, bit_reverse_ce_trace, prover.rs:88-91)
    comp_evals = E.eval(prog, pl, ch, hints, ce_blowup, 7, n_ce, lde_t.columns, bit_reversed=True)      # prover.rs:97-107
    lap("constraint evaluation")
    kept_evals = comp_evals.clone() if keep else None                          # the next two steps work in place
    comp_poly = Matrix([comp_evals]).bit_reverse_rows().into_polynomials(ce_dom).columns[0]             # prover.rs:111-112
    comp_polys = Matrix.from_chunks(comp_poly, ce_blowup)                      # prover.rs:113-121
    comp_lde = comp_polys.bit_reversed_evaluate(lde_dom)                       # prover.rs:122
    tree_c = MerkleTree.from_matrix(comp_lde, hash)                            # prover.rs:123-124
    out["composition_root"] = tree_c.root()
    lap("composition trace: iNTT + split + LDE + commit")
    composer = DeepPolyComposer(draws.trace_args, n_t, draws.z, base_polys, None, comp_polys)          # prover.rs:137-144
    out["ood"] = composer.get_ood_evals()                                      # prover.rs:145-146
    # prover.rs:149-152: deep_composition_poly = composer.into_deep_poly(coeffs); its bit-reversed evaluations over the LDE domain are
    # the first FRI layer.  Both committed LDEs are still resident (the queries need them), so those evaluations are computed where the
    # LDEs lie -- the quotient is a polynomial: same values -- instead of 9 coset transforms, the composition, an inverse transform and
    # an LDE (ms_deep_rows; tests/test_deep_parity.py checks it against the two-step form).  keep=True also forms the coefficients.
    deep_poly = composer.into_deep_poly(draws.deep) if keep else None
    deep = Matrix([composer.into_deep_evaluations(draws.deep, lde_t, None, comp_lde, n_lde)])
    lap("DEEP: OOD evaluations + composition + LDE")
    cur, n, roots, layers, fri_layers, fri_trees = deep.columns[0], n_lde, [], [], [], []     # fri.rs:179-231
    for alpha in draws.fri_alphas:
        tree = MerkleTree.from_fri_layer(cur, folding, hash)
        roots.append(tree.root())
        fri_layers.append(cur); fri_trees.append(tree)                        # FriLayer { merkle_tree, evaluations } (fri.rs:218-221)
        if keep:
            layers.append(cur)
        cur = apply_drp(cur, np.array([gl_to_mont(alpha)], dtype=np.uint64), folding, 1)
        n //= folding
    out["fri_roots"], out["remainder"] = roots, cur
    # FriProver::set_remainder (fri.rs:232-248): bit_reverse, iNTT over the subgroup of the remainder's size, keep n / blowup coefficients
    rem = Matrix([cur.clone()]).bit_reverse_rows().into_polynomials(Radix2EvaluationDomain(n)).columns[0]
    out["remainder_coeffs"] = rem.to_numpy()[: max(n // blowup, 1)]
    lap("FRI layers (commit + fold) + remainder")
    fine, tf = {}, time.perf_counter()

    def sub(name):                                                             # host-side split of the last phase (no synchronisation)
        nonlocal tf
        now = time.perf_counter()
        fine[name] = round((now - tf) * 1e3, 3)
        tf = now

    out["nonce"] = grind_proof_of_work(pl, roots[-1] if roots else out["composition_root"], grinding_bits)   # prover.rs:160
    sub("proof of work")
    from .api import GatherBatch
    batch = GatherBatch(pl)                                                    # every gather of the phase into one buffer: ONE download
    queries = Queries(lde_t, None, comp_lde, tree_t, None, tree_c, draws.positions, batch)                # prover.rs:163-173
    sub("trace openings: index walks + gather launches")
    # fri_prover.into_proof(&query_positions) (prover.rs:161, fri.rs:148-165): per layer the folded positions' rows and Merkle view
    pos, launched = sorted(set(int(p) for p in draws.positions)), []
    for layer, tree in zip(fri_layers, fri_trees):                            # all gathers first, then the download
        pos = fold_positions(pos, folding)
        launched.append((pos, fri_layer_rows_launch(layer, folding, pos, batch), tree.prove_launch(pos, batch)))
    sub("FRI openings: index walks + gather launches")
    batch.fetch()
    sub("wait + download")
    out["queries"] = queries.fetch()
    out["fri_openings"] = [{"positions": p, "rows": rows(), "proof": proof()} for p, rows, proof in launched]
    sub("assembly")
    out["openings_ms"] = fine
    lap("proof of work + queries")
    out["phases_ms"] = {k: round(v, 3) for k, v in phase.items()}
    if keep:
        out.update(base_polys=base_polys, lde=lde_t, comp_evals=kept_evals, comp_polys=comp_polys, comp_lde=comp_lde,
                   deep_poly=deep_poly, deep_lde=deep, fri_layers=layers)
    return out
