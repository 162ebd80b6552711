"""The data-parallel phases of `default_prove` (src/prover.rs:25-174) for an Fq = Fp AIR, device-resident,
sequenced over the mirror in api.py / expr.py / composer.py -- what BASELINE.json's "end-to-end prove time"
measures on this backend (configs[4], the C5 shape: fib-like trace, ProofOptions::new(32, 4, 8, 8, 64),
examples/fib/main.rs:225).

The Fiat-Shamir channel (src/channel.rs: SHA-256 over a few digests, host work in the reference too) is
replaced by draws the caller fixes in advance, so that the CPU oracle can follow the same transcript and every
intermediate commitment can be compared (tests/test_pipeline_parity.py); nothing else is left out:
    interpolate + LDE + commit            prover.rs:50-55
    constraint evaluation                 prover.rs:88-107   (on the committed bit-reversed layout)
    composition trace                     prover.rs:111-124  (iNTT, split into blowup columns, LDE, commit)
    DEEP composition + its LDE            prover.rs:137-152  (composer.rs:43-188)
    FRI layers: commit + fold             fri.rs:179-231
    proof of work, query openings         prover.rs:160-173
"""
import time

import numpy as np

from . import expr as E
from .api import (GL_P, GOLDILOCKS_FP, Matrix, MerkleTree, Queries, Radix2EvaluationDomain, apply_drp, gl_to_mont,
                  grind_proof_of_work)
from .composer import DeepCompositionCoeffs, DeepPolyComposer


def fib_constraints(n_trace, ncols=8):
    """The composition constraint of an `ncols`-column Fibonacci-style AIR in the shape of examples/fib/main.rs:73-140:
    transition constraints c_k = c_(k-2) + c_(k-1) across and along rows, each divided by the transition zerofier
    (X - 3) / (X^n - 1) and degree-adjusted by (alpha_k X^3 + beta_k)."""
    x = E.X()
    c = [lambda o=0, k=k: E.Trace(k, o) for k in range(ncols)]
    cons = [c[0](1) - (c[ncols - 2]() + c[ncols - 1]()), c[1](1) - (c[ncols - 1]() + c[0](1))]
    cons += [c[k]() - (c[k - 2]() + c[k - 1]()) for k in range(2, ncols)]
    zer = (x - E.Constant(3)) / (x ** n_trace - 1)
    comp = None
    for k, cn in enumerate(cons):
        term = cn * zer * (E.Challenge(2 * k) * x ** 3 + E.Challenge(2 * k + 1))
        comp = term if comp is None else comp + term
    return comp, 2 * len(cons)


class Draws:
    """What the verifier's coin would supply, fixed up front (canonical integers of Fp)."""

    def __init__(self, seed, ncols, nchallenges, blowup, nqueries, n_lde, nlayers):
        rng = np.random.default_rng(seed)
        r = lambda k: [int(v) for v in rng.integers(1, GL_P, size=k, dtype=np.uint64)]
        self.challenges = r(nchallenges)
        self.z = r(1)[0]
        self.trace_args = [(c, o) for c in range(ncols) for o in (0, 1)]        # every column at the current and the next row
        self.deep = DeepCompositionCoeffs(r(len(self.trace_args)), r(blowup), (r(1)[0], r(1)[0]))
        self.fri_alphas = r(nlayers)
        self.positions = [int(p) for p in rng.integers(0, n_lde, size=nqueries)]


def fri_num_layers(n_lde, blowup, folding, max_remainder_coeffs):
    """FriOptions::num_layers (src/fri.rs:49-56)."""
    layers, n = 0, n_lde
    while n > max_remainder_coeffs * blowup:
        n //= folding
        layers += 1
    return layers


def prove_phases(planner, trace, comp_expr, draws, blowup=4, folding=8, max_remainder_coeffs=64, grinding_bits=8, hash="sha256",
                 keep=False):
    """trace: Matrix of Fp columns (2^k rows).  Returns dict(roots=..., fri_roots=[...], remainder=GpuVec, nonce=int,
    queries=Queries, phases_ms={...}); with keep=True also the intermediate device objects (for parity tests)."""
    pl = planner
    n_t = trace.num_rows()
    n_lde = n_t * blowup
    trace_dom, lde_dom = Radix2EvaluationDomain(n_t), Radix2EvaluationDomain(n_lde, 7)
    prog = E.compile_expr(comp_expr, trace.num_cols(), False)
    ch = np.array([gl_to_mont(c) for c in draws.challenges], dtype=np.uint64).reshape(-1, 1)
    out, phase = {}, {}
    t = time.perf_counter()

    def lap(name):
        nonlocal t
        pl.sync()
        now = time.perf_counter()
        phase[name] = (now - t) * 1e3
        t = now

    base_polys = trace.interpolate(trace_dom)                                  # prover.rs:50
    lde_t = base_polys.bit_reversed_evaluate(lde_dom)                          # prover.rs:51
    tree_t = MerkleTree.from_matrix(lde_t, hash)                               # prover.rs:52-55
    out["base_root"] = tree_t.root()
    lap("base trace: interpolate + LDE + commit")
    comp_evals = E.eval(prog, pl, ch, ch[:1], blowup, 7, n_lde, lde_t.columns, bit_reversed=True)       # prover.rs:88-107
    lap("constraint evaluation")
    kept_evals = comp_evals.clone() if keep else None                          # the next two steps work in place
    comp_poly = Matrix([comp_evals]).bit_reverse_rows().into_polynomials(lde_dom).columns[0]            # prover.rs:111-112
    comp_polys = Matrix.from_chunks(comp_poly, blowup)                         # prover.rs:113-121
    comp_lde = comp_polys.bit_reversed_evaluate(lde_dom)                       # prover.rs:122
    tree_c = MerkleTree.from_matrix(comp_lde, hash)                            # prover.rs:123-124
    out["composition_root"] = tree_c.root()
    lap("composition trace: iNTT + split + LDE + commit")
    composer = DeepPolyComposer(draws.trace_args, n_t, draws.z, base_polys, None, comp_polys)          # prover.rs:137-144
    out["ood"] = composer.get_ood_evals()                                      # prover.rs:145-146
    deep_poly = composer.into_deep_poly(draws.deep)                            # prover.rs:149
    deep = Matrix([deep_poly.clone() if keep else deep_poly]).into_bit_reversed_evaluations(lde_dom)   # prover.rs:150-152
    lap("DEEP: OOD evaluations + composition + LDE")
    cur, n, roots, layers = deep.columns[0], n_lde, [], []                     # fri.rs:179-231
    for alpha in draws.fri_alphas:
        roots.append(MerkleTree.from_fri_layer(cur, folding, hash).root())
        if keep:
            layers.append(cur)
        cur = apply_drp(cur, np.array([gl_to_mont(alpha)], dtype=np.uint64), folding, 1)
        n //= folding
    out["fri_roots"], out["remainder"] = roots, cur
    lap("FRI layers (commit + fold)")
    out["nonce"] = grind_proof_of_work(pl, roots[-1] if roots else out["composition_root"], grinding_bits)   # prover.rs:160
    out["queries"] = Queries(lde_t, None, comp_lde, tree_t, None, tree_c, draws.positions)                # prover.rs:163-173
    lap("proof of work + queries")
    out["phases_ms"] = {k: round(v, 3) for k, v in phase.items()}
    if keep:
        out.update(base_polys=base_polys, lde=lde_t, comp_evals=kept_evals, comp_polys=comp_polys, comp_lde=comp_lde,
                   deep_poly=deep_poly, deep_lde=deep, fri_layers=layers)
    return out
