"""TEST INFRASTRUCTURE (oracle): the data-parallel chain of `default_prove` (src/prover.rs:25-174) on the CPU, phase by phase through
oracle/c -- every transform, hash, constraint evaluation (`eval_cpu::eval`, src/eval_cpu.rs:33-150), the DEEP composition
(src/composer.rs:43-188), the FRI layers (src/fri.rs:179-249) and the proof of work -- with fixed draws in place of the channel.
Used by tests/test_pipeline_parity.py as the checker of `pipeline.prove_phases` / `distributed.prove_sharded` and by bench.py's
`cpu_baseline` leg of the C5 object; never by the product."""
import numpy as np

from oracle.pyref.fields import GL


def c5_oracle_chain(cols, log_t, blowup, folding, draws, comp_expr, ce_blowup=None):
    """The same transcript on the CPU: oracle/c for every transform, hash, evaluation and the DEEP composition.
    ce_blowup: the AIR's ce_blowup_factor (constraint-evaluation domain = trace_len * ce_blowup points, src/air.rs:55-59)."""
    import hashlib
    from oracle import cref
    ce_blowup = blowup if ce_blowup is None else ce_blowup
    log_b = blowup.bit_length() - 1
    log_l = log_t + log_b
    log_ce = log_t + ce_blowup.bit_length() - 1
    n_t, n_l = 1 << log_t, 1 << log_l
    R = lambda v: np.array([cref.lib().oracle_gl_to_mont(int(v) % cref.GL_P)], dtype=np.uint64)
    out = {}
    polys = [cref.ntt(c.copy(), log_t, 1, True, 1) for c in cols]
    lde_nat = [cref.lde(c, log_t, log_b, 1, 7, False) for c in cols]
    lde_br = [cref.bit_reverse(c.copy(), log_l) for c in lde_nat]
    out["base_root"] = cref.sha256_merkle(cref.sha256_rows(lde_br, 1))[1].tobytes()
    ch = np.array([R(c)[0] for c in draws.challenges], dtype=np.uint64).reshape(-1, 1)
    hints = np.array([R(c)[0] for c in draws.hints], dtype=np.uint64).reshape(-1, 1)
    # the constraint-evaluation coset h<w_(n ce)> in natural order: every (blowup / ce_blowup)-th point of the LDE coset
    ce_nat = [np.ascontiguousarray(c[::blowup // ce_blowup]) for c in lde_nat]
    comp_nat = cref.eval_expr(comp_expr, log_ce, ce_blowup, 7, ce_nat, [], ch, hints, False)     # prover.rs:97-107 (eval_cpu::eval)
    out["comp_evals_br"] = cref.bit_reverse(comp_nat.copy(), log_ce)
    comp_poly = cref.ntt(comp_nat.copy(), log_ce, 1, True, 7)                                  # prover.rs:111-112
    comp_polys = [np.ascontiguousarray(comp_poly[c::ce_blowup]) for c in range(ce_blowup)]    # prover.rs:113-121
    out["comp_polys"] = comp_polys

    def evaluate_br(coeffs):                                                                 # bit_reversed_evaluate on the LDE coset
        a = np.zeros(n_l, dtype=np.uint64)
        a[:len(coeffs)] = coeffs
        return cref.bit_reverse(cref.ntt(a, log_l, 1, False, 7), log_l)
    comp_lde = [evaluate_br(p) for p in comp_polys]
    out["composition_root"] = cref.sha256_merkle(cref.sha256_rows(comp_lde, 1))[1].tobytes()
    # DEEP (composer.rs:43-188), Fq = Fp
    g = GL.root_of_unity(n_t)
    z = draws.z
    pt = lambda off: (z * pow(g, off, GL.p)) % GL.p
    z_n = pow(z, ce_blowup, GL.p)
    exec_ood = [cref.horner_eval(polys[c], 1, R(pt(o))) for c, o in draws.trace_args]
    comp_ood = [cref.horner_eval(p, 1, R(z_n)) for p in comp_polys]
    out["ood"] = ([GL.from_mont(int(v[0])) for v in exec_ood], [GL.from_mont(int(v[0])) for v in comp_ood])
    terms = []
    for c in range(len(polys)):
        zs = [R(pt(o))[0] for (cc, o) in draws.trace_args if cc == c]
        al = [R(a)[0] for (cc, o), a in zip(draws.trace_args, draws.deep.execution_trace) if cc == c]
        terms.append((np.array(zs, dtype=np.uint64), np.array(al, dtype=np.uint64)))
    for c in range(ce_blowup):
        terms.append((R(z_n), R(draws.deep.composition_trace[c])))
    deep_poly = cref.deep_compose(polys + comp_polys, [1] * (len(polys) + ce_blowup), terms, n_t, 1,
                                  (R(draws.deep.degree[0]), R(draws.deep.degree[1])))
    out["deep_poly"] = deep_poly
    layer = evaluate_br(deep_poly)
    out["fri_roots"], n, fri_layers = [], n_l, []
    for alpha in draws.fri_alphas:
        fri_layers.append(layer)
        rows = [np.ascontiguousarray(layer[k::folding]) for k in range(folding)]              # rows of `folding` consecutive evaluations
        out["fri_roots"].append(cref.sha256_merkle(cref.sha256_rows(rows, 1))[1].tobytes())
        layer = cref.fri_fold(layer, n.bit_length() - 1, 1, folding, R(alpha), 1)
        n //= folding
    out["remainder"] = layer
    out["fri_layers"] = fri_layers
    # FriProver::set_remainder (fri.rs:232-248)
    log_r = n.bit_length() - 1
    out["remainder_coeffs"] = cref.ntt(cref.bit_reverse(layer.copy(), log_r), log_r, 1, True, 1)[: max(n // blowup, 1)]
    seed = out["fri_roots"][-1]
    nonce = 1
    while int.from_bytes(hashlib.sha256(seed + nonce.to_bytes(8, "big")).digest()[:8], "big") >> (64 - 8):
        nonce += 1
    out["nonce"] = nonce
    out["lde_br"], out["comp_lde"] = lde_br, comp_lde
    return out
