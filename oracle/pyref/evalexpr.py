"""ORACLE (test infrastructure) -- point-wise evaluation of a constraint expression DAG.

Restates what `eval_cpu::eval` computes (src/eval_cpu.rs:33-150) the way the reference's own
tests check it: per LDE point, by direct recursion over the DAG (`Expr::eval`,
src/eval_gpu.rs:977-990), with big-int arithmetic:
  X -> offset * w^i ; Trace(col, off) -> column[(i + lde_step*off) mod n] (eval_cpu.rs:115-134);
  a node is Fp iff both operands are Fp, else Fq (eval_cpu.rs:306-428);
  x / y = x * y^-1 with 0^-1 = 0 (ark_ff::batch_inversion leaves zeros, eval_cpu.rs:280-294);
  the result is returned as Fq (into_fq_array, eval_cpu.rs:262-275).
Works on any DAG whose nodes expose `.kind` and `.args` (duck-typed; nothing is imported from the
product package).  Periodic columns follow eval_periodic_column (eval_cpu.rs:233-256).
"""
from .fields import FQ3, GL
from .ntt import Domain, fft

P = GL.p


def _is_q(v):
    return isinstance(v, tuple)


def _emb(v):
    return v if _is_q(v) else (v, 0, 0)


def _add(a, b):
    if not _is_q(a) and not _is_q(b):
        return (a + b) % P
    return FQ3.add(_emb(a), _emb(b))


def _mul(a, b):
    if not _is_q(a) and not _is_q(b):
        return (a * b) % P
    if _is_q(a) and _is_q(b):
        return FQ3.mul(a, b)
    if _is_q(a):
        return FQ3.mul_base(a, b)
    return FQ3.mul_base(b, a)


def _inv(a):
    if _is_q(a):
        return (0, 0, 0) if a == (0, 0, 0) else FQ3.inv(a)
    return 0 if a == 0 else pow(a, -1, P)


def _pow(a, e):
    if _is_q(a):
        return FQ3.pow(a, e)
    return pow(a, e, P)


def eval_points(expr, points, n, lde_step, domain_offset, base_cols, ext_cols, challenges, hints, fq_is_ext=True, field=GL):
    """Evaluate at each i in `points`.  Columns / challenges / hints are canonical (ints or
    3-tuples).  Returns a list of Fq values (3-tuples, or ints when fq_is_ext is False).
    `field`: the base PrimeField (GL, or F252 with fq_is_ext=False)."""
    global P
    P = field.p
    w = field.root_of_unity(n)
    trace_len = n // lde_step
    periodic_cache = {}

    def periodic(coeffs, interval):
        k = (coeffs, interval)
        if k not in periodic_cache:
            size = interval * lde_step
            d = Domain(field, size, pow(domain_offset, trace_len // interval, P))
            periodic_cache[k] = fft(d, list(coeffs))
        return periodic_cache[k]

    out = []
    for i in points:
        memo = {}

        def ev(e):
            if id(e) in memo:
                return memo[id(e)]
            k = e.kind
            if k == "x":
                r = (domain_offset * pow(w, i, P)) % P
            elif k == "const":
                r = tuple(c % P for c in e.args[1]) if isinstance(e.args[1], tuple) else e.args[1] % P
            elif k == "challenge":
                r = challenges[e.args[0]]
            elif k == "hint":
                r = hints[e.args[0]]
            elif k == "trace":
                col, off = e.args
                j = (i + lde_step * off) % n
                r = base_cols[col][j] if col < len(base_cols) else ext_cols[col - len(base_cols)][j]
            elif k == "periodic":
                t = periodic(e.args[0], e.args[1])
                r = t[i % len(t)]
            elif k == "neg":
                a = ev(e.args[0])
                r = FQ3.neg(a) if _is_q(a) else (-a) % P
            elif k == "add":
                r = _add(ev(e.args[0]), ev(e.args[1]))
            elif k == "mul":
                r = _mul(ev(e.args[0]), ev(e.args[1]))
            elif k == "div":
                r = _mul(ev(e.args[0]), _inv(ev(e.args[1])))
            elif k == "pow":
                r = _pow(ev(e.args[0]), e.args[1])
            else:
                raise ValueError(k)
            memo[id(e)] = r
            return r

        # recursion depth: DAGs in the tests are shallow enough; raise the limit in the caller if needed
        v = ev(expr)
        out.append(_emb(v) if fq_is_ext else v)
    return out
