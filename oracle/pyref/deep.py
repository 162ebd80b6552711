"""ORACLE (test infrastructure) -- literal restatement of the reference's DEEP composition:
horner_evaluate (src/utils.rs:124-133), divide_out_point_into / divide_out_points_into
(src/utils.rs:151-175), DeepPolyComposer::{get_ood_evals, into_deep_poly} (src/composer.rs:43-188),
with Python integers.  Fq elements are 3-tuples (or ints when Fq = Fp)."""
from .fields import FQ3, GL

P = GL.p


def _isq(v):
    return isinstance(v, tuple)


def _emb(v, q):
    return v if (_isq(v) or not q) else (v, 0, 0)


def _add(a, b):
    return FQ3.add(a, b) if _isq(a) else (a + b) % P


def _mul(a, b):
    return FQ3.mul(a, b) if _isq(a) else (a * b) % P


def _zero(q):
    return (0, 0, 0) if q else 0


def horner_evaluate(coeffs, point):
    q = _isq(point)
    acc = _zero(q)
    for c in reversed(coeffs):
        acc = _add(_mul(acc, point), _emb(c, q))
    return acc


def divide_out_point_into(coeffs, z, c):
    q = _isq(z)
    out = list(coeffs)
    rem = _zero(q)
    for i in range(len(out) - 1, -1, -1):
        tmp = _emb(out[i], q)
        out[i] = _mul(rem, c)
        rem = _add(_mul(rem, z), tmp)
    return out


def divide_out_points_into(coeffs, zs, cs):
    q = _isq(zs[0]) if zs else False
    out = [_emb(c, q) for c in coeffs]
    rems = [_zero(q)] * len(zs)
    for i in range(len(out) - 1, -1, -1):
        tmp = out[i]
        acc = _zero(q)
        for r, c in zip(rems, cs):
            acc = _add(acc, _mul(c, r))
        out[i] = acc
        rems = [_add(_mul(z, r), tmp) for r, z in zip(rems, zs)]
    return out


def point_for(z, g, g_inv, offset):
    gen = g if offset >= 0 else g_inv
    s = pow(gen, abs(offset), P)
    return tuple((c * s) % P for c in z) if _isq(z) else (z * s) % P


def qpow(z, e):
    return FQ3.pow(z, e) if _isq(z) else pow(z, e, P)


def get_ood_evals(z, g, g_inv, trace_arguments, base_polys, ext_polys, comp_polys):
    nbase = len(base_polys)
    execution = []
    for col, off in trace_arguments:
        x = point_for(z, g, g_inv, off)
        coeffs = base_polys[col] if col < nbase else ext_polys[col - nbase]
        execution.append(horner_evaluate(coeffs, x))
    z_n = qpow(z, len(comp_polys))
    return execution, [horner_evaluate(c, z_n) for c in comp_polys]


def into_deep_poly(z, g, g_inv, trace_arguments, base_polys, ext_polys, comp_polys, exec_alphas, comp_alphas, degree):
    q = _isq(z)
    n = len(base_polys[0])
    nbase = len(base_polys)
    z_n = qpow(z, len(comp_polys))
    quotients = [divide_out_point_into(c, z_n, a) for c, a in zip(comp_polys, comp_alphas)]
    for col_idx in range(nbase + len(ext_polys)):
        xs, als = [], []
        for (col, off), a in zip(trace_arguments, exec_alphas):
            if col == col_idx:
                xs.append(point_for(z, g, g_inv, off))
                als.append(a)
        coeffs = base_polys[col_idx] if col_idx < nbase else ext_polys[col_idx - nbase]
        quotients.append(divide_out_points_into(coeffs, xs, als) if xs else [_zero(q)] * n)
    combined = [_zero(q)] * n
    for col in quotients:
        combined = [_add(a, b) for a, b in zip(combined, col)]
    da, db = degree
    zero = _zero(q)
    if db == zero:
        return [_mul(c, da) for c in combined]
    out, last = [], zero
    for c in combined:
        out.append(_add(_mul(c, da), _mul(last, db)))
        last = c
    return out
