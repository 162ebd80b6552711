"""ORACLE (test infrastructure) -- radix-2 evaluation domains, NTT / iNTT, LDE.

Restates the semantics the reference takes from ark-poly 0.4.2
`Radix2EvaluationDomain` (an un-vendored dependency, Cargo.lock) as they are
consumed at the reference's own call sites:

  * domain constants (size, group_gen, group_gen_inv, offset, offset_inv,
    size_inv) .......................... gpu/src/plan.rs:386-423
  * forward: c_i *= offset^i, then DFT with group_gen, natural order in/out
    ................................... gpu/src/plan.rs:254-263, src/matrix.rs:166-190
  * inverse: DFT with group_gen_inv, then e_i *= size_inv * offset_inv^i
    ................................... gpu/src/plan.rs:300-309,417-423, src/matrix.rs:119-139
  * LDE: interpolate on subgroup(n), evaluate on coset(n*blowup, g),
    bit-reverse rows ................... src/prover.rs:50-51, src/matrix.rs:225-251,352-354

`dft_naive` is the O(n^2) definition; `ntt` is an iterative radix-2
transform used for larger sizes and cross-checked against the definition in
tests/test_oracle_kat.py.  Elements are canonical ints (Fp / Fp252) or
3-tuples (Fq3); twiddles always live in the base FFT field.
"""
from .fields import bit_reverse_index


class Domain:
    """Radix2EvaluationDomain::new / new_coset."""

    def __init__(self, field, size, offset=1):
        assert size & (size - 1) == 0 and size >= 1
        self.F = field
        self.size = size
        self.log_size = size.bit_length() - 1
        self.group_gen = field.root_of_unity(size)
        self.group_gen_inv = field.inv(self.group_gen)
        self.size_inv = field.inv(size % field.p)
        self.offset = offset % field.p
        self.offset_inv = field.inv(self.offset)

    def element(self, i):
        return self.F.mul(self.offset, self.F.pow(self.group_gen, i))

    def elements(self):
        F = self.F
        x = self.offset
        out = []
        for _ in range(self.size):
            out.append(x)
            x = F.mul(x, self.group_gen)
        return out


class _ScalarOps:
    def __init__(self, F):
        self.F = F
        self.zero = 0

    def add(self, a, b):
        return (a + b) % self.F.p

    def sub(self, a, b):
        return (a - b) % self.F.p

    def scale(self, a, s):
        return (a * s) % self.F.p


class _Ext3Ops:
    def __init__(self, F):
        self.F = F
        self.zero = (0, 0, 0)

    def add(self, a, b):
        p = self.F.p
        return ((a[0] + b[0]) % p, (a[1] + b[1]) % p, (a[2] + b[2]) % p)

    def sub(self, a, b):
        p = self.F.p
        return ((a[0] - b[0]) % p, (a[1] - b[1]) % p, (a[2] - b[2]) % p)

    def scale(self, a, s):
        p = self.F.p
        return ((a[0] * s) % p, (a[1] * s) % p, (a[2] * s) % p)


def _ops(F, v):
    for x in v:
        return _Ext3Ops(F) if isinstance(x, tuple) else _ScalarOps(F)
    return _ScalarOps(F)


def dft_naive(F, v, root):
    """y[k] = sum_j v[j] * root^(j*k)   -- the definition."""
    n = len(v)
    ops = _ops(F, v)
    out = []
    for k in range(n):
        acc = ops.zero
        wk = pow(root, k, F.p)
        w = 1
        for j in range(n):
            acc = ops.add(acc, ops.scale(v[j], w))
            w = (w * wk) % F.p
        out.append(acc)
    return out


def ntt(F, v, root):
    """Iterative radix-2 DIT, natural order in and out."""
    n = len(v)
    if n == 1:
        return list(v)
    ops = _ops(F, v)
    a = [v[bit_reverse_index(n, i)] for i in range(n)]
    m = 2
    while m <= n:
        wm = pow(root, n // m, F.p)
        half = m // 2
        tw = [1] * half
        for i in range(1, half):
            tw[i] = (tw[i - 1] * wm) % F.p
        for s in range(0, n, m):
            for i in range(half):
                u = a[s + i]
                t = ops.scale(a[s + i + half], tw[i])
                a[s + i] = ops.add(u, t)
                a[s + i + half] = ops.sub(u, t)
        m *= 2
    return a


def fft(domain, coeffs):
    """Radix2EvaluationDomain::fft: zero-pad to size, coset-scale, transform."""
    F = domain.F
    assert len(coeffs) <= domain.size
    ops = _ops(F, coeffs)
    v = list(coeffs) + [ops.zero] * (domain.size - len(coeffs))
    if domain.offset != 1:
        g = 1
        for i in range(len(v)):
            v[i] = ops.scale(v[i], g)
            g = (g * domain.offset) % F.p
    return ntt(F, v, domain.group_gen)


def ifft(domain, evals):
    """Radix2EvaluationDomain::ifft."""
    F = domain.F
    assert len(evals) <= domain.size
    ops = _ops(F, evals)
    v = list(evals) + [ops.zero] * (domain.size - len(evals))
    v = ntt(F, v, domain.group_gen_inv)
    g = domain.size_inv
    for i in range(len(v)):
        v[i] = ops.scale(v[i], g)
        g = (g * domain.offset_inv) % F.p
    return v


def horner(F, coeffs, x):
    """Polynomial evaluation, used as an independent spot check of fft()."""
    ops = _ops(F, coeffs)
    acc = ops.zero
    for c in reversed(coeffs):
        acc = ops.add(ops.scale(acc, x), c)
    return acc


def lde_bit_reversed(F, column, blowup, offset):
    """src/prover.rs:50-51: interpolate(trace_domain) then
    bit_reversed_evaluate(lde_domain)."""
    n = len(column)
    trace_domain = Domain(F, n)
    lde_domain = Domain(F, n * blowup, offset)
    coeffs = ifft(trace_domain, column)
    evals = fft(lde_domain, coeffs)
    N = len(evals)
    return [evals[bit_reverse_index(N, i)] for i in range(N)]
