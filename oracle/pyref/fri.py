"""ORACLE (test infrastructure) -- FRI degree-respecting projection.

Literal restatement of `apply_drp` (src/fri.rs:526-567) as called from
`build_layer` (src/fri.rs:199-231, domain_offset = ONE):

    bit_reverse(evals); coeffs = ifft(evals, coset(n, offset));
    coeffs *= folding_factor;
    drp_coeffs[i] = sum_k coeffs[i*ff + k] * alpha^k;
    evals' = fft(drp_coeffs, coset(n/ff, offset^ff)); bit_reverse(evals')

Elements may be Fp ints (alpha an int) or Fq3 tuples (alpha a tuple).
"""
from .fields import bit_reverse
from .ntt import Domain, fft, ifft


def apply_drp(F, ext, evals, domain_offset, alpha, folding_factor):
    """F: base PrimeField; ext: None (elements are ints) or a CubicExt."""
    n = len(evals)
    domain = Domain(F, n, domain_offset)
    coeffs = ifft(domain, bit_reverse(list(evals)))
    if ext is None:
        mul = F.mul
        add = F.add
        one, zero = 1, 0
        scale = lambda a, s: F.mul(a, s)
    else:
        mul = ext.mul
        add = ext.add
        one, zero = ext.one(), ext.zero()
        scale = lambda a, s: ext.mul_base(a, s)
    ff = folding_factor
    coeffs = [scale(c, ff % F.p) for c in coeffs]
    alpha_pows = [one]
    for _ in range(1, ff):
        alpha_pows.append(mul(alpha_pows[-1], alpha))
    drp = []
    for i in range(n // ff):
        acc = zero
        for k in range(ff):
            acc = add(acc, mul(coeffs[i * ff + k], alpha_pows[k]))
        drp.append(acc)
    drp_domain = Domain(F, n // ff, F.pow(domain_offset, ff))
    return bit_reverse(fft(drp_domain, drp))
