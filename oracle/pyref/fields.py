"""ORACLE (test infrastructure, not product code) -- big-integer field arithmetic.

Pure-Python restatement of the three fields the reference's gpu-poly path
instantiates.  Nothing here shares code with the HIP kernels or the C oracle:
it only uses Python integers and pow(x, e, p), so it is an independent
second opinion for both.

Follows (reference file:line):
  * Goldilocks Fp, Montgomery R = 2^64 ........ gpu/src/metal/felt_u64.h.metal:9-178
  * Fq3 = Fp[x]/(x^3 - 2) ...................... gpu/src/metal/felt_u64.h.metal:183-279,
                                                 gpu/src/fields.rs:78-97
  * Fp252 (StarkWare prime), R = 2^256 ........ gpu/src/metal/felt_u256.h.metal:9-204,
                                                 gpu/src/fields.rs:239-264

Memory format (what crosses the C ABI): arkworks' in-memory representation,
i.e. the Montgomery residue a*R mod p as little-endian u64 limbs
(gpu/src/fields.rs:82 "BigInt([8589934590]) = 2").  Every function below works
on *canonical* integers in [0, p); `to_mont` / `from_mont` move between the two.

Parity status: the constants are pinned against the literals the reference
carries in-tree (tests/test_oracle_kat.py); FFT/LDE/FRI outputs have no stored
vectors in the reference ("parity unpinned" by golden files -- they are pinned
by exact-arithmetic definitions instead).
"""

# ----------------------------------------------------------------------------
# Goldilocks
# ----------------------------------------------------------------------------
GL_P = (1 << 64) - (1 << 32) + 1
GL_R = (1 << 64) % GL_P            # = 2^32 - 1 ("ONE" felt_u64.h.metal:118)
GL_R2 = (GL_R * GL_R) % GL_P       # felt_u64.h.metal:127
GL_RINV = pow(GL_R, -1, GL_P)
GL_GENERATOR = 7                   # ark-ff-optimized fp64 FpParams::GENERATOR
GL_TWO_ADICITY = 32
GL_TWO_ADIC_ROOT = pow(GL_GENERATOR, (GL_P - 1) >> GL_TWO_ADICITY, GL_P)

# ----------------------------------------------------------------------------
# StarkWare 252-bit prime
# ----------------------------------------------------------------------------
F252_P = (1 << 251) + 17 * (1 << 192) + 1
F252_R = (1 << 256) % F252_P
F252_R2 = (F252_R * F252_R) % F252_P
F252_RINV = pow(F252_R, -1, F252_P)
F252_GENERATOR = 3                 # gpu/src/fields.rs:241
F252_TWO_ADICITY = 192
F252_TWO_ADIC_ROOT = pow(F252_GENERATOR, (F252_P - 1) >> F252_TWO_ADICITY, F252_P)


class PrimeField:
    """A prime field with a power-of-two Montgomery radix."""

    def __init__(self, name, p, rbits, generator, two_adicity):
        self.name = name
        self.p = p
        self.rbits = rbits
        self.R = (1 << rbits) % p
        self.Rinv = pow(self.R, -1, p)
        self.generator = generator
        self.two_adicity = two_adicity
        self.two_adic_root = pow(generator, (p - 1) >> two_adicity, p)
        self.nlimbs = rbits // 64
        self.nbytes = rbits // 8

    # representation ---------------------------------------------------------
    def to_mont(self, a):
        return (a * self.R) % self.p

    def from_mont(self, m):
        return (m * self.Rinv) % self.p

    # arithmetic on canonical integers ---------------------------------------
    def add(self, a, b):
        return (a + b) % self.p

    def sub(self, a, b):
        return (a - b) % self.p

    def neg(self, a):
        return (-a) % self.p

    def mul(self, a, b):
        return (a * b) % self.p

    def inv(self, a):
        if a % self.p == 0:
            raise ZeroDivisionError("inverse of zero")
        return pow(a, -1, self.p)

    def pow(self, a, e):
        return pow(a, e, self.p)

    def root_of_unity(self, n):
        """arkworks FftField::get_root_of_unity(n): TWO_ADIC_ROOT^(2^(s - log n))."""
        assert n & (n - 1) == 0 and n >= 1
        logn = n.bit_length() - 1
        assert logn <= self.two_adicity
        return pow(self.two_adic_root, 1 << (self.two_adicity - logn), self.p)

    # canonical little-endian serialisation (ark-serialize uncompressed) ------
    def to_bytes(self, a):
        return int(a).to_bytes(self.nbytes, "little")


GL = PrimeField("goldilocks_fp", GL_P, 64, GL_GENERATOR, GL_TWO_ADICITY)
F252 = PrimeField("stark252_fp", F252_P, 256, F252_GENERATOR, F252_TWO_ADICITY)


class CubicExt:
    """Fq3 = Fp[x]/(x^3 - nonresidue) over Goldilocks; elements are 3-tuples
    (c0, c1, c2) of canonical ints.  felt_u64.h.metal:183-279."""

    def __init__(self, base, nonresidue):
        self.base = base
        self.nr = nonresidue
        self.name = "goldilocks_fq3"
        self.p = base.p
        self.nbytes = 3 * base.nbytes

    def zero(self):
        return (0, 0, 0)

    def one(self):
        return (1, 0, 0)

    def embed(self, a):
        return (a % self.p, 0, 0)

    def add(self, a, b):
        p = self.p
        return ((a[0] + b[0]) % p, (a[1] + b[1]) % p, (a[2] + b[2]) % p)

    def sub(self, a, b):
        p = self.p
        return ((a[0] - b[0]) % p, (a[1] - b[1]) % p, (a[2] - b[2]) % p)

    def neg(self, a):
        p = self.p
        return ((-a[0]) % p, (-a[1]) % p, (-a[2]) % p)

    def mul(self, a, b):
        # schoolbook; x^3 = nr
        p, nr = self.p, self.nr
        a0, a1, a2 = a
        b0, b1, b2 = b
        c0 = a0 * b0 + nr * (a1 * b2 + a2 * b1)
        c1 = a0 * b1 + a1 * b0 + nr * (a2 * b2)
        c2 = a0 * b2 + a1 * b1 + a2 * b0
        return (c0 % p, c1 % p, c2 % p)

    def mul_base(self, a, s):
        p = self.p
        return ((a[0] * s) % p, (a[1] * s) % p, (a[2] * s) % p)

    def pow(self, a, e):
        r = self.one()
        base = a
        while e > 0:
            if e & 1:
                r = self.mul(r, base)
            base = self.mul(base, base)
            e >>= 1
        return r

    def inv(self, a):
        # a^(p^3 - 2); slow but independent of any formula we could get wrong
        if a == (0, 0, 0):
            raise ZeroDivisionError("inverse of zero")
        return self.pow(a, self.p ** 3 - 2)

    def to_mont(self, a):
        return tuple(self.base.to_mont(c) for c in a)

    def from_mont(self, a):
        return tuple(self.base.from_mont(c) for c in a)

    def to_bytes(self, a):
        return b"".join(self.base.to_bytes(c) for c in a)


FQ3 = CubicExt(GL, 2)


def bit_reverse_index(n, i):
    """gpu/src/utils.rs:4-7."""
    logn = n.bit_length() - 1
    r = 0
    for _ in range(logn):
        r = (r << 1) | (i & 1)
        i >>= 1
    return r


def bit_reverse(v):
    """gpu/src/utils.rs:32-41 (out-of-place restatement)."""
    n = len(v)
    return [v[bit_reverse_index(n, i)] for i in range(n)]
