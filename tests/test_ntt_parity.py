"""NTT / iNTT / LDE / bit-reverse parity: HIP kernels (through the C ABI) vs the oracle.

Shapes follow the reference's own GPU tests (gpu/tests/shaders.rs:17-117): sizes 2048,
4096, 65536, subgroup and coset (offset = GENERATOR = 7), Fp and Fq3, forward and
inverse -- bit-exact.  The `emu` variants run the same kernels under the simulator in
tests/emu on CPU (kernel-logic check, not gpu); the `hip` variants are the parity tests
proper on an MI355X.
"""
import os

import numpy as np
import pytest

from oracle import cref
from tests import backends
from ministark_amd import (GOLDILOCKS_FP, GOLDILOCKS_FQ3, GpuFft, GpuIfft, GpuVec, Matrix,
                           Radix2EvaluationDomain)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BACKENDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


def _rand(n_words, seed):
    return cref.random_elements(n_words, seed)


def _run(kind, field, log_n, inverse, offset, ncols=1, seed=1):
    pl = backends.planner(kind)
    V = 3 if field == GOLDILOCKS_FQ3 else 1
    n = 1 << log_n
    dom = Radix2EvaluationDomain(n, offset)
    cols = [_rand(n * V, seed + c) for c in range(ncols)]
    vecs = [GpuVec.from_numpy(pl, c, field) for c in cols]
    plan = (GpuIfft if inverse else GpuFft)(dom, field, pl)
    for v in vecs:
        plan.encode(v)
    plan.execute()
    for c, v in zip(cols, vecs):
        want = cref.ntt(c, log_n, V, inverse, offset)
        got = v.to_numpy()
        bad = np.nonzero(want != got)[0]
        assert bad.size == 0, f"mismatch at {bad[:8]} of {bad.size} (log_n={log_n} inv={inverse} off={offset} V={V})"
    plan.close()


# --- emu: every code path at the smallest size that reaches it -----------------------
@pytest.mark.parametrize("log_n", [0, 1, 3, 8, 11])
@pytest.mark.parametrize("inverse,offset", [(False, 1), (False, 7), (True, 1), (True, 7)])
def test_small_sizes_emu(log_n, inverse, offset):
    _run("emu", GOLDILOCKS_FP, log_n, inverse, offset, ncols=2)


@pytest.mark.parametrize("log_n", [12, 13, 14, 15, 16, 17])
def test_multipass_forward_emu(log_n):
    _run("emu", GOLDILOCKS_FP, log_n, False, 1)


@pytest.mark.parametrize("log_n,inverse,offset", [(12, False, 7), (12, True, 1), (12, True, 7), (16, True, 7),
                                                   (17, False, 7), (18, True, 7), (20, False, 7),
                                                   # (256, R, 256) plans, R = 2..16: the register-resident middle pass
                                                   (17, True, 1), (19, False, 7), (19, True, 7), (19, False, 1), (20, True, 7), (20, True, 1)])
def test_multipass_variants_emu(log_n, inverse, offset):
    _run("emu", GOLDILOCKS_FP, log_n, inverse, offset)


def _fused_small(kind, log_n):
    """Fp columns of 2^11 .. 2^14 points take ONE launch (ntt_fused_tiny / ntt_fused_small: the (256, n / 256) plan with the column in LDS between
    its two passes, the column pointers from a table; 2^9 and 2^10 stay with ntt_small, measured faster there) -- forward and inverse, subgroup and coset, in place and out of place, more columns than
    the other kernels take per launch -- and the kernel that ran is that one; Fq3 columns of the same length keep their own route.  Every word
    against the oracle."""
    pl = backends.planner(kind)
    n = 1 << log_n
    for inverse, offset, ncols in ((False, 7, 3), (False, 1, 2), (True, 7, 2), (True, 1, 258 if log_n == 12 or kind == "hip" else 5)):
        dom = Radix2EvaluationDomain(n, offset)
        cols = [cref.random_elements(n, 900 + 7 * c + offset) for c in range(ncols)]
        vecs = [GpuVec.from_numpy(pl, c, GOLDILOCKS_FP) for c in cols]
        outs = [GpuVec(pl, n, GOLDILOCKS_FP) for _ in cols]
        plan = (GpuIfft if inverse else GpuFft)(dom, GOLDILOCKS_FP, pl)
        pl.profile(True)
        plan.enqueue_to(vecs, outs)                                # out of place: the sources stay
        plan.enqueue(vecs)                                          # in place
        names = set(pl.profile_read())
        pl.profile(False)
        plan.close()
        assert names == {"ntt_fused_small" if log_n >= 12 else "ntt_fused_tiny" if log_n == 11 else "ntt_small"}, names
        for c, v, o in zip(cols, vecs, outs):
            want = cref.ntt(c, log_n, 1, inverse, offset)
            assert np.array_equal(v.to_numpy(), want) and np.array_equal(o.to_numpy(), want), (inverse, offset)
    q = cref.random_elements(3 * n, 77)
    v = GpuVec.from_numpy(pl, q, GOLDILOCKS_FQ3)
    plan = GpuFft(Radix2EvaluationDomain(n, 7), GOLDILOCKS_FQ3, pl)
    pl.profile(True)
    plan.enqueue([v])
    names = set(pl.profile_read())
    pl.profile(False)
    plan.close()
    assert not ({"ntt_fused_small", "ntt_fused_tiny"} & names) and names, names
    assert np.array_equal(v.to_numpy(), cref.ntt(q, log_n, 3, False, 7))


@pytest.mark.parametrize("log_n", [9, 10, 11, 12, 13, 14])
def test_fused_small_transform_emu(log_n):
    _fused_small("emu", log_n)


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [9, 10, 11, 12, 13, 14])
def test_fused_small_transform_hip(log_n):
    _fused_small("hip", log_n)


def _inverse_2_18(kind):
    """The inverse transforms of 2^18-point Fp columns run through the two FORWARD kernels of the two-pass plan (the column read backwards, the
    forward tables of the subgroup, n^-1 h^-k on the natural-order output): coset and subgroup, in place and out of place, every word."""
    pl = backends.planner(kind)
    n = 1 << 18
    for offset in (7, 1):
        cols = [cref.random_elements(n, 31 + c + offset) for c in range(2)]
        vecs = [GpuVec.from_numpy(pl, c, GOLDILOCKS_FP) for c in cols]
        outs = [GpuVec(pl, n, GOLDILOCKS_FP) for _ in cols]
        plan = GpuIfft(Radix2EvaluationDomain(n, offset), GOLDILOCKS_FP, pl)
        pl.profile(True)
        plan.enqueue_to(vecs, outs)
        plan.enqueue(vecs)
        names = set(pl.profile_read())
        pl.profile(False)
        plan.close()
        assert names == {"lde2_pass_a", "lde2_pass_b"}, names
        for c, v, o in zip(cols, vecs, outs):
            want = cref.ntt(c, 18, 1, True, offset)
            assert np.array_equal(v.to_numpy(), want) and np.array_equal(o.to_numpy(), want), offset
        back = GpuFft(Radix2EvaluationDomain(n, offset), GOLDILOCKS_FP, pl)          # and forward again: the column it started from
        back.enqueue(vecs)
        back.close()
        assert all(np.array_equal(v.to_numpy(), c) for v, c in zip(vecs, cols))


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [9, 11, 12])
def test_more_columns_than_one_pointer_table_hip(log_n):
    """5 000 short columns in ONE enqueue: the launches that read their column pointers from a table (ntt_small, ntt_fused_tiny, ntt_fused_small)
    take 4 096 columns each -- the second launch starts where the first one ended.  Sampled columns against the oracle, the rest against them
    (every column holds the same values, so every output must equal the sampled one)."""
    pl = backends.planner("hip")
    n, ncols = 1 << log_n, 5000
    col = cref.random_elements(n, 4242 + log_n)
    vecs = [GpuVec.from_numpy(pl, col, GOLDILOCKS_FP) for _ in range(ncols)]
    plan = GpuFft(Radix2EvaluationDomain(n, 7), GOLDILOCKS_FP, pl)
    plan.enqueue(vecs)
    plan.close()
    want = cref.ntt(col, log_n, 1, False, 7)
    for c in (0, 1, 4095, 4096, 4097, ncols - 1):
        assert np.array_equal(vecs[c].to_numpy(), want), c
    for c in range(0, ncols, 97):                                  # a spread of the others
        assert np.array_equal(vecs[c].to_numpy(), want), c
    for v in vecs:
        v.free()


@pytest.mark.gpu
def test_pointer_tables_survive_the_staging_ring_wrapping_hip():
    """The pointer tables of the short-column launches live in the context's pinned staging ring (1 MiB), read in place by the kernels: 400
    forward + inverse round trips over 600 columns of 2^12 points (9.4 KiB of table per launch) wrap the ring several times, with the launches
    queued back to back.  A table overwritten while a queued launch still reads it would transform the wrong columns: every column must come
    back as it started."""
    pl = backends.planner("hip")
    n, ncols = 1 << 12, 600
    cols = [cref.random_elements(n, 9000 + c % 7) for c in range(ncols)]
    vecs = [GpuVec.from_numpy(pl, c, GOLDILOCKS_FP) for c in cols]
    dom = Radix2EvaluationDomain(n, 7)
    fwd, inv = GpuFft(dom, GOLDILOCKS_FP, pl), GpuIfft(dom, GOLDILOCKS_FP, pl)
    from ministark_amd import ColumnSet
    cs = ColumnSet(vecs)
    for _ in range(400):
        fwd.enqueue(cs)
        inv.enqueue(cs)
    fwd.close(); inv.close()
    for c in (0, 1, 255, 256, 511, 599):
        assert np.array_equal(vecs[c].to_numpy(), cols[c]), c
    fwd = GpuFft(dom, GOLDILOCKS_FP, pl)
    fwd.enqueue(cs)
    fwd.close()
    assert np.array_equal(vecs[300].to_numpy(), cref.ntt(cols[300], 12, 1, False, 7))
    for v in vecs:
        v.free()


def test_inverse_2_18_two_pass_emu():
    _inverse_2_18("emu")


@pytest.mark.gpu
def test_inverse_2_18_two_pass_hip():
    _inverse_2_18("hip")


# three-pass plans whose last radix is >= 64 (2^22 .. 2^24): pass 1's inter-pass factor comes from wave-uniform tables and
# pass 2 applies the per-lane remainder on its loads (ntt2_first_pass<.., UNI>, ntt2_mid_pass<.., LOADQ>)
@pytest.mark.parametrize("field,log_n,inverse,offset", [(GOLDILOCKS_FP, 22, False, 7), (GOLDILOCKS_FP, 22, True, 1),
                                                        (GOLDILOCKS_FP, 22, False, 1), (GOLDILOCKS_FP, 22, True, 7),
                                                        (GOLDILOCKS_FQ3, 22, False, 7),
                                                        # (256, 32, 256) and (256, 128, 256): ntt2_mid_pass_r with T2 = 2, 8
                                                        (GOLDILOCKS_FP, 21, False, 7), (GOLDILOCKS_FP, 21, True, 7), (GOLDILOCKS_FP, 23, True, 7),
                                                        (GOLDILOCKS_FP, 23, False, 1), (GOLDILOCKS_FQ3, 21, True, 1)])
def test_uniform_interpass_factor_emu(field, log_n, inverse, offset):
    _run("emu", field, log_n, inverse, offset)


@pytest.mark.parametrize("field,log_n,log_b,bit_reversed", [(GOLDILOCKS_FQ3, 18, 4, True), (GOLDILOCKS_FQ3, 19, 3, False),
                                                            (GOLDILOCKS_FP, 21, 1, True)])
def test_uniform_interpass_factor_pruned_variants_emu(field, log_n, log_b, bit_reversed):
    # 2^22-point LDE domains that do not take the two-pass coset form: Fq3 with one / two non-zero inputs per first
    # network (blow-up 16 / 8), Fp with blow-up 2 (memset + full transform + bit reversal)
    _lde("emu", field, log_n, log_b, ncols=1, bit_reversed=bit_reversed)


def test_inverse_coset_scale_in_limb_last_pass_emu():
    # last radix 256 (2^16, 2^24): n^-1 h^-pos as a per-lane constant on the loads and a wave-uniform row factor
    _run("emu", GOLDILOCKS_FP, 16, True, 7, ncols=2)
    _run("emu", GOLDILOCKS_FQ3, 16, True, 7, ncols=1)


def test_lde_bit_reversed_limb_last_pass_emu():
    # a 2^24-point LDE domain = (8, 8, 8): pruned uniform-factor pass 1, load-factor pass 2, ntt2_last_pass_bitrev
    _lde("emu", GOLDILOCKS_FP, 22, 2, ncols=1)


def test_column_group_on_two_streams_emu():
    # MS_NTT_STREAMS=2 splits >= 2 columns of >= 2^20 points over two streams, [0, n/2) and [n/2, n): an odd count
    # exercises both ranges.  The switch is read once per process, so this runs in a child interpreter.
    import subprocess, sys, os
    code = ("import sys; sys.path.insert(0, %r); import tests.test_ntt_parity as t; "
            "t._run('emu', t.GOLDILOCKS_FP, 20, False, 7, ncols=3)") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, MS_NTT_STREAMS="2"))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_large_sizes_random_hip(seed):
    # 2^22 .. 2^24: random field, direction, offset and column count through the uniform-factor / permuted-row passes
    rng = np.random.default_rng(1000 + seed)
    log_n = int(rng.integers(22, 25))
    field = GOLDILOCKS_FQ3 if (log_n < 24 and rng.integers(0, 3) == 0) else GOLDILOCKS_FP
    offset = int([1, 7, int(rng.integers(2, cref.GL_P, dtype=np.uint64))][int(rng.integers(0, 3))])
    _run("hip", field, log_n, bool(rng.integers(0, 2)), offset, ncols=int(rng.integers(1, 4)), seed=int(rng.integers(1, 1 << 30)))


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", range(17, 24))
@pytest.mark.parametrize("seed", range(2))
def test_mid_sizes_random_hip(log_n, seed):
    # every (256, R, 256) plan, R = 2 .. 128 (ntt2_small_mid_pass, ntt2_mid_pass_r): random field, direction, offset, column count
    rng = np.random.default_rng(3000 + 16 * log_n + seed)
    field = GOLDILOCKS_FQ3 if rng.integers(0, 3) == 0 else GOLDILOCKS_FP
    offset = int([1, 7, int(rng.integers(2, cref.GL_P, dtype=np.uint64))][int(rng.integers(0, 3))])
    _run("hip", field, log_n, bool(rng.integers(0, 2)), offset, ncols=int(rng.integers(1, 4)), seed=int(rng.integers(1, 1 << 30)))


@pytest.mark.gpu
@pytest.mark.parametrize("log_dom", range(17, 22))
def test_mid_lde_random_hip(log_dom):
    # LDE domains of 2^17 .. 2^21 points that do not take the two-pass LDE (Fq3 columns, or fewer than 2^17 rows): pruned first
    # networks + the fused bit-reversed store over the (256, R, 256) plans
    rng = np.random.default_rng(4000 + log_dom)
    log_b = int(rng.integers(1, 5))
    for field in (GOLDILOCKS_FP, GOLDILOCKS_FQ3):
        _lde("hip", field, log_dom - log_b, log_b, ncols=int(rng.integers(1, 3)), bit_reversed=bool(rng.integers(0, 2)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_large_lde_random_hip(seed):
    # LDE domains of 2^22 .. 2^24 points: random field, blow-up (pruned first networks), output order, column count
    rng = np.random.default_rng(2000 + seed)
    log_dom = int(rng.integers(22, 25))
    log_b = int(rng.integers(1, 5))
    field = GOLDILOCKS_FQ3 if (log_dom < 24 and rng.integers(0, 3) == 0) else GOLDILOCKS_FP
    _lde("hip", field, log_dom - log_b, log_b, ncols=int(rng.integers(1, 3)), bit_reversed=bool(rng.integers(0, 2)))


@pytest.mark.gpu
def test_column_group_on_two_streams_hip():
    import subprocess, sys, os
    code = ("import sys; sys.path.insert(0, %r); import tests.test_ntt_parity as t; "
            "t._run('hip', t.GOLDILOCKS_FP, 22, False, 7, ncols=3); t._run('hip', t.GOLDILOCKS_FP, 24, True, 1, ncols=2)") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, MS_NTT_STREAMS="2"))


@pytest.mark.parametrize("log_b", [2, 3])
def test_uniform_interpass_factor_lde_emu(log_b):
    # the LDE's forward transform on 2^22 points with the pruned first network (zero-padded input) and the fused bit reversal
    _lde("emu", GOLDILOCKS_FP, 22 - log_b, log_b, ncols=1)
    if log_b == 2:      # natural order: the permuted-row variant (pass 2 out of place) with the pruned network
        _lde("emu", GOLDILOCKS_FP, 22 - log_b, log_b, ncols=2, bit_reversed=False)


@pytest.mark.parametrize("log_n,inverse,offset", [(5, False, 7), (11, True, 7), (12, False, 1), (13, True, 7), (17, False, 7),
                                                   (18, True, 7), (19, False, 1), (20, False, 7)])
def test_fq3_emu(log_n, inverse, offset):
    _run("emu", GOLDILOCKS_FQ3, log_n, inverse, offset, ncols=2)


# --- hip: the reference's shapes and beyond --------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("log_n,offset", [(11, 1), (12, 1), (16, 1), (11, 7), (12, 7)])
def test_fft_with_64_bit_field(log_n, offset):          # gpu/tests/shaders.rs:17-40
    _run("hip", GOLDILOCKS_FP, log_n, False, offset)


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,offset", [(11, 1), (12, 1), (16, 1), (11, 7), (12, 7)])
def test_fft_with_extension_field(log_n, offset):       # gpu/tests/shaders.rs:43-66
    _run("hip", GOLDILOCKS_FQ3, log_n, False, offset)


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,offset", [(11, 1), (12, 1), (11, 7), (12, 7)])
def test_ifft(log_n, offset):                           # gpu/tests/shaders.rs:94-117
    _run("hip", GOLDILOCKS_FP, log_n, True, offset)


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", list(range(0, 25)))
def test_all_sizes_forward_coset_hip(log_n):
    _run("hip", GOLDILOCKS_FP, log_n, False, 7, ncols=3 if log_n < 20 else 1)


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [9, 12, 13, 17, 20, 22, 24])
@pytest.mark.parametrize("offset", [1, 7])
def test_inverse_sizes_hip(log_n, offset):
    _run("hip", GOLDILOCKS_FP, log_n, True, offset)


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,inverse,offset", [(13, False, 7), (17, True, 7), (20, False, 7), (22, True, 1), (22, False, 7), (23, False, 7), (23, True, 7)])
def test_fq3_sizes_hip(log_n, inverse, offset):
    _run("hip", GOLDILOCKS_FQ3, log_n, inverse, offset, ncols=2)


@pytest.mark.gpu
def test_many_columns_hip():
    _run("hip", GOLDILOCKS_FP, 16, False, 7, ncols=37)   # more than one launch group


def _adversarial(n, V, kind):
    P = cref.GL_P
    edge = np.array([0, 1, P - 1, P - 2, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, P - (1 << 32), (1 << 63), (1 << 63) - 1,
                     0xFFFFFFFE00000001, 0xFFFFFFFEFFFFFFFF, 0x00000000FFFFFFFE, 0xFFFFFFFF00000000], dtype=np.uint64)
    if kind == 0:
        return np.full(n * V, P - 1, dtype=np.uint64)
    if kind == 1:
        a = np.zeros(n * V, dtype=np.uint64); a[::2] = P - 1
        return a
    if kind == 2:
        return edge[np.arange(n * V) % edge.size]
    rng = np.random.default_rng(9)
    a = edge[rng.integers(0, edge.size, size=n * V)]
    return a


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("pattern", [0, 1, 2, 3])
def test_adversarial_values(kind, pattern):
    # carries / borrows / p-1 everywhere: the lazy ("weak") arithmetic must still land on canonical results
    pl = backends.planner(kind)
    for log_n, V, inverse, offset in ((13, 1, False, 7), (12, 3, True, 7), (16 if kind == "hip" else 12, 1, True, 1), (17 if kind == "hip" else 13, 1, False, 1)):
        n = 1 << log_n
        field = GOLDILOCKS_FQ3 if V == 3 else GOLDILOCKS_FP
        x = _adversarial(n, V, pattern)
        v = GpuVec.from_numpy(pl, x, field)
        plan = (GpuIfft if inverse else GpuFft)(Radix2EvaluationDomain(n, offset), field, pl)
        plan.encode(v); plan.execute()
        assert np.array_equal(v.to_numpy(), cref.ntt(x, log_n, V, inverse, offset)), (log_n, V, inverse, offset)
    c = _adversarial(1 << 12, 1, pattern)
    out = Matrix.from_numpy(pl, [c]).lde(8, 7, True).to_numpy()[0]
    assert np.array_equal(out, cref.lde(c, 12, 3, 1, 7, True))


@pytest.mark.gpu
def test_four_pass_sizes_hip():
    # 2^25 and beyond take four passes (256 x ...): checked against the oracle at 2^25, round trip at 2^27
    _run("hip", GOLDILOCKS_FP, 25, False, 7)
    _run("hip", GOLDILOCKS_FP, 25, True, 7)
    pl = backends.planner("hip")
    n = 1 << 27
    x = _rand(n, 123)
    v = GpuVec.from_numpy(pl, x)
    dom = Radix2EvaluationDomain(n, 7)
    f = GpuFft(dom, GOLDILOCKS_FP, pl); f.encode(v); f.execute()
    g = GpuIfft(dom, GOLDILOCKS_FP, pl); g.encode(v); g.execute()
    assert np.array_equal(v.to_numpy(), x)


# --- round trip / properties at BASELINE's full size ----------------------------------
@pytest.mark.gpu
def test_roundtrip_2_24_hip():
    pl = backends.planner("hip")
    n = 1 << 24
    x = _rand(n, 99)
    v = GpuVec.from_numpy(pl, x)
    dom = Radix2EvaluationDomain(n, 7)
    f = GpuFft(dom, GOLDILOCKS_FP, pl); f.encode(v); f.execute()
    y = v.to_numpy()
    assert not np.array_equal(x, y)
    g = GpuIfft(dom, GOLDILOCKS_FP, pl); g.encode(v); g.execute()
    assert np.array_equal(v.to_numpy(), x)


@pytest.mark.gpu
def test_linearity_and_evaluation_2_24_hip():
    # size-independent properties at BASELINE's full size: NTT(a + c*b) = NTT(a) + c*NTT(b), and the
    # transform is polynomial evaluation: y[k] = sum_j a_j (h w^k)^j checked at a few k by Horner on the host
    from ministark_amd import stages as S
    pl = backends.planner("hip")
    log_n, n = 24, 1 << 24
    a, b = _rand(n, 1), _rand(n, 2)
    c = _rand(1, 3)
    dom = Radix2EvaluationDomain(n, 7)
    A, B = GpuVec.from_numpy(pl, a), GpuVec.from_numpy(pl, b)
    Cmb = GpuVec(pl, n)
    S.MulIntoConstStage(pl, n, GOLDILOCKS_FP).encode(Cmb, B, c)
    S.AddAssignStage(pl, n, GOLDILOCKS_FP).encode(Cmb, A)            # a + c*b
    f = GpuFft(dom, GOLDILOCKS_FP, pl)
    for v in (A, B, Cmb):
        f.encode(v)
    f.execute()
    S.MulAssignConstStage(pl, n, GOLDILOCKS_FP).encode(B, c)
    S.AddAssignStage(pl, n, GOLDILOCKS_FP).encode(B, A)              # NTT(a) + c*NTT(b)
    assert np.array_equal(B.to_numpy(), Cmb.to_numpy())
    ya = A.to_numpy()
    P = cref.GL_P
    Rinv = pow((1 << 64) % P, -1, P)
    w = pow(1753635133440165772, 1 << (32 - log_n), P)
    L = cref.lib()
    for k in (0, 1, 12345, n - 1):
        x = (7 * pow(w, k, P)) % P
        xm = (x * ((1 << 64) % P)) % P
        acc = 0
        # Horner in Montgomery form through the C oracle's scalar ops (fast enough for 2^24 terms x 4 points)
        acc = int(_horner_mont(a, xm))
        assert acc == int(ya[k]), k


def _horner_mont(coeffs, x_mont):
    """sum_j coeffs[j] * x^j with Montgomery words, vectorised by splitting into 2^12 blocks."""
    P = cref.GL_P
    R = (1 << 64) % P
    Rinv = pow(R, -1, P)
    x = (int(x_mont) * Rinv) % P
    blk = 1 << 12
    xs = np.array([pow(x, j, P) for j in range(blk)], dtype=object)
    acc, xb, cur = 0, pow(x, blk, P), 1
    for s in range(0, len(coeffs), blk):
        part = int(np.dot(coeffs[s:s + blk].astype(object), xs) % P)
        acc = (acc + part * cur) % P
        cur = (cur * xb) % P
    return acc          # coefficients are Montgomery words: sum (c_j R) x^j = (P(x)) R


# --- bit reversal ------------------------------------------------------------------------
def _bitrev(kind, field, log_n, ncols=2):
    pl = backends.planner(kind)
    V = 3 if field == GOLDILOCKS_FQ3 else 1
    cols = [_rand((1 << log_n) * V, 5 + c) for c in range(ncols)]
    m = Matrix.from_numpy(pl, cols, field)
    m.bit_reverse_rows()
    for c, got in zip(cols, m.to_numpy()):
        assert np.array_equal(got, cref.bit_reverse(c, log_n, V))


@pytest.mark.parametrize("log_n", [1, 4, 9, 10, 11, 13])
@pytest.mark.parametrize("field", [GOLDILOCKS_FP, GOLDILOCKS_FQ3])
def test_bit_reverse_emu(log_n, field):
    _bitrev("emu", field, log_n)


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [0, 4, 10, 15, 20, 23])
@pytest.mark.parametrize("field", [GOLDILOCKS_FP, GOLDILOCKS_FQ3])
def test_bit_reverse_hip(log_n, field):
    _bitrev("hip", field, log_n)


# --- LDE -----------------------------------------------------------------------------------
def _lde(kind, field, log_n, log_b, ncols=2, bit_reversed=True):
    pl = backends.planner(kind)
    V = 3 if field == GOLDILOCKS_FQ3 else 1
    cols = [_rand((1 << log_n) * V, 50 + c) for c in range(ncols)]
    m = Matrix.from_numpy(pl, cols, field)
    out = m.lde(1 << log_b, 7, bit_reversed)
    for c, keep, got in zip(cols, m.to_numpy(), out.to_numpy()):
        assert np.array_equal(keep, c), "input column must be preserved"
        assert np.array_equal(got, cref.lde(c, log_n, log_b, V, 7, bit_reversed))


@pytest.mark.parametrize("log_n,log_b", [(4, 1), (8, 3), (9, 3), (10, 4), (12, 1), (13, 3), (12, 3), (14, 3), (10, 2), (11, 4), (12, 0)])
def test_lde_emu(log_n, log_b):
    _lde("emu", GOLDILOCKS_FP, log_n, log_b)


@pytest.mark.parametrize("field,log_n,log_b,bit_reversed", [(GOLDILOCKS_FP, 15, 2, True), (GOLDILOCKS_FP, 16, 3, True), (GOLDILOCKS_FP, 16, 4, False),
                                                            (GOLDILOCKS_FP, 16, 5, True), (GOLDILOCKS_FQ3, 17, 2, True), (GOLDILOCKS_FQ3, 16, 1, False)])
def test_lde_over_register_resident_middle_pass_emu(field, log_n, log_b, bit_reversed):
    # LDE domains of 2^17 .. 2^21 points outside the two-pass LDE: pruned first network, (256, R, 256) plan, fused bit reversal
    _lde("emu", field, log_n, log_b, ncols=1, bit_reversed=bit_reversed)


@pytest.mark.parametrize("offset", [1, 7])
@pytest.mark.parametrize("log_b", [2, 3, 4])
def test_lde_pruned_first_pass_emu(offset, log_b):
    # blow-up 4 / 8 / 16 take the pruned radix-16 network; offset 1 the non-coset instantiation
    pl = backends.planner("emu")
    c = _rand(1 << 10, 91)
    out = Matrix.from_numpy(pl, [c]).lde(1 << log_b, offset, False).to_numpy()[0]
    assert np.array_equal(out, cref.lde(c, 10, log_b, 1, offset, False))


def test_lde_fq3_emu():
    _lde("emu", GOLDILOCKS_FQ3, 10, 3)
    _lde("emu", GOLDILOCKS_FQ3, 11, 3)
    _lde("emu", GOLDILOCKS_FQ3, 12, 2)
    _lde("emu", GOLDILOCKS_FP, 10, 2, bit_reversed=False)


@pytest.mark.parametrize("log_n,log_b,bit_reversed", [(17, 1, True), (17, 2, False),
                                                      # T = 4, 8, 16: the uniform split of pass A's factor (per-lane remainder in pass B)
                                                      (18, 2, True), (19, 1, True), (20, 1, False),
                                                      # T = 32, 64 (round 4): rows of 8192 / 16384 words, radix T = 16 x T1 with a third exchange
                                                      (21, 1, True), (22, 1, True)])
def test_lde_two_pass_cosets_emu(log_n, log_b, bit_reversed):
    """lde2_kernels.h (columns of 2^17..2^22 rows: beta coset transforms in two passes each), smallest blow-ups."""
    _lde("emu", GOLDILOCKS_FP, log_n, log_b, ncols=2, bit_reversed=bit_reversed)


@pytest.mark.parametrize("log_n,log_b,bit_reversed", [(18, 1, True), (18, 2, False), (19, 1, False), (20, 1, True)])
def test_lde_two_pass_cosets_fq3_emu(log_n, log_b, bit_reversed):
    """Fq3 columns through the two-pass coset LDE (round 6: lde2_strided_pass<.., 3> reads one word plane of the interleaved coefficients,
    lde2_rows_pass<.., 3> brings the three planes of a row together in LDS and stores whole interleaved runs), rows of 512 .. 4096 elements
    (T = 4, 8, 16; T = 2 in the child-process test below), natural and bit-reversed order, the input column preserved."""
    _lde("emu", GOLDILOCKS_FQ3, log_n, log_b, ncols=2 if log_n < 19 else 1, bit_reversed=bit_reversed)


def test_evaluate_two_pass_cosets_fq3_emu():
    _evaluate("emu", GOLDILOCKS_FQ3, 18, 1, ncols=2)


@pytest.mark.parametrize("switch,log_n", [("0", 18), ("all", 17)])
def test_fq3_lde_route_switch_emu(switch, log_n):
    """MS_LDE2_FQ3 (read once per process: a child).  "0" sends Fq3 columns down the (256, R, 256) plan again -- the route of 2^21 / 2^22-row
    extension columns and of before / after timings; "all" takes the two-pass kernels at 2^17 rows too (T = 2: not the default there, the
    three-pass plan is faster).  The words are the same."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\nfrom tests.test_ntt_parity import _lde\nfrom ministark_amd import GOLDILOCKS_FQ3\n"
            "_lde('emu', GOLDILOCKS_FQ3, %d, 1, ncols=1)\nprint('ok')\n") % (ROOT, log_n)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, MS_LDE2_FQ3=switch), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,log_b", [(17, 2), (18, 3), (19, 1), (20, 3), (20, 1), (18, 5)])
def test_lde_two_pass_cosets_fq3_hip(log_n, log_b):
    """the Fq3 passes at every row length they take by default (T = 4, 8, 16; 2^17 rows: the three-pass plan), blow-ups 2 .. 32"""
    _lde("hip", GOLDILOCKS_FQ3, log_n, log_b, ncols=3)
    _lde("hip", GOLDILOCKS_FQ3, log_n, log_b, ncols=1, bit_reversed=False)
    if log_b == 1:
        _evaluate("hip", GOLDILOCKS_FQ3, log_n, log_b, ncols=2)


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,log_b", [(17, 2), (18, 3), (19, 1), (19, 4), (20, 2), (18, 5)])
def test_lde_two_pass_cosets_hip(log_n, log_b):
    """every row-length instantiation of lde2_rows_pass (T = 2, 4, 8, 16) and blow-ups 2..32"""
    _lde("hip", GOLDILOCKS_FP, log_n, log_b, ncols=3)
    _lde("hip", GOLDILOCKS_FP, log_n, log_b, ncols=1, bit_reversed=False)


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,log_b", [(21, 1), (21, 3), (22, 2), (22, 1)])
def test_two_pass_lde_of_long_rows_hip(log_n, log_b):
    """the two-pass coset LDE on rows of 8192 / 16384 words (lde2_rows_pass<32 / 64>): BASELINE configs[4]'s 2^22-row trace at blow-up 4
    (a 2^24-point domain), its neighbours, natural and bit-reversed order, and ms_evaluate's entry on the same kernels"""
    _lde("hip", GOLDILOCKS_FP, log_n, log_b, ncols=3)
    _lde("hip", GOLDILOCKS_FP, log_n, log_b, ncols=1, bit_reversed=False)
    if log_b == 2:
        _evaluate("hip", GOLDILOCKS_FP, log_n, log_b, ncols=2)


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,log_b", [(9, 4), (11, 3), (16, 3), (20, 3), (18, 0), (15, 2), (14, 4), (17, 1), (13, 3)])
def test_lde_hip(log_n, log_b):
    _lde("hip", GOLDILOCKS_FP, log_n, log_b, ncols=3)


@pytest.mark.gpu
def test_lde_fq3_hip():
    _lde("hip", GOLDILOCKS_FQ3, 14, 3)


# --- evaluation of short coefficient columns (Matrix::into_evaluations with its resize, src/matrix.rs:193-251)
def _evaluate(kind, field, log_n, log_b, offset=7, ncols=2):
    from ministark_amd import Radix2EvaluationDomain
    pl = backends.planner(kind)
    V = 3 if field == GOLDILOCKS_FQ3 else 1
    n, N = 1 << log_n, 1 << (log_n + log_b)
    cols = [_rand(n * V, 70 + c) for c in range(ncols)]
    want = []
    for c in cols:
        padded = np.zeros(N * V, dtype=np.uint64)
        padded[: n * V] = c
        want.append(cref.ntt(padded, log_n + log_b, V, False, offset))
    dom = Radix2EvaluationDomain(N, offset)
    m = Matrix.from_numpy(pl, cols, field)
    nat = m.evaluate(dom)
    for c, keep, got, w in zip(cols, m.to_numpy(), nat.to_numpy(), want):
        assert np.array_equal(keep, c), "evaluate() must not touch its input"
        assert np.array_equal(got, w)
    br = m.bit_reversed_evaluate(dom)
    for got, w in zip(br.to_numpy(), want):
        assert np.array_equal(got, cref.bit_reverse(w, log_n + log_b, V))


@pytest.mark.parametrize("log_n,log_b", [(3, 1), (8, 2), (10, 2), (10, 3), (9, 4), (8, 5), (12, 0), (11, 1)])
def test_evaluate_short_columns_emu(log_n, log_b):
    _evaluate("emu", GOLDILOCKS_FP, log_n, log_b)


def test_evaluate_short_columns_fq3_and_subgroup_emu():
    _evaluate("emu", GOLDILOCKS_FQ3, 10, 2)
    _evaluate("emu", GOLDILOCKS_FP, 10, 3, offset=1)


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,log_b", [(18, 2), (16, 3), (14, 4), (15, 1), (13, 6)])
def test_evaluate_short_columns_hip(log_n, log_b):
    _evaluate("hip", GOLDILOCKS_FP, log_n, log_b, ncols=3)
    if log_b == 2:
        _evaluate("hip", GOLDILOCKS_FQ3, 14, 2)


@pytest.mark.parametrize("kind", [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)])
def test_composition_poly_chunks(kind):                 # src/prover.rs:113-121
    pl = backends.planner(kind)
    from ministark_amd import STARK252_FP
    for field, V in ((GOLDILOCKS_FP, 1), (GOLDILOCKS_FQ3, 3), (STARK252_FP, 4)):
        n, k = 1 << 12, 4
        poly = _rand(n * V, 5)
        cols = Matrix.from_chunks(GpuVec.from_numpy(pl, poly, field), k).to_numpy()
        elems = poly.reshape(n, V)
        for c in range(k):
            assert np.array_equal(cols[c].reshape(-1, V), elems[c::k])


# --- the largest domains: known-answer transforms generated and sampled on the device ---------------
def _big_known_answer(log_n):
    """NTT of the polynomial c0 + c1*X + c2*X^(n-1) on coset(n, 7) is c0 + c1*x_i + c2*x_i^(n-1): checked at
    sampled positions (8-byte downloads), then the inverse must restore the three coefficients and zeros."""
    import ctypes
    from oracle.pyref.fields import GL
    from ministark_amd import stages as S, gl_to_mont
    pl = backends.planner("hip")
    L = pl.lib
    n = 1 << log_n
    v = GpuVec(pl, n, GOLDILOCKS_FP)
    S.FillBuffStage(pl, n, GOLDILOCKS_FP).encode(v, np.zeros(1, dtype=np.uint64))
    c0, c1, c2 = 0x1234567, 0xabcdef0123, 0x77777777777
    for pos, c in ((0, c0), (1, c1), (n - 1, c2)):
        w = ctypes.c_uint64(gl_to_mont(c))
        L.check(L.ms_upload(pl.handle, v.ptr + 8 * pos, ctypes.byref(w), 8))
    dom = Radix2EvaluationDomain(n, 7)
    f = GpuFft(dom, GOLDILOCKS_FP, pl); f.encode(v); f.execute(); f.close()
    rng = np.random.default_rng(log_n)
    wn = GL.root_of_unity(n)
    for i in [0, 1, n - 1, n // 2, n // 2 + 1] + [int(x) for x in rng.integers(0, n, size=40)]:
        out = ctypes.c_uint64(0)
        L.check(L.ms_download(pl.handle, ctypes.byref(out), v.ptr + 8 * i, 8))
        x = 7 * pow(wn, i, GL.p) % GL.p
        assert GL.from_mont(out.value) == (c0 + c1 * x + c2 * pow(x, n - 1, GL.p)) % GL.p, f"position {i}"
    g = GpuIfft(dom, GOLDILOCKS_FP, pl); g.encode(v); g.execute(); g.close()
    for pos, c in ((0, c0), (1, c1), (n - 1, c2), (2, 0), (n // 2, 0), (n - 2, 0), (12345678 % n, 0)):
        out = ctypes.c_uint64(1)
        L.check(L.ms_download(pl.handle, ctypes.byref(out), v.ptr + 8 * pos, 8))
        assert GL.from_mont(out.value) == c, f"coefficient {pos}"


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [30, 32])
def test_known_answer_largest_domains_hip(log_n):
    # 2^32 is the field's whole two-adic subgroup (32 GiB column + 32 GiB of scratch): the maximum size there is
    _big_known_answer(log_n)


@pytest.mark.parametrize("kind", ["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def test_plan_outlives_its_context_safely(kind):
    """A GpuFft collected / used after its Planner was closed: the library releases the caller's plan objects
    together with the context; destroying one afterwards is a no-op and using one is an error, not a read of freed memory."""
    import ctypes
    from ministark_amd import Planner
    lib = backends.planner(kind).lib
    pl = Planner(0, lib)
    dom = Radix2EvaluationDomain(1 << 13, 7)
    f = GpuFft(dom, GOLDILOCKS_FP, pl)
    raw = ctypes.c_void_p(f.handle.value)
    col = GpuVec.from_numpy(pl, cref.random_elements(1 << 13, 5))
    f.encode(col); f.execute()
    handle = pl.handle
    f2 = GpuIfft(dom, GOLDILOCKS_FP, pl)
    raw2 = ctypes.c_void_p(f2.handle.value)
    pl._plans.clear()                                     # as if the Python side had lost track: the C side must cope alone
    col.free()
    lib.check(lib.ms_ctx_destroy(handle)); pl.handle = None
    assert lib.ms_ntt_execute(raw2) != 0                  # an error, reported
    assert lib.ms_ntt_plan_destroy(raw) == 0 and lib.ms_ntt_plan_destroy(raw2) == 0      # no-ops
    f.handle = None; f2.handle = None


# A batch larger than one launch holds (msntt::MAXC = 128 columns per launch): the groups must tile the batch exactly.
def test_batch_wider_than_a_launch_emu():
    _run("emu", GOLDILOCKS_FP, 12, False, 7, ncols=259)                      # a launch takes 256 columns (msntt::MAXC)


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,inverse,ncols", [(10, False, 520), (12, False, 259), (14, True, 257), (16, False, 258)])
def test_batch_wider_than_a_launch_hip(log_n, inverse, ncols):
    _run("hip", GOLDILOCKS_FP, log_n, inverse, 7, ncols=ncols)


@pytest.mark.gpu
def test_two_pass_2_18_handles_do_not_leak_hip():
    """Handles of the 2^18 two-pass forward route (a fresh GpuFft per Matrix.evaluate) use the CACHED plan's tables: creating and
    destroying many of them neither rebuilds nor leaks device memory (round-4 advisor finding, ms_ntt.cpp plan_run)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    free_b, total_b = ctypes.c_size_t(0), ctypes.c_size_t(0)

    def free_bytes():
        assert hip.hipMemGetInfo(ctypes.byref(free_b), ctypes.byref(total_b)) == 0
        return free_b.value
    pl = backends.planner("hip")
    log_n = 18
    x = _rand(1 << log_n, 5)
    want = cref.ntt(x, log_n, 1, False, 7)
    dom = Radix2EvaluationDomain(1 << log_n, 7)

    def once():
        v = GpuVec.from_numpy(pl, x, GOLDILOCKS_FP)
        plan = GpuFft(dom, GOLDILOCKS_FP, pl)
        plan.encode(v)
        plan.execute()
        got = v.to_numpy()
        plan.close()
        v.free()
        return got
    assert np.array_equal(once(), want)
    pl.sync()
    before = free_bytes()
    for _ in range(300):
        got = once()
    pl.sync()
    assert np.array_equal(got, want)
    assert before - free_bytes() < (8 << 20), f"device memory shrank by {before - free_bytes()} bytes over 300 handles"


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("field,log_n,inverse", [(GOLDILOCKS_FP, 8, False), (GOLDILOCKS_FP, 13, True), (GOLDILOCKS_FQ3, 12, True), (GOLDILOCKS_FP, 17, True),
                                                   (GOLDILOCKS_FP, 18, False), (GOLDILOCKS_FQ3, 17, False), (GOLDILOCKS_FP, 20, True)])
def test_out_of_place_transform_leaves_its_source_alone(kind, field, log_n, inverse):
    """ms_ntt_enqueue_to (Matrix::interpolate / evaluate = `self.clone().into_...`, src/matrix.rs:155-163, 237-243, without the device copy):
    dst = transform(src) equals the oracle, src keeps every word, and dst may be src (then it is the in-place transform).  Every plan kind:
    one workgroup in LDS, two passes, three passes, the two-pass 2^18 forward plan, Fq3."""
    pl = backends.planner(kind)
    V = 3 if field == GOLDILOCKS_FQ3 else 1
    n = 1 << log_n
    dom = Radix2EvaluationDomain(n, 7)
    cols = [_rand(n * V, 300 + c) for c in range(3)]
    src = [GpuVec.from_numpy(pl, c, field) for c in cols]
    dst = [GpuVec(pl, n, field) for _ in cols]
    plan = (GpuIfft if inverse else GpuFft)(dom, field, pl)
    plan.enqueue_to(src, dst)
    pl.sync()
    for c, s, d in zip(cols, src, dst):
        assert np.array_equal(s.to_numpy(), c), "the source column must be preserved"
        assert np.array_equal(d.to_numpy(), cref.ntt(c, log_n, V, inverse, 7))
    plan.enqueue_to(src[:1], src[:1])                          # aliasing: the in-place transform
    pl.sync()
    assert np.array_equal(src[0].to_numpy(), cref.ntt(cols[0], log_n, V, inverse, 7))
    plan.close()
    m = Matrix.from_numpy(pl, cols[1:], field)
    polys = m.interpolate(dom)                                  # the mirror's user of it
    assert all(np.array_equal(k, c) for k, c in zip(m.to_numpy(), cols[1:]))
    assert all(np.array_equal(g, cref.ntt(c, log_n, V, True, 7)) for g, c in zip(polys.to_numpy(), cols[1:]))
