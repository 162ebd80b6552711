"""Fused constraint evaluation vs the oracle, bit-exact.  Cases follow the reference's
eval_gpu unit tests (src/eval_gpu.rs:917-1082): an X-only expression with pow and div; mixed
Fp/Fq3 columns with curr/next offsets; 1/X; the Fibonacci AIR (examples/fib/main.rs:73-140);
plus rotations with lde_step > 1 and wrap-around, periodic columns, challenges and hints."""
import sys

import numpy as np
import pytest

from oracle import cref
from oracle.pyref import evalexpr
from oracle.pyref.fields import GL
from tests import backends
from ministark_amd import GOLDILOCKS_FP as FP, GOLDILOCKS_FQ3 as FQ3, GpuVec
from ministark_amd import expr as E

P = cref.GL_P
KINDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
sys.setrecursionlimit(10000)


def _canon(arr, V=1):
    a = [GL.from_mont(int(x)) for x in arr]
    return a if V == 1 else [tuple(a[3 * i:3 * i + 3]) for i in range(len(a) // 3)]


def _check(kind, expr, log_n, lde_step, nbase, next_, nch=0, nh=0, fq_is_ext=True, offset=7, npoints=48, seed=1):
    pl = backends.planner(kind)
    n = 1 << log_n
    base = [cref.random_elements(n, seed + c) for c in range(nbase)]
    ext = [cref.random_elements(3 * n, seed + 50 + c) for c in range(next_)]
    qw = 3 if fq_is_ext else 1
    ch = cref.random_elements(max(nch, 1) * qw, seed + 90).reshape(-1, qw)
    hi = cref.random_elements(max(nh, 1) * qw, seed + 91).reshape(-1, qw)
    prog = E.compile_expr(expr, nbase, fq_is_ext)
    out = E.eval(prog, pl, ch, hi, lde_step, offset, n, [GpuVec.from_numpy(pl, c, FP) for c in base],
                 [GpuVec.from_numpy(pl, c, FQ3) for c in ext]).to_numpy()
    rng = np.random.default_rng(seed)
    pts = sorted(set([0, 1, n - 1, n - 2, n // 2] + [int(x) for x in rng.integers(0, n, size=npoints)]))
    qc = (lambda r: tuple(GL.from_mont(int(x)) for x in r)) if fq_is_ext else (lambda r: GL.from_mont(int(r[0])))
    want = evalexpr.eval_points(expr, pts, n, lde_step, offset, [_canon(c) for c in base], [_canon(c, 3) for c in ext],
                                [qc(r) for r in ch], [qc(r) for r in hi], fq_is_ext)
    V = 3 if fq_is_ext else 1
    for i, w in zip(pts, want):
        got = tuple(GL.from_mont(int(x)) for x in out[V * i:V * i + V])
        assert got == (w if fq_is_ext else (w,)), f"point {i}"
    return prog


@pytest.mark.parametrize("kind", KINDS)
def test_x_only_pow_div(kind):                     # src/eval_gpu.rs:917-950
    x = E.X()
    expr = (x ** 9 - 1) / (x - E.Constant(3)) + x * x ** 3 + 5
    _check(kind, expr, 12, 1, 0, 0)


@pytest.mark.parametrize("kind", KINDS)
def test_one_over_x(kind):                         # src/eval_gpu.rs:993-1020
    _check(kind, E.Constant(1) / E.X(), 12, 1, 0, 0)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("lde_step", [1, 4])
def test_mixed_fields_curr_next(kind, lde_step):   # src/eval_gpu.rs:953-990 scaled up
    x = E.X()
    b0, b1, b2 = (lambda o=0, c=c: E.Trace(c, o) for c in range(3))
    e0, e1 = (lambda o=0, c=c: E.Trace(3 + c, o) for c in range(2))
    expr = (b0(1) - b0() * b1() + e0(1) * e1() - e0() * b2(-1)) * (x - 1) / (x ** 8 - 1) \
        + E.Challenge(0) * e1(2) + E.Hint(1) * b1(1) + e0() ** 3 + (b2() + E.Challenge(1)) / (e1() - E.Hint(0))
    prog = _check(kind, expr, 12, lde_step, 3, 2, nch=2, nh=2)
    assert prog.max_p <= 16 and prog.max_q <= 8


@pytest.mark.parametrize("kind", KINDS)
def test_fibonacci_air(kind):                      # examples/fib/main.rs:73-140 shape: Fq = Fp, 8 columns
    x = E.X()
    c = [lambda o=0, k=k: E.Trace(k, o) for k in range(8)]
    cons = [c[0](1) - (c[6]() + c[7]()), c[1](1) - (c[7]() + c[0](1))]
    cons += [c[k]() - (c[k - 2]() + c[k - 1]()) for k in range(2, 8)]
    n_trace = 1 << 10
    g_inv = pow(GL.root_of_unity(n_trace), -1, P)
    zer = (x - E.Constant(g_inv)) / (x ** n_trace - 1)
    comp = None
    for k, cn in enumerate(cons):
        term = cn * zer * (E.Challenge(2 * k) * x ** 3 + E.Challenge(2 * k + 1))
        comp = term if comp is None else comp + term
    comp = comp + (c[0]() - 1) / (x - 1) * E.Challenge(16)
    _check(kind, comp, 12, 4, 8, 0, nch=17, fq_is_ext=False)


@pytest.mark.parametrize("kind", KINDS)
def test_periodic_column_and_const_fq(kind):
    x = E.X()
    per = E.Periodic([3, 1, 4, 1, 5, 9, 2, 6], 8)
    expr = per * E.Trace(0) + E.Constant((5, 6, 7)) * x - per ** 2 + E.Trace(1, 3)
    _check(kind, expr, 12, 2, 1, 1)


@pytest.mark.parametrize("kind", KINDS)
def test_zerofier_tables_and_x_power_lookup(kind):
    # the library hoists short-period sub-expressions (zerofier inverses, periodic columns times x^N) into
    # tables and turns long x^e chains into twiddle lookups (csrc/eval_opt.h): results must not change
    x = E.X()
    per = E.Periodic([2, 7, 1, 8], 4)
    zer_inv = 1 / ((x ** 256 - 1) * (x ** 512 - E.Constant(5)))
    expr = (E.Trace(0, 1) - E.Trace(0) * E.Trace(1)) * zer_inv * (E.Challenge(0) * x ** 1000003 + E.Challenge(1)) \
        + per * x ** 1024 * E.Trace(2) + (x ** 4096) ** 3 * E.Trace(1, -2) + x ** 12 + E.Challenge(1) * E.Challenge(0) * E.Challenge(1) * E.Trace(3)
    pl = backends.planner(kind)
    pl.profile(True)
    _check(kind, expr, 14, 4, 2, 2, nch=2)
    _check(kind, expr, 14, 4, 4, 0, nch=2, fq_is_ext=False, offset=3)
    prof = pl.profile_read()
    pl.profile(False)
    assert "eval_prologue" in prof and "eval_program" in prof


@pytest.mark.parametrize("kind,log_n", [pytest.param("emu", 10, id="emu"), pytest.param("hip", 12, id="hip", marks=pytest.mark.gpu),
                                        pytest.param("hip", 16, id="hip-specialised", marks=pytest.mark.gpu)])
def test_bit_reversed_storage(kind, log_n):
    # MS_EVAL_BIT_REVERSED: evaluate straight on the committed (bit-reversed) LDE layout, on the first n entries of
    # longer columns as bit_reverse_ce_trace does (src/prover.rs:185-194); must equal the natural-order evaluation,
    # bit-reversed.  The natural-order path is checked against the oracle by the other tests.
    pl = backends.planner(kind)
    n, N = 1 << log_n, 4 << log_n
    x = E.X()
    per = E.Periodic([3, 1, 4, 1, 5, 9, 2, 6], 8)
    b0, b1, b2 = (lambda o=0, c=c: E.Trace(c, o) for c in range(3))
    e0, e1 = (lambda o=0, c=c: E.Trace(3 + c, o) for c in range(2))
    expr = (b0(1) - b0() * b1(-1) + e0(1) * e1() - e0(2) * b2(-2)) * (x - 1) / (x ** (n // 4) - 1) + per * e1(1) + E.Challenge(0) * b1(3) \
        + x ** 12345 * b2() + E.Hint(0)
    prog = E.compile_expr(expr, 3, True)
    base = [cref.random_elements(N, 300 + c) for c in range(3)]
    ext = [cref.random_elements(3 * N, 350 + c) for c in range(2)]
    ch = cref.random_elements(3, 390).reshape(-1, 3)
    full = [GpuVec.from_numpy(pl, c, FP) for c in base], [GpuVec.from_numpy(pl, c, FQ3) for c in ext]
    got = E.eval(prog, pl, ch, ch, 4, 7, n, full[0], full[1], bit_reversed=True).to_numpy()
    nat_b = [GpuVec.from_numpy(pl, cref.bit_reverse(c[:n], log_n, 1), FP) for c in base]
    nat_e = [GpuVec.from_numpy(pl, cref.bit_reverse(c[:3 * n], log_n, 3), FQ3) for c in ext]
    want = E.eval(prog, pl, ch, ch, 4, 7, n, nat_b, nat_e).to_numpy()
    assert np.array_equal(got, cref.bit_reverse(want, log_n, 3))


def test_many_registers_emu():
    # a wide sum of products keeps many values alive -> larger register files
    terms = [E.Trace(k) * E.Trace(k + 1, 1) for k in range(0, 40, 2)]
    prods = [E.Trace(k) for k in range(40)]
    expr = terms[0]
    for t in terms[1:]:
        expr = expr + t
    keep = [p * p for p in prods]            # 40 squares kept alive until the end
    tail = keep[0]
    for kq in keep[1:]:
        tail = tail * kq + expr
    _check("emu", tail, 8, 1, 40, 0, npoints=6)


def test_invalid_programs_rejected_emu():
    import ctypes
    pl = backends.planner("emu")
    L = pl.lib
    out = GpuVec(pl, 16, FP)
    bad = np.array([[E.OP_ADD_PP, 0, 1, 2], [E.OP_STORE_P, 0, 0, 0]], dtype=np.uint32)   # reads unwritten registers
    rc = L.ms_eval_program(pl.handle, bad.ctypes.data, 2, None, 0, 4, 1, None, out.ptr, None, 0, None, 0, None, None, 0, FP, out.ptr)
    assert rc == -1 and b"invalid instruction" in L.ms_last_error()
    bad = np.array([[E.OP_TRACE_P, 0, 3, 0], [E.OP_STORE_P, 0, 0, 0]], dtype=np.uint32)    # column out of range
    rc = L.ms_eval_program(pl.handle, bad.ctypes.data, 2, None, 0, 4, 1, None, out.ptr, None, 0, None, 0, None, None, 0, FP, out.ptr)
    assert rc == -1


@pytest.mark.gpu
def test_eval_2_20_vs_sampled_oracle_hip():
    x = E.X()
    expr = (E.Trace(0, 1) - E.Trace(0) * E.Trace(1)) / (x ** 1024 - 1) + E.Trace(2) * E.Challenge(0) + E.Trace(3, -1) ** 5
    _check("hip", expr, 20, 8, 2, 2, nch=1, npoints=24)


def test_specialised_kernel_compiles_for_gfx950():
    # hiprtc needs no device: the straight-line kernel generated for a program (csrc/eval_jit.h) must
    # compile for gfx950 from the headers embedded in the library
    import ctypes
    from ministark_amd import STARK252_FP
    from ministark_amd._lib import Lib
    L = Lib()
    x = E.X()
    expr = (E.Trace(0, 1) - E.Trace(0) * E.Trace(1) + E.Trace(2) * E.Trace(3, -1)) / (x ** 64 - 1) + E.Challenge(0) * x ** 3 + E.Trace(2) ** 7
    for prog, field in ((E.compile_expr(expr, 2, True), FQ3), (E.compile_expr(expr, 4, False), FP), (E.compile_expr(expr, 4, False, STARK252_FP), STARK252_FP)):
        code = np.array(prog.instrs, dtype=np.uint32).reshape(-1, 4)
        size = ctypes.c_size_t(0)
        rc = L.ms_eval_jit_check(code.ctypes.data, len(code), field, ctypes.byref(size))
        assert rc == 0, L.ms_last_error().decode()[:2000]
        assert size.value > 1000


@pytest.mark.gpu
def test_specialised_kernels_2_16_hip():
    # domains of >= 2^16 points run the hiprtc-compiled straight-line kernel (csrc/eval_jit.h)
    pl = backends.planner("hip")
    x = E.X()
    b0, b1, b2 = (lambda o=0, c=c: E.Trace(c, o) for c in range(3))
    e0, e1 = (lambda o=0, c=c: E.Trace(3 + c, o) for c in range(2))
    mixed = (b0(1) - b0() * b1() + e0(1) * e1() - e0() * b2(-1)) * (x - 1) / (x ** 8 - 1) \
        + E.Challenge(0) * e1(2) + E.Hint(1) * b1(1) + e0() ** 3 + (b2() + E.Challenge(1)) / (e1() - E.Hint(0))
    per = E.Periodic([3, 1, 4, 1, 5, 9, 2, 6], 8)
    periodic = per * E.Trace(0) + E.Constant((5, 6, 7)) * x - per ** 2 + E.Trace(1, 3) + x ** 1000003 * E.Trace(0, -1) / (x ** 4096 - 1)
    pl.profile(True)
    _check("hip", mixed, 16, 4, 3, 2, nch=2, nh=2, npoints=24)
    _check("hip", periodic, 16, 2, 1, 1, npoints=24)
    _check("hip", (b0(1) - b0() * b1()) / (x ** 1024 - 1) + b2() ** 5 * E.Challenge(0), 16, 8, 3, 0, nch=1, fq_is_ext=False, npoints=24)
    prof = pl.profile_read()
    pl.profile(False)
    assert prof.get("eval_program_jit", {}).get("calls", 0) >= 3, prof.keys()


def _check252(kind, log_n, small_period=False):
    from oracle.pyref.fields import F252
    from ministark_amd import STARK252_FP, f252_from_mont_limbs, f252_to_mont_limbs
    pl = backends.planner(kind)
    lde_step, ncols = 2, 3
    n = 1 << log_n
    rng = np.random.default_rng(3)
    raw = rng.integers(0, 1 << 62, size=(ncols, n, 4), dtype=np.uint64)       # any 4-limb pattern < 2^254 is reduced below
    cols = [[(int(r[0]) | int(r[1]) << 64 | int(r[2]) << 128 | (int(r[3]) & ((1 << 58) - 1)) << 192) % F252.p for r in raw[c]] for c in range(ncols)]
    ch = [int.from_bytes(rng.bytes(32), "little") % F252.p for _ in range(2)]
    x = E.X()
    expr = (E.Trace(0, 1) - E.Trace(0) * E.Trace(1)) / (x ** 4 - 1) + E.Trace(2, -1) ** 3 * E.Challenge(1) + E.Constant(12345678901234567890123) / x + E.Challenge(0) \
        + E.Trace(1) / ((x ** 128 - 1) * (x ** 8 - 7)) + x ** 77777 * E.Trace(0)       # hoisted table (128 points: device prologue) + x^e lookup (csrc/eval_opt.h)
    if small_period:
        # zerofier of a blow-up-4 LDE: 4 distinct values, tabulated on the host (the same fp252.h functions)
        N = n // 4
        expr = (E.Trace(0, 1) - E.Trace(0) * E.Trace(1)) * (x - 5) / (x ** N - 1) * (E.Challenge(0) * x ** 3 + E.Challenge(1)) + E.Trace(2) / (x ** (2 * N) + 3)
    prog = E.compile_expr(expr, ncols, fq_is_ext=False, base_field=STARK252_FP)
    dev = [GpuVec.from_numpy(pl, np.concatenate([f252_to_mont_limbs(v) for v in c]), STARK252_FP) for c in cols]
    chm = np.stack([f252_to_mont_limbs(v) for v in ch])
    pl.profile(True)
    out = E.eval(prog, pl, chm, chm[:1], lde_step, 3, n, dev).to_numpy().reshape(n, 4)
    prof = pl.profile_read()
    pl.profile(False)
    pts = [0, 1, 5, n // 2, n - 1, 777]
    want = evalexpr.eval_points(expr, pts, n, lde_step, 3, cols, [], ch, ch[:1], False, F252)
    for i, w in zip(pts, want):
        assert f252_from_mont_limbs(out[i]) == w, f"point {i}"
    return prof


@pytest.mark.parametrize("kind", KINDS)
def test_fp252_program(kind):                      # src/eval_gpu.rs:1054-1082: constants / columns on Fp252
    prof = _check252(kind, 10)
    assert "eval_prologue" in prof and "eval_program252" in prof


@pytest.mark.parametrize("kind", KINDS)
def test_fp252_small_tables_from_the_host(kind):
    prof = _check252(kind, 10, small_period=True)
    assert "eval_prologue" not in prof and "eval_program252" in prof


@pytest.mark.gpu
def test_fp252_specialised_kernel_2_16_hip():
    assert "eval_program252_jit" in _check252("hip", 16)


@pytest.mark.parametrize("kind", ["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def test_x_only_denominators_are_batch_inverted(kind, monkeypatch):
    """eval_opt.h split_inversions: the (X - 1), (X - g^-1) divisions of the reference's fib AIR (examples/fib/main.rs:73-140) become
    full-length tables inverted with Montgomery's trick; results must equal the per-point evaluation of eval_cpu::eval -- also
    on the bit-reversed layout, over the 252-bit field, and with a denominator that is zero at a domain point (0^-1 = 0)."""
    from ministark_amd import STARK252_FP, pipeline
    pl = backends.planner(kind)
    log_n = 12 if kind == "emu" else 17
    n = 1 << log_n
    base = [cref.random_elements(n, 100 + k) for k in range(8)]
    for lde_step in (1, 4):
        comp, ce, nch = pipeline.fib_constraints(n // lde_step)
        ch = cref.random_elements(nch, 5).reshape(-1, 1)
        h = cref.random_elements(1, 6).reshape(-1, 1)
        prog = E.compile_expr(comp, 8, False)
        cols = [GpuVec.from_numpy(pl, c, FP) for c in base]
        want = cref.eval_expr(comp, log_n, lde_step, 7, base, [], ch, h, False)
        assert np.array_equal(E.eval(prog, pl, ch, h, lde_step, 7, n, cols).to_numpy(), want)
        br = [GpuVec.from_numpy(pl, cref.bit_reverse(c.copy(), log_n), FP) for c in base]
        got = E.eval(prog, pl, ch, h, lde_step, 7, n, br, bit_reversed=True).to_numpy()
        assert np.array_equal(got, cref.bit_reverse(want.copy(), log_n))
    # a denominator with a root ON the domain (offset 1: x_0 = 1 makes X - 1 vanish) and an Fq3 numerator
    x = E.X()
    expr = (E.Trace(0) * E.Challenge(0) + E.Trace(8)) / (x - E.Constant(1)) + E.Trace(1, 1) / (x * x - E.Constant(4))
    ext = [cref.random_elements(3 * n, 77)]
    chq = cref.random_elements(3, 8).reshape(-1, 3)
    prog = E.compile_expr(expr, 8, True)
    got = E.eval(prog, pl, chq, chq[:1], 1, 1, n, [GpuVec.from_numpy(pl, c, FP) for c in base], [GpuVec.from_numpy(pl, ext[0], FQ3)]).to_numpy()
    assert np.array_equal(got, cref.eval_expr(expr, log_n, 1, 1, base, ext, chq, chq[:1], True))
    # the 252-bit field
    comp, ce, nch = pipeline.fib_constraints(n // 4, 8, STARK252_FP)
    rng = np.random.default_rng(252)
    cols = [rng.integers(0, 1 << 63, size=4 * n, dtype=np.uint64) for _ in range(8)]
    for c in cols:
        c[3::4] >>= np.uint64(4)
    ch = rng.integers(0, 1 << 59, size=(nch, 4), dtype=np.uint64)
    prog = E.compile_expr(comp, 8, False, STARK252_FP)
    got = E.eval(prog, pl, ch, ch[:1], 4, 3, n, [GpuVec.from_numpy(pl, c, STARK252_FP) for c in cols]).to_numpy()
    want = cref.eval_expr(comp, log_n, 4, 3, cols, [], ch, ch[:1], False, field="f252")
    assert np.array_equal(got, want)
    if kind == "hip":
        # from 2^16 points the 252-bit tables take the two-level scheme; its middle level takes 32 products per Fermat inverse only on
        # domains of 2^23 points and more -- reached here through the knob ms_eval.cpp reads per call
        monkeypatch.setenv("MS_EVAL_INV_MIN_LANES", "1")
        got = E.eval(prog, pl, ch, ch[:1], 4, 3, n, [GpuVec.from_numpy(pl, c, STARK252_FP) for c in cols]).to_numpy()
        assert np.array_equal(got, want)


@pytest.mark.parametrize("field", ["goldilocks", "f252", "mixed"])
def test_sums_of_products_pass_on_the_reference_airs_emu(field):
    """csrc/eval_regroup.h on the shapes it was written for -- the reference's fib AIR over Goldilocks and over the 252-bit field, the
    17 Fp + 9 Fq3 composition -- on the simulator (the interpreter executes the accumulator opcodes): every output equals the C oracle's
    (the switch MS_EVAL_REGROUP is read once per process: on / off / forced are compared by tests/test_fuzz_gpu.py in their own processes)."""
    import numpy as np
    from oracle import cref
    from ministark_amd import GOLDILOCKS_FP, GOLDILOCKS_FQ3, STARK252_FP, GpuVec, pipeline
    pl = backends.planner("emu")
    log_n = 10
    n = 1 << log_n
    rng = np.random.default_rng(77)
    P = (1 << 64) - (1 << 32) + 1
    if field == "goldilocks":
        step, off = 1, 7
        comp, _, nch = pipeline.fib_constraints(n)
        base = [rng.integers(0, P, size=n, dtype=np.uint64) for _ in range(8)]
        ext, ch = [], rng.integers(1, P, size=(nch, 1), dtype=np.uint64)
        prog, bf, ext_flag, kw = E.compile_expr(comp, 8, False, GOLDILOCKS_FP), GOLDILOCKS_FP, False, {}
    elif field == "f252":
        step, off = 4, 3
        comp, _, nch = pipeline.fib_constraints(n // step, 8, STARK252_FP)
        base = [rng.integers(0, 1 << 63, size=4 * n, dtype=np.uint64) for _ in range(8)]
        for c in base:
            c[3::4] >>= np.uint64(4)
        ext, ch = [], rng.integers(0, 1 << 59, size=(nch, 4), dtype=np.uint64)
        prog, bf, ext_flag, kw = E.compile_expr(comp, 8, False, STARK252_FP), STARK252_FP, False, {"field": "f252"}
    else:
        step, off = 2, 7
        comp, nch = pipeline.mixed_air_constraints()
        base = [rng.integers(0, P, size=n, dtype=np.uint64) for _ in range(17)]
        ext = [rng.integers(0, P, size=3 * n, dtype=np.uint64) for _ in range(9)]
        ch = rng.integers(1, P, size=(nch, 3), dtype=np.uint64)
        prog, bf, ext_flag, kw = E.compile_expr(comp, 17, True, GOLDILOCKS_FP), GOLDILOCKS_FP, True, {}
    want = cref.eval_expr(comp, log_n, step, off, base, ext, ch, ch[:1], ext_flag, **kw)
    dbase = [GpuVec.from_numpy(pl, c, bf) for c in base]
    dext = [GpuVec.from_numpy(pl, c, GOLDILOCKS_FQ3) for c in ext]
    assert np.array_equal(E.eval(prog, pl, ch, ch[:1], step, off, n, dbase, dext).to_numpy(), want)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("field", ["goldilocks", "f252"])
def test_denominators_at_other_trace_rows_share_one_table(kind, field, capfd, monkeypatch):
    """csrc/eval_shift.h: boundary denominators X - g^r for several rows r are rotations of ONE inverse table (1 / (x - g^r) = g^(r0 - r) times
    the entry r0 - r trace rows further on).  Rows within reach of each other (16), one out of reach (its own table), a denominator that is not
    X - a (left alone), the domain WITHOUT an offset (x_i = g^r for some i: the zero denominators, 0^-1 = 0, must agree), natural and
    bit-reversed layouts, lde_step 1 / 4 -- the C oracle's per-point evaluation (eval_cpu::eval restated) is the reference."""
    from ministark_amd import STARK252_FP
    from ministark_amd.api import Radix2EvaluationDomain
    pl = backends.planner(kind)
    log_n = 12 if kind == "emu" else 16
    n = 1 << log_n
    f252 = field == "f252"
    bf = STARK252_FP if f252 else FP
    rng = np.random.default_rng(77)
    if f252:
        cols = [rng.integers(0, 1 << 63, size=4 * n, dtype=np.uint64) for _ in range(4)]
        for c in cols:
            c[3::4] >>= np.uint64(4)
        ch = rng.integers(0, 1 << 59, size=(8, 4), dtype=np.uint64)
        kw = {"field": "f252"}
    else:
        cols = [cref.random_elements(n, 300 + k) for k in range(4)]
        ch = cref.random_elements(8, 301).reshape(-1, 1)
        kw = {}
    monkeypatch.setenv("MS_EVAL_DEBUG", "1")
    for lde_step, offset in ((1, 7), (4, 3), (4, 1)):
        dom = Radix2EvaluationDomain(n // lde_step, 1, bf)
        g = dom.group_gen
        x = E.X()
        rows = [0, 1, -1, 5, -16, 40]                              # 40 is out of reach of every other row: a table of its own
        expr = None
        for j, r in enumerate(rows):
            a = pow(g, r % (n // lde_step), dom.p)
            t = (E.Trace(j % 4, 0) * E.Challenge(j) - E.Trace((j + 1) % 4, 1)) / (x - E.Constant(a))
            expr = t if expr is None else expr + t
        expr = expr + E.Trace(2, 0) / (x * x - E.Constant(9)) + E.Trace(3, 1) * E.Challenge(7) / (x - E.Constant(1))   # not X - a; X - 1 a second time
        prog = E.compile_expr(expr, 4, False, bf)
        want = cref.eval_expr(expr, log_n, lde_step, offset, cols, [], ch, ch[:1], False, **kw)
        capfd.readouterr()
        got = E.eval(prog, pl, ch, ch[:1], lde_step, offset, n, [GpuVec.from_numpy(pl, c, bf) for c in cols]).to_numpy()
        err = capfd.readouterr().err
        assert err.count("shared tables:") == 4, err[-2000:]          # rows 1, -1, 5, -16 ride on row 0's table (the second X - 1 IS the first: one node)
        assert np.array_equal(got, want)
        Vw = 4 if f252 else 1
        perm = np.array([int(format(i, f"0{log_n}b")[::-1], 2) for i in range(n)])
        brc = [np.ascontiguousarray(c.reshape(n, Vw)[perm].reshape(-1)) for c in cols]
        got = E.eval(prog, pl, ch, ch[:1], lde_step, offset, n, [GpuVec.from_numpy(pl, c, bf) for c in brc], bit_reversed=True).to_numpy()
        assert np.array_equal(got.reshape(n, Vw), want.reshape(n, Vw)[perm])


@pytest.mark.parametrize("kind", KINDS)
def test_denominator_built_on_another_inverse(kind):
    """a / (b / (x - c)) next to d / (x - c): the inverse of x - c is a hoisted table AND an operand of the second denominator.  The
    denominators' program used to run the inversion (in place: dst == a) before it stored x - c, and the table held (x - c)^-1 before the batch
    inversion -- found by the constraint fuzzer once it drew boundary-style divisors (round 5)."""
    x = E.X()
    a = E.Constant(628335062820441145)
    expr = ((x ** 2) / (E.Constant(2242740954004980262) / (x - a) / (-E.Challenge(1)))) ** 0 \
        + (E.Trace(1, 2) - (x / (x - a)) * ((x ** 64) * E.Constant(4605101940728796340)) + E.Trace(0, -2) * (E.Constant(2298312417385081279) + x))
    expr = expr + E.Trace(2, 1) / (E.Constant(3) / (x - a) + x)
    pl = backends.planner(kind)
    log_n = 12 if kind == "emu" else 16
    n = 1 << log_n
    base = [cref.random_elements(n, 10 + c) for c in range(3)]
    ch = cref.random_elements(2, 5).reshape(-1, 1)
    prog = E.compile_expr(expr, 3, False)
    out = E.eval(prog, pl, ch, ch[:1], 2, 7, n, [GpuVec.from_numpy(pl, c, FP) for c in base], []).to_numpy()
    assert np.array_equal(out, cref.eval_expr(expr, log_n, 2, 7, base, [], ch, ch[:1], False))


def test_shared_tables_in_a_program_with_extension_columns_emu(capfd, monkeypatch):
    """the same rotation of one inverse table inside a Q-typed program (Fq3 columns and challenges: the table and its scaling constant stay in
    the base field, the product with the numerator is an Fq3 x Fp one) -- against the C oracle, natural and bit-reversed layouts"""
    from ministark_amd.api import Radix2EvaluationDomain
    pl = backends.planner("emu")
    log_n, lde_step, offset = 12, 2, 7
    n = 1 << log_n
    g = Radix2EvaluationDomain(n // lde_step, 1, FP).group_gen
    x = E.X()
    expr = None
    for j, r in enumerate([0, 1, -1, 3]):
        a = pow(g, r % (n // lde_step), P)
        t = (E.Trace(2 + j % 2, 0) * E.Challenge(j % 2) - E.Trace(j % 2, 1)) / (x - E.Constant(a))
        expr = t if expr is None else expr + t * E.Trace(3, 1)
    base = [cref.random_elements(n, 400 + k) for k in range(2)]
    ext = [cref.random_elements(3 * n, 410 + k) for k in range(2)]
    ch = cref.random_elements(6, 420).reshape(-1, 3)
    prog = E.compile_expr(expr, 2, True)
    want = cref.eval_expr(expr, log_n, lde_step, offset, base, ext, ch, ch[:1], True)
    monkeypatch.setenv("MS_EVAL_DEBUG", "1")
    capfd.readouterr()
    got = E.eval(prog, pl, ch, ch[:1], lde_step, offset, n, [GpuVec.from_numpy(pl, c, FP) for c in base], [GpuVec.from_numpy(pl, c, FQ3) for c in ext]).to_numpy()
    assert capfd.readouterr().err.count("shared tables:") == 3
    assert np.array_equal(got, want)
    got = E.eval(prog, pl, ch, ch[:1], lde_step, offset, n, [GpuVec.from_numpy(pl, cref.bit_reverse(c.copy(), log_n, 1), FP) for c in base],
                 [GpuVec.from_numpy(pl, cref.bit_reverse(c.copy(), log_n, 3), FQ3) for c in ext], bit_reversed=True).to_numpy()
    assert np.array_equal(got, cref.bit_reverse(want.copy(), log_n, 3))


def test_plain_flag_gives_the_same_words_emu():
    """MS_EVAL_PLAIN (flags bit 1 of ms_eval_program_ex): the program exactly as given on the interpreter -- no tables, no shared inverses, no sums
    of products -- must give the words of the default path (which rewrites this program: a boundary-style and a periodic denominator, a sum of
    products); the library's MS_EVAL_SELFCHECK compares the same two evaluations internally."""
    import ctypes
    pl = backends.planner("emu")
    log_n, lde_step, n = 12, 2, 1 << 12
    x = E.X()
    expr = (E.Trace(0, 1) - E.Trace(0) * E.Trace(1)) / (x ** (n // lde_step) - 1) + (E.Trace(1) - E.Challenge(0)) / (x - 1) + E.Trace(0) * E.Trace(1, -1) * E.Challenge(0)
    base = [cref.random_elements(n, 640 + c) for c in range(2)]
    ch = cref.random_elements(1, 650).reshape(-1, 1)
    prog = E.compile_expr(expr, 2, False)
    cols = [GpuVec.from_numpy(pl, c, FP) for c in base]
    want = E.eval(prog, pl, ch, ch[:1], lde_step, 7, n, cols, []).to_numpy()
    assert np.array_equal(want, cref.eval_expr(expr, log_n, lde_step, 7, base, [], ch, ch[:1], False))
    consts = np.array(prog.consts, dtype=np.uint64)
    for idx, off in prog.challenge_slots.items():
        consts[off] = ch[idx][0]
    for idx, off in prog.hint_slots.items():
        consts[off] = ch[idx][0]
    code = np.array(prog.instrs, dtype=np.uint32).reshape(-1, 4)
    out = GpuVec(pl, n, FP)
    off = np.array([E.gl_to_mont(7)], dtype=np.uint64)
    VP = ctypes.c_void_p
    arr = (VP * 2)(*[c.ptr for c in cols])
    none = (VP * 1)()
    plen = (ctypes.c_uint * 1)()
    L = pl.lib
    for flags in (2, 3 & ~1):                                    # MS_EVAL_PLAIN
        L.check(L.ms_eval_program_ex(pl.handle, code.ctypes.data, len(code), consts.ctypes.data, consts.size, log_n, lde_step, off.ctypes.data, None,
                                     arr, 2, none, 0, none, plen, 0, FP, out.ptr, flags))
        pl.sync()
        assert np.array_equal(out.to_numpy(), want)
    assert L.ms_eval_program_ex(pl.handle, code.ctypes.data, len(code), consts.ctypes.data, consts.size, log_n, lde_step, off.ctypes.data, None,
                                arr, 2, none, 0, none, plen, 0, FP, out.ptr, 8) != 0      # unknown flag bits are refused
