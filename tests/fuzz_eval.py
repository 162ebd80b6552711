#!/usr/bin/env python3
"""Random constraint expressions through compile_expr -> ms_eval_program (table hoisting, x^e lookups,
interpreter on small domains, hiprtc-specialised kernels on 2^16 points) against the oracle's direct
evaluation at sampled points:   python tests/fuzz_eval.py [seconds] [seed]
MS_FUZZ_BACKEND=emu: on the simulator build (small domains).  MS_FUZZ_FIELD=f252: programs over the 252-bit field (Fq = Fp; every output
against the C oracle).  MS_EVAL_REGROUP=force: the sums-of-products pass applied wherever it can be.  Divisors are random expressions or, about
as often, boundary-style X - g^r for a few trace rows r (the shared inverse tables of csrc/eval_shift.h)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.setrecursionlimit(10000)
from oracle import cref  # noqa: E402  (the checker)
from oracle.pyref import evalexpr  # noqa: E402
from oracle.pyref.fields import GL  # noqa: E402
from ministark_amd import GOLDILOCKS_FP as FP, GOLDILOCKS_FQ3 as FQ3, GpuVec, Planner  # noqa: E402
from ministark_amd.api import Radix2EvaluationDomain  # noqa: E402
from ministark_amd import expr as E  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
EMU = os.environ.get("MS_FUZZ_BACKEND") == "emu"          # the simulator build (CPU): the rewriting passes and the interpreter, small domains
if EMU:
    from tests import backends  # noqa: E402
    pl = backends.planner("emu")
else:
    pl = Planner(0)
P = GL.p
F252 = os.environ.get("MS_FUZZ_FIELD") == "f252"
if F252:
    from ministark_amd import STARK252_FP  # noqa: E402


BOUNDARY = []                                  # g^r for a few trace rows r of the case at hand


def rand_expr(depth, nbase, next_, nch, log_n):
    r = rng.random()
    if depth == 0 or r < 0.18:
        k = int(rng.integers(0, 6))
        if k == 0:
            return E.X()
        if k == 1:
            return E.Constant(int(rng.integers(0, 1 << 62)))
        if k == 2 and nch:
            return E.Challenge(int(rng.integers(0, nch)))
        if k == 3 and next_:
            return E.Trace(nbase + int(rng.integers(0, next_)), int(rng.integers(-2, 3)))
        if k == 4:
            return E.X() ** int(rng.choice([1, 2, 3, 8, 64, 1 << int(rng.integers(3, log_n + 2)), int(rng.integers(5, 1 << 20))]))
        return E.Trace(int(rng.integers(0, nbase)), int(rng.integers(-2, 3)))
    a = rand_expr(depth - 1, nbase, next_, nch, log_n)
    if r < 0.28:
        return -a
    if r < 0.36:
        return a ** int(rng.integers(0, 9))
    b = rand_expr(depth - 1, nbase, next_, nch, log_n)
    if r < 0.62:
        return a + b
    if r < 0.72:
        return a - b
    if r < 0.92:
        return a * b
    if BOUNDARY and rng.random() < 0.45:       # a boundary-style divisor X - g^r (csrc/eval_shift.h: rotations of one inverse table)
        return a / (E.X() - E.Constant(BOUNDARY[int(rng.integers(0, len(BOUNDARY)))]))
    return a / b                               # 0^-1 = 0 on both sides


def canon(arr, V):
    a = [GL.from_mont(int(x)) for x in arr]
    return a if V == 1 else [tuple(a[3 * i:3 * i + 3]) for i in range(len(a) // 3)]


t0, count, jit = time.time(), 0, 0
while time.time() - t0 < budget:
    log_n = int(rng.choice([6, 8, 9, 12] if EMU else [6, 9, 12, 12, 13, 16]))
    n = 1 << log_n
    fq_is_ext = bool(rng.integers(0, 2)) and not F252
    nbase, next_, nch = int(rng.integers(1, 4)), (int(rng.integers(0, 3)) if fq_is_ext else 0), int(rng.integers(0, 3))
    lde_step, offset = int(rng.choice([1, 2, 4, 8])), int(rng.choice([1, 3, 7]))
    dom = Radix2EvaluationDomain(max(n // lde_step, 1), 1, STARK252_FP if F252 else FP)
    BOUNDARY[:] = [pow(dom.group_gen, int(r) % max(n // lde_step, 1), dom.p) for r in rng.integers(-20, 21, size=4)]
    expr = rand_expr(int(rng.integers(2, 6)), nbase, next_, nch, log_n)
    if F252:                                    # 4-word elements below 2^251 < p; the C oracle checks every output
        def el(k, sd):
            r = np.random.default_rng(sd)
            a = r.integers(0, 1 << 63, size=4 * k, dtype=np.uint64)
            a[3::4] >>= np.uint64(4)
            return a
        base = [el(n, seed * 1000 + count * 7 + c) for c in range(nbase)]
        ch = el(max(nch, 1), count + 90).reshape(-1, 4)
        prog = E.compile_expr(expr, nbase, False, STARK252_FP)
        out = E.eval(prog, pl, ch, ch[:1], lde_step, offset, n, [GpuVec.from_numpy(pl, c, STARK252_FP) for c in base], []).to_numpy()
        want_all = cref.eval_expr(expr, log_n, lde_step, offset, base, [], ch, ch[:1], False, field="f252")
        if not np.array_equal(out, want_all):
            bad = np.nonzero(out != want_all)[0]
            print(f"MISMATCH (252-bit) case {count} (seed {seed}): log_n={log_n} lde_step={lde_step} offset={offset}: {bad.size} words differ, first at {bad[:4]}; {len(prog.instrs)} instructions")
            sys.exit(1)
        count += 1
        jit += log_n >= 16
        continue
    qw = 3 if fq_is_ext else 1
    base = [cref.random_elements(n, seed * 1000 + count * 7 + c) for c in range(nbase)]
    ext = [cref.random_elements(3 * n, seed * 1000 + count * 7 + 50 + c) for c in range(next_)]
    ch = cref.random_elements(max(nch, 1) * qw, count + 90).reshape(-1, qw)
    prog = E.compile_expr(expr, nbase, fq_is_ext)
    out = E.eval(prog, pl, ch, ch[:1], lde_step, offset, n, [GpuVec.from_numpy(pl, c, FP) for c in base],
                 [GpuVec.from_numpy(pl, c, FQ3) for c in ext]).to_numpy()
    if rng.random() < 0.3:                      # the committed-layout mode must give the same values, bit-reversed
        V3 = 3 if fq_is_ext else 1
        br = E.eval(prog, pl, ch, ch[:1], lde_step, offset, n, [GpuVec.from_numpy(pl, cref.bit_reverse(c, log_n, 1), FP) for c in base],
                    [GpuVec.from_numpy(pl, cref.bit_reverse(c, log_n, 3), FQ3) for c in ext], bit_reversed=True).to_numpy()
        if not np.array_equal(br, cref.bit_reverse(out, log_n, V3)):
            print(f"MISMATCH (bit-reversed layout) case {count} (seed {seed}): log_n={log_n} fq_is_ext={fq_is_ext} lde_step={lde_step} offset={offset}")
            sys.exit(1)
    # every output against the C restatement of eval_cpu::eval (chunks of 512, batch inversion) ...
    want_all = cref.eval_expr(expr, log_n, lde_step, offset, base, ext, ch, ch[:1], fq_is_ext)
    if not np.array_equal(out, want_all):
        bad = np.nonzero(out != want_all)[0]
        print(f"MISMATCH case {count} (seed {seed}): log_n={log_n} fq_is_ext={fq_is_ext} lde_step={lde_step} offset={offset}: {bad.size} words differ, first at {bad[:4]}; {len(prog.instrs)} instructions")
        sys.exit(1)
    # ... and a few points against the independent per-point big-integer evaluator
    pts = sorted(set([0, n - 1] + [int(x) for x in rng.integers(0, n, size=3)]))
    qc = (lambda r: tuple(GL.from_mont(int(x)) for x in r)) if fq_is_ext else (lambda r: GL.from_mont(int(r[0])))
    need = sorted({(i + lde_step * o) % n for i in pts for o in range(-2, 3)})
    cb = [{j: GL.from_mont(int(c[j])) for j in need} for c in base]
    ce = [{j: tuple(GL.from_mont(int(x)) for x in c[3 * j:3 * j + 3]) for j in need} for c in ext]
    want = evalexpr.eval_points(expr, pts, n, lde_step, offset, cb, ce, [qc(r) for r in ch], [qc(r) for r in ch[:1]], fq_is_ext)
    for i, w in zip(pts, want):
        got = tuple(GL.from_mont(int(x)) for x in out[qw * i:qw * i + qw])
        if got != (w if fq_is_ext else (w,)):
            print(f"MISMATCH (python oracle) case {count} (seed {seed}): log_n={log_n} fq_is_ext={fq_is_ext} lde_step={lde_step} offset={offset} point {i}")
            sys.exit(1)
    count += 1
    jit += log_n >= 16
st = pl.jit_stats()                                # what the library says it did, not what this script expects of it
print(f"fuzz_eval ok: {count} random programs ({jit} on domains of 2^16 points; specialised kernels: {st['kernels_compiled']} compiled, "
      f"{st['kernels_from_disk']} from the cache, {st['compile_failures']} failed) in {time.time() - t0:.0f} s (seed {seed})")
if st["compile_failures"]:
    print("FAILED: a generated kernel did not compile (the interpreter ran in its place)")
    sys.exit(1)
