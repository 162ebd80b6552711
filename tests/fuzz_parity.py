#!/usr/bin/env python3
"""Randomised differential run of the library against the C oracle (oracle/c) on the GPU:
    python tests/fuzz_parity.py [seconds] [seed]
Random sizes (2^0 .. 2^21: every plan family incl. the (256, R, 256) ones), fields, directions, coset offsets, blow-ups, folding factors, shifts and column
counts; values are a mix of uniform elements and edge values (0, 1, p-1, 2^32-1, 2^32, p-2^32 ...).
Complements tests/ (fixed shapes): any mismatch prints the failing case and exits non-zero.  MS_FUZZ_BACKEND=emu: the same on the simulator build
(CPU; transforms to 2^18, LDEs to 2^14 rows)."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import cref  # noqa: E402  (the checker)
from ministark_amd import (GOLDILOCKS_FP as FP, GOLDILOCKS_FQ3 as FQ3, GpuFft, GpuIfft, GpuVec, Matrix, MerkleTree, Planner,  # noqa: E402
                           Radix2EvaluationDomain, apply_drp, gl_to_mont)
from ministark_amd import stages as S  # noqa: E402

P = cref.GL_P
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
EMU = __import__("os").environ.get("MS_FUZZ_BACKEND") == "emu"      # the simulator build (CPU): the same kernels, smaller upper sizes
if EMU:
    from tests import backends  # noqa: E402
    pl = backends.planner("emu")
else:
    pl = Planner(0)
TOP_NTT, TOP_LDE = (19, 15) if EMU else (22, 18)
EDGE = np.array([gl_to_mont(v % P) for v in (0, 1, 2, P - 1, P - 2, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, P - (1 << 32), (1 << 63), 7)], dtype=np.uint64)
RAW_EDGE = np.array([0, 1, P - 1, P - 2, 0xFFFFFFFF, 1 << 32, 0xFFFFFFFF00000000], dtype=np.uint64)   # canonical Montgomery words


def values(n_words):
    a = rng.integers(0, P, size=n_words, dtype=np.uint64)
    mode = rng.integers(0, 4)
    if mode == 1:
        m = rng.random(n_words) < 0.3
        a[m] = rng.choice(np.concatenate([EDGE, RAW_EDGE]), size=int(m.sum()))
    elif mode == 2:
        a[:] = rng.choice(np.concatenate([EDGE, RAW_EDGE]), size=n_words)
    return a


def offset():
    return int(rng.choice([1, 7, 3, P - 1, int(rng.integers(2, 1 << 62))]))


def case_ntt():
    log_n, V, inv, off = int(rng.integers(0, TOP_NTT)), int(rng.choice([1, 3])), bool(rng.integers(0, 2)), offset()
    field = FQ3 if V == 3 else FP
    x = values((1 << log_n) * V)
    v = GpuVec.from_numpy(pl, x, field)
    f = (GpuIfft if inv else GpuFft)(Radix2EvaluationDomain(1 << log_n, off), field, pl)
    f.encode(v); f.execute(); f.close()
    return np.array_equal(v.to_numpy(), cref.ntt(x, log_n, V, inv, off)), f"ntt log_n={log_n} V={V} inv={inv} off={off}"


def case_lde():
    log_n, log_b, V, off, br = int(rng.integers(0, TOP_LDE)), int(rng.integers(0, 6)), int(rng.choice([1, 3])), offset(), bool(rng.integers(0, 2))
    field = FQ3 if V == 3 else FP
    cols = [values((1 << log_n) * V) for _ in range(int(rng.integers(1, 4)))]
    out = Matrix.from_numpy(pl, cols, field).lde(1 << log_b, off, br).to_numpy()
    ok = all(np.array_equal(o, cref.lde(c, log_n, log_b, V, off, br)) for o, c in zip(out, cols))
    return ok, f"lde log_n={log_n} log_b={log_b} V={V} off={off} bit_reversed={br}"


def case_evaluate():
    log_n, log_b, off, br = int(rng.integers(0, 14)), int(rng.integers(0, 6)), offset(), bool(rng.integers(0, 2))
    n, N = 1 << log_n, 1 << (log_n + log_b)
    c = values(n)
    padded = np.zeros(N, dtype=np.uint64); padded[:n] = c
    want = cref.ntt(padded, log_n + log_b, 1, False, off)
    if br:
        want = cref.bit_reverse(want, log_n + log_b, 1)
    m = Matrix.from_numpy(pl, [c], FP)
    got = (m.bit_reversed_evaluate if br else m.evaluate)(Radix2EvaluationDomain(N, off)).to_numpy()[0]
    return np.array_equal(got, want), f"evaluate log_n={log_n} log_b={log_b} off={off} bit_reversed={br}"


def case_fri():
    ff = int(rng.choice([2, 4, 8, 16]))
    log_n, V, off = int(rng.integers(ff.bit_length() - 1, 17)), int(rng.choice([1, 3])), offset()
    field = FQ3 if V == 3 else FP
    ev, alpha = values((1 << log_n) * V), values(V)
    got = apply_drp(GpuVec.from_numpy(pl, ev, field), alpha, ff, off).to_numpy()
    return np.array_equal(got, cref.fri_fold(ev, log_n, V, ff, alpha, off)), f"fri log_n={log_n} V={V} ff={ff} off={off}"


def case_commit():
    log_n, ncols, V = int(rng.integers(1, 13)), int(rng.integers(1, 40)), int(rng.choice([1, 3]))
    field = FQ3 if V == 3 else FP
    cols = [values((1 << log_n) * V) for _ in range(ncols)]
    tree = MerkleTree.from_matrix(Matrix.from_numpy(pl, cols, field))
    want = cref.sha256_merkle(cref.sha256_rows(cols, V))
    return tree.root() == want[1].tobytes() and np.array_equal(tree.nodes_numpy()[1:], want[1:]), f"commit log_n={log_n} ncols={ncols} V={V}"


def case_stage():
    log_n = int(rng.integers(0, 15))
    n = 1 << log_n
    lf, rf = [(FP, FP), (FQ3, FQ3), (FQ3, FP)][int(rng.integers(0, 3))]
    VL, VR = (3 if lf == FQ3 else 1), (3 if rf == FQ3 else 1)
    a, b, shift, e = values(n * VL), values(n * VR), int(rng.integers(-2 * n, 2 * n + 1)), int(rng.integers(0, 40))
    A, B, D = GpuVec.from_numpy(pl, a, lf), GpuVec.from_numpy(pl, b, rf), GpuVec(pl, n, lf)
    S.MulIntoStage(pl, n, lf, rf).encode(D, A, B, shift)
    ok = np.array_equal(D.to_numpy(), cref.binary(1, VL, VR, a, b, shift))
    S.AddIntoStage(pl, n, lf, rf).encode(D, A, B, shift)
    ok &= np.array_equal(D.to_numpy(), cref.binary(0, VL, VR, a, b, shift))
    S.MulPowStage(pl, n, lf, rf).encode(A, B, e, shift)
    ok &= np.array_equal(A.to_numpy(), cref.mul_pow(VL, VR, a, b, e, shift))
    S.InverseIntoStage(pl, n, lf).encode(D, A)
    ok &= np.array_equal(D.to_numpy(), cref.unary(1, VL, A.to_numpy(), 0))
    return bool(ok), f"stage log_n={log_n} fields=({VL},{VR}) shift={shift} e={e}"


CASES = [case_ntt, case_lde, case_evaluate, case_fri, case_commit, case_stage]
t0, count = time.time(), 0
while time.time() - t0 < budget:
    fn = CASES[int(rng.integers(0, len(CASES)))]
    ok, what = fn()
    count += 1
    if not ok:
        print(f"MISMATCH after {count} cases (seed {seed}): {what}")
        sys.exit(1)
print(f"fuzz ok: {count} random cases in {time.time() - t0:.0f} s (seed {seed})")
