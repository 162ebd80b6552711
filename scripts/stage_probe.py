#!/usr/bin/env python3
"""Streaming rate of the element-wise stages (a9) and the small kernels around the transforms at 2^24 elements: kernel time from the
library's per-launch events, algorithmic bytes = every operand once + the result once.  One JSON line per kernel."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ministark_amd import GOLDILOCKS_FP as FP, GOLDILOCKS_FQ3 as FQ3, GpuVec, Matrix, Planner  # noqa: E402
from ministark_amd import stages as S  # noqa: E402

P = (1 << 64) - (1 << 32) + 1
pl = Planner(0)
rng = np.random.default_rng(3)
n = 1 << 24


def timed(fn, reps=5):
    for _ in range(3):
        fn()
    pl.sync()
    pl.profile(True)
    for _ in range(reps):
        fn()
    pl.sync()
    prof = pl.profile_read()
    pl.profile(False)
    return sum(v["total_us"] for v in prof.values()) / reps, list(prof)


def emit(name, us, kernels, nbytes, note=""):
    print(json.dumps({"kernel": name, "launches": kernels, "us": round(us, 1), "algorithmic_bytes": nbytes, "GBps": round(nbytes / us / 1e3, 1),
                      "hbm_frac": round(nbytes / us / 1e3 / 8000.0, 3), "note": note}), flush=True)


for field, V, fname in ((FP, 1, "Fp"), (FQ3, 3, "Fq3")):
    a = GpuVec.from_numpy(pl, rng.integers(1, P, size=n * V, dtype=np.uint64), field)
    b = GpuVec.from_numpy(pl, rng.integers(1, P, size=n * V, dtype=np.uint64), field)
    c = GpuVec(pl, n, field)
    w = n * V * 8
    const = rng.integers(1, P, size=V, dtype=np.uint64)
    for name, fn, streams in ((f"MulAssignStage<{fname}>", lambda: S.MulAssignStage(pl, n, field, field).encode(a, b), 3),
                              (f"MulIntoStage<{fname}>", lambda: S.MulIntoStage(pl, n, field, field).encode(c, a, b), 3),
                              (f"AddAssignStage<{fname}>", lambda: S.AddAssignStage(pl, n, field, field).encode(a, b), 3),
                              (f"AddAssignStage<{fname}> shift 5", lambda: S.AddAssignStage(pl, n, field, field).encode(a, b, 5), 3),
                              (f"MulAssignConstStage<{fname}>", lambda: S.MulAssignConstStage(pl, n, field, field).encode(a, const), 2),
                              (f"NegInPlaceStage<{fname}>", lambda: S.NegInPlaceStage(pl, n, field).encode(a), 2),
                              (f"InverseIntoStage<{fname}>", lambda: S.InverseIntoStage(pl, n, field).encode(c, a), 2),
                              (f"ExpIntoStage<{fname}> ^7", lambda: S.ExpIntoStage(pl, n, field).encode(c, a, 7), 2)):
        us, k = timed(fn)
        emit(name, us, k, streams * w, "Montgomery batch inversion: 3 products per element + one Fermat inverse per 16 (Fp) / 8 (Fq3) elements" if "Inverse" in name else "")
    if V == 1:
        q = GpuVec(pl, n, FQ3)
        us, k = timed(lambda: S.ConvertIntoStage(pl, n, FQ3, FP).encode(q, a))
        emit("ConvertIntoStage<Fq3 <- Fp>", us, k, 4 * w)
        m = Matrix([a, b])
        us, k = timed(lambda: S.sum_columns(m))
        emit("sum_columns, 2 columns", us, k, 3 * w)
        us, k = timed(lambda: Matrix([a]).bit_reverse_rows())
        emit("bit_reverse_rows, 1 column", us, k, 2 * w)
        us, k = timed(lambda: Matrix.from_chunks(a, 4))
        emit("from_chunks (deinterleave) into 4 columns", us, k, 2 * w)
