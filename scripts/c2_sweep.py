#!/usr/bin/env python3
"""C2 sweep at the column counts a prover has (2^20 x 32, 2^22 x 16, 2^24 x 8): forward coset NTT per column, by column-group
size (MS_NTT_GROUP_BYTES is read at context creation: one process per setting).  One JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ministark_amd import GOLDILOCKS_FP as FP, ColumnSet, GpuFft, GpuVec, Planner, Radix2EvaluationDomain  # noqa: E402

P = (1 << 64) - (1 << 32) + 1
pl = Planner(0)
rng = np.random.default_rng(1)
out = {"group_bytes": os.environ.get("MS_NTT_GROUP_BYTES", "default"), "streams": os.environ.get("MS_NTT_STREAMS", "1")}
SHAPES = ((17, 64), (18, 64), (19, 64), (20, 32), (21, 32), (22, 16), (23, 16), (24, 8)) if "--all" in sys.argv else ((20, 32), (22, 16), (24, 8))
if "--small" in sys.argv:          # the two-pass sizes below the (256, R, 256) plans
    SHAPES = ((9, 512), (10, 512), (11, 512), (12, 512), (13, 512), (14, 256), (15, 256), (16, 128))
INVERSE = "--inverse" in sys.argv
if INVERSE:
    from ministark_amd import GpuIfft as GpuFft  # noqa: E402,F811
    out["direction"] = "inverse"
for log_n, ncol in SHAPES:
    n = 1 << log_n
    cols = ColumnSet([GpuVec.from_numpy(pl, rng.integers(0, P, size=n, dtype=np.uint64), FP) for _ in range(ncol)])   # the pointer table built once: the clock sees launches, not list marshalling
    plan = GpuFft(Radix2EvaluationDomain(n, 7), FP, pl)
    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end:
        plan.enqueue(cols); pl.sync()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.enqueue(cols)
    pl.sync()
    us = (time.perf_counter() - t0) / reps / ncol * 1e6
    out[f"2^{log_n}x{ncol}"] = {"us_per_column": round(us, 2), "hbm_frac": round(2.0 * n * 8 / (us * 1e-6) / 8e12, 4)}
    del cols, plan
print(json.dumps(out))
