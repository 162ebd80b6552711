#!/usr/bin/env python3
"""Where a fresh process's first proof goes: every primitive of the prover chain at configs[4]'s sizes, first call against second call
(plans and their tables, lazily loaded kernel code, pool growth).  GPU box: python scripts/cold_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.perf_counter()
from ministark_amd import GOLDILOCKS_FP, GpuFft, GpuIfft, GpuVec, Matrix, MerkleTree, Planner, Radix2EvaluationDomain  # noqa: E402
t1 = time.perf_counter()
pl = Planner(0)
pl.sync()
t2 = time.perf_counter()
print(f"import {1e3 * (t1 - t0):.1f} ms, context {1e3 * (t2 - t1):.1f} ms")
P = (1 << 64) - (1 << 32) + 1
rng = np.random.default_rng(1)
n = 1 << 22
host = [rng.integers(0, P, size=n, dtype=np.uint64) for _ in range(8)]


def timed(name, fn, reps=3):
    ts = []
    for _ in range(reps):
        a = time.perf_counter()
        r = fn()
        pl.sync()
        ts.append((time.perf_counter() - a) * 1e3)
    print(f"{name:40s} first {ts[0]:8.2f} ms   then {min(ts[1:]):8.2f} ms   cold excess {ts[0] - min(ts[1:]):8.2f} ms")
    return r


trace = timed("upload 8 x 2^22", lambda: Matrix.from_numpy(pl, host, GOLDILOCKS_FP))
dom = Radix2EvaluationDomain(n, 1)
timed("GpuIfft(2^22) plan + run 8 columns", lambda: GpuIfft(dom, GOLDILOCKS_FP, pl).enqueue([c.clone() for c in trace.columns]))
lde = timed("lde 2^22 x 8, blow-up 4 (bit-reversed)", lambda: trace.lde(4, 7, True))
tree = timed("MerkleTree.from_matrix (2^24 rows)", lambda: MerkleTree.from_matrix(lde))
timed("root download", lambda: tree.root())
dom24 = Radix2EvaluationDomain(1 << 24, 7)
col = lde.columns[0].clone()
timed("GpuFft(2^24 coset) plan + run 1 column", lambda: GpuFft(dom24, GOLDILOCKS_FP, pl).enqueue([col]))
