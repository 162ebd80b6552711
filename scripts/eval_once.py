"""C4(i): the reference's fib AIR (8 Fp columns, Fq = Fp) evaluated on 2^23 points, a few times (for counter collection with scripts/sq_probe.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ministark_amd import GOLDILOCKS_FP, GpuVec, Planner, expr as E, pipeline

pl = Planner(0)
n = 1 << 23
P = (1 << 64) - (1 << 32) + 1
rng = np.random.default_rng(23)
if os.environ.get("FIELD") == "f252":                      # C4(iii): the same AIR over the 252-bit field, lde_step 4 (bench.py's case)
    from ministark_amd import STARK252_FP
    comp, _, nch = pipeline.fib_constraints(n >> 2, 8, STARK252_FP)
    cols = [rng.integers(0, 1 << 63, size=4 * n, dtype=np.uint64) for _ in range(8)]
    for c in cols:
        c[3::4] >>= np.uint64(4)
    base = [GpuVec.from_numpy(pl, c, STARK252_FP) for c in cols]
    ch = rng.integers(0, 1 << 59, size=(nch, 4), dtype=np.uint64)
    prog = E.compile_expr(comp, 8, False, STARK252_FP)
    step, off = 4, 3
else:
    comp, _, nch = pipeline.fib_constraints(n)
    base = [GpuVec.from_numpy(pl, rng.integers(0, P, size=n, dtype=np.uint64), GOLDILOCKS_FP) for _ in range(8)]
    ch = rng.integers(1, P, size=(nch, 1), dtype=np.uint64)
    prog = E.compile_expr(comp, 8, False, GOLDILOCKS_FP)
    step, off = 1, 7
for _ in range(int(os.environ.get("REPS", "4"))):
    out = E.eval(prog, pl, ch, ch[:1], step, off, n, base, [])
pl.sync()
