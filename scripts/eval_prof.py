import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ministark_amd import GOLDILOCKS_FP, GpuVec, Planner, expr as E, pipeline
pl = Planner(0)
n = 1 << 23
P = (1 << 64) - (1 << 32) + 1
rng = np.random.default_rng(23)
comp, _, nch = pipeline.fib_constraints(n)
base = [GpuVec.from_numpy(pl, rng.integers(0, P, size=n, dtype=np.uint64), GOLDILOCKS_FP) for _ in range(8)]
ch = rng.integers(1, P, size=(nch, 1), dtype=np.uint64)
prog = E.compile_expr(comp, 8, False, GOLDILOCKS_FP)
for _ in range(3):
    out = E.eval(prog, pl, ch, ch[:1], 1, 7, n, base, [])
pl.sync(); pl.profile(True)
for _ in range(5):
    out = E.eval(prog, pl, ch, ch[:1], 1, 7, n, base, [])
rec = pl.profile_read(); pl.profile(False)
print("jit us", {k: round(v["avg_us"], 1) for k, v in rec.items()})
