"""C4 (iii) kernel times: the fib AIR over the 252-bit field on 2^23 points, lde_step 4 (bench.py's case), per kernel (hipEvents)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ministark_amd import STARK252_FP, GpuVec, Planner, expr as E, pipeline
pl = Planner(0)
n = 1 << 23
rng = np.random.default_rng(23)
comp, _, nch = pipeline.fib_constraints(n >> 2, 8, STARK252_FP)
cols = [rng.integers(0, 1 << 63, size=4 * n, dtype=np.uint64) for _ in range(8)]
for c in cols:
    c[3::4] >>= np.uint64(4)
base = [GpuVec.from_numpy(pl, c, STARK252_FP) for c in cols]
ch = rng.integers(0, 1 << 59, size=(nch, 4), dtype=np.uint64)
prog = E.compile_expr(comp, 8, False, STARK252_FP)
for _ in range(3):
    out = E.eval(prog, pl, ch, ch[:1], 4, 3, n, base, [])
pl.sync(); pl.profile(True)
for _ in range(5):
    out = E.eval(prog, pl, ch, ch[:1], 4, 3, n, base, [])
rec = pl.profile_read(); pl.profile(False)
k = {k: round(v["total_us"] / 5, 1) for k, v in rec.items()}
print("waves", os.environ.get("MS_EVAL_JIT_WAVES"), k, "total", round(sum(k.values()), 1))
