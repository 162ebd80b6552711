"""C4 (ii) kernel times: 17 Fp + 9 Fq3 columns (the brainfuck shape) on 2^23 points, lde_step 2 (bench.py's case), per kernel (hipEvents).
REPS=1 PROFILE=0: one evaluation only (for counter collection / MS_EVAL_DUMP)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ministark_amd import GOLDILOCKS_FP, GOLDILOCKS_FQ3, GpuVec, Planner, expr as E, pipeline
pl = Planner(0)
n = 1 << 23
P = (1 << 64) - (1 << 32) + 1
rng = np.random.default_rng(23)
comp, nch = pipeline.mixed_air_constraints()
base = [GpuVec.from_numpy(pl, rng.integers(0, P, size=n, dtype=np.uint64), GOLDILOCKS_FP) for _ in range(17)]
ext = [GpuVec.from_numpy(pl, rng.integers(0, P, size=3 * n, dtype=np.uint64), GOLDILOCKS_FQ3) for _ in range(9)]
ch = rng.integers(1, P, size=(nch, 3), dtype=np.uint64)
prog = E.compile_expr(comp, 17, True, GOLDILOCKS_FP)
reps = int(os.environ.get("REPS", "5"))
if os.environ.get("PROFILE", "1") == "0":
    for _ in range(reps):
        out = E.eval(prog, pl, ch, ch[:1], 2, 7, n, base, ext)
    pl.sync()
    sys.exit(0)
for _ in range(3):
    out = E.eval(prog, pl, ch, ch[:1], 2, 7, n, base, ext)
pl.sync(); pl.profile(True)
for _ in range(reps):
    out = E.eval(prog, pl, ch, ch[:1], 2, 7, n, base, ext)
rec = pl.profile_read(); pl.profile(False)
k = {k: round(v["total_us"] / reps, 1) for k, v in rec.items()}
print(k, "total", round(sum(k.values()), 1))
