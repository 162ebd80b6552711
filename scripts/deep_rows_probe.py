"""Kernel time of the DEEP composition on the committed LDE rows (ms_deep_rows) at configs[4]'s shape: 2^24 rows, 8 base columns + 1 composition column."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ministark_amd import GpuVec, Matrix, Planner, pipeline

pl = Planner(0)
log_rows, ncols = 22, 8
n_t = 1 << log_rows
P = (1 << 64) - (1 << 32) + 1
trace = Matrix([GpuVec.from_numpy(pl, np.random.default_rng(c).integers(0, P, size=n_t, dtype=np.uint64)) for c in range(ncols)])
comp, ce, nch = pipeline.fib_constraints(n_t, ncols)
draws = pipeline.Draws(0xC5, ncols, nch, ce, 32, n_t * 4, pipeline.fri_num_layers(n_t * 4, 4, 8, 64))
for _ in range(2):
    pipeline.prove_phases(pl, trace, comp, draws, 4, 8, 64, 8, ce_blowup=ce)
pl.sync(); pl.profile(True)
for _ in range(5):
    pipeline.prove_phases(pl, trace, comp, draws, 4, 8, 64, 8, ce_blowup=ce)
rec = pl.profile_read(); pl.profile(False)
print("deep_rows us:", round(rec["deep_rows"]["avg_us"], 1), "PTS", os.environ.get("MS_AB_DEEP_PTS", "4"))
