"""A few runs of pipeline.prove_phases at configs[4]'s size (for counter collection with scripts/sq_probe.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ministark_amd import GpuVec, Matrix, Planner, pipeline

pl = Planner(0)
log_rows, ncols = int(os.environ.get("LOG_ROWS", "22")), 8
n_t = 1 << log_rows
P = (1 << 64) - (1 << 32) + 1
trace = Matrix([GpuVec.from_numpy(pl, np.random.default_rng(c).integers(0, P, size=n_t, dtype=np.uint64)) for c in range(ncols)])
comp, ce, nch = pipeline.fib_constraints(n_t, ncols)
draws = pipeline.Draws(0xC5, ncols, nch, ce, 32, n_t * 4, pipeline.fri_num_layers(n_t * 4, 4, 8, 64))
for _ in range(int(os.environ.get("REPS", "3"))):
    pipeline.prove_phases(pl, trace, comp, draws, 4, 8, 64, 8, ce_blowup=ce)
pl.sync()
