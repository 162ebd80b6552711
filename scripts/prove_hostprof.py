"""Host-side profile of pipeline.prove_phases on one GPU at configs[4]'s size: where the wall time beyond the kernels goes."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ministark_amd import GpuVec, Matrix, Planner, pipeline

pl = Planner(0)
log_rows, ncols = int(os.environ.get("LOG_ROWS", "22")), 8
n_t = 1 << log_rows
P = (1 << 64) - (1 << 32) + 1
cols = [GpuVec.from_numpy(pl, np.random.default_rng(c).integers(0, P, size=n_t, dtype=np.uint64)) for c in range(ncols)]
comp, ce, nch = pipeline.fib_constraints(n_t, ncols)
draws = pipeline.Draws(0xC5, ncols, nch, ce, 32, n_t * 4, pipeline.fri_num_layers(n_t * 4, 4, 8, 64))
trace = Matrix(cols)
run = lambda: pipeline.prove_phases(pl, trace, comp, draws, 4, 8, 64, 8, ce_blowup=ce)
for _ in range(2):
    run()
pl.sync()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); out = run(); pl.sync(); ts.append((time.perf_counter() - t0) * 1e3)
print("wall ms:", [round(t, 2) for t in ts], out["phases_ms"], out.get("openings_ms"))
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    run()
pl.sync(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45); print(s.getvalue()[:9000])
