"""Where the host time of the C5-shaped prover chain goes: wall without per-launch events, then a cProfile of the calls."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ministark_amd import GOLDILOCKS_FP, Matrix, Planner, pipeline

pl = Planner(0)
log_t, blowup, folding, ncols = 22, 4, 8, 8
n_t = 1 << log_t
rng = np.random.default_rng(5)
P = (1 << 64) - (1 << 32) + 1
trace = Matrix.from_numpy(pl, [rng.integers(0, P, size=n_t, dtype=np.uint64) for _ in range(ncols)], GOLDILOCKS_FP)
comp, ce, nch = pipeline.fib_constraints(n_t, ncols)
draws = pipeline.Draws(0xC5, ncols, nch, ce, 32, n_t * blowup, pipeline.fri_num_layers(n_t * blowup, blowup, folding, 64))
run = lambda: pipeline.prove_phases(pl, trace, comp, draws, blowup, folding, 64, 8, ce_blowup=ce)
for _ in range(2):
    r = run()
pl.sync()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); r = run(); pl.sync(); ts.append((time.perf_counter() - t0) * 1e3)
print("wall ms (no per-launch events):", [round(t, 2) for t in ts], "phases:", r["phases_ms"])
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    run()
pl.sync()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
