"""Wall time of pipeline.prove_phases at configs[4]'s size with and without the waits at the phase boundaries (medians of 9), and the kernel sum."""
import os, sys, time
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
for timed in (True, False, True, False):
    ts = []
    for it in range(12):
        pl.sync()
        t0 = time.perf_counter()
        out = pipeline.prove_phases(pl, trace, comp, draws, 4, 8, 64, 8, ce_blowup=ce, time_phases=timed)
        pl.sync()
        if it >= 3:
            ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print(f"time_phases={timed}: median {ts[len(ts) // 2]:.3f} ms, best {ts[0]:.3f}", out.get("phases_ms", ""), out.get("openings_ms"))
pl.profile(True)
for _ in range(3):
    pipeline.prove_phases(pl, trace, comp, draws, 4, 8, 64, 8, ce_blowup=ce, time_phases=False)
pl.sync()
pr = pl.profile_read()
print("kernel sum ms:", round(sum(v["total_us"] for v in pr.values()) / 3e3, 3))
