"""Host-side profile of distributed.prove_sharded on one GPU (world 1, RCCL communicator of one rank): where the wall time of its phases goes."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ministark_amd import GpuVec, Planner, pipeline
from ministark_amd.distributed import RcclComm, prove_sharded

pl = Planner(0)
comm = RcclComm(pl, 0, 1, RcclComm.unique_id(pl.lib))
log_rows, ncols = 22, 8
n_t = 1 << log_rows
P = (1 << 64) - (1 << 32) + 1
cols = [GpuVec.from_numpy(pl, np.random.default_rng(c).integers(0, P, size=n_t, dtype=np.uint64)) for c in range(ncols)]
comp, ce, nch = pipeline.fib_constraints(n_t, ncols)
draws = pipeline.Draws(0xC5, ncols, nch, ce, 32, n_t * 4, pipeline.fri_num_layers(n_t * 4, 4, 8, 64))
run = lambda ph=None: prove_sharded(pl, comm, cols, ncols, log_rows, comp, draws, 4, 8, 64, 8, ce_blowup=ce, phases_ms=ph)
for _ in range(2):
    run()
pl.sync()
ts = []
for _ in range(5):
    ph = {}
    t0 = time.perf_counter(); run(ph); pl.sync(); ts.append((time.perf_counter() - t0) * 1e3)
print("wall ms:", [round(t, 2) for t in ts], {k: round(v, 3) for k, v in ph.items()})
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    run()
pl.sync(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40); print(s.getvalue()[:7000])
comm.close()
