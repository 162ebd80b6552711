"""Phase 1 of the prover (interpolate + LDE + commit of the base trace, 2^22 x 8, blow-up 4) on one GPU: wall time against the sum of its kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ministark_amd import GpuVec, Matrix, MerkleTree, Planner, Radix2EvaluationDomain

pl = Planner(0)
log_rows, ncols, blowup = 22, 8, 4
n = 1 << log_rows
P = (1 << 64) - (1 << 32) + 1
trace = Matrix([GpuVec.from_numpy(pl, np.random.default_rng(c).integers(0, P, size=n, dtype=np.uint64)) for c in range(ncols)])
tdom, ldom = Radix2EvaluationDomain(n), Radix2EvaluationDomain(n * blowup, 7)

def phase():
    polys = trace.interpolate(tdom)
    lde = polys.bit_reversed_evaluate(ldom)
    tree = MerkleTree.from_matrix(lde, "sha256")
    return tree.root()

for _ in range(3):
    phase()
pl.sync()
ts = []
for _ in range(7):
    t0 = time.perf_counter(); phase(); ts.append((time.perf_counter() - t0) * 1e3)
pl.profile(True)
for _ in range(3):
    phase()
recs = pl.profile_read(); pl.profile(False)
print("wall ms (7 runs):", [round(t, 3) for t in ts])
print("kernel ms per run: %.3f" % (sum(v["total_us"] for v in recs.values()) / 3e3), {k: (v["calls"] // 3, round(v["total_us"] / 3, 1)) for k, v in recs.items()})
# the same phase split at every host-visible step
def timed():
    t = [time.perf_counter()]
    polys = trace.interpolate(tdom); pl.sync(); t.append(time.perf_counter())
    lde = polys.bit_reversed_evaluate(ldom); pl.sync(); t.append(time.perf_counter())
    tree = MerkleTree.from_matrix(lde, "sha256"); pl.sync(); t.append(time.perf_counter())
    tree.root(); t.append(time.perf_counter())
    return [round((b - a) * 1e3, 3) for a, b in zip(t, t[1:])]
timed()
print("interpolate / LDE / commit / root ms (synchronised after each):", timed(), timed())
