import json, sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from ministark_amd import GOLDILOCKS_FP as FP, GpuFft, GpuVec, Planner, Radix2EvaluationDomain
from ministark_amd.api import _ptr_array
P = (1 << 64) - (1 << 32) + 1
pl = Planner(0); rng = np.random.default_rng(1)
for log_n, ncol in ((12, 512), (13, 512), (14, 256), (15, 256), (16, 128), (17, 64), (18, 64)):
    n = 1 << log_n
    cols = [GpuVec.from_numpy(pl, rng.integers(0, P, size=n, dtype=np.uint64), FP) for _ in range(ncol)]
    plan = GpuFft(Radix2EvaluationDomain(n, 7), FP, pl)
    arr = _ptr_array(cols)
    for _ in range(20): plan.enqueue(cols)
    pl.sync()
    t0 = time.perf_counter()
    for _ in range(20): pl.lib.ms_ntt_enqueue(plan.handle, arr, ncol)
    pl.sync()
    wall = (time.perf_counter() - t0) / 20 * 1e6
    pl.profile(True)
    for _ in range(5): pl.lib.ms_ntt_enqueue(plan.handle, arr, ncol)
    pl.sync()
    recs = pl.profile_read(); pl.profile(False)
    print(log_n, ncol, 'wall us/batch %.1f' % wall, 'per col %.3f' % (wall / ncol), {k: (v['calls'] // 5, round(v['total_us'] / 5, 1)) for k, v in recs.items()})
