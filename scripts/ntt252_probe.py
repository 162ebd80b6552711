"""Fp252 NTT / LDE timing probe: per-kernel microseconds (hipEvents) and wall time per column.
(The round-2 A/B switch MS_NTT252_RADIX2 is gone: profiles/r02_ntt252_probe.txt keeps that comparison.)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ministark_amd import STARK252_FP, GpuFft, GpuVec, Matrix, Planner, Radix2EvaluationDomain

pl = Planner(0)
rng = np.random.default_rng(1)


def timed(fn, reps=5):
    fn(); pl.sync()
    pl.profile(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    pl.sync()
    wall = (time.perf_counter() - t0) / reps
    prof = pl.profile_read()
    pl.profile(False)
    return wall, {k: round(v["total_us"] / reps, 1) for k, v in sorted(prof.items())}


for log_n, ncols in ((20, 1), (20, 2), (20, 8), (22, 2), (16, 16)):
    cols = [GpuVec.from_numpy(pl, rng.integers(0, 1 << 59, size=4 << log_n, dtype=np.uint64), STARK252_FP) for _ in range(ncols)]
    fft = GpuFft(Radix2EvaluationDomain(1 << log_n, 3, STARK252_FP), STARK252_FP, pl)

    def run():
        for c in cols:
            fft.encode(c)
        fft.execute()
    wall, k = timed(run)
    print(json.dumps({"what": f"Fp252 NTT 2^{log_n} x{ncols}, coset 3", "us_per_column": round(wall * 1e6 / ncols, 1),
                      "kernel_us_per_column": round(sum(k.values()) / ncols, 1), "kernel_us": k}))
    del cols, fft
for log_n, log_b, ncols in ((18, 2, 4), (16, 3, 8)):
    m = Matrix([GpuVec.from_numpy(pl, rng.integers(0, 1 << 59, size=4 << log_n, dtype=np.uint64), STARK252_FP) for _ in range(ncols)])
    wall, k = timed(lambda: m.lde(1 << log_b, 3, True))
    print(json.dumps({"what": f"Fp252 LDE 2^{log_n} x{ncols} blow-up {1 << log_b}, bit-reversed", "us_per_column": round(wall * 1e6 / ncols, 1),
                      "kernel_us_per_column": round(sum(k.values()) / ncols, 1), "kernel_us": k}))
