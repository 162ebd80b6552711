"""C3 LDE (2^20 x 32, blow-up 8) kernel-time probe: per-pass kernel time for the column-group size in MS_NTT_GROUP_BYTES."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ministark_amd import GOLDILOCKS_FP, GOLDILOCKS_FQ3, Matrix, Planner

if os.environ.get("MS_LIB"):                      # a second build of the library (before / after timings)
    from ministark_amd import _lib
    pl = Planner(0, _lib.Lib(os.environ["MS_LIB"]))
else:
    pl = Planner(0)
log_n, log_b, ncols = int(os.environ.get("LOGN", 20)), int(os.environ.get("LOGB", 3)), int(os.environ.get("NCOLS", 32))
rng = np.random.default_rng(3)
P = (1 << 64) - (1 << 32) + 1
FIELD = GOLDILOCKS_FQ3 if os.environ.get("FQ3") else GOLDILOCKS_FP          # FQ3=1: extension-field columns (three words per element)
V = 3 if os.environ.get("FQ3") else 1
trace = Matrix.from_numpy(pl, [rng.integers(0, P, size=V << log_n, dtype=np.uint64) for _ in range(ncols)], FIELD)
for _ in range(3):
    lde = trace.lde(1 << log_b, 7, True); del lde
pl.sync()
pl.profile(True)
for _ in range(5):
    lde = trace.lde(1 << log_b, 7, True); del lde
pl.sync()
prof = pl.profile_read()
pl.profile(False)
k = {name: round(v["total_us"] / 5, 1) for name, v in sorted(prof.items())}
print(json.dumps({"group_bytes": os.environ.get("MS_NTT_GROUP_BYTES"), "kernel_us": k, "lde_ms": round(sum(k.values()) / 1e3, 3)}))
