#!/usr/bin/env python3
"""Commit phase only (SHA-256 row hashing + Merkle levels) on an HBM-resident column-major matrix:
a fixed target for rocprofv3 (kernel trace or --pmc) when tuning the hash kernels.
    python scripts/bench_commit.py [log_rows] [ncols] [reps]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ministark_amd import GOLDILOCKS_FP as FP, Matrix, MerkleTree, Planner  # noqa: E402

log_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 23
ncols = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
P = (1 << 64) - (1 << 32) + 1
pl = Planner(0)
rng = np.random.default_rng(3)
base = rng.integers(0, P, size=1 << log_rows, dtype=np.uint64)
m = Matrix.from_numpy(pl, [np.roll(base, c) for c in range(ncols)], FP)
root = None


def commit():
    global root
    root = MerkleTree.from_matrix(m).root()


t_settle = time.perf_counter()                     # settle clocks before timing
while time.perf_counter() - t_settle < 0.5:
    commit()
    pl.sync()
pl.profile(True)
t0 = time.perf_counter()
for _ in range(reps):
    commit()
pl.sync()
wall = (time.perf_counter() - t0) / reps
prof = {k: round(v["total_us"] / reps, 1) for k, v in pl.profile_read().items()}
nrows = 1 << log_rows
blocks_rows = nrows * ((ncols + 2 + 7) // 8)
blocks_tree = 2 * (nrows - 1)
print(json.dumps({"config": f"commit 2^{log_rows} rows x {ncols} Fp columns", "wall_ms": round(wall * 1e3, 3), "kernel_us": prof,
                  "rows_Gcompress_per_s": round(blocks_rows / prof["sha256_rows"] / 1e3, 2),
                  "tree_Gcompress_per_s": round(blocks_tree / prof["sha256_merkle_level"] / 1e3, 2),
                  "root": root.hex()}))
