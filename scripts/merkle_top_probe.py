"""Latency of the upper part of a SHA-256 Merkle tree (sha256_merkle_top): trees of 2^9 leaves (one workgroup, 8 levels) and 2^18 leaves (512 workgroups x 9
levels, then the 2^9-leaf remainder), kernel time from the library's per-launch events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np
from ministark_amd import Planner
from ministark_amd.api import DeviceBytes

pl = Planner(0)
L = pl.lib
for log_leaves in (9, 13, 18, 21, 23, 24):
    n = 1 << log_leaves
    leaves = DeviceBytes(pl, 32 * n); nodes = DeviceBytes(pl, 32 * n)
    host = np.random.default_rng(log_leaves).integers(0, 256, size=32 * n, dtype=np.uint8)
    L.check(L.ms_upload(pl.handle, leaves.ptr, host.ctypes.data, host.nbytes))
    for _ in range(20):
        L.check(L.ms_sha256_merkle(pl.handle, n, leaves.ptr, nodes.ptr))
    pl.sync()
    pl.profile(True)
    for _ in range(20):
        L.check(L.ms_sha256_merkle(pl.handle, n, leaves.ptr, nodes.ptr))
    rec = pl.profile_read(); pl.profile(False)
    print(f"2^{log_leaves} leaves:", {k: (v["calls"] // 20, round(v["total_us"] / 20, 1)) for k, v in rec.items()})
