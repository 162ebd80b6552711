"""Column-sharded LDE + row-sharded Merkle commitment over 2 ranks (gloo, CPU, kernels under the
simulator): every rank must end with the root a single device computes for the whole matrix."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import cref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _expected():
    out = []
    for V, total_cols, log_n, log_b in ((1, 5, 6, 2), (3, 3, 5, 3), (1, 1, 7, 1)):
        cols = [cref.lde(cref.random_elements((1 << log_n) * V, 1000 + c), log_n, log_b, V, 7, True) for c in range(total_cols)]
        out.append(cref.sha256_merkle(cref.sha256_rows(cols, V))[1].tobytes().hex())
    return out


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_commit_matches_single_device_gloo(tmp_path, world):
    port = str(29500 + (os.getpid() % 400) + world)
    procs, files = [], []
    for r in range(world):
        f = str(tmp_path / f"r{r}.txt")
        files.append(f)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r), str(world), port, "emu", f],
                                      cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()[-2000:]
    want = _expected()
    for f in files:
        assert open(f).read().split("\n") == want
