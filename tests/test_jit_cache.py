"""The on-disk cache of the specialised constraint kernels (csrc/jit_cache.h, ms_eval_jit_stats).

The reference pays no run-time compilation (gpu/src/plan.rs:30: the metallib is a build artefact); a process here pays hiprtc once per
program EVER -- later processes load the code object.  hiprtc needs no device, so the cache itself is tested in the GPU-less container
through ms_eval_jit_check; the GPU test proves that a damaged entry costs a recompilation and never a different output word."""
import ctypes
import glob
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from ministark_amd import GOLDILOCKS_FP as FP, GOLDILOCKS_FQ3 as FQ3, STARK252_FP
from ministark_amd import expr as E
from ministark_amd._lib import Lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _programs():
    x = E.X()
    expr = (E.Trace(0, 1) - E.Trace(0) * E.Trace(1) + E.Trace(2) * E.Trace(3, -1)) / (x ** 64 - 1) + E.Challenge(0) * x ** 3 + E.Trace(2) ** 7
    return [(E.compile_expr(expr, 2, True), FQ3), (E.compile_expr(expr, 4, False), FP), (E.compile_expr(expr, 4, False, STARK252_FP), STARK252_FP)]


def _check(L, prog, field):
    code = np.array(prog.instrs, dtype=np.uint32).reshape(-1, 4)
    size = ctypes.c_size_t(0)
    t0 = time.perf_counter()
    rc = L.ms_eval_jit_check(code.ctypes.data, len(code), field, ctypes.byref(size))
    assert rc == 0, L.ms_last_error().decode()[:2000]
    return size.value, (time.perf_counter() - t0) * 1e3


@pytest.fixture
def cache_dir(tmp_path, monkeypatch):
    d = tmp_path / "jit"
    monkeypatch.setenv("MS_JIT_CACHE", str(d))
    return d


def test_second_lookup_loads_from_disk(cache_dir):
    L = Lib()
    before = L.jit_stats()
    sizes = [_check(L, p, f)[0] for p, f in _programs()]
    mid = L.jit_stats()
    assert mid["kernels_compiled"] - before["kernels_compiled"] == 3 and mid["kernels_from_disk"] == before["kernels_from_disk"]
    files = sorted(glob.glob(str(cache_dir / "*.co")))
    assert len(files) == 3 and not glob.glob(str(cache_dir / ".tmp-*"))
    again = [_check(L, p, f) for p, f in _programs()]
    after = L.jit_stats()
    assert [s for s, _ in again] == sizes
    assert after["kernels_from_disk"] - mid["kernels_from_disk"] == 3 and after["kernels_compiled"] == mid["kernels_compiled"]
    assert all(ms < 20.0 for _, ms in again), again                       # second-process cost < 20 ms per program
    assert after["compile_failures"] == before["compile_failures"] == 0


def test_a_second_process_compiles_nothing(cache_dir):
    L = Lib()
    prog, field = _programs()[1]
    _check(L, prog, field)
    child = ("import sys, ctypes, numpy as np; sys.path.insert(0, %r)\n"
             "from tests.test_jit_cache import _programs, _check\n"
             "from ministark_amd._lib import Lib\n"
             "L = Lib(); p, f = _programs()[1]; _check(L, p, f); s = L.jit_stats(); print(s['kernels_compiled'], s['kernels_from_disk'])\n") % ROOT
    r = subprocess.run([sys.executable, "-c", child], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split() == ["0", "1"]


@pytest.mark.parametrize("damage", ["flip", "truncate", "append", "magic", "empty"])
def test_damaged_entries_are_dropped_and_recompiled(cache_dir, damage):
    L = Lib()
    prog, field = _programs()[1]
    size, _ = _check(L, prog, field)
    (path,) = glob.glob(str(cache_dir / "*.co"))
    good = open(path, "rb").read()
    assert len(good) == 48 + size
    bad = bytearray(good)
    if damage == "flip":
        bad[48 + size // 2] ^= 0x40
    elif damage == "truncate":
        bad = bad[: 48 + size // 2]
    elif damage == "append":
        bad += b"\0"
    elif damage == "magic":
        bad[0:8] = b"NOTMINE1"
    else:
        bad = bytearray()
    open(path, "wb").write(bytes(bad))
    before = L.jit_stats()
    size2, _ = _check(L, prog, field)
    after = L.jit_stats()
    assert size2 == size
    assert after["damaged_entries"] - before["damaged_entries"] == 1 and after["kernels_compiled"] - before["kernels_compiled"] == 1
    assert open(path, "rb").read() == good                               # the entry is whole again (hiprtc is deterministic for one source)


def test_cache_can_be_switched_off(tmp_path, monkeypatch):
    monkeypatch.setenv("MS_JIT_CACHE", "0")
    monkeypatch.setenv("HOME", str(tmp_path))
    L = Lib()
    before = L.jit_stats()
    prog, field = _programs()[1]
    _check(L, prog, field)
    _check(L, prog, field)
    after = L.jit_stats()
    assert after["kernels_compiled"] - before["kernels_compiled"] == 2 and after["kernels_from_disk"] == before["kernels_from_disk"]
    assert not os.path.exists(tmp_path / ".cache")


def test_unwritable_cache_directory_is_not_an_error(tmp_path, monkeypatch):
    blocker = tmp_path / "file"
    blocker.write_text("x")
    monkeypatch.setenv("MS_JIT_CACHE", str(blocker / "sub"))               # cannot be created: a file is in the way
    L = Lib()
    prog, field = _programs()[1]
    assert _check(L, prog, field)[0] > 1000


@pytest.mark.gpu
def test_damaged_cache_entry_never_changes_an_output_word_hip(cache_dir):
    """A fresh context (its in-memory table is empty) evaluates a program on 2^16 points three times: compiling, loading from disk, and
    with every cache file damaged -- the three outputs are the same words, equal to the oracle's, and the statistics say what happened."""
    from oracle import cref
    from ministark_amd import GpuVec, Planner
    x = E.X()
    c = [lambda o=0, k=k: E.Trace(k, o) for k in range(4)]
    expr = (c[0](1) - c[0]() * c[1]() + c[2]() * c[3](-1)) / (x ** 1024 - 1) + E.Challenge(0) * c[2]() ** 5 + (c[1]() - c[3]()) / (x - 1)
    prog = E.compile_expr(expr, 4, False)
    log_n, n = 16, 1 << 16
    base = [cref.random_elements(n, 600 + k) for k in range(4)]
    ch = cref.random_elements(1, 700).reshape(-1, 1)
    want = cref.eval_expr(expr, log_n, 2, 7, base, [], ch, ch[:1], False)
    outs, stats = [], []
    for round_ in range(3):
        if round_ == 2:
            files = glob.glob(str(cache_dir / "*.co"))
            assert files
            for k, path in enumerate(files):
                b = bytearray(open(path, "rb").read())
                if k % 2:
                    b = b[: len(b) // 2]
                else:
                    b[len(b) // 2] ^= 1
                open(path, "wb").write(bytes(b))
        pl = Planner(0)
        try:
            cols = [GpuVec.from_numpy(pl, b, FP) for b in base]
            outs.append(E.eval(prog, pl, ch, ch[:1], 2, 7, n, cols, []).to_numpy())
            stats.append(pl.jit_stats())
        finally:
            pl.close()
    assert all(np.array_equal(o, want) for o in outs)
    assert stats[0]["kernels_compiled"] >= 1 and stats[0]["kernels_from_disk"] == 0
    assert stats[1]["kernels_compiled"] == 0 and stats[1]["kernels_from_disk"] == stats[0]["kernels_compiled"]
    assert stats[2]["damaged_entries"] == stats[0]["kernels_compiled"] == stats[2]["kernels_compiled"]
    assert all(s["compile_failures"] == 0 for s in stats)
    assert stats[1]["load_ms"] < 20.0 * max(1, stats[1]["kernels_from_disk"])


@pytest.mark.gpu
@pytest.mark.parametrize("bit_reversed", [False, True])
def test_lds_staging_of_twice_read_columns_252_hip(tmp_path, bit_reversed):
    """MS_EVAL_STAGE=1 (off by default: fewer HBM fetches, slower) -- the 252-bit specialised kernel stages the columns and the
    full-length inverse table it reads at several row offsets in LDS once per workgroup (csrc/eval_jit.h, ev252_stage).  A child process
    (the switch is part of the generated source) evaluates a program with next / previous-row reads and boundary + terminal denominators
    on 2^16 points, natural layout (staged) and bit-reversed layout (the staged kernel's global-load branch): every word equals the C oracle's,
    and the domain's wrap-around rows are among the staged ones."""
    child = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from oracle import cref
from ministark_amd import STARK252_FP, GpuVec, Planner, expr as E
from ministark_amd.api import Radix2EvaluationDomain
bitrev = %r
log_n, lde_step, offset = 16, 4, 3
n = 1 << log_n
dom = Radix2EvaluationDomain(n // lde_step, 1, STARK252_FP)
g = dom.group_gen
X = E.X()
c = [lambda o=0, k=k: E.Trace(k, o) for k in range(3)]
expr = (c[0](1) - c[0]() * c[1](-1) + c[2](2)) / (X - E.Constant(1)) + (c[1]() - c[2](1)) / (X - E.Constant(pow(g, dom.p - 2, dom.p))) + E.Challenge(0) * c[2]() * c[0](1)
prog = E.compile_expr(expr, 3, False, STARK252_FP)
def el(k, sd):
    r = np.random.default_rng(sd)
    a = r.integers(0, 1 << 63, size=4 * k, dtype=np.uint64)
    a[3::4] >>= np.uint64(4)
    return a
base = [el(n, 900 + k) for k in range(3)]
ch = el(1, 990).reshape(-1, 4)
want = cref.eval_expr(expr, log_n, lde_step, offset, base, [], ch, ch[:1], False, field="f252")
pl = Planner(0)
cols = [GpuVec.from_numpy(pl, cref.bit_reverse(b, log_n, 4) if bitrev else b, STARK252_FP) for b in base]
out = E.eval(prog, pl, ch, ch[:1], lde_step, offset, n, cols, [], bit_reversed=bitrev).to_numpy()
if bitrev:
    want = cref.bit_reverse(want, log_n, 4)
st = pl.jit_stats()
print("equal", bool(np.array_equal(out, want)), "compiled", st["kernels_compiled"], "failed", st["compile_failures"])
''' % (ROOT, bit_reversed)
    env = dict(os.environ, MS_EVAL_STAGE="1", MS_JIT_CACHE=str(tmp_path / "jit"), MS_EVAL_DUMP=str(tmp_path / "src.hip"))
    r = subprocess.run([sys.executable, "-c", child], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "equal True" in r.stdout and "failed 0" in r.stdout, r.stdout
    assert "ev252_stage(" in open(tmp_path / "src.hip").read()           # the staged form was generated, not skipped
