"""`python bench.py --gpus N` rehearsed at the bench level: the DEFAULT mode (the driver's command) and --mode lde-commit at
toy sizes, N = 2 / 4 / 8 ranks started by bench.py itself, on the simulator build of the library with gloo for torch.distributed and the
shared-memory stand-in for RCCL (tests/emu/fake_rccl.cpp) -- so that the first run on an 8-GPU node is not the first execution of this path.
Checked: exactly one JSON line, below 4 KiB, n_gpus = N, the sharded commitment's root equal to the one-rank root, and the two bail-out
paths (a rank that dies before the exchange; the in-process timer).  The multi-GPU NUMBERS remain unmeasured on hardware."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOY = ["--log-n", "12", "--cols", "2", "--steps", "2", "--warmup", "1", "--settle", "0", "--log-rows", "8", "--total-cols", "8", "--no-cpu-baseline"]


def _env(**extra):
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    env = dict(os.environ, MS_BENCH_LIB=build_emu.build(), MS_BENCH_DIST_BACKEND="gloo", MS_RCCL_LIB=build_emu.build_fake_rccl(), MS_BENCH_NO_PMC="1", **extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _run(args, env, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    return r, lines


_one_rank = {}


def _baseline(mode_args):
    key = tuple(mode_args)
    if key not in _one_rank:
        r, lines = _run(mode_args + TOY + ["--gpus", "1"], _env())
        assert r.returncode == 0 and len(lines) == 1, r.stderr.decode()[-2000:]
        _one_rank[key] = json.loads(lines[0])
    return _one_rank[key]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_lde_commit_mode_n_ranks(world):
    one = _baseline(["--mode", "lde-commit"])
    r, lines = _run(["--mode", "lde-commit"] + TOY + ["--gpus", str(world)], _env())
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert len(lines) == 1 and len(lines[0]) < 4096
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["scaling"] == "strong"
    sh = line["sharded_lde_commit"]
    assert sh["n_gpus"] == world and sh["root"] == one["sharded_lde_commit"]["root"] is not None
    assert sh["prove"]["base_root"] == one["sharded_lde_commit"]["prove"]["base_root"]


@pytest.mark.parametrize("world", [2, 4])
def test_default_mode_n_ranks(world):
    """the driver's own command shape (`bench.py --gpus N --steps K --warmup W`), toy sizes: weak-scaling headline + the sharded objects"""
    one = _baseline(["--mode", "lde-commit"])              # (the default mode with ONE rank also runs every single-GPU object: not on the simulator)
    r, lines = _run(TOY + ["--gpus", str(world)], _env())
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert len(lines) == 1 and len(lines[0]) < 4096
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["scaling"] == "weak" and line["steps"] == 2 and line["warmup"] == 1
    assert "NTT" in line["metric"] and line["unit"] == "GB/s" and line["value"] > 0
    assert line["config"]["parallelism"] == f"columns x{world}"
    assert "sharded_error" not in line
    assert line["sharded_root"] == one["sharded_lde_commit"]["root"][:16]  # the same commitment whatever the number of ranks
    assert line["sharded_prove_ms"] > 0


def test_a_rank_that_dies_before_the_exchange_still_leaves_one_line():
    """rank 1 exits where the sharded phase starts: the launcher terminates the others, rank 0 prints the headline it had measured with
    the reason recorded; the exit code says the run failed"""
    r, lines = _run(TOY + ["--gpus", "2"], _env(MS_BENCH_TEST_FAIL_RANK="1"), timeout=300)
    assert r.returncode != 0
    assert len(lines) == 1, (r.stdout.decode()[-1000:], r.stderr.decode()[-2000:])
    line = json.loads(lines[0])
    # (how rank 0 learns of it depends on the transport: gloo reports the reset connection -> the recorded exception; RCCL would block ->
    # the launcher's SIGTERM -> "terminated by the launcher"; either way the headline is printed with the reason)
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["sharded_error"]


def test_the_in_process_timer_prints_the_headline_when_a_collective_hangs():
    """a rank that STALLS (no launcher notices anything): rank 0's own timer (180 s in production, 5 s here) fires inside the stuck exchange
    and prints the line"""
    env = _env(MS_BENCH_TEST_FAIL_RANK="1", MS_BENCH_TEST_FAIL_MODE="hang", MS_BENCH_BAIL_S="5", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29731")
    p0 = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + TOY + ["--gpus", "2"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    p1 = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + TOY + ["--gpus", "2"], cwd=ROOT, env=dict(env, RANK="1", LOCAL_RANK="1"),
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        out, err = p0.communicate(timeout=300)
    finally:
        for p in (p0, p1):
            if p.poll() is None:
                p.kill()
    lines = [ln for ln in out.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, err.decode()[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and "timed out" in line["sharded_error"] and line["value"] > 0
