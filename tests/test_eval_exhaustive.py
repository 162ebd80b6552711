"""The constraint evaluator against an EXHAUSTIVE universe of programs (tests/exhaustive_eval.py) -- a net that does not depend on a random
generator's taste (a rewriting-pass bug lived through two rounds because the fuzzer never drew its shape).

Every expression DAG with <= 3 operator nodes over {neg, pow, add, mul, div} and <= 4 over {add, div} (depth <= 4, three leaves), for five
leaf triples, under four settings of the rewriting switches; each evaluation is checked twice: by the library itself (MS_EVAL_SELFCHECK:
the rewritten program against the original on the plain interpreter, word by word -- the pairwise comparison) and against the C oracle.
With the round-3 bug put back (commit 3571d78 reverted) the smallest universe already fails: (C / X) / (C / X), program 155 of
`--nodes 2 --ops add,div` (verified when this test was written).  Larger universes (<= 4 nodes over all operators: 2 x 10^5 programs per
triple; <= 5 over {add, div}: 6 x 10^5) run offline: profiles/r06_exhaustive_eval.log."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tests", "exhaustive_eval.py")
ALL_OPS = "neg,pow,add,mul,div"
SETTINGS = {
    "default": {},
    "regroup-forced": {"MS_EVAL_REGROUP": "force"},
    "regroup-off": {"MS_EVAL_REGROUP": "0"},
    "tables-plain": {"MS_EVAL_SHARE_TABLES": "0", "MS_EVAL_FUSE_DENOMINATORS": "0", "MS_EVAL_HOST_TABLES": "0"},
}


def _run(args, env_extra, timeout=1500):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, SCRIPT] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    return r.returncode, r.stdout.decode()[-1500:]


def _build_emu():
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    build_emu.build()


def test_exhaustive_universes_emu():
    """One pool for all of it (each universe x leaf triple x setting is its own process: the switches are read once per process):
    <= 3 nodes over all operators, five leaf triples, three settings (11 529 programs each); <= 4 nodes over {add, div} -- denominators built
    on other denominators, the shape of the round-3 bug and of boundary / terminal constraints that share a zerofier -- for the trace-bearing
    triple under the default switches and the x-only triple with the table rewrites off (56 835 programs each); regroup-off on one triple."""
    _build_emu()
    jobs = [(["--nodes", "4", "--ops", "add,div", "--leaves", "xtc"], "default", 56835), (["--nodes", "4", "--ops", "add,div", "--leaves", "xcc"], "tables-plain", 56835)]
    jobs += [(["--nodes", "3", "--ops", ALL_OPS, "--leaves", lv], st, 11529) for st in ("default", "regroup-forced", "tables-plain") for lv in ("xtc", "xtn", "xcc", "ttc", "xqc")]
    jobs += [(["--nodes", "3", "--ops", ALL_OPS, "--leaves", "xtc"], "regroup-off", 11529)]
    with ThreadPoolExecutor(min(8, os.cpu_count() or 4)) as ex:
        for (rc, out), (args, st, count) in zip(ex.map(lambda j: _run(j[0], SETTINGS[j[1]]), jobs), jobs):
            assert rc == 0 and f"exhaustive_eval ok: {count} programs" in out, (args, st, out)


@pytest.mark.gpu
@pytest.mark.parametrize("leaves", ["xtc", "xqc"])
def test_small_universe_through_the_specialised_kernels_hip(leaves):
    """2^16 points: every program of the two-node universe is compiled by hiprtc and runs as a specialised kernel; the self-check compares
    it with the interpreter on the original program, the oracle checks all 65536 outputs (the generated source of every opcode the rewriting
    passes emit -- csrc/eval_jit.h -- is exercised here)."""
    rc, out = _run(["--nodes", "2", "--ops", "add,mul,div", "--leaves", leaves, "--backend", "hip", "--log-n", "16"], {"MS_EXHAUSTIVE_OMP_THREADS": "8"})
    assert rc == 0 and "exhaustive_eval ok: 336 programs" in out, out
