"""The GENERATED SOURCE of the specialised constraint kernels, checked without a GPU (csrc/eval_jit.h's text was only exercised on
hardware).  The simulator build compiles the same text the product hands to hiprtc (csrc/eval_jit_source.h) with g++ against tests/emu's
hip_runtime.h (tests/emu/emu_jit.h, MS_EMU_JIT=1), launches it through the fiber scheduler and every output word is compared with the C oracle:
a generator that emits the wrong helper, operand order, accumulator opcode or row offset for an instruction fails here.

Each case runs in its own process: the switches (MS_EMU_JIT, MS_EVAL_JIT_MIN_LOG_N, the rewriting passes') are read once per process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JIT_ENV = {"MS_EMU_JIT": "1", "MS_EVAL_JIT_MIN_LOG_N": "0", "MS_EVAL_SPLIT_MIN_LOG_N": "6", "OMP_NUM_THREADS": "2"}

AIRS = r"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from tests import backends
from oracle import cref
from ministark_amd import GOLDILOCKS_FP, GOLDILOCKS_FQ3, STARK252_FP, GpuVec, expr as E, pipeline
pl = backends.planner("emu")
P = (1 << 64) - (1 << 32) + 1
rng = np.random.default_rng(7)

def f252_cols(k, n):
    cols = [rng.integers(0, 1 << 63, size=4 * n, dtype=np.uint64) for _ in range(k)]
    for c in cols:
        c[3::4] >>= np.uint64(4)
    return cols

def case(name, comp, nch, log_n, lde_step, offset, field, fq_ext, base, ext, ch):
    n = 1 << log_n
    prog = E.compile_expr(comp, len(base), fq_ext, field)
    before = pl.jit_stats()
    out = E.eval(prog, pl, ch, ch[:1], lde_step, offset, n, [GpuVec.from_numpy(pl, c, field) for c in base],
                 [GpuVec.from_numpy(pl, c, GOLDILOCKS_FQ3) for c in ext]).to_numpy()
    after = pl.jit_stats()
    got = after["kernels_compiled"] + after["kernels_from_disk"] - before["kernels_compiled"] - before["kernels_from_disk"]
    assert after["compile_failures"] == 0, "a generated kernel did not compile"
    assert got >= 1, f"{name}: no generated kernel ran ({after})"
    kw = {"field": "f252"} if field == STARK252_FP else {}
    want = cref.eval_expr(comp, log_n, lde_step, offset, base, ext, ch, ch[:1], fq_ext, **kw)
    assert np.array_equal(out, want), f"{name}: {int((out != want).sum())} words differ from the oracle"
    # other challenges, the same kernels: the generated source depends on the program, not on the values of its constants (a prover
    # draws new challenges for every proof and must not meet the compiler again -- ADVICE r5 on csrc/ms_eval.cpp)
    ch2 = (ch + np.uint64(12345)) % np.uint64(P) if field != STARK252_FP else ch + np.uint64(977)
    out2 = E.eval(prog, pl, ch2, ch2[:1], lde_step, offset, n, [GpuVec.from_numpy(pl, c, field) for c in base],
                  [GpuVec.from_numpy(pl, c, GOLDILOCKS_FQ3) for c in ext]).to_numpy()
    again = pl.jit_stats()
    assert (again["kernels_compiled"], again["kernels_from_disk"]) == (after["kernels_compiled"], after["kernels_from_disk"]), f"{name}: new challenges, new kernels ({after} -> {again})"
    assert np.array_equal(out2, cref.eval_expr(comp, log_n, lde_step, offset, base, ext, ch2, ch2[:1], fq_ext, **kw)), f"{name}: second set of challenges"
    print(f"{name}: {got} generated kernel(s), {out.size} words equal the oracle's, reused for a second set of challenges")

log_n = 8
n = 1 << log_n
comp, _, nch = pipeline.fib_constraints(n)                                                   # configs[3] (i): lde_step 1
case("fib_air_fp", comp, nch, log_n, 1, 7, GOLDILOCKS_FP, False, [rng.integers(0, P, size=n, dtype=np.uint64) for _ in range(8)], [],
     rng.integers(1, P, size=(nch, 1), dtype=np.uint64))
comp, nch = pipeline.mixed_air_constraints()                                                  # (ii): 17 Fp + 9 Fq3 columns, lde_step 2
case("mixed_17fp_9fq3", comp, nch, log_n, 2, 7, GOLDILOCKS_FP, True, [rng.integers(0, P, size=n, dtype=np.uint64) for _ in range(17)],
     [rng.integers(0, P, size=3 * n, dtype=np.uint64) for _ in range(9)], rng.integers(1, P, size=(nch, 3), dtype=np.uint64))
log_n = 6
n = 1 << log_n
comp, _, nch = pipeline.fib_constraints(n >> 2, 8, STARK252_FP)                               # (iii): the 252-bit field, lde_step 4
case("fib_air_fp252", comp, nch, log_n, 4, 3, STARK252_FP, False, f252_cols(8, n), [], rng.integers(0, 1 << 59, size=(nch, 4), dtype=np.uint64))
print("generated kernels ok")
"""


def _run(args, env, timeout=900):
    p = subprocess.run(args, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout, env=dict(os.environ, **env))
    return p.returncode, p.stdout.decode()


def test_generated_kernels_of_the_bench_airs_equal_the_oracle_emu():
    """the three AIRs of configs[3] -- fib over Goldilocks, 17 Fp + 9 Fq3 columns, fib over the 252-bit field -- through source generation,
    g++ and the simulator: every output word, and the library must report that generated kernels ran"""
    rc, text = _run([sys.executable, "-c", AIRS], JIT_ENV)
    assert rc == 0 and "generated kernels ok" in text, text[-3000:]


@pytest.mark.parametrize("env", [{}, {"MS_EVAL_REGROUP": "force"}, {"MS_EVAL_REGROUP": "force", "MS_FUZZ_FIELD": "f252"}], ids=["default", "regroup-forced", "regroup-forced-252"])
def test_fuzz_through_generated_kernels_emu(env):
    """tests/fuzz_eval.py's random programs, every one through a generated kernel (the accumulator opcodes of the regrouping pass included):
    the same outputs as the C oracle, and no generated source that fails to compile"""
    rc, text = _run([sys.executable, os.path.join(ROOT, "tests", "fuzz_eval.py"), os.environ.get("MS_JIT_FUZZ_SECONDS", "20"), "31"],
                    dict(JIT_ENV, MS_FUZZ_BACKEND="emu", **env))
    assert rc == 0 and "fuzz_eval ok" in text, text[-3000:]
    last = text.strip().splitlines()[-1]
    compiled = int(last.split("specialised kernels: ")[1].split(" compiled")[0]) + int(last.split("compiled, ")[1].split(" from the cache")[0])
    assert compiled >= 5, last                                # the run went through generated kernels, not around them
