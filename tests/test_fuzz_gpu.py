"""The two randomised differential runs (tests/fuzz_parity.py: transforms, LDE, FRI, commitments, stages against the C
oracle; tests/fuzz_eval.py: random constraint programs against the C and the Python evaluators) as GPU tests with
fixed seeds and a fixed time budget.  Their output is appended to gpurun_out/r02_fuzz.log when that directory exists
(the summary kept under profiles/ is copied from there)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUDGET = os.environ.get("MS_FUZZ_SECONDS", "20")


def _run(script, seed, env=None):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", script), BUDGET, str(seed)], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=dict(os.environ, **(env or {})))
    text = p.stdout.decode()
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "r02_fuzz.log"), "a") as f:
            f.write(f"$ {' '.join(k + '=' + v for k, v in (env or {}).items())} python tests/{script} {BUDGET} {seed}\n{text.strip().splitlines()[-1] if text.strip() else '(no output)'}\n")
    assert p.returncode == 0, text[-3000:]
    assert "ok:" in text, text[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_parity_hip(seed):
    _run("fuzz_parity.py", seed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [21, 22])
def test_fuzz_eval_hip(seed):
    _run("fuzz_eval.py", seed)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"MS_EVAL_REGROUP": "force"}, {"MS_EVAL_REGROUP": "force", "MS_FUZZ_FIELD": "f252"}, {"MS_EVAL_REGROUP": "0"},
                                 {"MS_EVAL_SHARE_TABLES": "0", "MS_EVAL_FUSE_DENOMINATORS": "0", "MS_EVAL_HOST_TABLES": "0"}],
                         ids=["regroup-forced", "regroup-forced-252", "regroup-off", "tables-plain"])
def test_fuzz_eval_rewriting_pass_hip(env):
    """the sums-of-products pass (csrc/eval_regroup.h) applied to EVERY random program it can be applied to -- Goldilocks with and without
    Fq3 values, and the 252-bit field -- and switched off; "tables-plain": every inverse table computed, stored and inverted on its own on the device
    (csrc/eval_shift.h and the fused / host-side tables off).  The same outputs as the C oracle every way"""
    _run("fuzz_eval.py", 23, env)
