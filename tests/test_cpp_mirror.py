"""The C++ host mirror (ministark_amd/csrc/host/ministark.hpp) compiles against the C header alone
(CPU check) and, on the GPU box, passes its parity program against the C oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "_build", "test_host_mirror")


def _build():
    from ministark_amd import build
    from oracle import cref
    so = build.build(verbose=False)
    osso = cref.build()
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", SRC, "-o", BIN, so, osso, "-Wl,-rpath," + os.path.dirname(so), "-Wl,-rpath," + os.path.dirname(osso),
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-fopenmp"]
    subprocess.check_call(cmd)
    return BIN


def test_cpp_mirror_compiles_and_links():
    assert os.path.exists(_build())


def test_cpp_mirror_parity_under_the_simulator():
    # the same parity program linked against the simulator build of the library (tests/emu): every mirror class against the C oracle
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    from oracle import cref
    so, osso = build_emu.build(), cref.build()
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "test_host_mirror_emu")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", SRC, "-o", exe, so, osso, "-Wl,-rpath," + os.path.dirname(so),
                           "-Wl,-rpath," + os.path.dirname(osso), "-fopenmp"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "cpp host mirror ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_mirror_parity_on_gpu():
    exe = _build()
    from oracle import cref
    env = dict(os.environ, OMP_NUM_THREADS=str(cref._cpu_budget()))
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "cpp host mirror ok" in out.stdout, out.stdout + out.stderr


EXAMPLE = os.path.join(ROOT, "examples", "fib_prover.cpp")
EXAMPLE_BIN = os.path.join(ROOT, "tests", "cpp", "_build", "fib_prover")


def _build_example():
    from ministark_amd import build
    so = build.build(verbose=False)
    os.makedirs(os.path.dirname(EXAMPLE_BIN), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", EXAMPLE, "-o", EXAMPLE_BIN, so, "-Wl,-rpath," + os.path.dirname(so), "-Wl,-rpath,/opt/rocm/lib"])
    return EXAMPLE_BIN


def test_fib_prover_example_compiles():
    assert os.path.exists(_build_example())


def test_fib_prover_example_under_the_simulator():
    # the same program linked against the simulator build of the library (tests/emu): the valid 2^12-row trace must give a
    # composition polynomial of degree < n over the whole LDE coset and a FRI remainder without high coefficients (fri.rs:244)
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    so = build_emu.build()
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "fib_prover_emu")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", EXAMPLE, "-o", exe, so, "-Wl,-rpath," + os.path.dirname(so)])
    for log_rows in ("10", "12"):
        out = subprocess.run([exe, log_rows, "1"], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "fib prover pipeline ok" in out.stdout, out.stdout + out.stderr
        assert "ce_blowup_factor 1 -> 197 instructions" in out.stdout, out.stdout       # the same program as pipeline.fib_constraints
    # the verifier relations recomputed inside the example are not vacuous: one altered opened value and the run fails
    for how, what in (("tamper-ood", "out-of-domain consistency"), ("tamper-row", "DEEP composition at query 17")):
        out = subprocess.run([exe, "10", "1", how], capture_output=True, text=True, timeout=600)
        assert out.returncode == 1 and "FAILED: " + what in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_fib_prover_example_on_gpu():
    # examples/fib on the device end to end: 2^14 rows here (the composition polynomial of the valid trace must have
    # degree < n: checked inside), then once at 2^18 rows so that the specialised evaluator kernel is exercised
    exe = _build_example()
    for log_rows in ("14", "18"):
        out = subprocess.run([exe, log_rows, "1"], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "fib prover pipeline ok" in out.stdout, out.stdout + out.stderr
