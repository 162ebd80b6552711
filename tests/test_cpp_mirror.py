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


@pytest.mark.gpu
def test_cpp_mirror_parity_on_gpu():
    exe = _build()
    from oracle import cref
    env = dict(os.environ, OMP_NUM_THREADS=str(cref._cpu_budget()))
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "cpp host mirror ok" in out.stdout, out.stdout + out.stderr
