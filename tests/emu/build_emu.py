"""TEST INFRASTRUCTURE ONLY: compile the product sources with g++ against the HIP
execution-model simulator (tests/emu/hip/hip_runtime.h) into
tests/emu/_build/libministark_emu.so, so kernel logic can be checked against the
oracle without a GPU.  Never imported by the ministark_amd package."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "ministark_amd", "csrc")
SO = os.path.join(HERE, "_build", "libministark_emu.so")


def build(force=False):
    import sys
    sys.path.insert(0, ROOT)
    from ministark_amd.build import SOURCES                    # the same translation units as the product library
    srcs = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "emu_runtime.cpp")]
    deps = list(srcs) + [os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(HERE, "emu_jit.h"), os.path.join(ROOT, "include", "ministark_hip.h")]
    for d, _, files in os.walk(CSRC):
        deps += [os.path.join(d, f) for f in files]
    newest = max(os.path.getmtime(p) for p in deps)
    if not force and os.path.exists(SO) and os.path.getmtime(SO) >= newest:
        return SO
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + HERE, "-I" + CSRC, "-Wall", "-Wno-unused-function",
           "-Wno-unknown-pragmas", "-DMS_NO_JIT", f'-DMS_EMU_DIR="{HERE}"', f'-DMS_CSRC_DIR="{CSRC}"'] + srcs + ["-o", SO, "-ldl"]
    subprocess.check_call(cmd)
    return SO


def build_fake_rccl():
    """tests/emu/fake_rccl.cpp -> _build/libfake_rccl.so: the NCCL entry points ms_comm.cpp binds, between CPU processes over
    shared memory (MS_RCCL_LIB points the library at it)."""
    src = os.path.join(HERE, "fake_rccl.cpp")
    so = os.path.join(HERE, "_build", "libfake_rccl.so")
    if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(src):
        return so
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", src, "-o", so, "-lrt", "-pthread"])
    return so


if __name__ == "__main__":
    print(build(force=True))
