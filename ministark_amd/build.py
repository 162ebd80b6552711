"""Build libministark_hip.so (hipcc, gfx950) in-tree.

    python -m ministark_amd.build            # build if sources are newer
    python -m ministark_amd.build --force

The shared object is git-ignored but travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libministark_hip.so")
SOURCES = ["ms_core.cpp", "ms_ntt.cpp", "ms_stage.cpp", "ms_hash.cpp", "ms_eval.cpp", "ms_deep.cpp", "ms_comm.cpp"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-Wall", "-Wno-unused-function"]
OBJDIR = os.path.join(HERE, "_obj")


def _deps():
    out = []
    for d, _, files in os.walk(CSRC):
        out += [os.path.join(d, f) for f in files if f.endswith((".h", ".cpp", ".hip", ".inc"))]
    out.append(os.path.join(ROOT, "include", "ministark_hip.h"))
    return out


JIT_HEADERS = ["gl.h", "gl_dev.h", "fp252.h", "stage_kernels.h", "eval_kernels.h"]
EMBED = os.path.join(CSRC, "_embedded_headers.inc")


def embed_headers():
    """The device headers the specialised constraint kernels include (csrc/eval_jit.h) travel inside
    the library as strings: hiprtc has no include path on the box that runs it."""
    parts = []
    for name in JIT_HEADERS:
        text = open(os.path.join(CSRC, name)).read()
        assert ')MSHDR"' not in text
        parts.append('{"%s", R"MSHDR(%s)MSHDR"},\n' % (name, text))
    new = "".join(parts)
    if not os.path.exists(EMBED) or open(EMBED).read() != new:
        with open(EMBED, "w") as f:
            f.write(new)


STAMP = SO + ".srchash"


def source_hash(cmd):
    """Content hash of every source the library is compiled from, plus the command line: the binary is rebuilt
    when this changes, whatever the file times say (the .so is git-ignored and travels with snapshots)."""
    import hashlib
    h = hashlib.sha256(" ".join(cmd).encode())
    for p in sorted(_deps()):
        h.update(os.path.relpath(p, ROOT).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()


def build(force=False, verbose=True):
    """One hipcc -c per translation unit, in parallel, then one link: the library is seven units (context / NTT / stages /
    hashes / constraint evaluation / DEEP / RCCL exchange) instead of one 2 400-line file."""
    from concurrent.futures import ThreadPoolExecutor
    embed_headers()
    extra = os.environ.get("MS_HIPCC_FLAGS", "").split()
    cmd_id = [HIPCC] + FLAGS + extra + SOURCES + ["-lhiprtc", "-ldl"]
    want = source_hash(cmd_id)
    have = open(STAMP).read().strip() if os.path.exists(STAMP) else None
    if not force and os.path.exists(SO) and have == want:
        return SO
    if not os.path.exists(HIPCC):
        # a box without the compiler (none is expected): keep a binary that is at least present -- and say so when it is stale
        if os.path.exists(SO):
            if have != want:
                print(f"[ministark_amd.build] WARNING: {HIPCC} not found and {SO} was built from different sources "
                      f"(stamp {str(have)[:12]} != {want[:12]}); using it as it is", file=sys.stderr, flush=True)
            return SO
        raise FileNotFoundError(f"{HIPCC} not found and {SO} has not been built")
    os.makedirs(OBJDIR, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(OBJDIR, src.replace(".cpp", ".o"))
        cmd = [HIPCC] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[ministark_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    # the link step sees the same architecture / code-generation options as the units (MS_HIPCC_FLAGS may override the arch);
    # options that only make sense for a compile (-x hip, -c, warnings, -std) stay out
    link_flags = [f for f in FLAGS + extra if f.startswith(("--offload-arch", "-O", "-f", "-m", "-g"))]
    link = [HIPCC] + link_flags + ["-shared"] + objs + ["-o", SO, "-lhiprtc", "-ldl"]
    if verbose:
        print("[ministark_amd.build]", " ".join(link), flush=True)
    subprocess.check_call(link)
    with open(STAMP, "w") as f:
        f.write(want + "\n")
    return SO


if __name__ == "__main__":
    build(force="--force" in sys.argv)
