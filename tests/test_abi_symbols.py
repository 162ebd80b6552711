"""The C-ABI library loads (no GPU needed for dlopen) and exports every symbol that
include/ministark_hip.h declares; the ctypes binding declares the same set."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ministark_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ms_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_by_hip_library():
    from ministark_amd import build
    so = build.build(verbose=False)
    lib = ctypes.CDLL(so)
    names = _declared()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"not exported: {missing}"


def test_ctypes_binding_covers_header():
    from ministark_amd import _lib
    L = _lib.Lib()
    assert sorted(L.sigs) == _declared()


def test_no_cpu_fallback_in_package():
    # the package must never reach for the oracle or the simulator
    pkg = os.path.join(ROOT, "ministark_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".cpp")):
                text = open(os.path.join(d, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "libministark_emu" not in text, f
