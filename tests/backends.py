"""Backends the parity tests run against.

  "hip": the product library on a real MI355X (tests marked gpu)
  "emu": the same sources compiled by g++ against the HIP execution-model simulator
         in tests/emu (kernel-logic tests in the GPU-less container; test infra only)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ministark_amd import _lib, api  # noqa: E402

_cache = {}


def planner(kind, device=0):
    """One Planner per (backend, GPU): a rank of a multi-GPU test must run its kernels on ITS device."""
    key = (kind, device)
    if key not in _cache:
        if kind == "hip":
            lib = _lib.Lib()
        elif kind == "emu":
            sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
            import build_emu
            lib = _lib.Lib(build_emu.build())
        else:
            raise ValueError(kind)
        _cache[key] = api.Planner(device, lib)
    return _cache[key]
