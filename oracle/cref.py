"""ORACLE (test infrastructure) -- ctypes loader for the C restatement
(oracle/c/oracle.c).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product package never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

u64p = ctypes.POINTER(ctypes.c_uint64)
u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    src = os.path.join(_HERE, "c", "oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        # -march=native must not travel to another host: build generic x86-64-v2
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(
            ["gcc", "-O3", "-fopenmp", "-fPIC", "-std=c11", "-shared", "-o", _SO, src])
    return _SO


def _cpu_budget():
    """CPUs this process may really use: min(affinity, cgroup quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def lib():
    global _lib
    if _lib is None:
        build()
        os.environ.setdefault("OMP_NUM_THREADS", str(_cpu_budget()))
        L = ctypes.CDLL(_SO)
        L.oracle_gl_to_mont.restype = ctypes.c_uint64
        L.oracle_gl_to_mont.argtypes = [ctypes.c_uint64]
        L.oracle_gl_from_mont.restype = ctypes.c_uint64
        L.oracle_gl_from_mont.argtypes = [ctypes.c_uint64]
        for nm in ("oracle_gl_mul", "oracle_gl_add", "oracle_gl_sub", "oracle_gl_pow"):
            getattr(L, nm).restype = ctypes.c_uint64
            getattr(L, nm).argtypes = [ctypes.c_uint64, ctypes.c_uint64]
        L.oracle_gl_inv.restype = ctypes.c_uint64
        L.oracle_gl_inv.argtypes = [ctypes.c_uint64]
        L.oracle_gl_root_of_unity.restype = ctypes.c_uint64
        L.oracle_gl_root_of_unity.argtypes = [ctypes.c_uint]
        L.oracle_fq3_mul.argtypes = [u64p, u64p, u64p]
        L.oracle_fq3_inv.argtypes = [u64p, u64p]
        L.oracle_bit_reverse.argtypes = [u64p, ctypes.c_uint, ctypes.c_uint]
        L.oracle_ntt.argtypes = [u64p, ctypes.c_uint, ctypes.c_uint, ctypes.c_int, ctypes.c_uint64]
        L.oracle_lde.argtypes = [u64p, u64p, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint,
                                 ctypes.c_uint64, ctypes.c_int]
        L.oracle_sha256.argtypes = [u8p, ctypes.c_size_t, u8p]
        L.oracle_sha256_rows.argtypes = [ctypes.POINTER(u64p), ctypes.c_uint, ctypes.c_uint,
                                         ctypes.c_size_t, u8p]
        L.oracle_sha256_merkle.argtypes = [u8p, ctypes.c_size_t, u8p]
        L.oracle_fri_fold.argtypes = [u64p, u64p, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint,
                                      u64p, ctypes.c_uint64]
        L.oracle_binary.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_uint, ctypes.c_size_t,
                                    u64p, u64p, u64p, ctypes.c_size_t]
        L.oracle_binary_const.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_uint,
                                          ctypes.c_size_t, u64p, u64p, u64p]
        L.oracle_mul_pow.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_size_t, u64p, u64p,
                                     u64p, ctypes.c_uint, ctypes.c_size_t]
        L.oracle_unary.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_size_t, u64p, u64p,
                                   ctypes.c_uint]
        L.oracle_sum_columns.argtypes = [ctypes.POINTER(u64p), ctypes.c_uint, ctypes.c_uint,
                                         ctypes.c_size_t, u64p]
        L.oracle_num_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def _p8(a):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u8p)


GL_P = (1 << 64) - (1 << 32) + 1


def to_mont(a):
    """numpy u64 canonical -> Montgomery (vectorised through Python ints for
    exactness; used on test-sized arrays only)."""
    R = (1 << 64) % GL_P
    flat = [(int(x) * R) % GL_P for x in np.asarray(a, dtype=np.uint64).ravel()]
    return np.array(flat, dtype=np.uint64).reshape(np.shape(a))


def from_mont(a):
    Rinv = pow((1 << 64) % GL_P, -1, GL_P)
    flat = [(int(x) * Rinv) % GL_P for x in np.asarray(a, dtype=np.uint64).ravel()]
    return np.array(flat, dtype=np.uint64).reshape(np.shape(a))


def random_elements(n, seed, V=1):
    """Uniform canonical Goldilocks residues, returned in MONTGOMERY form is
    unnecessary: the uniform distribution is invariant under *R, so a uniform
    value in [0,p) is used directly as the stored (Montgomery) word."""
    rng = np.random.default_rng(seed)
    out = rng.integers(0, GL_P, size=n * V, dtype=np.uint64, endpoint=False)
    return out


def ntt(a, log_n, V=1, inverse=False, offset=1):
    out = np.ascontiguousarray(a, dtype=np.uint64).copy()
    lib().oracle_ntt(_p(out), log_n, V, 1 if inverse else 0, offset)
    return out


def bit_reverse(a, log_n, V=1):
    out = np.ascontiguousarray(a, dtype=np.uint64).copy()
    lib().oracle_bit_reverse(_p(out), log_n, V)
    return out


def lde(a, log_n, log_blowup, V=1, offset=7, bit_reversed=True):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty((a.size << log_blowup,), dtype=np.uint64)
    lib().oracle_lde(_p(a), _p(out), log_n, log_blowup, V, offset, 1 if bit_reversed else 0)
    return out


def sha256_rows(cols, V=1):
    cols = [np.ascontiguousarray(c, dtype=np.uint64) for c in cols]
    nrows = cols[0].size // V
    arr = (u64p * len(cols))(*[_p(c) for c in cols])
    out = np.empty((nrows, 32), dtype=np.uint8)
    lib().oracle_sha256_rows(arr, len(cols), V, nrows, _p8(out))
    return out


def sha256_merkle(leaves):
    leaves = np.ascontiguousarray(leaves, dtype=np.uint8)
    n = leaves.shape[0]
    nodes = np.empty((n, 32), dtype=np.uint8)
    lib().oracle_sha256_merkle(_p8(leaves), n, _p8(nodes))
    return nodes


def fri_fold(evals, log_n, V, ff, alpha, offset=1):
    evals = np.ascontiguousarray(evals, dtype=np.uint64)
    alpha = np.ascontiguousarray(alpha, dtype=np.uint64)
    out = np.empty((evals.size // ff,), dtype=np.uint64)
    lib().oracle_fri_fold(_p(evals), _p(out), log_n, V, ff, _p(alpha), offset)
    return out


def binary(op, VL, VR, lhs, rhs, shift=0):
    lhs = np.ascontiguousarray(lhs, dtype=np.uint64)
    rhs = np.ascontiguousarray(rhs, dtype=np.uint64)
    n = lhs.size // VL
    dst = np.empty_like(lhs)
    lib().oracle_binary(op, VL, VR, n, _p(dst), _p(lhs), _p(rhs), shift % n)
    return dst


def binary_const(op, VL, VR, lhs, c):
    lhs = np.ascontiguousarray(lhs, dtype=np.uint64)
    c = np.ascontiguousarray(c, dtype=np.uint64)
    dst = np.empty_like(lhs)
    lib().oracle_binary_const(op, VL, VR, lhs.size // VL, _p(dst), _p(lhs), _p(c))
    return dst


def mul_pow(VL, VR, lhs, rhs, e, shift=0):
    lhs = np.ascontiguousarray(lhs, dtype=np.uint64)
    rhs = np.ascontiguousarray(rhs, dtype=np.uint64)
    n = lhs.size // VL
    dst = np.empty_like(lhs)
    lib().oracle_mul_pow(VL, VR, n, _p(dst), _p(lhs), _p(rhs), e, shift % n)
    return dst


def unary(op, V, src, e=0):
    src = np.ascontiguousarray(src, dtype=np.uint64)
    dst = np.empty_like(src)
    lib().oracle_unary(op, V, src.size // V, _p(dst), _p(src), e)
    return dst


def sum_columns(cols, V=1):
    cols = [np.ascontiguousarray(c, dtype=np.uint64) for c in cols]
    arr = (u64p * len(cols))(*[_p(c) for c in cols])
    dst = np.empty_like(cols[0])
    lib().oracle_sum_columns(arr, len(cols), V, cols[0].size // V, _p(dst))
    return dst


def num_threads():
    return lib().oracle_num_threads()
