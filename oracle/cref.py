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
        i32p = ctypes.POINTER(ctypes.c_int32)
        L.oracle_eval_expr.restype = ctypes.c_int
        L.oracle_eval_expr.argtypes = [i32p, ctypes.c_uint, u64p, ctypes.c_int, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint,
                                       ctypes.c_uint64, u64p, ctypes.POINTER(u64p), ctypes.c_uint, ctypes.POINTER(u64p),
                                       u64p, u64p, ctypes.POINTER(u64p), ctypes.POINTER(ctypes.c_uint32), u64p, u64p]
        L.oracle_f252_mul.argtypes = [u64p, u64p, u64p]
        L.oracle_f252_inv.argtypes = [u64p, u64p]
        L.oracle_f252_ntt.argtypes = [u64p, ctypes.c_uint, ctypes.c_int, u64p]
        L.oracle_f252_lde.argtypes = [u64p, u64p, ctypes.c_uint, ctypes.c_uint, u64p, ctypes.c_int]
        L.oracle_horner_eval.argtypes = [u64p, ctypes.c_size_t, ctypes.c_uint, u64p, ctypes.c_uint, u64p]
        L.oracle_divide_out_points_acc.argtypes = [u64p, ctypes.c_size_t, ctypes.c_uint, u64p, u64p, ctypes.c_uint, ctypes.c_uint, u64p]
        L.oracle_degree_adjust.argtypes = [u64p, ctypes.c_size_t, ctypes.c_uint, u64p, u64p, u64p]
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


F252_ONE_MONT = np.array([18446744073709551585, 18446744073709551615, 18446744073709551615, 576460752303422960], dtype=np.uint64)


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


def ntt252(a, log_n, inverse=False, offset_mont=None):
    """Fp252 column (4 Montgomery words per element); offset_mont: 4 words, default R mod p (the subgroup)."""
    out = np.ascontiguousarray(a, dtype=np.uint64).copy()
    off = np.ascontiguousarray(F252_ONE_MONT if offset_mont is None else offset_mont, dtype=np.uint64)
    lib().oracle_f252_ntt(_p(out), log_n, 1 if inverse else 0, _p(off))
    return out


def lde252(a, log_n, log_blowup, offset_mont, bit_reversed=True):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty((a.size << log_blowup,), dtype=np.uint64)
    off = np.ascontiguousarray(offset_mont, dtype=np.uint64)
    lib().oracle_f252_lde(_p(a), _p(out), log_n, log_blowup, _p(off), 1 if bit_reversed else 0)
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


# ---- constraint evaluation (oracle_eval_expr: eval_cpu::eval, 512-point chunks + batch inversion) -------------
F252_P = (1 << 251) + 17 * (1 << 192) + 1
_F252_R = (1 << 256) % F252_P


def _f252_words(x):
    m = (int(x) * _F252_R) % F252_P
    return [(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def _flatten(expr, mode, fq_words):
    """DAG (objects with .kind / .args, nothing imported from the product) -> topologically ordered
    (kind, a, b) triples, shared sub-expressions once; constants as Montgomery words."""
    KIND = {"x": 0, "const": 1, "challenge": 2, "hint": 3, "trace": 4, "periodic": 5, "neg": 6, "add": 7, "mul": 8, "div": 9, "pow": 10}
    nodes, consts, periodic, index = [], [], [], {}
    R = (1 << 64) % GL_P

    def mont(v):
        return (int(v) * R) % GL_P

    def visit(e):
        if id(e) in index:
            return index[id(e)]
        k = e.kind
        if k in ("neg", "pow"):
            a = visit(e.args[0])
            rec = (KIND[k], a, e.args[1] if k == "pow" else 0)
        elif k in ("add", "mul", "div"):
            a, b = visit(e.args[0]), visit(e.args[1])
            rec = (KIND[k], a, b)
        elif k == "const":
            v = e.args[1]
            off = len(consts)
            if mode == 1:
                consts.extend(_f252_words(v))
                rec = (1, off, 0)
            elif isinstance(v, tuple):
                consts.extend(mont(c) for c in v)
                rec = (1, off, 1)
            else:
                consts.append(mont(v % GL_P))
                rec = (1, off, 0)
        elif k == "periodic":
            periodic.append((e.args[0], e.args[1]))
            rec = (5, len(periodic) - 1, 0)
        elif k == "trace":
            rec = (4, e.args[0], e.args[1])
        elif k in ("challenge", "hint"):
            rec = (KIND[k], e.args[0], 0)
        elif k == "x":
            rec = (0, 0, 0)
        else:
            raise ValueError(k)
        nodes.append(rec)
        index[id(e)] = len(nodes) - 1
        return index[id(e)]

    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 100000))
    visit(expr)
    return np.array(nodes, dtype=np.int32).reshape(-1, 3), np.array(consts + [0], dtype=np.uint64), periodic


def eval_expr(expr, log_n, lde_step, offset, base_cols, ext_cols, challenges, hints, fq_is_ext=True, field="goldilocks"):
    """All 2^log_n evaluations of the constraint DAG (Montgomery words in, Montgomery words out).
    base_cols / ext_cols: numpy u64 columns (V = 1 / 3 words per row, 4 for the 252-bit field);
    challenges / hints: numpy u64 arrays of Fq elements.  Returns numpy u64 [n * fq_words]."""
    L = lib()
    mode = 1 if field == "f252" else 0
    fq_words = 4 if mode else (3 if fq_is_ext else 1)
    nodes, consts, periodic = _flatten(expr, mode, fq_words)
    n = 1 << log_n
    trace_len = n // lde_step
    tabs = []
    for coeffs, interval in periodic:                      # eval_periodic_column (eval_cpu.rs:233-256)
        if mode:
            raise NotImplementedError("periodic columns on the 252-bit field")
        size = interval * lde_step
        t = np.zeros(size, dtype=np.uint64)
        t[:len(coeffs)] = to_mont(np.array([c % GL_P for c in coeffs], dtype=np.uint64))
        off = pow(offset, trace_len // interval, GL_P)
        tabs.append(ntt(t, size.bit_length() - 1, 1, False, off))
    P8 = ctypes.POINTER(ctypes.c_uint64)
    base = (P8 * max(1, len(base_cols)))(*[_p(c) for c in base_cols])
    ext = (P8 * max(1, len(ext_cols)))(*[_p(c) for c in ext_cols])
    per = (P8 * max(1, len(tabs)))(*[_p(t) for t in tabs])
    plen = (ctypes.c_uint32 * max(1, len(tabs)))(*[len(t) for t in tabs])
    ch = np.ascontiguousarray(np.asarray(challenges, dtype=np.uint64).ravel()) if len(challenges) else np.zeros(4, dtype=np.uint64)
    hi = np.ascontiguousarray(np.asarray(hints, dtype=np.uint64).ravel()) if len(hints) else np.zeros(4, dtype=np.uint64)
    out = np.empty(n * fq_words, dtype=np.uint64)
    off252 = np.array(_f252_words(offset) if mode else [0, 0, 0, 0], dtype=np.uint64)
    w252 = np.zeros(4, dtype=np.uint64)
    if mode:
        g = pow(3, (F252_P - 1) >> 192, F252_P)
        w252 = np.array(_f252_words(pow(g, 1 << (192 - log_n), F252_P)), dtype=np.uint64)
    flat = np.ascontiguousarray(nodes.ravel())
    rc = L.oracle_eval_expr(flat.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), len(nodes), _p(consts), mode, fq_words, log_n, lde_step,
                            int(offset) % GL_P if not mode else 0, _p(off252), base, len(base_cols), ext, _p(ch), _p(hi), per, plen, _p(w252), _p(out))
    assert rc == 0
    return out


# ---- DEEP composition (src/utils.rs:124-175, src/composer.rs:43-188) in Montgomery words -----------------------
def horner_eval(coeffs, V, point):
    """coeffs: numpy u64 (V words per coefficient); point: numpy u64 of PW words (Fq).  -> PW words."""
    point = np.ascontiguousarray(point, dtype=np.uint64)
    out = np.zeros(len(point), dtype=np.uint64)
    lib().oracle_horner_eval(_p(coeffs), len(coeffs) // V, V, _p(point), len(point), _p(out))
    return out


def deep_compose(polys, Vs, terms_by_poly, n, PW, degree):
    """sum over polynomials of divide_out_points_into(poly, zs, alphas), then the degree adjustment.
    polys: numpy columns; Vs: words per coefficient of each; terms_by_poly[p] = (zs, alphas) numpy arrays of
    k x PW words (k may be 0); degree = (alpha, beta) PW-word arrays.  -> numpy u64 [n * PW]."""
    L = lib()
    acc = np.zeros(n * PW, dtype=np.uint64)
    for poly, V, (zs, cs) in zip(polys, Vs, terms_by_poly):
        k = len(zs) // PW if len(zs) else 0
        if k:
            zs = np.ascontiguousarray(zs, dtype=np.uint64); cs = np.ascontiguousarray(cs, dtype=np.uint64)
            L.oracle_divide_out_points_acc(_p(poly), n, V, _p(zs), _p(cs), k, PW, _p(acc))
    out = np.empty(n * PW, dtype=np.uint64)
    da = np.ascontiguousarray(degree[0], dtype=np.uint64); db = np.ascontiguousarray(degree[1], dtype=np.uint64)
    L.oracle_degree_adjust(_p(acc), n, PW, _p(da), _p(db), _p(out))
    return out
