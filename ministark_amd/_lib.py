"""ctypes loader for libministark_hip.so -- the ONLY native library the package loads.

There is no CPU fallback: if the hipcc-built library is missing, or no AMD GPU is
visible when a context is created, this raises.  (tests/emu builds a g++ simulator
of the same sources for kernel-logic tests; that library is loaded by the tests
through `Lib(path)` explicitly and is never looked for here.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_SO = os.path.join(_HERE, "libministark_hip.so")

c_void_pp = ctypes.POINTER(ctypes.c_void_p)


class MsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ministark_hip error {code}: {msg}")
        self.code = code


class JitStats(ctypes.Structure):
    """`ms_jit_stats` of include/ministark_hip.h."""
    _fields_ = [("kernels_compiled", ctypes.c_uint64), ("kernels_from_disk", ctypes.c_uint64), ("compile_failures", ctypes.c_uint64),
                ("damaged_entries", ctypes.c_uint64), ("compile_ms", ctypes.c_double), ("load_ms", ctypes.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Lib:
    """Typed view of the C ABI in include/ministark_hip.h."""

    def __init__(self, path=None):
        path = path or DEFAULT_SO
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} not found: build it with `python -m ministark_amd.build` (hipcc, gfx950). "
                "There is no CPU fallback.")
        self.path = path
        L = ctypes.CDLL(path)
        self.L = L
        vp, u, i, sz = ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_size_t
        sigs = {
            "ms_ctx_create": (i, [i, c_void_pp]),
            "ms_ctx_destroy": (i, [vp]),
            "ms_sync": (i, [vp]),
            "ms_ctx_stream": (vp, [vp]),
            "ms_last_error": (ctypes.c_char_p, []),
            "ms_field_bytes": (sz, [i]),
            "ms_profile_enable": (i, [vp, i]),
            "ms_profile_read": (i, [vp, ctypes.c_char_p, sz]),
            "ms_alloc": (i, [vp, sz, c_void_pp]),
            "ms_free": (i, [vp, vp]),
            "ms_copy": (i, [vp, vp, vp, sz]),
            "ms_upload": (i, [vp, vp, vp, sz]),
            "ms_download": (i, [vp, vp, vp, sz]),
            "ms_ntt_plan_create": (i, [vp, i, u, i, vp, vp, c_void_pp]),
            "ms_ntt_plan_destroy": (i, [vp]),
            "ms_ntt_encode": (i, [vp, vp]),
            "ms_ntt_execute": (i, [vp]),
            "ms_ntt_enqueue": (i, [vp, c_void_pp, u]),
            "ms_ntt_enqueue_to": (i, [vp, c_void_pp, c_void_pp, u]),
            "ms_bit_reverse": (i, [vp, i, u, c_void_pp, u]),
            "ms_lde": (i, [vp, i, u, u, vp, c_void_pp, c_void_pp, u, i]),
            "ms_evaluate": (i, [vp, i, u, u, vp, c_void_pp, c_void_pp, u, i]),
            "ms_deinterleave": (i, [vp, i, sz, u, vp, c_void_pp]),
            "ms_binary": (i, [vp, i, i, i, sz, vp, vp, vp, ctypes.c_long]),
            "ms_binary_const": (i, [vp, i, i, i, sz, vp, vp, vp]),
            "ms_mul_pow": (i, [vp, i, i, sz, vp, vp, vp, u, ctypes.c_long]),
            "ms_unary": (i, [vp, i, i, sz, vp, vp, u]),
            "ms_convert": (i, [vp, i, i, sz, vp, vp]),
            "ms_fill": (i, [vp, i, sz, vp, vp]),
            "ms_sum_columns": (i, [vp, i, sz, c_void_pp, u, vp]),
            "ms_eval_program": (i, [vp, vp, u, vp, u, u, u, vp, vp, c_void_pp, u, c_void_pp, u, c_void_pp, vp, u, i, vp]),
            "ms_eval_program_ex": (i, [vp, vp, u, vp, u, u, u, vp, vp, c_void_pp, u, c_void_pp, u, c_void_pp, vp, u, i, vp, u]),
            "ms_eval_jit_check": (i, [vp, u, i, vp]),
            "ms_eval_jit_stats": (i, [vp, ctypes.POINTER(JitStats)]),
            "ms_scan_affine": (i, [vp, i, sz, vp, vp, vp, i, vp]),
            "ms_gather_rows": (i, [vp, i, sz, c_void_pp, u, vp, sz, vp]),
            "ms_gather_digests": (i, [vp, sz, vp, vp, sz, vp]),
            "ms_gather_digests_multi": (i, [vp, u, c_void_pp, ctypes.POINTER(sz), vp, ctypes.POINTER(sz), c_void_pp]),
            "ms_merkle_view_ids": (i, [sz, vp, sz, vp, vp, vp, vp, vp]),
            "ms_fri_fold": (i, [vp, i, u, u, vp, vp, vp, vp]),
            "ms_fri_fold_rows": (i, [vp, i, u, u, vp, vp, sz, sz, vp, vp]),
            "ms_sha256_rows": (i, [vp, i, sz, c_void_pp, u, vp]),
            "ms_sha256_merkle": (i, [vp, sz, vp, vp]),
            "ms_horner_eval": (i, [vp, i, i, sz, c_void_pp, u, vp, vp, u, vp]),
            "ms_deep_compose": (i, [vp, i, u, vp, c_void_pp, u, c_void_pp, u, vp, u, vp, vp, vp, vp, u, vp, vp, vp]),
            "ms_deep_rows": (i, [vp, i, u, vp, sz, sz, c_void_pp, u, c_void_pp, u, vp, u, vp, vp, vp, vp, u, vp, vp, vp]),
            "ms_sha256_pow_grind": (i, [vp, vp, u, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]),
            "ms_rpo256_rows": (i, [vp, sz, c_void_pp, u, vp]),
            "ms_rpo256_rows_row_major": (i, [vp, sz, u, vp, vp]),
            "ms_rpo256_merkle": (i, [vp, sz, vp, vp]),
            "ms_rpo256_rows_field": (i, [vp, i, sz, c_void_pp, u, vp]),
            "ms_comm_unique_id": (i, [vp]),
            "ms_comm_init": (i, [vp, i, i, vp]),
            "ms_comm_destroy": (i, [vp]),
            "ms_comm_rank": (i, [vp, ctypes.POINTER(i), ctypes.POINTER(i)]),
            "ms_cols_to_rows_alltoall": (i, [vp, i, sz, c_void_pp, u, u, c_void_pp]),
            "ms_allgather_digests": (i, [vp, vp, vp]),
            "ms_cols_to_rows_schedule": (i, [u, u, u, u, sz, vp, sz, ctypes.POINTER(sz)]),
            "ms_p2p_batch": (i, [vp, vp, sz]),
            "ms_sha256_rows_row_major": (i, [vp, i, sz, u, vp, vp]),
        }
        self.optional = {}
        for name, (res, args) in sigs.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        self.sigs = sigs

    def declare(self, name, res, args):
        fn = getattr(self.L, name)
        fn.restype = res
        fn.argtypes = args
        return fn

    def jit_stats(self, handle=None):
        """Specialised constraint kernels of a context (None: of the whole process): compiled / loaded from the disk cache / left to the
        interpreter, and the milliseconds each cost (ms_eval_jit_stats)."""
        st = JitStats()
        self.check(self.L.ms_eval_jit_stats(handle, ctypes.byref(st)))
        return st.as_dict()

    def check(self, rc):
        if rc != 0:
            raise MsError(rc, self.L.ms_last_error().decode())

    def __getattr__(self, name):
        return getattr(self.L, name)


_default = None


def lib():
    global _default
    if _default is None:
        _default = Lib()
    return _default
