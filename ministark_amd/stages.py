"""The reference's element-wise stage structs (gpu/src/stage.rs:115-1155), same names and
argument order: `XStage(planner, n, lhs_field[, rhs_field])` then `.encode(buffers..., scalars...)`.
The reference encodes into a caller-owned command buffer and the caller commits/waits; here encode
enqueues on the planner's stream and `planner.sync()` is the wait."""
import ctypes

import numpy as np

from .api import FIELD_WORDS, GOLDILOCKS_FP, GpuVec

ADD, MUL = 0, 1
NEG, INV, EXP = 0, 1, 2


def _const(value, field):
    a = np.ascontiguousarray(value, dtype=np.uint64).ravel()
    assert a.size == FIELD_WORDS[field], "constant has the wrong number of limbs for its field"
    return a


class _Stage:
    def __init__(self, planner, n, lhs_field=GOLDILOCKS_FP, rhs_field=None):
        # stage.rs:55-59: n must be a power of two (and >= 2048 there; no lower bound here)
        if n < 1 or n & (n - 1):
            raise ValueError("n must be a power of two")
        self.planner, self.n = planner, n
        self.lf = lhs_field
        self.rf = lhs_field if rhs_field is None else rhs_field

    def _chk(self, *vecs):
        for v in vecs:
            if len(v) != self.n:
                raise ValueError("buffer length differs from the stage's n")


class MulAssignStage(_Stage):        # stage.rs:176-233
    def encode(self, lhs, rhs, shift=0):
        self._chk(lhs, rhs)
        L = self.planner.lib
        L.check(L.ms_binary(self.planner.handle, MUL, self.lf, self.rf, self.n, lhs.ptr, lhs.ptr, rhs.ptr, shift))


class MulIntoStage(_Stage):          # stage.rs:115-174
    def encode(self, dst, lhs, rhs, shift=0):
        self._chk(dst, lhs, rhs)
        L = self.planner.lib
        L.check(L.ms_binary(self.planner.handle, MUL, self.lf, self.rf, self.n, dst.ptr, lhs.ptr, rhs.ptr, shift))


class AddAssignStage(_Stage):        # stage.rs:393-455
    def encode(self, lhs, rhs, shift=0):
        self._chk(lhs, rhs)
        L = self.planner.lib
        L.check(L.ms_binary(self.planner.handle, ADD, self.lf, self.rf, self.n, lhs.ptr, lhs.ptr, rhs.ptr, shift))


class AddIntoStage(_Stage):          # stage.rs:457-521
    def encode(self, dst, lhs, rhs, shift=0):
        self._chk(dst, lhs, rhs)
        L = self.planner.lib
        L.check(L.ms_binary(self.planner.handle, ADD, self.lf, self.rf, self.n, dst.ptr, lhs.ptr, rhs.ptr, shift))


class _ConstStage(_Stage):
    OP = ADD

    def _run(self, dst, lhs, value):
        self._chk(dst, lhs)
        c = _const(value, self.rf)
        L = self.planner.lib
        L.check(L.ms_binary_const(self.planner.handle, self.OP, self.lf, self.rf, self.n, dst.ptr, lhs.ptr, c.ctypes.data))


class AddIntoConstStage(_ConstStage):    # stage.rs:523-579
    OP = ADD

    def encode(self, dst, lhs, value):
        self._run(dst, lhs, value)


class AddAssignConstStage(_ConstStage):  # stage.rs:637-692
    OP = ADD

    def encode(self, lhs, value):
        self._run(lhs, lhs, value)


class MulIntoConstStage(_ConstStage):    # stage.rs:694-750
    OP = MUL

    def encode(self, dst, lhs, value):
        self._run(dst, lhs, value)


class MulAssignConstStage(_ConstStage):  # stage.rs:752-806
    OP = MUL

    def encode(self, lhs, value):
        self._run(lhs, lhs, value)


class MulPowStage(_Stage):           # stage.rs:334-391: lhs *= rhs[(i+shift)%n]^power
    def encode(self, lhs, rhs, power, shift=0):
        self._chk(lhs, rhs)
        L = self.planner.lib
        L.check(L.ms_mul_pow(self.planner.handle, self.lf, self.rf, self.n, lhs.ptr, lhs.ptr, rhs.ptr, power, shift))


class ConvertIntoStage(_Stage):      # stage.rs:581-635 (lhs_field = destination, rhs_field = source)
    def encode(self, dst, src):
        self._chk(dst, src)
        L = self.planner.lib
        L.check(L.ms_convert(self.planner.handle, self.lf, self.rf, self.n, dst.ptr, src.ptr))


class _UnaryStage(_Stage):
    OP = NEG

    def _run(self, dst, src, e=0):
        self._chk(dst, src)
        L = self.planner.lib
        L.check(L.ms_unary(self.planner.handle, self.OP, self.lf, self.n, dst.ptr, src.ptr, e))


class NegInPlaceStage(_UnaryStage):      # stage.rs:855-900
    OP = NEG

    def encode(self, buf):
        self._run(buf, buf)


class NegIntoStage(_UnaryStage):         # stage.rs:902-947
    OP = NEG

    def encode(self, dst, src):
        self._run(dst, src)


class InverseInPlaceStage(_UnaryStage):  # stage.rs:808-853
    OP = INV

    def encode(self, buf):
        self._run(buf, buf)


class InverseIntoStage(_UnaryStage):     # stage.rs:949-997
    OP = INV

    def encode(self, dst, src):
        self._run(dst, src)


class ExpInPlaceStage(_UnaryStage):      # stage.rs:1056-1109
    OP = EXP

    def encode(self, buf, exponent):
        self._run(buf, buf, exponent)


class ExpIntoStage(_UnaryStage):         # stage.rs:999-1054
    OP = EXP

    def encode(self, dst, src, exponent):
        self._run(dst, src, exponent)


class FillBuffStage(_Stage):             # stage.rs:1111-1155
    def encode(self, dst, value):
        self._chk(dst)
        c = _const(value, self.lf)
        L = self.planner.lib
        L.check(L.ms_fill(self.planner.handle, self.lf, self.n, dst.ptr, c.ctypes.data))


def sum_columns(matrix):
    """`Matrix::sum_columns` (src/matrix.rs:357-394) -> a one-column GpuVec."""
    pl = matrix.planner
    n = matrix.num_rows()
    out = GpuVec(pl, n, matrix.field)
    arr = (ctypes.c_void_p * matrix.num_cols())(*[c.ptr for c in matrix.columns])
    pl.lib.check(pl.lib.ms_sum_columns(pl.handle, matrix.field, n, arr, matrix.num_cols(), out.ptr))
    return out
