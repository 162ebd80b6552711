"""ORACLE (test infrastructure) -- the sequential loops that build extension columns.

Follows examples/brainfuck/trace.rs:108-289: every extension column is
    state = init;  for row: ext[row] = state;  state = a_row * state + b_row
(a running product when b = 0: instruction / memory permutation columns :131-145; a running
evaluation state * gamma + value: input / output evaluation columns :147-159; padding or
non-matching rows leave the state unchanged: a = 1, b = 0)."""
from .fields import GL, FQ3


def _ops(ext):
    if ext:
        return FQ3.mul, FQ3.add, FQ3.one(), FQ3.zero()
    return GL.mul, GL.add, 1, 0


def scan_affine(a, b, init, n, ext=False, inclusive=False):
    """a, b: lists of canonical elements (ints, or 3-tuples when ext) or None."""
    mul, add, one, zero = _ops(ext)
    out, state = [], init
    for i in range(n):
        if not inclusive:
            out.append(state)
        state = add(mul(a[i] if a is not None else one, state), b[i] if b is not None else zero)
        if inclusive:
            out.append(state)
    return out
