"""Host side of the fused constraint evaluator: the reference's expression DAG and its
lowering to the register program of include/ministark_hip.h ("constraint program").

Mirrors `Expr<AlgebraicItem<FieldVariant<Fp, Fq>>>` (src/expression.rs:33-40,
src/constraints.rs:21-28): nodes Leaf / Neg / Add / Mul / Div / Pow; `a - b` is
Add(a, Neg(b)) (src/expression.rs:573-580); leaves X, Constant, Challenge, Hint, Trace(col,
offset), Periodic.  `eval(...)` mirrors `eval_gpu::eval` / `eval_cpu::eval`
(src/eval_gpu.rs:46-54, src/eval_cpu.rs:33-42).  Only bookkeeping happens here (hash-consing,
typing, register allocation); every field operation runs in the HIP kernel.
"""
import ctypes

import numpy as np

from .api import (FIELD_WORDS, GOLDILOCKS_FP, GOLDILOCKS_FQ3, STARK252_FP, GL_P, F252_P, GpuFft, GpuVec, Radix2EvaluationDomain,
                  gl_to_mont, f252_to_mont_limbs)

FP, FQ = "fp", "fq"
(OP_X_P, OP_CONST_P, OP_CONST_Q, OP_TRACE_P, OP_TRACE_Q, OP_PERIODIC_P, OP_PERIODIC_Q, OP_NEG_P, OP_NEG_Q,
 OP_ADD_PP, OP_ADD_QQ, OP_ADD_QP, OP_MUL_PP, OP_MUL_QQ, OP_MUL_QP, OP_INV_P, OP_INV_Q, OP_POW_P, OP_POW_Q,
 OP_EMBED, OP_STORE_Q, OP_STORE_P) = range(22)


class Expr:
    """kind in {"x","const","challenge","hint","trace","periodic","neg","add","mul","div","pow"}."""
    __slots__ = ("kind", "args")

    def __init__(self, kind, *args):
        self.kind, self.args = kind, args

    def __add__(self, o): return Expr("add", self, _lift(o))
    def __radd__(self, o): return Expr("add", _lift(o), self)
    def __neg__(self): return Expr("neg", self)
    def __sub__(self, o): return Expr("add", self, Expr("neg", _lift(o)))          # expression.rs:573-580
    def __rsub__(self, o): return Expr("add", _lift(o), Expr("neg", self))
    def __mul__(self, o): return Expr("mul", self, _lift(o))
    def __rmul__(self, o): return Expr("mul", _lift(o), self)
    def __truediv__(self, o): return Expr("div", self, _lift(o))
    def __rtruediv__(self, o): return Expr("div", _lift(o), self)
    def __pow__(self, e): return Expr("pow", self, int(e))


def _lift(v):
    return v if isinstance(v, Expr) else Constant(v)


def X():
    return Expr("x")


def Constant(value, field=FP):
    """value: canonical int (Fp) or 3-tuple of canonical ints (Fq).  Integers are kept as given
    and reduced by the base field of the program they are compiled into."""
    if isinstance(value, tuple):
        return Expr("const", FQ, tuple(int(v) for v in value))
    return Expr("const", field, int(value)) if field == FP else Expr("const", FQ, (int(value), 0, 0))


def Challenge(i): return Expr("challenge", int(i))
def Hint(i): return Expr("hint", int(i))
def Trace(col, offset=0): return Expr("trace", int(col), int(offset))


def Periodic(coeffs, interval_size=None):
    """`PeriodicColumn::new(coeffs, interval_size)`: polynomial coefficients (canonical ints, Fp)
    of the column over one interval of the trace (src/constraints.rs PeriodicColumn)."""
    coeffs = tuple(int(c) % GL_P for c in coeffs)
    return Expr("periodic", coeffs, int(interval_size or len(coeffs)))


class Program:
    """The lowered program + what it needs at run time."""

    def __init__(self):
        self.instrs = []          # (op, dst, a, b)
        self.consts = []          # u64 words, Montgomery
        self.nchallenges = 0
        self.nhints = 0
        self.challenge_slots = {}  # index -> const word offset
        self.hint_slots = {}
        self.periodic = []        # (coeffs, interval)
        self.out_field = None
        self.max_p = 0
        self.max_q = 0


def compile_expr(expr, num_base_columns, fq_is_ext=True, base_field=GOLDILOCKS_FP):
    """Lower `expr` to a Program.  Trace columns < num_base_columns are Fp, the rest Fq
    (eval_cpu.rs:103-134).  Challenges and hints are Fq (eval_cpu.rs:111-113); with
    fq_is_ext=False (Fq = Fp AIRs such as examples/fib) they are Fp.  base_field = STARK252_FP
    compiles for the 252-bit field (Fq = Fp, 4-word elements)."""
    prog = Program()
    if base_field == STARK252_FP:
        if fq_is_ext:
            raise ValueError("the 252-bit field has no extension here: pass fq_is_ext=False")
        pwords, pmod = 4, F252_P
        to_words = lambda v: [int(w) for w in f252_to_mont_limbs(v % F252_P)]
    else:
        pwords, pmod = 1, GL_P
        to_words = lambda v: [gl_to_mont(v % GL_P)]
    prog.base_field = base_field
    memo = {}          # structural key -> (type, virtual register)
    nodes = []         # virtual instructions: [op, vdst, va, vb, type, imm]
    qtype = FQ if fq_is_ext else FP

    def const_slot(words):
        off = len(prog.consts)
        prog.consts.extend(words)
        return off

    def emit(op, typ, a=None, b=None, imm=0):
        v = len(nodes)
        nodes.append([op, v, a, b, typ, imm])
        return v

    def key_of(e, kids):
        if e.kind in ("neg", "add", "mul", "div"):
            ks = kids
            if e.kind in ("add", "mul"):
                ks = tuple(sorted(kids))
            return (e.kind,) + tuple(ks)
        if e.kind == "pow":
            return ("pow", kids[0], e.args[1])
        return (e.kind,) + tuple(e.args)

    # iterative post-order (DAGs can be deep)
    result_of = {}
    stack = [(expr, False)]
    while stack:
        e, ready = stack.pop()
        if id(e) in result_of:
            continue
        child = [a for a in e.args if isinstance(a, Expr)]
        if not ready and child:
            stack.append((e, True))
            for c in child:
                if id(c) not in result_of:
                    stack.append((c, False))
            continue
        kids = tuple(result_of[id(c)] for c in child)
        k = key_of(e, kids)
        if k in memo:
            result_of[id(e)] = memo[k]
            continue
        kd = e.kind
        if kd == "x":
            v = emit(OP_X_P, FP)
        elif kd == "const":
            if e.args[0] == FP:
                v = emit(OP_CONST_P, FP, imm=const_slot(to_words(e.args[1])))
            else:
                if pwords != 1:
                    raise ValueError("Fq constants need the Goldilocks extension")
                v = emit(OP_CONST_Q, FQ, imm=const_slot([gl_to_mont(c % GL_P) for c in e.args[1]]))
        elif kd in ("challenge", "hint"):
            table = prog.challenge_slots if kd == "challenge" else prog.hint_slots
            idx = e.args[0]
            if idx not in table:
                table[idx] = const_slot([0] * (3 if qtype == FQ else pwords))
            if kd == "challenge":
                prog.nchallenges = max(prog.nchallenges, idx + 1)
            else:
                prog.nhints = max(prog.nhints, idx + 1)
            v = emit(OP_CONST_Q if qtype == FQ else OP_CONST_P, qtype, imm=table[idx])
        elif kd == "trace":
            col, off = e.args
            if col < num_base_columns:
                v = emit(OP_TRACE_P, FP, imm=(col, off))
            else:
                v = emit(OP_TRACE_Q if qtype == FQ else OP_TRACE_P, qtype, imm=(col - num_base_columns if qtype == FQ else col, off))
        elif kd == "periodic":
            pid = len(prog.periodic)
            for j, pc in enumerate(prog.periodic):
                if pc == (e.args[0], e.args[1]):
                    pid = j
            if pid == len(prog.periodic):
                prog.periodic.append((e.args[0], e.args[1]))
            v = emit(OP_PERIODIC_P, FP, imm=pid)
        elif kd == "neg":
            t = nodes[kids[0]][4]
            v = emit(OP_NEG_P if t == FP else OP_NEG_Q, t, kids[0])
        elif kd in ("add", "mul", "div"):
            a, b = kids
            if kd == "div":
                tb = nodes[b][4]
                if ("inv", b) not in memo:
                    memo[("inv", b)] = emit(OP_INV_P if tb == FP else OP_INV_Q, tb, b)
                b = memo[("inv", b)]
            ta, tb = nodes[a][4], nodes[b][4]
            base = OP_ADD_PP if kd == "add" else OP_MUL_PP
            if ta == FP and tb == FP:
                v = emit(base, FP, a, b)
            elif ta == FQ and tb == FQ:
                v = emit(base + 1, FQ, a, b)
            else:
                if ta == FP:
                    a, b = b, a
                v = emit(base + 2, FQ, a, b)
        elif kd == "pow":
            t = nodes[kids[0]][4]
            ex = e.args[1]
            if ex < 0 or ex >= 1 << 32:
                raise ValueError("exponent out of range")
            v = emit(OP_POW_P if t == FP else OP_POW_Q, t, kids[0], imm=ex)
        else:
            raise ValueError(kd)
        memo[k] = v
        result_of[id(e)] = v

    root = result_of[id(expr)]
    # result is always Fq (into_fq_array, eval_cpu.rs:262-275); for Fq = Fp AIRs that is Fp
    if qtype == FQ and nodes[root][4] == FP:
        root = emit(OP_EMBED, FQ, root)
    prog.out_field = GOLDILOCKS_FQ3 if qtype == FQ else base_field

    # ---- register allocation: linear scan over the (already topological) node list
    last_use = {}
    for n in nodes:
        for opnd in (n[2], n[3]):
            if opnd is not None:
                last_use[opnd] = n[1]
    last_use[root] = len(nodes)
    free = {FP: [], FQ: []}
    nxt = {FP: 0, FQ: 0}
    reg = {}
    for n in nodes:
        op, v, a, b, typ, imm = n
        # operands that die here free their register before the destination is chosen
        for opnd in {a, b} - {None}:
            if last_use[opnd] == v:
                free[nodes[opnd][4]].append(reg[opnd])
        if free[typ]:
            r = free[typ].pop()
        else:
            r = nxt[typ]
            nxt[typ] += 1
        reg[v] = r
        if op in (OP_X_P,):
            prog.instrs.append((op, r, 0, 0))
        elif op in (OP_CONST_P, OP_CONST_Q, OP_PERIODIC_P, OP_PERIODIC_Q):
            prog.instrs.append((op, r, imm, 0))
        elif op in (OP_TRACE_P, OP_TRACE_Q):
            prog.instrs.append((op, r, imm[0], imm[1] & 0xFFFFFFFF))
        elif op in (OP_POW_P, OP_POW_Q):
            prog.instrs.append((op, r, reg[a], imm))
        elif b is None:
            prog.instrs.append((op, r, reg[a], 0))
        else:
            prog.instrs.append((op, r, reg[a], reg[b]))
        if v not in last_use:                 # dead value (cannot happen for a DAG reachable from root)
            free[typ].append(r)
    prog.instrs.append((OP_STORE_Q if qtype == FQ else OP_STORE_P, 0, reg[root], 0))
    prog.max_p, prog.max_q = nxt[FP], nxt[FQ]
    if prog.max_p > 256 or prog.max_q > 128:
        raise ValueError(f"program needs {prog.max_p} Fp and {prog.max_q} Fq registers (limits 256 / 128)")
    return prog


def periodic_lde(planner, coeffs, interval_size, domain_offset, trace_len, lde_step):
    """`eval_periodic_column` (src/eval_cpu.rs:233-256): evaluations of the column's polynomial on
    coset(interval_size * blowup, offset^(trace_len / interval_size))."""
    size = interval_size * lde_step
    off = pow(domain_offset, trace_len // interval_size, GL_P)
    a = np.zeros(size, dtype=np.uint64)
    a[: len(coeffs)] = [gl_to_mont(c) for c in coeffs]
    v = GpuVec.from_numpy(planner, a)
    f = GpuFft(Radix2EvaluationDomain(size, off), GOLDILOCKS_FP, planner)
    f.encode(v)
    f.execute()
    f.close()
    return v


def eval(prog, planner, challenges, hints, lde_step, domain_offset, n, base_cols, ext_cols=(), x_lde=None, bit_reversed=False):
    """`eval_cpu::eval(expr, challenges, hints, lde_step, domain_offset, x_lde, base, ext)`
    (src/eval_cpu.rs:33-42) -> one GpuVec of n elements of Fq.  challenges / hints: numpy u64
    limbs (Montgomery), one row per element.  bit_reversed: the columns (their first n entries) and the
    result are in bit-reversed order -- the committed LDE layout; replaces the reference's
    bit_reverse_ce_trace round trip (src/prover.rs:88-91, 126-129)."""
    qwords = FIELD_WORDS[prog.out_field]
    is252 = prog.out_field == STARK252_FP
    consts = np.array(prog.consts, dtype=np.uint64)
    for table, vals in ((prog.challenge_slots, challenges), (prog.hint_slots, hints)):
        vals = np.ascontiguousarray(vals, dtype=np.uint64).reshape(-1, qwords) if len(table) else None
        for idx, off in table.items():
            consts[off:off + qwords] = vals[idx]
    trace_len = n // lde_step
    if is252 and prog.periodic:
        raise ValueError("periodic columns are not implemented for the 252-bit field")
    per = [periodic_lde(planner, c, iv, domain_offset, trace_len, lde_step) for (c, iv) in prog.periodic]
    code = np.array(prog.instrs, dtype=np.uint32).reshape(-1, 4)
    out = GpuVec(planner, n, prog.out_field)
    L = planner.lib
    off = f252_to_mont_limbs(domain_offset) if is252 else np.array([gl_to_mont(domain_offset)], dtype=np.uint64)
    VP = ctypes.c_void_p
    base_arr = (VP * max(1, len(base_cols)))(*[c.ptr for c in base_cols])
    ext_arr = (VP * max(1, len(ext_cols)))(*[c.ptr for c in ext_cols])
    per_arr = (VP * max(1, len(per)))(*[p.ptr for p in per])
    per_len = (ctypes.c_uint * max(1, len(per)))(*[len(p) for p in per])
    L.check(L.ms_eval_program_ex(planner.handle, code.ctypes.data, len(code), consts.ctypes.data if consts.size else None, consts.size,
                                 n.bit_length() - 1, lde_step, off.ctypes.data, x_lde.ptr if x_lde is not None else None,
                                 base_arr, len(base_cols), ext_arr, len(ext_cols), per_arr, per_len, len(per),
                                 prog.out_field, out.ptr, 1 if bit_reversed else 0))
    planner.sync()
    return out
