#!/usr/bin/env python3
"""EXHAUSTIVE structural test of the constraint evaluator: every expression DAG with at most K operator nodes and depth <= 4
over three leaves, through compile_expr -> ms_eval_program_ex on a 64-point domain, with every rewriting pass able to fire
(MS_EVAL_SPLIT_MIN_LOG_N=6) and the library's own self-check on (MS_EVAL_SELFCHECK=1: rewritten program against the original on the plain
interpreter, every word), and every output word against the C oracle (oracle/c: eval_cpu::eval restated).  Independent of tests/fuzz_eval.py's
random generator: nothing is drawn, everything in the universe runs.

    python tests/exhaustive_eval.py --nodes 3 --ops neg,pow,add,mul,div --leaves xtc [--backend emu|hip] [--log-n 6]

leaf triples (lde_step 2, so the trace generator g = w^2):
    xtc   X, Trace(0), Constant(-g)            boundary-style denominators X - g next to trace terms
    xtn   X, Trace(0, next row), Constant(-1/g) the terminal-style root: its table is a rotation of xtc's (csrc/eval_shift.h)
    xcc   X, Constant(-g), Constant(-g^3)      x-only programs, two roots that share one inverse table
    ttc   Trace(0), Trace(1, next), Constant(3) no x at all
    xqc   X, Trace(ext column 0) in Fq3, Constant(-g)   (fq_is_ext: the Fq3 accumulators of csrc/eval_regroup.h)
The environment decides which passes run (MS_EVAL_REGROUP=force / 0, MS_EVAL_SHARE_TABLES=0, MS_EVAL_FUSE_DENOMINATORS=0,
MS_EVAL_HOST_TABLES=0): tests/test_eval_exhaustive.py runs the universe once per setting, each in its own process (the switches are read once).
Prints one summary line; exit code 1 and the offending expression on the first difference.
"""
import argparse
import os
import sys
import time

# 64-point evaluations: the oracle's thread pool would cost 100 x the work -- and eight of these processes side by side, each with eight
# spinning OpenMP threads, take minutes instead of seconds (the test suite's environment may carry an OMP_NUM_THREADS of its own)
os.environ["OMP_NUM_THREADS"] = os.environ.get("MS_EXHAUSTIVE_OMP_THREADS", "1")
os.environ.setdefault("MS_EVAL_SELFCHECK", "1")
os.environ.setdefault("MS_EVAL_SPLIT_MIN_LOG_N", "6")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.setrecursionlimit(10000)

import numpy as np  # noqa: E402

from oracle import cref  # noqa: E402  (the checker)
from ministark_amd import GOLDILOCKS_FP as FP, GOLDILOCKS_FQ3 as FQ3, GpuVec  # noqa: E402
from ministark_amd import expr as E  # noqa: E402
from ministark_amd.api import Radix2EvaluationDomain  # noqa: E402

UNARY = {"neg": lambda a: -a, "pow": lambda a: a ** 3}
BINARY = {"add": lambda a, b: a + b, "mul": lambda a, b: a * b, "div": lambda a, b: a / b}
COMMUTATIVE = {"add", "mul"}
MAX_DEPTH = 4


def universe(max_nodes, ops):
    """Every DAG: nodes 0..2 are the leaves, node k >= 3 is (op, a, b) over earlier nodes; the last node is the output and every internal
    node is used.  Two DAGs with the same tree expansion (up to the order of commutative operands) are one program.  Yields (key, nodes)."""
    un = [o for o in ops if o in UNARY]
    bi = [o for o in ops if o in BINARY]
    seen = set()

    def key_of(nodes, i, memo):
        if i in memo:
            return memo[i]
        if i < 3:
            k = "L%d" % i
        else:
            op, a, b = nodes[i]
            if b < 0:
                k = "%s(%s)" % (op, key_of(nodes, a, memo))
            else:
                ka, kb = key_of(nodes, a, memo), key_of(nodes, b, memo)
                if op in COMMUTATIVE and kb < ka:
                    ka, kb = kb, ka
                k = "%s(%s,%s)" % (op, ka, kb)
        memo[i] = k
        return k

    def rec(nodes, used, depth):
        m = len(nodes)
        k = m - 3
        if k >= 1 and all(used[3:m - 1]):
            key = key_of(nodes, m - 1, {})
            if key not in seen:
                seen.add(key)
                yield key, list(nodes)
        if k == max_nodes:
            return
        left = max_nodes - k - 1                               # nodes that can still be added after this one
        for op in un + bi:
            for a in range(m):
                for b in (range(m) if op in BINARY else (-1,)):
                    if op in COMMUTATIVE and b < a:
                        continue
                    d = 1 + max(depth[a], depth[b] if b >= 0 else 0)
                    if d > MAX_DEPTH:
                        continue
                    nu = list(used)
                    nu[a] = True
                    if b >= 0:
                        nu[b] = True
                    unused = sum(1 for i in range(3, m) if not nu[i])
                    if unused > left + 1 and unused > 2 * left:   # cannot all be consumed any more
                        continue
                    yield from rec(nodes + [(op, a, b)], nu + [False], depth + [d])
    yield from rec([None, None, None], [False] * 3, [0, 0, 0])


def build(nodes, leaves):
    vals = list(leaves)
    for op, a, b in nodes[3:]:
        vals.append(UNARY[op](vals[a]) if b < 0 else BINARY[op](vals[a], vals[b]))
    return vals[-1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=3)
    ap.add_argument("--ops", default="neg,pow,add,mul,div")
    ap.add_argument("--leaves", default="xtc")
    ap.add_argument("--backend", default="emu")
    ap.add_argument("--log-n", type=int, default=6)
    ap.add_argument("--stride", type=int, default=1, help="run every stride-th program of the universe (1 = all of it)")
    ap.add_argument("--phase", type=int, default=0)
    args = ap.parse_args()
    if args.backend == "emu":
        from tests import backends
        pl = backends.planner("emu")
    else:
        from ministark_amd import Planner
        pl = Planner(0)
    log_n, lde_step, offset = args.log_n, 2, 7
    n = 1 << log_n
    dom = Radix2EvaluationDomain(n // lde_step, 1, FP)
    P = dom.p
    g = dom.group_gen
    X = E.X()
    fq_is_ext, nbase = False, 2
    if args.leaves == "xtc":
        leaves = (X, E.Trace(0), E.Constant(P - g))
    elif args.leaves == "xtn":
        leaves = (X, E.Trace(0, 1), E.Constant(P - pow(g, P - 2, P)))
    elif args.leaves == "xcc":
        leaves = (X, E.Constant(P - g), E.Constant(P - pow(g, 3, P)))
    elif args.leaves == "ttc":
        leaves = (E.Trace(0), E.Trace(1, 1), E.Constant(3))
    elif args.leaves == "xqc":
        leaves, fq_is_ext = (X, E.Trace(nbase), E.Constant(P - g)), True
    else:
        raise SystemExit("unknown leaf triple")
    base = [cref.random_elements(n, 4100 + c) for c in range(nbase)]
    ext = [cref.random_elements(3 * n, 4200)] if fq_is_ext else []
    qw = 3 if fq_is_ext else 1
    ch = cref.random_elements(qw, 4300).reshape(-1, qw)
    dbase = [GpuVec.from_numpy(pl, c, FP) for c in base]
    dext = [GpuVec.from_numpy(pl, c, FQ3) for c in ext]
    t0, count = time.time(), 0
    for idx, (key, nodes) in enumerate(universe(args.nodes, args.ops.split(","))):
        if idx % args.stride != args.phase:
            continue
        expr = build(nodes, leaves)
        prog = E.compile_expr(expr, nbase, fq_is_ext)
        try:
            out = E.eval(prog, pl, ch, ch[:1], lde_step, offset, n, dbase, dext).to_numpy()
        except Exception as e:                                 # noqa: BLE001 -- the library's self-check reports through its error code
            print(f"FAILED program {idx}: {key} (leaves {args.leaves}): {e}")
            sys.exit(1)
        want = cref.eval_expr(expr, log_n, lde_step, offset, base, ext, ch, ch[:1], fq_is_ext)
        if not np.array_equal(out, want):
            bad = np.nonzero(out != want)[0]
            print(f"MISMATCH against the oracle, program {idx}: {key} (leaves {args.leaves}): {bad.size} words differ, first at {bad[:4]}")
            sys.exit(1)
        count += 1
    sw = {k: os.environ[k] for k in ("MS_EVAL_REGROUP", "MS_EVAL_SHARE_TABLES", "MS_EVAL_FUSE_DENOMINATORS", "MS_EVAL_HOST_TABLES", "MS_EVAL_JIT") if k in os.environ}
    print(f"exhaustive_eval ok: {count} programs (<= {args.nodes} operator nodes over {args.ops}, leaves {args.leaves}, 2^{log_n} points, {args.backend}, "
          f"self-check {os.environ['MS_EVAL_SELFCHECK']}, switches {sw or 'default'}) in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
