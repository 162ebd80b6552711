#!/usr/bin/env python3
"""Timings of the other BASELINE.json configs on one MI355X (hipEvents around every launch via the
library's profiling hooks).  Parity for these shapes is asserted in tests/test_baseline_configs.py;
this script only measures.  One JSON line per config."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ministark_amd import (GOLDILOCKS_FP as FP, GOLDILOCKS_FQ3 as FQ3, STARK252_FP, GpuFft, GpuIfft, GpuVec, Matrix, MerkleTree,  # noqa: E402
                           Planner, Radix2EvaluationDomain, apply_drp)
from ministark_amd import expr as E  # noqa: E402

P = (1 << 64) - (1 << 32) + 1
pl = Planner(0)
rng = np.random.default_rng(1)


def rand(n_words):
    return rng.integers(0, P, size=n_words, dtype=np.uint64)


def timed(fn, reps=3):
    t_settle = time.perf_counter()                 # plans, specialised kernels, pool and clocks settle before timing
    while True:
        fn()
        pl.sync()
        if time.perf_counter() - t_settle > 0.4:
            break
    pl.profile(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    pl.sync()
    wall = (time.perf_counter() - t0) / reps
    prof = pl.profile_read()
    pl.profile(False)
    return wall, {k: round(v["total_us"] / reps, 1) for k, v in prof.items()}


def emit(name, wall, kernels, **kw):
    print(json.dumps({"config": name, "wall_ms": round(wall * 1e3, 3), "kernel_us": kernels, **kw}), flush=True)


# ---- C2 sweep: forward coset NTT 2^20 / 2^22 / 2^24, inverse 2^24, Fq3 2^22.  A 2^20 column is 64 tiles and the chip has 512
# workgroup slots: 4 columns per launch leave it half empty, so the small sizes are also shown at the column counts a prover
# has (32 at 2^20 = the C3 matrix, 16 at 2^22).
for log_n, field, inv, ncol in ((20, FP, False, 4), (20, FP, False, 32), (22, FP, False, 4), (22, FP, False, 16), (24, FP, False, 4),
                                (24, FP, False, 8), (24, FP, True, 4), (22, FQ3, False, 4)):
    V = 3 if field == FQ3 else 1
    n = 1 << log_n
    cols = [GpuVec.from_numpy(pl, rand(n * V), field) for _ in range(ncol)]
    plan = (GpuIfft if inv else GpuFft)(Radix2EvaluationDomain(n, 7), field, pl)
    wall, k = timed(lambda: plan.enqueue(cols), reps=5)
    alg = 2.0 * n * V * 8 * len(cols)
    emit(f"C2 {'iNTT' if inv else 'NTT'} 2^{log_n} {'Fq3' if V == 3 else 'Fp'} x{ncol} columns, coset 7", wall, k,
         us_per_column=round(wall / len(cols) * 1e6, 1), algorithmic_GBps=round(alg / wall / 1e9, 1), hbm_frac=round(alg / wall / 8e12, 4))
    del cols, plan

# ---- C3: LDE 2^20 x 32, blowup 8, + Merkle commit
cols = [rand(1 << 20) for _ in range(32)]
m = Matrix.from_numpy(pl, cols, FP)
state = {}


def c3():
    state["lde"] = m.lde(8, 7, True)
    state["tree"] = MerkleTree.from_matrix(state["lde"])
    state["tree"].root()


wall, k = timed(c3, reps=2)
lde_bytes = 32 * (8 + 64) * (1 << 20) * 1.0
emit("C3 LDE 2^20 x 32, blowup 8 (iNTT + coset NTT + bit-reverse) + SHA-256 rows + Merkle", wall, k,
     lde_algorithmic_bytes=lde_bytes, note="wall includes 32 output allocations and one 32-byte download")
lde = state["lde"]
del m

# ---- C4: constraint evaluation on 2^23 points (i) fib AIR Fp=Fq
x = E.X()
c = [lambda o=0, kk=kk: E.Trace(kk, o) for kk in range(8)]
cons = [c[0](1) - (c[6]() + c[7]()), c[1](1) - (c[7]() + c[0](1))] + [c[kk]() - (c[kk - 2]() + c[kk - 1]()) for kk in range(2, 8)]
n_trace = 1 << 21
zer = (x - E.Constant(3)) / (x ** n_trace - 1)
comp = None
for kk, cn in enumerate(cons):
    term = cn * zer * (E.Challenge(2 * kk) * x ** 3 + E.Challenge(2 * kk + 1))
    comp = term if comp is None else comp + term
prog = E.compile_expr(comp, 8, False)
ch = rand(16).reshape(-1, 1)
base = lde.columns[:8]
wall, k = timed(lambda: E.eval(prog, pl, ch, ch[:1], 4, 7, 1 << 23, base), reps=3)
emit("C4(i) fib AIR 8 Fp columns, 2^23 points, lde_step 4", wall, k, ninstr=len(prog.instrs), regs=[prog.max_p, prog.max_q],
     algorithmic_GBps=round(9 * 8 * (1 << 23) / wall / 1e9, 1))

# (ii) mixed 17 Fp + 9 Fq3 at 2^23
b = [lambda o=0, kk=kk: E.Trace(kk, o) for kk in range(17)]
e = [lambda o=0, kk=kk: E.Trace(17 + kk, o) for kk in range(9)]
expr = None
for kk in range(9):
    t = (e[kk](1) - e[kk]() * (E.Challenge(kk % 4) - b[kk]() * E.Challenge((kk + 1) % 4) - b[kk + 8](1))) * (x - 1) / (x ** 64 - 1)
    expr = t if expr is None else expr + t * E.Challenge(kk % 4)
prog2 = E.compile_expr(expr, 17, True)
ext = [GpuVec.from_numpy(pl, rand(3 << 23), FQ3) for _ in range(9)]
ch3 = rand(12).reshape(-1, 3)
wall, k = timed(lambda: E.eval(prog2, pl, ch3, ch3[:1], 2, 7, 1 << 23, lde.columns[:17], ext), reps=3)
emit("C4(ii) mixed 17 Fp + 9 Fq3 columns, 2^23 points", wall, k, ninstr=len(prog2.instrs), regs=[prog2.max_p, prog2.max_q],
     algorithmic_GBps=round((17 * 8 + 9 * 24 + 24) * (1 << 23) / wall / 1e9, 1))
del ext

# (iii) Fp252 (the reference's only 256-bit field; Fq = Fp), 8 columns, at 2^20 and at BASELINE's 2^23 points
prog3 = E.compile_expr(comp, 8, False, STARK252_FP)
ch252 = rng.integers(0, 1 << 59, size=(16, 4), dtype=np.uint64)
for lg in (20, 23):
    cols252 = [GpuVec.from_numpy(pl, rng.integers(0, 1 << 59, size=4 << lg, dtype=np.uint64), STARK252_FP) for _ in range(8)]
    wall, k = timed(lambda: E.eval(prog3, pl, ch252, ch252[:1], 4, 3, 1 << lg, cols252), reps=2)
    emit(f"C4(iii) fib AIR on Fp252 (Fq = Fp), 8 columns, 2^{lg} points", wall, k, ninstr=len(prog3.instrs),
         algorithmic_GBps=round(9 * 32 * (1 << lg) / wall / 1e9, 1))
    if lg == 23:
        del cols252
cols252 = [GpuVec.from_numpy(pl, rng.integers(0, 1 << 59, size=4 << 20, dtype=np.uint64), STARK252_FP) for _ in range(2)]
plan252 = GpuFft(Radix2EvaluationDomain(1 << 20, 3, STARK252_FP), STARK252_FP, pl)
wall, k = timed(lambda: plan252.enqueue(cols252[:2]), reps=2)
emit("Fp252 NTT 2^20 x2 columns, coset 3", wall, k, us_per_column=round(wall / 2 * 1e6, 1))
del cols252

# ---- FRI: fold a 2^23 Fq3 layer by 8 down to 2^8 (build_layers), committing each layer
layer = GpuVec.from_numpy(pl, rand(3 << 23), FQ3)
alpha = rand(3)


def fri():
    cur, n = layer, 1 << 23
    while n > 256:
        cosets = Matrix([cur])      # the reference commits rows of ff elements; here: hash the layer as n/ff rows is a re-view
        cur = apply_drp(cur, alpha, 8, 1)
        n //= 8
    pl.sync()


wall, k = timed(fri, reps=3)
emit("FRI folds 2^23 -> 2^8 by 8 (Fq3), 5 layers", wall, k, first_layer_GBps=round((24 * (1 << 23) * 9 / 8) / (k.get("fri_fold", 1) * 1e-6) / 1e9, 1))

# ---- C5-shaped pipeline on ONE GPU: every data-parallel phase of default_prove (src/prover.rs:25-174), device-resident,
# on a 2^22-row x 8-column fib-shaped trace, ProofOptions::new(32, 4, 8, 8, 64) (examples/fib/main.rs:225): 32 queries,
# blow-up 4, grinding 8 bits, FRI folding 8, remainder <= 64.  The channel (Fiat-Shamir hashing of a few digests) is
# replaced by fixed pseudo-random challenges: timings only; every phase's parity is asserted in tests/.
from ministark_amd import Queries, grind_proof_of_work   # noqa: E402
from ministark_amd.composer import DeepCompositionCoeffs, DeepPolyComposer   # noqa: E402

del lde, state
log_t, blow = 22, 4
n_t, n_lde = 1 << log_t, 1 << (log_t + 2)
trace = Matrix.from_numpy(pl, [rand(n_t) for _ in range(8)], FP)
prog_c5 = E.compile_expr(comp, 8, False)
trace_args = [(c, o) for c in range(8) for o in (0, 1)]          # every column at the current and the next row
coeffs = DeepCompositionCoeffs([int(v) for v in rand(len(trace_args))], [int(v) for v in rand(blow)], (int(rand(1)[0]), int(rand(1)[0])))
positions = [int(p) for p in rng.integers(0, n_lde, size=32)]
phase = {}


def c5():
    t = time.perf_counter()
    trace_dom, lde_dom = Radix2EvaluationDomain(n_t), Radix2EvaluationDomain(n_lde, 7)
    base_polys = trace.interpolate(trace_dom)                          # prover.rs:50
    lde_t = base_polys.bit_reversed_evaluate(lde_dom)                  # prover.rs:51
    tree_t = MerkleTree.from_matrix(lde_t); tree_t.root()              # prover.rs:52-55
    phase["base trace: interpolate + LDE + commit"] = time.perf_counter() - t; t = time.perf_counter()
    comp_evals = E.eval(prog_c5, pl, ch, ch[:1], blow, 7, n_lde, lde_t.columns, bit_reversed=True)   # prover.rs:88-107, on the committed layout
    phase["constraint evaluation"] = time.perf_counter() - t; t = time.perf_counter()
    comp_poly = Matrix([comp_evals]).bit_reverse_rows().into_polynomials(lde_dom).columns[0]   # prover.rs:111-112
    comp_polys = Matrix.from_chunks(comp_poly, blow)                   # prover.rs:113-121
    comp_lde = comp_polys.bit_reversed_evaluate(lde_dom)               # prover.rs:122
    tree_c = MerkleTree.from_matrix(comp_lde); tree_c.root()           # prover.rs:123-124
    phase["composition trace: iNTT + split + LDE + commit"] = time.perf_counter() - t; t = time.perf_counter()
    composer = DeepPolyComposer(trace_args, n_t, 0x1234567890abcdef % P, base_polys, None, comp_polys)   # prover.rs:137-144
    composer.get_ood_evals()                                           # prover.rs:145-146
    deep = Matrix([composer.into_deep_poly(coeffs)]).into_bit_reversed_evaluations(lde_dom)   # prover.rs:149-152
    phase["DEEP: OOD evaluations + composition + LDE"] = time.perf_counter() - t; t = time.perf_counter()
    cur, n = deep.columns[0], n_lde                                    # fri.rs:179-231
    alpha1 = rand(1)
    last_root = None
    while n > 64 * blow:
        last_root = MerkleTree.from_fri_layer(cur, 8).root()
        cur = apply_drp(cur, alpha1, 8, 1)
        n //= 8
    pl.sync()
    phase["FRI layers (commit + fold)"] = time.perf_counter() - t; t = time.perf_counter()
    grind_proof_of_work(pl, last_root, 8)                              # prover.rs:160 (grinding factor 8)
    Queries(lde_t, None, comp_lde, tree_t, None, tree_c, positions)    # prover.rs:163-173
    phase["proof of work + queries"] = time.perf_counter() - t


wall, k = timed(c5, reps=2)
emit("C5-shaped single-GPU pipeline: 2^22 rows x 8 cols, blow-up 4, FRI fold 8, 32 queries, 8 grinding bits (fixed challenges instead of the channel)", wall, k,
     phases_ms={kk: round(v * 1e3, 2) for kk, v in phase.items()})

# ---- C1-shaped pipeline at scale: examples/brainfuck's column mix (17 base Fp + 9 extension Fq3 columns, blow-up 16,
# FRI folding 16, examples/brainfuck/main.rs:92-105) on 2^16 rows (LDE 2^20), every data-parallel phase on the device;
# the extension columns are built on the device as running products of challenge-weighted base columns
# (examples/brainfuck/trace.rs:131-145).  Bit-exactness of this chain is asserted in tests/test_pipeline_parity.py.
from ministark_amd import running_product   # noqa: E402
from ministark_amd import stages as S       # noqa: E402

del trace
log_r, blow1, fold1 = 16, 16, 16
n_r, n_l = 1 << log_r, (1 << log_r) * 16
base1 = Matrix.from_numpy(pl, [rand(n_r) for _ in range(17)], FP)
x1 = E.X()
b1 = lambda c, o=0: E.Trace(c, o)
e1 = lambda c, o=0: E.Trace(17 + c, o)
expr1 = None
for kk in range(9):
    t1 = (e1(kk, 1) - e1(kk) * (E.Challenge(kk % 4) - b1(kk) * E.Challenge((kk + 1) % 4) - b1(kk + 8, 1))) * (x1 - 1) / (x1 ** n_r - 1)
    expr1 = t1 if expr1 is None else expr1 + t1 * E.Challenge(kk % 4)
prog1 = E.compile_expr(expr1, 17, True)
ch1 = rand(12).reshape(-1, 3)
args1 = [(c, o) for c in range(26) for o in (0, 1)]
q3 = lambda: tuple(int(v) for v in rand(3))
coeffs1 = DeepCompositionCoeffs([q3() for _ in args1], [q3() for _ in range(blow1)], (q3(), q3()))
pos1 = [int(p) for p in rng.integers(0, n_l, size=32)]
one3 = np.array([0xFFFFFFFF, 0, 0], dtype=np.uint64)          # 1 in Montgomery form
phase1 = {}


def c1():
    t = time.perf_counter()
    td, ld = Radix2EvaluationDomain(n_r), Radix2EvaluationDomain(n_l, 7)
    base_polys = base1.interpolate(td)
    base_lde = base_polys.bit_reversed_evaluate(ld)
    base_tree = MerkleTree.from_matrix(base_lde); base_tree.root()
    phase1["base trace: interpolate + LDE + commit"] = time.perf_counter() - t; t = time.perf_counter()
    ext_cols = []                                                       # trace.rs:131-145: p[i+1] = p[i] * (alpha - a*col_k[i] - b*col_(k+8)[i])
    for kk in range(9):
        f = GpuVec(pl, n_r, FQ3)
        S.ConvertIntoStage(pl, n_r, FQ3, FP).encode(f, base1.columns[kk])
        S.MulAssignConstStage(pl, n_r, FQ3, FQ3).encode(f, ch1[(kk + 1) % 4])
        g = GpuVec(pl, n_r, FQ3)
        S.ConvertIntoStage(pl, n_r, FQ3, FP).encode(g, base1.columns[kk + 8])
        S.AddAssignStage(pl, n_r, FQ3, FQ3).encode(f, g)
        S.NegInPlaceStage(pl, n_r, FQ3).encode(f)
        S.AddAssignConstStage(pl, n_r, FQ3, FQ3).encode(f, ch1[kk % 4])
        ext_cols.append(running_product(f, one3))
    ext_polys = Matrix(ext_cols).interpolate(td)
    ext_lde = ext_polys.bit_reversed_evaluate(ld)
    ext_tree = MerkleTree.from_matrix(ext_lde); ext_tree.root()
    phase1["extension trace: build (stages + scans) + LDE + commit"] = time.perf_counter() - t; t = time.perf_counter()
    comp_evals = E.eval(prog1, pl, ch1, ch1[:1], blow1, 7, n_l, base_lde.columns, ext_lde.columns, bit_reversed=True)
    phase1["constraint evaluation"] = time.perf_counter() - t; t = time.perf_counter()
    comp_poly = Matrix([comp_evals]).bit_reverse_rows().into_polynomials(ld).columns[0]
    comp_polys = Matrix.from_chunks(comp_poly, blow1)
    comp_lde = comp_polys.bit_reversed_evaluate(ld)
    comp_tree = MerkleTree.from_matrix(comp_lde); comp_tree.root()
    phase1["composition trace: iNTT + split + LDE + commit"] = time.perf_counter() - t; t = time.perf_counter()
    composer = DeepPolyComposer(args1, n_r, q3(), base_polys, ext_polys, comp_polys)
    composer.get_ood_evals()
    layer = Matrix([composer.into_deep_poly(coeffs1)]).into_bit_reversed_evaluations(ld).columns[0]
    phase1["DEEP: OOD evaluations + composition + LDE"] = time.perf_counter() - t; t = time.perf_counter()
    size, last_root = n_l, None
    while size > 64:
        last_root = MerkleTree.from_fri_layer(layer, fold1).root()
        layer = apply_drp(layer, rand(3), fold1, 1)
        size //= fold1
    pl.sync()
    phase1["FRI layers (commit + fold)"] = time.perf_counter() - t; t = time.perf_counter()
    grind_proof_of_work(pl, last_root, 8)
    Queries(base_lde, ext_lde, comp_lde, base_tree, ext_tree, comp_tree, pos1)
    phase1["proof of work + queries"] = time.perf_counter() - t


wall, k = timed(c1, reps=2)
emit("C1-shaped pipeline at scale: 2^16 rows, 17 Fp + 9 Fq3 columns (extension columns built on the device), blow-up 16, FRI fold 16, 32 queries", wall, k,
     phases_ms={kk: round(v * 1e3, 2) for kk, v in phase1.items()})
