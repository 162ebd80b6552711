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

# ---- C4: constraint evaluation on 2^23 points: the same three AIRs as bench.py's `constraint_eval` object (ministark_amd/pipeline.py)
from ministark_amd import pipeline   # noqa: E402
N23 = 1 << 23
for lde_step in (1, 4):           # (i) the reference's fib AIR (examples/fib/main.rs:73-140); 1 = its ce_blowup_factor
    comp, _, nch = pipeline.fib_constraints(N23 // lde_step)
    prog = E.compile_expr(comp, 8, False)
    ch = rand(nch).reshape(-1, 1)
    base = lde.columns[:8]
    wall, k = timed(lambda: E.eval(prog, pl, ch, ch[:1], lde_step, 7, N23, base), reps=3)
    emit(f"C4(i) fib AIR (FibAirConfig::constraints, 17 constraints), 8 Fp columns, 2^23 points, lde_step {lde_step}", wall, k, ninstr=len(prog.instrs),
         regs=[prog.max_p, prog.max_q], algorithmic_GBps=round(9 * 8 * N23 / (sum(k.values()) * 1e-6) / 1e9, 1))
expr2, nch2 = pipeline.mixed_air_constraints()                       # (ii) 17 Fp + 9 Fq3
prog2 = E.compile_expr(expr2, 17, True)
ext = [GpuVec.from_numpy(pl, rand(3 << 23), FQ3) for _ in range(9)]
ch3 = rand(3 * nch2).reshape(-1, 3)
wall, k = timed(lambda: E.eval(prog2, pl, ch3, ch3[:1], 2, 7, N23, lde.columns[:17], ext), reps=3)
emit("C4(ii) mixed 17 Fp + 9 Fq3 columns, 2^23 points", wall, k, ninstr=len(prog2.instrs), regs=[prog2.max_p, prog2.max_q],
     algorithmic_GBps=round((17 * 8 + 9 * 24 + 24) * N23 / (sum(k.values()) * 1e-6) / 1e9, 1))
del ext
for lg in (20, 23):               # (iii) the fib AIR over the 252-bit field, at 2^20 and at BASELINE's 2^23 points
    comp3, _, nch3 = pipeline.fib_constraints((1 << lg) // 4, 8, STARK252_FP)
    prog3 = E.compile_expr(comp3, 8, False, STARK252_FP)
    ch252 = rng.integers(0, 1 << 59, size=(nch3, 4), dtype=np.uint64)
    cols252 = [GpuVec.from_numpy(pl, rng.integers(0, 1 << 59, size=4 << lg, dtype=np.uint64), STARK252_FP) for _ in range(8)]
    wall, k = timed(lambda: E.eval(prog3, pl, ch252, ch252[:1], 4, 3, 1 << lg, cols252), reps=2)
    emit(f"C4(iii) fib AIR on Fp252 (Fq = Fp), 8 columns, 2^{lg} points", wall, k, ninstr=len(prog3.instrs),
         algorithmic_GBps=round(9 * 32 * (1 << lg) / (sum(k.values()) * 1e-6) / 1e9, 1))
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

# ---- C5 on ONE GPU: ministark_amd/pipeline.py (every data-parallel phase of default_prove, src/prover.rs:25-174) on a
# 2^22-row x 8-column trace with the reference's fib AIR, ProofOptions::new(32, 4, 8, 8, 64) (examples/fib/main.rs:225); the
# channel is replaced by fixed draws: timings only, every phase's parity is asserted in tests/test_pipeline_parity.py.
from ministark_amd.composer import DeepCompositionCoeffs, DeepPolyComposer   # noqa: E402
from ministark_amd import Queries, grind_proof_of_work   # noqa: E402

del lde, state
log_t, blow = 22, 4
n_t = 1 << log_t
trace = Matrix.from_numpy(pl, [rand(n_t) for _ in range(8)], FP)
comp5, ce5, nch5 = pipeline.fib_constraints(n_t)
draws5 = pipeline.Draws(0xC5, 8, nch5, ce5, 32, n_t * blow, pipeline.fri_num_layers(n_t * blow, blow, 8, 64))
res5 = {}


def c5():
    res5.update(pipeline.prove_phases(pl, trace, comp5, draws5, blow, 8, 64, 8, ce_blowup=ce5))


wall, k = timed(c5, reps=2)
emit("C5 on one GPU: 2^22 rows x 8 cols, fib AIR, blow-up 4, FRI fold 8, 32 queries, 8 grinding bits (fixed draws instead of the channel)", wall, k,
     phases_ms=res5["phases_ms"])

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


# ---- SURVEY.md 8(f), the "next" rows, each on its own at the C5 shape (2^22 rows x 8 columns, blow-up 4) ---------------------------
# Bounds: DEEP / scans / gathers are streaming (HBM, algorithmic bytes = every input column once + the output once); RPO-256,
# SHA-256 and the proof-of-work search are integer-ALU work (rates in hashes per second, no byte roofline).
HBM = 8000.0
n5, N5 = 1 << 22, 1 << 24
polys5 = Matrix.from_numpy(pl, [rand(n5) for _ in range(8)], FP)
comp5 = Matrix.from_numpy(pl, [rand(n5)], FP)
args5 = [(c, o) for c in range(8) for o in (0, 1)]
r1 = lambda: int(rng.integers(1, P, dtype=np.uint64))
coeffs5 = DeepCompositionCoeffs([r1() for _ in args5], [r1()], (r1(), r1()))
z5 = r1()


def deep5():
    c = DeepPolyComposer(args5, n5, z5, polys5, None, comp5)
    c.get_ood_evals()
    c.into_deep_poly(coeffs5)


wall, k = timed(deep5)
alg = (8 + 1) * n5 * 8 * 2 + n5 * 8                       # Horner pass over the 9 polynomials, composition pass over them again, one output
emit("f1 DEEP composition: 17 out-of-domain evaluations (Horner) + into_deep_poly, 8 + 1 polynomials of 2^22 coefficients", wall, k,
     algorithmic_bytes=alg, algorithmic_GBps=round(alg / (sum(k.values()) * 1e-6) / 1e9, 1), hbm_frac=round(alg / (sum(k.values()) * 1e-6) / 1e9 / HBM, 4),
     bound="hbm (streams every coefficient twice; 16 Horner points per base column share one pass)")

rows5 = Matrix.from_numpy(pl, [rand(1 << 20) for _ in range(8)], FP)
for h in ("sha256", "rpo256"):
    wall, k = timed(lambda: MerkleTree.from_matrix(rows5, h).root())
    hashes = (1 << 20) + (1 << 20) - 1                    # one leaf per row (8 elements = one rate block / one 64-byte message) + the tree
    emit(f"f2 commitment of 2^20 rows x 8 Fp columns with {h}: rows + Merkle tree", wall, k,
         hashes_per_s=round(hashes / (sum(k.values()) * 1e-6) / 1e9, 2), unit="G leaf-or-node hashes / s",
         bound="integer ALU (SHA-256: 2 compressions per leaf and per node; RPO-256: one 7-round permutation of 12 Goldilocks elements each, x^7 and x^(1/7) S-boxes)")

seed5 = bytes(range(32))
for bits in (16, 20):
    t0 = time.perf_counter()
    nonce = grind_proof_of_work(pl, seed5, bits)
    dt = time.perf_counter() - t0
    emit(f"f3 proof-of-work grinding, {bits} bits (src/random.rs:48-58)", dt, {}, nonce=nonce, tries_per_s=round(nonce / dt / 1e9, 3), unit="G SHA-256(seed || nonce) / s incl. launch and read-back",
         bound="integer ALU; latency-bound below ~2^24 tries (one launch wave finds the nonce)")

for field, V, name in ((FP, 1, "Fp"), (FQ3, 3, "Fq3")):
    fac = GpuVec.from_numpy(pl, rand(n5 * V), field)
    one = np.array([0xFFFFFFFF] + [0] * (V - 1), dtype=np.uint64)          # Montgomery 1
    wall, k = timed(lambda: running_product(fac, one))
    alg = 2 * n5 * V * 8
    emit(f"f4 running product over 2^22 {name} elements (extension-column scan, examples/brainfuck/trace.rs:131-145)", wall, k,
         algorithmic_bytes=alg, algorithmic_GBps=round(alg / (sum(k.values()) * 1e-6) / 1e9, 1), hbm_frac=round(alg / (sum(k.values()) * 1e-6) / 1e9 / HBM, 4),
         bound="hbm for the two streaming passes + a serial walk over the block aggregates (latency)" if V == 1 else
               "integer ALU: three to four Fq3 products (nine Goldilocks products each) per element")

lde5 = Matrix.from_numpy(pl, [rand(N5) for _ in range(8)], FP)
tree5 = MerkleTree.from_matrix(lde5)
clde5 = Matrix.from_numpy(pl, [rand(N5)], FP)
ctree5 = MerkleTree.from_matrix(clde5)
pos5 = [int(v) for v in rng.integers(0, N5, size=32)]
wall, k = timed(lambda: Queries(lde5, None, clde5, tree5, None, ctree5, pos5))
emit("f4 query extraction: 32 positions, rows of the 8-column and the 1-column LDE (2^24 rows) + both Merkle openings", wall, k,
     bound="latency (a few hundred 32-byte gathers and two host round trips)")
