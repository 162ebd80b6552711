"""BASELINE.json configs C3 / C4 at full size on the GPU, checked against the oracle.
(C2 is tests/test_ntt_parity.py; C5's exchange step is tests/test_distributed.py.)"""
import numpy as np
import pytest

from oracle import cref
from oracle.pyref.fields import GL
from tests import backends
from ministark_amd import GOLDILOCKS_FP as FP, GOLDILOCKS_FQ3 as FQ3, GpuVec, Matrix, MerkleTree
from ministark_amd import expr as E

P = cref.GL_P


@pytest.mark.gpu
def test_c3_lde_2_20_x_32_blowup_8_and_commit():
    # "Trace LDE: 2^20 rows x 32 columns, blowup 8, coset NTT + Merkle commit on 1 MI355X"
    pl = backends.planner("hip")
    log_n, log_b, ncols = 20, 3, 32
    cols = [cref.random_elements(1 << log_n, 0x6D696E69 + c) for c in range(ncols)]
    lde = Matrix.from_numpy(pl, cols, FP).lde(1 << log_b, 7, True)
    tree = MerkleTree.from_matrix(lde)
    root = tree.root()
    want_cols = [cref.lde(c, log_n, log_b, 1, 7, True) for c in cols]
    # 1000 sampled elements of the LDE, then the whole thing through the root
    rng = np.random.default_rng(1)
    got0 = lde.columns[0].to_numpy()
    got31 = lde.columns[31].to_numpy()
    idx = rng.integers(0, 1 << (log_n + log_b), size=1000)
    assert np.array_equal(got0[idx], want_cols[0][idx]) and np.array_equal(got31[idx], want_cols[31][idx])
    want_root = cref.sha256_merkle(cref.sha256_rows(want_cols, 1))[1].tobytes()
    assert root == want_root


def _full_eval(pl, expr, log_n, lde_step, base, ext, ch, fq_is_ext, hints=None):
    """ALL 2^log_n outputs of the device evaluator against the C restatement of eval_cpu::eval
    (oracle_eval_expr: 512-point chunks, batch inversion), Montgomery words, bit for bit."""
    n = 1 << log_n
    hints = ch[:1] if hints is None else hints
    prog = E.compile_expr(expr, len(base), fq_is_ext)
    got = E.eval(prog, pl, ch, hints, lde_step, 7, n, [GpuVec.from_numpy(pl, c, FP) for c in base],
                 [GpuVec.from_numpy(pl, c, FQ3) for c in ext]).to_numpy()
    want = cref.eval_expr(expr, log_n, lde_step, 7, base, ext, ch, hints, fq_is_ext)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, f"{bad.size} of {want.size} words differ, first at {bad[:4]}"


@pytest.mark.gpu
def test_c4_fib_air_2_23():
    # "Constraint composition eval on 2^23-row synthetic AIR" (i): the reference's fib AIR exactly -- FibAirConfig::constraints
    # (examples/fib/main.rs:73-140: 8 boundary / (X - 1), 1 terminal / (X - g^-1), 8 multiplicative transitions
    # (X - g^-1) / (X^n - 1)) under AirConfig::composition_constraint (src/air.rs:50-82) -- on 8 random Fp columns, Fq = Fp
    from ministark_amd import pipeline
    pl = backends.planner("hip")
    log_n = 23
    base = [cref.random_elements(1 << log_n, 100 + k) for k in range(8)]
    for lde_step in (1, 4):       # 1 = its ce_blowup_factor, what src/prover.rs:103 passes for this AIR; 4 exercises strided rotations
        comp, ce, nch = pipeline.fib_constraints((1 << log_n) // lde_step)
        assert ce == 1 and nch == 34
        ch = cref.random_elements(nch, 5).reshape(-1, 1)
        _full_eval(pl, comp, log_n, lde_step, base, [], ch, False, hints=cref.random_elements(1, 6).reshape(-1, 1))


@pytest.mark.gpu
def test_c4_additive_air_2_23():
    # a second, cheaper Fp shape (the stand-in of rounds 1-2): 8 additive transitions, one zerofier, degree adjustment X^3
    from ministark_amd import pipeline
    pl = backends.planner("hip")
    log_n, lde_step = 23, 4
    comp, _, nch = pipeline.additive_constraints((1 << log_n) // lde_step, 8, lde_step)
    base = [cref.random_elements(1 << log_n, 100 + k) for k in range(8)]
    ch = cref.random_elements(nch, 5).reshape(-1, 1)
    _full_eval(pl, comp, log_n, lde_step, base, [], ch, False)


@pytest.mark.gpu
def test_c4_mixed_17_fp_9_fq3_2_23():
    # (ii): the brainfuck shape, 17 Fp + 9 Fq3 columns (examples/brainfuck/air.rs:26-27), at BASELINE's 2^23 points
    pl = backends.planner("hip")
    from ministark_amd import pipeline
    log_n, lde_step = 23, 2
    expr, _ = pipeline.mixed_air_constraints()
    base = [cref.random_elements(1 << log_n, 200 + k) for k in range(17)]
    ext = [cref.random_elements(3 << log_n, 300 + k) for k in range(9)]
    ch = cref.random_elements(12, 6).reshape(-1, 3)
    _full_eval(pl, expr, log_n, lde_step, base, ext, ch, True)


@pytest.mark.gpu
def test_c4_fib_air_on_the_256_bit_field_2_23():
    # (iii): "256-bit Fq" = the reference's only 256-bit field, Fp252 with Fq = Fp (src/eval_gpu.rs:1054-1082), 8 columns
    from ministark_amd import STARK252_FP
    pl = backends.planner("hip")
    from ministark_amd import pipeline
    log_n, lde_step = 23, 4
    comp, ce, nch = pipeline.fib_constraints(1 << (log_n - 2), 8, STARK252_FP)        # FibAirConfig::constraints over the 252-bit field
    assert ce == 1 and nch == 34
    rng = np.random.default_rng(252)
    # uniformly random canonical residues below 2^251 (< p) as the stored Montgomery words
    cols = [rng.integers(0, 1 << 63, size=4 << log_n, dtype=np.uint64) for _ in range(8)]
    for col in cols:
        col[3::4] >>= np.uint64(4)
    ch = rng.integers(0, 1 << 59, size=(34, 4), dtype=np.uint64)
    prog = E.compile_expr(comp, 8, False, STARK252_FP)
    got = E.eval(prog, pl, ch, ch[:1], lde_step, 3, 1 << log_n, [GpuVec.from_numpy(pl, col, STARK252_FP) for col in cols]).to_numpy()
    want = cref.eval_expr(comp, log_n, lde_step, 3, cols, [], ch, ch[:1], False, field="f252")
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, f"{bad.size} of {want.size} words differ, first at {bad[:4]}"
