"""FRI fold (apply_drp, src/fri.rs:526-567) vs the oracle's literal restatement
(bit_reverse -> ifft -> *ff -> alpha-combine -> fft -> bit_reverse), bit-exact, for every folding
factor the reference supports (src/fri.rs:186-192), Fp and Fq3, offset 1 (what build_layer passes)
and offset 7; plus a chain of layers down to a remainder as build_layers does (src/fri.rs:179-195)."""
import numpy as np
import pytest

from oracle import cref
from tests import backends
from ministark_amd import GOLDILOCKS_FP as FP, GOLDILOCKS_FQ3 as FQ3, GpuVec, apply_drp

KINDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


def _fold(kind, field, log_n, ff, offset, seed=1):
    pl = backends.planner(kind)
    V = 3 if field == FQ3 else 1
    ev = cref.random_elements((1 << log_n) * V, seed)
    alpha = cref.random_elements(V, seed + 100)
    got = apply_drp(GpuVec.from_numpy(pl, ev, field), alpha, ff, offset).to_numpy()
    want = cref.fri_fold(ev, log_n, V, ff, alpha, offset)
    assert np.array_equal(got, want), f"log_n={log_n} ff={ff} V={V} offset={offset}"


@pytest.mark.parametrize("ff", [2, 4, 8, 16])
@pytest.mark.parametrize("offset", [1, 7])
def test_fold_fp_emu(ff, offset):
    _fold("emu", FP, 8, ff, offset)
    _fold("emu", FP, 4, ff, offset)           # tiny layers (n = ff .. 16)


@pytest.mark.parametrize("ff", [2, 4, 8, 16])
def test_fold_fq3_emu(ff):
    _fold("emu", FQ3, 7, ff, 1)


def test_fold_large_table_emu():
    _fold("emu", FP, 14, 4, 1)                # exponent above the low table


@pytest.mark.gpu
@pytest.mark.parametrize("ff", [2, 4, 8, 16])
@pytest.mark.parametrize("field", [FP, FQ3])
def test_fold_hip(ff, field):
    _fold("hip", field, 16, ff, 1)
    _fold("hip", field, 11, ff, 7)


@pytest.mark.gpu
def test_fold_chain_hip():
    # build_layers: fold by 8 until the layer is small; each layer checked against the oracle
    pl = backends.planner("hip")
    log_n, ff = 21, 8
    ev = cref.random_elements(1 << log_n, 5)
    cur = GpuVec.from_numpy(pl, ev, FP)
    host = ev
    k = 0
    while log_n >= 9:
        alpha = cref.random_elements(1, 300 + k)
        cur = apply_drp(cur, alpha, ff, 1)
        host = cref.fri_fold(host, log_n, 1, ff, alpha, 1)
        log_n -= 3
        k += 1
        assert np.array_equal(cur.to_numpy(), host)
