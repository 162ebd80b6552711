"""Extension-column scans, query-row gathers and batched Merkle openings
against the oracle's sequential restatements, bit-exact.  Shapes follow the reference: running
products with masked (padding) rows and running evaluations state*gamma + value
(examples/brainfuck/trace.rs:108-289); Queries::new (src/trace.rs:113-157); MerkleTreeImpl::prove /
verify and their tests (src/merkle.rs:149-287, 519-580: all leaves, single leaf, sibling pairs)."""
import numpy as np
import pytest

from oracle import cref
from oracle.pyref import merkle as omerkle
from oracle.pyref import scan as oscan
from oracle.pyref.fields import GL, FQ3
from tests import backends
from ministark_amd import GOLDILOCKS_FP as FP, GOLDILOCKS_FQ3 as FQ3F, GpuVec, Matrix, MerkleTree, Queries, scan_affine, running_product

KINDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


def _canon(arr, ext):
    a = [GL.from_mont(int(x)) for x in arr]
    return [tuple(a[3 * i:3 * i + 3]) for i in range(len(a) // 3)] if ext else a


def _mont(vals, ext):
    flat = [c for v in vals for c in v] if ext else vals
    return np.array([GL.to_mont(c) for c in flat], dtype=np.uint64)


def _scan_case(kind, n, ext, has_a, has_b, inclusive, seed, mask_every=3):
    pl = backends.planner(kind)
    V = 3 if ext else 1
    a = cref.random_elements(n * V, seed) if has_a else None
    b = cref.random_elements(n * V, seed + 1) if has_b else None
    one = _mont([FQ3.one()] if ext else [1], ext)
    # padding / non-matching rows leave the state unchanged: a = 1, b = 0 (trace.rs:134, :150-158)
    for i in range(0, n, mask_every):
        if a is not None:
            a[V * i:V * i + V] = one
        if b is not None:
            b[V * i:V * i + V] = 0
    init = cref.random_elements(V, seed + 2)
    da = GpuVec.from_numpy(pl, a, FQ3F if ext else FP) if has_a else None
    db = GpuVec.from_numpy(pl, b, FQ3F if ext else FP) if has_b else None
    got = _canon(scan_affine(da, db, init, inclusive).to_numpy(), ext)
    want = oscan.scan_affine(_canon(a, ext) if has_a else None, _canon(b, ext) if has_b else None,
                             _canon(init, ext)[0], n, ext, inclusive)
    assert got == want


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("n", [1, 5, 4096, 4097, 3 * 4096 + 17])
@pytest.mark.parametrize("ext", [False, True])
def test_running_product_masked(kind, n, ext):        # permutation columns, trace.rs:131-145
    _scan_case(kind, n, ext, True, False, False, seed=n)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("ext", [False, True])
@pytest.mark.parametrize("inclusive", [False, True])
def test_running_evaluation(kind, ext, inclusive):     # input / output evaluation columns, trace.rs:147-159
    _scan_case(kind, 9000, ext, True, True, inclusive, seed=7)


@pytest.mark.parametrize("kind", KINDS)
def test_running_sum_and_in_place(kind):
    _scan_case(kind, 5000, True, False, True, True, seed=11)
    pl = backends.planner(kind)
    n = 6000
    a = cref.random_elements(n, 3)
    da = GpuVec.from_numpy(pl, a, FP)
    init = cref.random_elements(1, 4)
    L = pl.lib
    L.check(L.ms_scan_affine(pl.handle, FP, n, da.ptr, None, init.ctypes.data, 0, da.ptr))      # out aliases a
    assert _canon(da.to_numpy(), False) == oscan.scan_affine(_canon(a, False), None, _canon(init, False)[0], n)


def test_scan_rejects_bad_arguments_emu():
    pl = backends.planner("emu")
    L = pl.lib
    v = GpuVec(pl, 8, FP)
    init = np.zeros(1, dtype=np.uint64)
    assert L.ms_scan_affine(pl.handle, FP, 8, None, None, init.ctypes.data, 0, v.ptr) == -1
    assert L.ms_scan_affine(pl.handle, 7, 8, v.ptr, None, init.ctypes.data, 0, v.ptr) == -1      # unknown field
    assert L.ms_scan_affine(pl.handle, FP, 0, v.ptr, None, init.ctypes.data, 0, v.ptr) == 0


@pytest.mark.parametrize("kind", KINDS)
def test_long_fp_column_takes_sixteen_rows_per_lane(kind):
    # from 2^20 rows an Fp scan composes 16 consecutive rows per lane (the block's 4096 rows cross LDS once, coalesced both ways);
    # ragged length, multipliers and addends, exclusive and inclusive
    n = (1 << 20) + 4097
    for inclusive in ((False, True) if kind == "hip" else (False,)):
        _scan_case(kind, n, False, True, True, inclusive, seed=31 + inclusive, mask_every=5)


@pytest.mark.gpu
def test_running_product_2_22_hip():
    pl = backends.planner("hip")
    n = 1 << 22
    a = cref.random_elements(3 * n, 21)
    init = cref.random_elements(3, 22)
    out = running_product(GpuVec.from_numpy(pl, a, FQ3F), init).to_numpy()
    # the sequential loop on the last 70 000 rows, seeded with the device's own state there, plus the
    # total product through a pairwise (different-order) reduction of the whole column
    ca = _canon(a, True)
    start = n - 70000
    want = oscan.scan_affine(ca[start:], None, _canon(out[3 * start:3 * start + 3], True)[0], 70000, True)
    assert _canon(out[3 * start:], True) == want
    prod = ca[:]
    while len(prod) > 1:
        prod = [FQ3.mul(prod[2 * i], prod[2 * i + 1]) for i in range(len(prod) // 2)]
    total = FQ3.mul(_canon(init, True)[0], prod[0])
    last = FQ3.mul(_canon(out[3 * (n - 1):], True)[0], ca[n - 1])
    assert last == total


# ---- queries -------------------------------------------------------------------------------------
def _tree(kind, log_n, ncols, ext=False, seed=5):
    pl = backends.planner(kind)
    n = 1 << log_n
    V = 3 if ext else 1
    cols = [cref.random_elements(n * V, seed + c) for c in range(ncols)]
    m = Matrix.from_numpy(pl, cols, FQ3F if ext else FP)
    tree = MerkleTree.from_matrix(m)
    field = FQ3 if ext else GL
    canon = [_canon(c, ext) for c in cols]
    leaves = omerkle.hash_rows(field, canon)
    nodes = omerkle.build_merkle_nodes(leaves)
    return m, tree, cols, leaves, nodes


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("indices", [[0, 1, 2, 3, 4, 5, 6, 7], [3], [0, 7], [2, 3], [5, 4, 5, 1], [6, 1, 0]])
def test_merkle_openings_small(kind, indices):          # src/merkle.rs:519-580
    m, tree, cols, leaves, nodes = _tree(kind, 3, 2)
    got = tree.prove(indices)
    want = omerkle.prove(leaves, nodes, indices)
    assert got == want
    assert omerkle.verify(nodes[1], got, indices)
    assert tree.root() == nodes[1]


@pytest.mark.parametrize("kind", KINDS)
def test_queries_rows_and_openings(kind):               # src/trace.rs:113-157
    rng = np.random.default_rng(9)
    mb, tb, cb, lb, nb = _tree(kind, 10, 5, False, seed=30)
    me, te, ce, le, ne = _tree(kind, 10, 3, True, seed=40)
    mc, tc, cc, lc, nc = _tree(kind, 10, 2, True, seed=50)
    positions = [int(p) for p in rng.integers(0, 1 << 10, size=40)] + [0, 1023, 1022]
    q = Queries(mb, me, mc, tb, te, tc, positions)
    for rows, cols, V in ((q.base_trace_values, cb, 1), (q.extension_trace_values, ce, 3), (q.composition_trace_values, cc, 3)):
        assert rows.shape == (len(positions), len(cols) * V)
        for k, p in enumerate(positions):
            assert list(rows[k]) == [int(x) for c in cols for x in c[V * p:V * p + V]]
    for proof, leaves, nodes in ((q.base_trace_proof, lb, nb), (q.extension_trace_proof, le, ne), (q.composition_trace_proof, lc, nc)):
        assert proof == omerkle.prove(leaves, nodes, positions)
        assert omerkle.verify(nodes[1], proof, positions)
    # a tampered opening must not verify
    bad = dict(q.base_trace_proof)
    bad["nodes"] = [bytes(32)] + bad["nodes"][1:]
    assert not omerkle.verify(nb[1], bad, positions)
    with pytest.raises(IndexError):
        tb.prove([1 << 10])


@pytest.mark.parametrize("kind", KINDS)
def test_running_evaluation_252(kind):
    from oracle.pyref.fields import F252
    from ministark_amd import STARK252_FP, f252_to_mont_limbs, f252_from_mont_limbs
    pl = backends.planner(kind)
    n = 4500
    rng = np.random.default_rng(8)
    draw = lambda: [int.from_bytes(rng.bytes(32), "little") % F252.p for _ in range(n)]
    a, b, init = draw(), draw(), int.from_bytes(rng.bytes(32), "little") % F252.p
    for i in range(0, n, 5):
        a[i], b[i] = 1, 0
    dev = lambda v: GpuVec.from_numpy(pl, np.concatenate([f252_to_mont_limbs(x) for x in v]), STARK252_FP)
    got = scan_affine(dev(a), dev(b), f252_to_mont_limbs(init)).to_numpy().reshape(n, 4)
    state = init
    for i in range(n):
        assert f252_from_mont_limbs(got[i]) == state, f"row {i}"
        state = (a[i] * state + b[i]) % F252.p


def test_long_column_rows_per_lane_emu():
    # columns of >= 2^20 rows take 16 rows per lane (4 below): both instantiations must agree with the loop
    _scan_case("emu", (1 << 20) + 5, False, True, False, False, seed=5, mask_every=7)


# The gathers read their index lists from the library's pinned staging ring (1 MiB): enough launches to wrap it several times without
# any download in between, then a list too long for a ring slot (pooled device copy instead), every result checked.
@pytest.mark.parametrize("kind", KINDS)
def test_index_lists_outlive_the_staging_ring(kind):
    from ministark_amd.api import GatherBatch
    pl = backends.planner(kind)
    n, ncols = 1 << 12, 3
    cols = [cref.random_elements(n, 900 + c) for c in range(ncols)]
    m = Matrix([GpuVec.from_numpy(pl, c, FP) for c in cols])
    rng = np.random.default_rng(5)
    batch = GatherBatch(pl, capacity=8 << 20)
    pending = []
    for _ in range(700):                                 # 700 lists of 256 positions = 1.4 MiB of indices, 4.2 MiB of rows
        pos = rng.integers(0, n, size=256).tolist()
        pending.append((pos, m.get_rows_launch(pos, batch)))
    for pos, fetch in pending:
        got = np.asarray(fetch()).reshape(len(pos), ncols)
        for c in range(ncols):
            assert np.array_equal(got[:, c], cols[c][pos])
    long_pos = rng.integers(0, n, size=40000).tolist()   # 320 KB of indices: more than a slot of the ring
    got = np.asarray(m.get_rows(long_pos)).reshape(len(long_pos), ncols)
    for c in range(ncols):
        assert np.array_equal(got[:, c], cols[c][long_pos])


@pytest.mark.parametrize("kind", KINDS)
def test_gather_digests_multi_equals_the_single_gathers(kind):
    """ms_gather_digests_multi: the digest gathers of several trees in ONE launch (what GatherBatch.flush issues for a proof's openings) give the
    bytes of one ms_gather_digests per tree; empty segments are allowed; an index past its own segment's array is refused before anything runs."""
    import ctypes
    from ministark_amd.api import DeviceBytes
    pl = backends.planner(kind)
    L = pl.lib
    rng = np.random.default_rng(77)
    sizes = [1 << 10, 1 << 4, 1 << 13, 2]
    arrays = [rng.integers(0, 256, size=32 * n, dtype=np.uint8) for n in sizes]
    devs = [DeviceBytes.from_numpy(pl, a) if hasattr(DeviceBytes, "from_numpy") else None for a in arrays]
    if devs[0] is None:
        devs = []
        for a in arrays:
            d = DeviceBytes(pl, a.size)
            L.check(L.ms_upload(pl.handle, d.ptr, a.ctypes.data, a.size))
            devs.append(d)
    idx = [rng.integers(0, n, size=k, dtype=np.uint64) for n, k in zip(sizes, (37, 0, 100, 5))]
    outs = [DeviceBytes(pl, max(32, 32 * len(i))) for i in idx]
    VP, SZ = ctypes.c_void_p, ctypes.c_size_t
    n = len(sizes)
    allidx = np.ascontiguousarray(np.concatenate(idx), dtype=np.uint64)
    L.check(L.ms_gather_digests_multi(pl.handle, n, (VP * n)(*[d.ptr for d in devs]), (SZ * n)(*sizes), allidx.ctypes.data,
                                      (SZ * n)(*[len(i) for i in idx]), (VP * n)(*[o.ptr for o in outs])))
    pl.sync()
    for a, i, o in zip(arrays, idx, outs):
        want = a.reshape(-1, 32)[i.astype(np.int64)].reshape(-1)
        assert np.array_equal(o.to_numpy(32 * len(i)), want)
    bad = allidx.copy()
    bad[37] = sizes[2]                                            # first index of the third segment (the second is empty): one past its array
    assert L.ms_gather_digests_multi(pl.handle, n, (VP * n)(*[d.ptr for d in devs]), (SZ * n)(*sizes), bad.ctypes.data,
                                     (SZ * n)(*[len(i) for i in idx]), (VP * n)(*[o.ptr for o in outs])) != 0
    assert b"out of range" in L.ms_last_error()
