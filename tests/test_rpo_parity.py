"""RPO-256 (a16): device vs the oracle's straightforward restatement, bit-exact, plus what the
reference itself asserts (gpu/tests/rpo.rs:88-92: identical rows give identical digests) and the
algebraic facts that can be checked without external vectors."""
import numpy as np
import pytest

from oracle import cref
from oracle.pyref import rpo as pyrpo
from oracle.pyref.fields import GL
from tests import backends
from ministark_amd import GpuRpo256ColumnMajor, GpuRpo256RowMajor, GpuVec, gen_rpo_merkle_tree

KINDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
P = GL.p


def test_rpo_parameters():
    # MDS first row de-Montgomerised (hash_shaders.h.metal:41-54), S-box exponents inverse mod p-1
    assert [GL.from_mont(x) for x in (30064771065, 98784247785, 34359738360, 111669149670)] == [7, 23, 8, 26]
    assert (7 * pyrpo.INV7) % (P - 1) == 1
    x = 0x123456789ABCDEF % P
    assert pow(pow(x, 7, P), pyrpo.INV7, P) == x
    assert len(pyrpo.RC0) == 84 and len(pyrpo.RC1) == 84 and all(c < P for c in pyrpo.RC0_MONT + pyrpo.RC1_MONT)


def _canon(a):
    return [GL.from_mont(int(x)) for x in a]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("ncols", [1, 7, 8, 9, 16, 17])
def test_rows_column_major(kind, ncols):
    pl = backends.planner(kind)
    n = 64 if kind == "hip" else 6
    cols = [cref.random_elements(n, 40 + c) for c in range(ncols)]
    h = GpuRpo256ColumnMajor(n, ncols % 8 != 0, pl)
    for c in cols:
        h.update(GpuVec.from_numpy(pl, c))
    got = h.finish().to_numpy().reshape(n, 4)
    for r in range(n):
        want = pyrpo.hash_row([GL.from_mont(int(c[r])) for c in cols])
        assert _canon(got[r]) == want


@pytest.mark.parametrize("kind", KINDS)
def test_rows_row_major_and_merkle(kind):
    pl = backends.planner(kind)
    n = 32 if kind == "hip" else 8
    rows = cref.random_elements(n * 8, 7)
    h = GpuRpo256RowMajor(n, False, pl)
    h.update(GpuVec.from_numpy(pl, rows))
    leaves = h.finish()
    got = leaves.to_numpy().reshape(n, 4)
    want_leaves = [pyrpo.hash_row(_canon(rows[8 * r:8 * r + 8])) for r in range(n)]
    assert [_canon(g) for g in got] == want_leaves
    nodes = gen_rpo_merkle_tree(leaves).to_numpy().reshape(n, 4)
    want_nodes = pyrpo.merkle_nodes(want_leaves)
    assert [_canon(g) for g in nodes] == want_nodes


@pytest.mark.gpu
def test_identical_rows_identical_digests_hip():       # gpu/tests/rpo.rs:62-92 (all-ones rows)
    pl = backends.planner("hip")
    n = 1 << 16
    ones = np.full(n * 8, 4294967295, dtype=np.uint64)
    h = GpuRpo256RowMajor(n, False, pl)
    h.update(GpuVec.from_numpy(pl, ones))
    d = h.finish().to_numpy().reshape(n, 4)
    assert (d == d[0]).all()
    assert _canon(d[0]) == pyrpo.hash_row([1] * 8)


# ---- external known-answer vectors (tests/golden/rpo256_miden_kat.json) -----------------------------
def _kat():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rpo256_miden_kat.json")))


def test_oracle_matches_external_kat():
    """Pins the oracle (restated from hash_shaders.h.metal) to vectors that do not come from the reference tree."""
    k = _kat()
    assert pyrpo.RC0[:12] == k["ark1_first_row"] and pyrpo.RC1[:12] == k["ark2_first_row"]
    assert pyrpo.MDS_ROW == k["mds_first_row"]
    for v in k["hash_elements"]:
        assert pyrpo.hash_row(v["input"]) == v["digest"]


@pytest.mark.parametrize("kind", KINDS)
def test_device_matches_external_kat(kind):
    pl = backends.planner(kind)
    for v in _kat()["hash_elements"]:
        cols = [np.array([GL.to_mont(x)] * 4, dtype=np.uint64) for x in v["input"]]    # 4 identical rows
        h = GpuRpo256ColumnMajor(4, len(cols) % 8 != 0, pl)
        for c in cols:
            h.update(GpuVec.from_numpy(pl, c))
        got = h.finish().to_numpy().reshape(4, 4)
        for r in range(4):
            assert _canon(got[r]) == v["digest"]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("field_name", ["fp", "fq3"])
def test_matrix_merkle_tree_over_rpo(kind, field_name):
    """MerkleTree.from_matrix(matrix, hash="rpo256") (f2): leaves, every node, the root and a batched opening
    against the oracle; Fq3 columns absorb c0, c1, c2 in serialisation order."""
    from ministark_amd import GOLDILOCKS_FP, GOLDILOCKS_FQ3, Matrix, MerkleTree
    pl = backends.planner(kind)
    field, V = (GOLDILOCKS_FP, 1) if field_name == "fp" else (GOLDILOCKS_FQ3, 3)
    n, ncols = (256, 5) if kind == "hip" else (16, 3)
    cols = [cref.random_elements(n * V, 900 + c) for c in range(ncols)]
    m = Matrix.from_numpy(pl, cols, field)
    tree = MerkleTree.from_matrix(m, hash="rpo256")
    want_leaves = []
    for r in range(n):
        row = []
        for c in cols:
            row += [GL.from_mont(int(x)) for x in c[r * V:(r + 1) * V]]
        want_leaves.append(pyrpo.hash_row(row))
    want_nodes = pyrpo.merkle_nodes(want_leaves)
    got_leaves = tree.leaves.to_numpy().view(np.uint64).reshape(n, 4)
    got_nodes = tree.nodes.to_numpy().view(np.uint64).reshape(n, 4)
    assert [_canon(g) for g in got_leaves] == want_leaves
    assert [_canon(g) for g in got_nodes[1:]] == want_nodes[1:]
    assert _canon(np.frombuffer(tree.root(), dtype=np.uint64)) == want_nodes[1]
    view = tree.prove([3, 4, n - 1])
    assert view["height"] == n.bit_length() - 1 and len(view["initial_leaves"]) == 3
    assert _canon(np.frombuffer(view["initial_leaves"][0], dtype=np.uint64)) == want_leaves[3]
    with pytest.raises(ValueError):
        MerkleTree.from_matrix(m, hash="md5")


@pytest.mark.parametrize("kind", KINDS)
def test_fri_layer_commitment_over_rpo(kind):
    from ministark_amd import GOLDILOCKS_FQ3, MerkleTree
    pl = backends.planner(kind)
    n, ff = (512, 8) if kind == "hip" else (32, 4)
    ev = cref.random_elements(n * 3, 31)
    tree = MerkleTree.from_fri_layer(GpuVec.from_numpy(pl, ev, GOLDILOCKS_FQ3), ff, hash="rpo256")
    rows = n // ff
    want_leaves = [pyrpo.hash_row([GL.from_mont(int(x)) for x in ev[r * ff * 3:(r + 1) * ff * 3]]) for r in range(rows)]
    want_nodes = pyrpo.merkle_nodes(want_leaves)
    assert _canon(np.frombuffer(tree.root(), dtype=np.uint64)) == want_nodes[1]


@pytest.mark.parametrize("kind", KINDS)
def test_tall_tree_uses_both_merge_kernels(kind):
    """2^17 leaves: the levels of more than 2^15 nodes run one lane per node, the rest sixteen lanes per node (rpo_kernels.h);
    every level is sampled against the oracle's merge of the two children the device holds, and the nodes above the level of 2^15
    are rebuilt as a tree of their own -- same digests."""
    pl = backends.planner(kind)
    n = 1 << 17 if kind == "hip" else 64                    # (the simulator only checks the bookkeeping of this test)
    leaves = GpuVec.from_numpy(pl, cref.random_elements(n * 4, 77))
    nodes = gen_rpo_merkle_tree(leaves).to_numpy().reshape(n, 4)
    lv = leaves.to_numpy().reshape(n, 4)
    rng = np.random.default_rng(5)
    count = n // 2
    while count >= 1:
        for i in set(int(x) for x in rng.integers(0, count, size=6)):
            k = count + i                                   # node k = merge(children 2k, 2k + 1); the leaves sit under level n/2
            left, right = (lv[2 * i], lv[2 * i + 1]) if count == n // 2 else (nodes[2 * k], nodes[2 * k + 1])
            assert _canon(nodes[k]) == pyrpo.merge(_canon(left), _canon(right)), (count, i)
        count //= 2
    # the level of n/4 nodes as the leaves of a tree of its own: everything above it must come out the same
    q = n // 4
    sub = GpuVec.from_numpy(pl, nodes[q:2 * q].reshape(-1).copy())
    sub_nodes = gen_rpo_merkle_tree(sub).to_numpy().reshape(q, 4)
    assert np.array_equal(sub_nodes[1:], nodes[1:q])
