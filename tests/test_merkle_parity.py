"""SHA-256 row hashing + Merkle tree (src/merkle.rs:412-508, src/hash.rs:58-100): device vs oracle,
byte-exact.  Shapes follow benches/merkle_tree.rs:17-45 (3 Goldilocks columns, depth 14-17) plus
ragged block boundaries (1..9 columns cross the 55/56/64-byte padding edges) and Fq3 rows."""
import hashlib

import numpy as np
import pytest

from oracle import cref
from tests import backends
from ministark_amd import GOLDILOCKS_FP, GOLDILOCKS_FQ3, Matrix, MerkleTree

BACKENDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


def _commit(kind, field, log_rows, ncols, seed=3):
    pl = backends.planner(kind)
    V = 3 if field == GOLDILOCKS_FQ3 else 1
    n = 1 << log_rows
    cols = [cref.random_elements(n * V, seed + c) for c in range(ncols)]
    m = Matrix.from_numpy(pl, cols, field)
    leaves = m.hash_rows().to_numpy().reshape(n, 32)
    want_leaves = cref.sha256_rows(cols, V)
    assert np.array_equal(leaves, want_leaves), "row digests differ"
    if n >= 2:
        tree = MerkleTree.from_matrix(m)
        want_nodes = cref.sha256_merkle(want_leaves)
        assert np.array_equal(tree.nodes_numpy(), want_nodes), "merkle nodes differ"
        assert tree.root() == want_nodes[1].tobytes()


@pytest.mark.parametrize("ncols", [1, 2, 3, 6, 7, 8, 9, 15, 16, 17, 32])
def test_rows_block_edges_emu(ncols):
    _commit("emu", GOLDILOCKS_FP, 5, ncols)


def test_fq3_rows_emu():
    _commit("emu", GOLDILOCKS_FQ3, 6, 3)
    _commit("emu", GOLDILOCKS_FQ3, 4, 9)


def test_single_row_and_known_digest_emu():
    # one element == 1 (Montgomery 2^32-1): the leaf is SHA-256 of 01 00 00 00 00 00 00 00
    pl = backends.planner("emu")
    m = Matrix.from_numpy(pl, [np.array([4294967295], dtype=np.uint64)])
    leaf = m.hash_rows().to_numpy().tobytes()
    assert leaf == hashlib.sha256((1).to_bytes(8, "little")).digest()


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [14, 15, 16, 17])
def test_merkle_tree_bench_shapes_hip(depth):            # benches/merkle_tree.rs:17-45
    _commit("hip", GOLDILOCKS_FP, depth, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("ncols", [1, 7, 8, 9, 17, 26, 32, 100])
def test_rows_block_edges_hip(ncols):
    _commit("hip", GOLDILOCKS_FP, 10, ncols)


@pytest.mark.gpu
def test_fq3_rows_hip():
    _commit("hip", GOLDILOCKS_FQ3, 12, 9)


@pytest.mark.gpu
def test_commit_2_20_x_32_hip():
    _commit("hip", GOLDILOCKS_FP, 20, 32)


# every shape of the upper levels (sha256_merkle_top): one workgroup only (<= 2^9 leaves), subtrees of 256 parents on <= 256
# workgroups (2^10 .. 2^17 leaves), two parents per lane at 2^17 parents (2^18 leaves and more), all nodes compared
@pytest.mark.parametrize("kind,log_rows", [("emu", 18)] + [pytest.param("hip", d, marks=pytest.mark.gpu) for d in (9, 10, 13, 18, 19, 21)])
def test_upper_levels_of_tall_trees(kind, log_rows):
    _commit(kind, GOLDILOCKS_FP, log_rows, 1)


@pytest.mark.parametrize("kind", BACKENDS)
@pytest.mark.parametrize("field,ff", [(GOLDILOCKS_FP, 2), (GOLDILOCKS_FP, 8), (GOLDILOCKS_FQ3, 4), (GOLDILOCKS_FQ3, 16)])
def test_fri_layer_commit_row_major(kind, field, ff):       # src/fri.rs:213-216
    from ministark_amd import GpuVec
    pl = backends.planner(kind)
    V = 3 if field == GOLDILOCKS_FQ3 else 1
    n = 1 << 9
    ev = cref.random_elements(n * V, 11)
    tree = MerkleTree.from_fri_layer(GpuVec.from_numpy(pl, ev, field), ff)
    rows = ev.reshape(n // ff, ff, V)
    cols = [np.ascontiguousarray(rows[:, k, :]).ravel() for k in range(ff)]      # Matrix::from_arrays: column k = k-th element of every coset
    want = cref.sha256_merkle(cref.sha256_rows(cols, V))
    assert np.array_equal(tree.nodes_numpy(), want)


def test_clone_is_device_side_emu():
    from ministark_amd import GpuVec
    pl = backends.planner("emu")
    a = cref.random_elements(100, 3)
    v = GpuVec.from_numpy(pl, a)
    w = v.clone()
    assert w.ptr != v.ptr and np.array_equal(w.to_numpy(), a)


def _lz(d):
    z = 0
    for b in d:
        if b == 0:
            z += 8
            continue
        z += 8 - b.bit_length()
        break
    return z


@pytest.mark.parametrize("kind", BACKENDS)
def test_pow_grind_matches_sequential_search(kind):          # src/random.rs:48-55, 129-132
    from ministark_amd import grind_proof_of_work
    pl = backends.planner(kind)
    bits = 20 if kind == "hip" else 9
    for s in range(3):
        seed = hashlib.sha256(bytes([s])).digest()
        got = grind_proof_of_work(pl, seed, bits, 1 << 32)
        nonce = 1
        while _lz(hashlib.sha256(seed + nonce.to_bytes(8, "big")).digest()) < bits:
            nonce += 1
        assert got == nonce
    assert grind_proof_of_work(pl, seed, 0) == 1


def test_view_index_walk_in_the_library_matches_the_queue_form():
    """ms_merkle_view_ids (MerkleTreeImpl::prove's index walk, src/merkle.rs:149-206, as a host helper of the C ABI) against the two
    queues written out in Python, on random and on clustered index sets, trees of 2 .. 2^14 leaves."""
    from ministark_amd.api import merkle_view_ids, merkle_view_ids_py
    lib = backends.planner("emu").lib
    rng = np.random.default_rng(9)
    for trial in range(1500):
        n = 1 << int(rng.integers(1, 15))
        k = int(rng.integers(1, 40))
        if trial % 3 == 0:
            base = int(rng.integers(0, n))
            idx = [min(n - 1, base + int(d)) for d in rng.integers(0, 5, size=k)]
        else:
            idx = [int(x) for x in rng.integers(0, n, size=k)]
        got, want = merkle_view_ids(n, idx, lib), merkle_view_ids_py(n, idx)
        assert [list(x) for x in got] == [list(x) for x in want], (n, idx)
    with pytest.raises(IndexError):
        merkle_view_ids(8, [8], lib)
