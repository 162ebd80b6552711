"""ORACLE (test infrastructure) -- SHA-256 row hashing and Merkle tree.

Follows:
  * leaf  = SHA-256( || over columns of serialize_uncompressed(M[c][r]) )
            src/merkle.rs:412-436, src/hash.rs:92-99.  ark-serialize 0.4.2
            (un-vendored) writes the canonical (non-Montgomery) integer in
            little-endian, 8 bytes for Goldilocks, 32 for Fp252, c0||c1||c2
            for Fq3 -- format assumed, see SURVEY.md 8(c).
  * node  = SHA-256(left || right); nodes[k] has children 2k, 2k+1; leaves
            pair into nodes[n/2..n); nodes[1] is the root; nodes[0] unused
            (left as 32 zero bytes)          src/merkle.rs:485-508, src/hash.rs:77-82.
"""
import hashlib


def hash_rows(field, columns):
    """columns: list of equal-length lists of canonical elements."""
    nrows = len(columns[0])
    out = []
    for r in range(nrows):
        h = hashlib.sha256()
        for col in columns:
            h.update(field.to_bytes(col[r]))
        out.append(h.digest())
    return out


def build_merkle_nodes(leaves):
    n = len(leaves)
    assert n >= 2 and n & (n - 1) == 0
    nodes = [bytes(32)] * n
    for i in range(n // 2):
        nodes[n // 2 + i] = hashlib.sha256(leaves[2 * i] + leaves[2 * i + 1]).digest()
    for k in range(n // 2 - 1, 0, -1):
        nodes[k] = hashlib.sha256(nodes[2 * k] + nodes[2 * k + 1]).digest()
    return nodes


def merkle_root(leaves):
    return build_merkle_nodes(leaves)[1]


def prove(leaves, nodes, indices):
    """MerkleTreeImpl::prove (src/merkle.rs:149-206): the batched opening ("MerkleView") of the
    sorted, de-duplicated `indices` -> dict(nodes, initial_leaves, sibling_leaves, height)."""
    from collections import deque
    num_leaves = len(leaves)
    for i in indices:
        if i >= num_leaves:
            raise IndexError(f"leaf index {i} out of bounds ({num_leaves})")          # Error::LeafIndexOutOfBounds
    idx = sorted(set(indices))
    initial_leaves, sibling_leaves = [], []
    node_queue = deque()
    leaf_queue = deque(idx)
    while leaf_queue:                                                               # merkle.rs:166-182
        index = leaf_queue.popleft()
        initial_leaves.append(leaves[index])
        node_queue.append((num_leaves + index) >> 1)
        if leaf_queue and (index ^ 1) == leaf_queue[0]:
            initial_leaves.append(leaves[leaf_queue.popleft()])
            continue
        sibling_leaves.append(leaves[index ^ 1])
    out_nodes = []
    while node_queue:                                                               # merkle.rs:185-198
        index = node_queue.popleft()
        if index > 2:
            node_queue.append(index >> 1)
        if node_queue and (index ^ 1) == node_queue[0]:
            node_queue.popleft()
            continue
        out_nodes.append(nodes[index ^ 1])
    return {"nodes": out_nodes, "initial_leaves": initial_leaves, "sibling_leaves": sibling_leaves,
            "height": num_leaves.bit_length() - 1}


def verify(root, proof, indices):
    """MerkleTreeImpl::verify (src/merkle.rs:208-287) with HashedLeafConfig (hash_leaves = hash_nodes =
    SHA-256(l || r), src/merkle.rs:397-405).  Returns True iff the opening is consistent with `root`."""
    from collections import deque
    merge = lambda l, r: hashlib.sha256(l + r).digest()
    height = proof["height"]
    num_leaves = 1 << height
    if any(i >= num_leaves for i in indices):
        raise IndexError("leaf index out of bounds")
    idx = sorted(set(indices))
    node_queue = deque()
    siblings = deque(proof["sibling_leaves"])
    leaf_queue = deque(zip(idx, proof["initial_leaves"]))
    while leaf_queue:
        index, leaf = leaf_queue.popleft()
        node_index = (num_leaves + index) >> 1
        if leaf_queue and (index ^ 1) == leaf_queue[0][0]:
            _, nxt = leaf_queue.popleft()
            node_queue.append((node_index, merge(leaf, nxt)))
            continue
        sib = siblings.popleft()
        node_queue.append((node_index, merge(leaf, sib) if index % 2 == 0 else merge(sib, leaf)))
    assert not siblings
    nodes = deque(proof["nodes"])
    while node_queue:
        index, h = node_queue.popleft()
        if index.bit_length() - 1 == 0:
            assert not node_queue
            return h == root
        if node_queue and (index ^ 1) == node_queue[0][0]:
            _, nh = node_queue.popleft()
            node_queue.append((index >> 1, merge(h, nh)))
            continue
        sib = nodes.popleft()
        node_queue.append((index >> 1, merge(h, sib) if index % 2 == 0 else merge(sib, h)))
    return True
