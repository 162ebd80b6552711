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
