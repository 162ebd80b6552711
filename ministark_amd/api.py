"""Host-side mirror of the reference's gpu-poly interface, over the C ABI.

Same names, argument meaning and error behaviour as the Rust items they mirror
(reference path:line in each docstring), so the parity tests read like the
reference's own tests (gpu/tests/shaders.rs).  Python is only plumbing here: all
arithmetic happens in the HIP kernels behind include/ministark_hip.h.  (The
reference is Rust; there is no Rust toolchain in the build image, so this file and
ministark_amd/csrc/host/ministark.hpp stand where the `cfg(feature = "hip")`
shim of INTEGRATION.md would.)
"""
import ctypes
import weakref
from collections import deque

import numpy as np

from . import _lib

GOLDILOCKS_FP, GOLDILOCKS_FQ3, STARK252_FP = 0, 1, 2
FIELD_WORDS = {GOLDILOCKS_FP: 1, GOLDILOCKS_FQ3: 3, STARK252_FP: 4}

GL_P = (1 << 64) - (1 << 32) + 1
_GL_R = (1 << 64) % GL_P
_GL_TWO_ADIC_ROOT = 1753635133440165772
GL_GENERATOR = 7


def gl_to_mont(x):
    return (int(x) * _GL_R) % GL_P


def gl_from_mont(x):
    return (int(x) * pow(_GL_R, -1, GL_P)) % GL_P


# the 252-bit StarkWare prime (gpu/src/fields.rs:239-264): generator 3, two-adicity 192, R = 2^256
F252_P = (1 << 251) + 17 * (1 << 192) + 1
_F252_R = (1 << 256) % F252_P
F252_GENERATOR = 3


def f252_to_mont_limbs(x):
    """canonical int -> 4 little-endian u64 limbs of the Montgomery residue."""
    m = (int(x) * _F252_R) % F252_P
    return np.array([(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def f252_from_mont_limbs(limbs):
    m = sum(int(v) << (64 * i) for i, v in enumerate(limbs))
    return (m * pow(_F252_R, -1, F252_P)) % F252_P


class Planner:
    """`Planner` / `get_planner()` (gpu/src/plan.rs:327-351, 464-469): owns the device,
    the kernel library and the command queue (here: a HIP stream)."""

    def __init__(self, device=0, lib=None):
        self.lib = lib or _lib.lib()
        h = ctypes.c_void_p()
        self.lib.check(self.lib.ms_ctx_create(device, ctypes.byref(h)))
        self.handle = h
        self.device = device
        self._plans = weakref.WeakSet()                  # GpuFft / GpuIfft objects created on this context

    def sync(self):
        self.lib.check(self.lib.ms_sync(self.handle))

    @property
    def stream(self):
        return self.lib.ms_ctx_stream(self.handle)

    def profile(self, on=True):
        """Bracket every kernel launch with hipEvents on the context's stream."""
        self.lib.check(self.lib.ms_profile_enable(self.handle, 1 if on else 0))

    def profile_read(self):
        """-> {kernel: {"calls", "total_us", "avg_us", "bytes_per_call"}} (blocks)."""
        buf = ctypes.create_string_buffer(1 << 16)
        self.lib.check(self.lib.ms_profile_read(self.handle, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, calls, us, b = line.split()
            out[name] = {"calls": int(calls), "total_us": float(us), "avg_us": float(us) / int(calls),
                         "bytes_per_call": float(b)}
        return out

    def jit_stats(self):
        """-> dict: this context's specialised constraint kernels -- compiled, loaded from the on-disk cache, failed (interpreter) -- and
        what that cost in milliseconds.  A non-zero `compile_failures` is a performance bug worth reporting, never a wrong result."""
        return self.lib.jit_stats(self.handle)

    def close(self):
        if self.handle:
            for plan in list(self._plans):               # plans refer to the context: release them first
                plan.close()
            self.lib.ms_ctx_destroy(self.handle)         # (the library also releases any plan it still knows of)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_PLANNER = None


def get_planner():
    global _PLANNER
    if _PLANNER is None:
        _PLANNER = Planner()
    return _PLANNER


class GpuVec:
    """`GpuVec<F>` (src/utils.rs:438-470): a column of `n` field elements.  The reference
    aliases a page-aligned host Vec (unified memory); here the column lives in HBM and
    `to_numpy()` / `from_numpy()` are the explicit mirror."""

    def __init__(self, planner, n, field=GOLDILOCKS_FP, ptr=None, owner=True):
        self.planner = planner
        self.n = n
        self.field = field
        self.words = n * FIELD_WORDS[field]
        self.owner = owner and ptr is None
        if ptr is None:
            p = ctypes.c_void_p()
            planner.lib.check(planner.lib.ms_alloc(planner.handle, max(self.words * 8, 8), ctypes.byref(p)))
            ptr = p.value
        self.ptr = ptr

    @classmethod
    def from_numpy(cls, planner, arr, field=GOLDILOCKS_FP):
        arr = np.ascontiguousarray(arr, dtype=np.uint64).ravel()
        V = FIELD_WORDS[field]
        assert arr.size % V == 0
        v = cls(planner, arr.size // V, field)
        if arr.size:
            planner.lib.check(planner.lib.ms_upload(planner.handle, v.ptr, arr.ctypes.data, arr.size * 8))
        return v

    def to_numpy(self):
        out = np.empty(self.words, dtype=np.uint64)
        if self.words:
            self.planner.lib.check(self.planner.lib.ms_download(self.planner.handle, out.ctypes.data, self.ptr, self.words * 8))
        return out

    def clone(self):
        out = GpuVec(self.planner, self.n, self.field)
        self.planner.lib.check(self.planner.lib.ms_copy(self.planner.handle, out.ptr, self.ptr, self.words * 8))
        return out

    def free(self):
        if self.owner and self.ptr:
            self.planner.lib.ms_free(self.planner.handle, self.ptr)
            self.ptr = None

    def __len__(self):
        return self.n

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBytes:
    """Raw device allocation (digest arrays)."""

    def __init__(self, planner, nbytes):
        self.planner = planner
        self.nbytes = nbytes
        p = ctypes.c_void_p()
        planner.lib.check(planner.lib.ms_alloc(planner.handle, max(nbytes, 32), ctypes.byref(p)))
        self.ptr = p.value

    def to_numpy(self, nbytes=None):
        """the first `nbytes` bytes (default: all) on the host"""
        nbytes = self.nbytes if nbytes is None else nbytes
        out = np.empty(nbytes, dtype=np.uint8)
        if nbytes:
            self.planner.lib.check(self.planner.lib.ms_download(self.planner.handle, out.ctypes.data, self.ptr, nbytes))
        return out

    def __del__(self):
        try:
            if self.ptr:
                self.planner.lib.ms_free(self.planner.handle, self.ptr)
                self.ptr = None
        except Exception:
            pass


class GatherBatch:
    """One device buffer for every gather of a phase (the query phase of a proof opens two trace trees and every FRI layer: some twenty
    small gathers): each gather writes its slice, the first fetch downloads the buffer ONCE and every fetch reads its slice of the host
    copy -- one device-to-host copy and one wait instead of twenty (each costs a round trip of 25-40 us however small it is)."""

    def __init__(self, planner, capacity=1 << 20):
        self.planner, self.capacity, self.used = planner, capacity, 0
        self.buf = DeviceBytes(planner, capacity)
        self._host = None
        self._digest_segments = []                             # (digests ptr, ndigests, index array, output ptr): launched together (flush)

    def defer_digests(self, digests_ptr, ndigests, idx, out_ptr):
        """a digest gather into a slice of this batch: it joins the ONE launch that `flush` issues (ms_gather_digests_multi)"""
        self._digest_segments.append((digests_ptr, ndigests, idx, out_ptr))

    def flush(self):
        """launch the digest gathers collected so far (before the download, or to start them early)"""
        segs, self._digest_segments = self._digest_segments, []
        if not segs:
            return
        L, n = self.planner.lib, len(segs)
        VP, SZ = ctypes.c_void_p, ctypes.c_size_t
        allidx = np.ascontiguousarray(np.concatenate([s[2] for s in segs]), dtype=np.uint64)      # (kept alive through the call)
        L.check(L.ms_gather_digests_multi(self.planner.handle, n, (VP * n)(*[s[0] for s in segs]), (SZ * n)(*[s[1] for s in segs]),
                                          allidx.ctypes.data, (SZ * n)(*[len(s[2]) for s in segs]), (VP * n)(*[s[3] for s in segs])))

    def reserve(self, nbytes):
        """-> (device pointer, reader) for a slice of nbytes, or None when the batch is full or has already been fetched"""
        nbytes = (nbytes + 31) & ~31
        if self._host is not None or self.used + nbytes > self.capacity:
            return None
        off = self.used
        self.used += nbytes

        def read():
            return self.fetch()[off:off + nbytes]
        return self.buf.ptr + off, read

    def fetch(self):
        """the used part of the buffer on the host: ONE wait and one copy of `used` bytes (not of the capacity); closes the batch"""
        if self._host is None:
            self.flush()
            self._host = self.buf.to_numpy(self.used)
        return self._host


def _gather_slot(planner, nbytes, batch):
    """(device pointer, function returning the bytes, keep-alive) for a gather's output: a slice of `batch` if there is room, else its own buffer"""
    slot = batch.reserve(nbytes) if batch is not None else None
    if slot is not None:
        return slot[0], slot[1], batch
    out = DeviceBytes(planner, nbytes)
    return out.ptr, out.to_numpy, out


def _gather_digests_launch(planner, digests, ndigests, ids, batch=None):
    """Launch the gather of the 32-byte records `ids` of a device digest array; returns a function that downloads them as a
    list of bytes.  (Launch every gather of an opening first, fetch afterwards: one wait instead of one per call; with a
    GatherBatch one download for the whole phase.)"""
    if len(ids) == 0:
        return lambda: []
    idx = np.ascontiguousarray(ids, dtype=np.uint64)
    ptr, read, keep = _gather_slot(planner, 32 * len(ids), batch)
    L = planner.lib
    if keep is batch and batch is not None:                    # a slice of the batch: one launch for all of the batch's digest gathers
        if idx.size and int(idx.max()) >= ndigests:
            raise _lib.MsError(-1, f"digest {int(idx.max())} out of range ({ndigests})")
        batch.defer_digests(digests.ptr, ndigests, idx, ptr)
    else:
        L.check(L.ms_gather_digests(planner.handle, ndigests, digests.ptr, idx.ctypes.data, len(ids), ptr))

    def fetch(_keep=keep):
        return np.ascontiguousarray(read()[: 32 * len(ids)]).view("V32").tolist()      # a list of 32-byte `bytes`, built in one C loop
    return fetch


def _gather_digests(planner, digests, ndigests, ids):
    return _gather_digests_launch(planner, digests, ndigests, ids)()


HASHES = ("sha256", "rpo256")


def merkle_view_ids(n, indices, lib=None):
    """The index walk of `MerkleTreeImpl::prove` (src/merkle.rs:149-206) over a tree of n leaves: -> (leaf_ids, initial, sibling,
    node_ids): the leaves to fetch (initial / sibling: which of them are the queried ones / their siblings) and the internal nodes of
    the batched opening, in the reference's order.  Indices only -- a single-device tree and a row-sharded one walk the same lists.
    With `lib` the walk runs in the library (ms_merkle_view_ids: 8 us against 100 us for 32 queries of a 2^24-leaf tree in the
    interpreter -- eight trees per proof); without it, the same two queues in Python (`merkle_view_ids_py`, the comparison in the tests)."""
    if lib is None:
        return merkle_view_ids_py(n, indices)
    leaf, initial, sibling, node = _merkle_view_ids_arrays(n, indices, lib)
    return leaf.tolist(), initial.tolist(), sibling.tolist(), node.tolist()


def _merkle_view_ids_arrays(n, indices, lib):
    """merkle_view_ids through the library, as numpy arrays (what the gathers take: no list round trip)"""
    if isinstance(indices, np.ndarray) and indices.dtype.kind == "u":
        idx = np.ascontiguousarray(indices, dtype=np.uint64)
    else:                                                                         # a negative index is out of bounds, not an OverflowError of a
        try:                                                                      # conversion to unsigned: through int64 (one C loop, not a Python one)
            signed = np.asarray(indices, dtype=np.int64)
        except OverflowError:
            raise IndexError(f"leaf index out of bounds ({n})") from None         # Error::LeafIndexOutOfBounds
        if signed.size and int(signed.min()) < 0:
            raise IndexError(f"leaf index {int(signed.min())} out of bounds ({n})")
        idx = signed.astype(np.uint64)
    if idx.size and int(idx.max()) >= n:
        raise IndexError(f"leaf index {int(idx.max())} out of bounds ({n})")     # Error::LeafIndexOutOfBounds
    m = max(1, idx.size)
    leaf = np.empty(2 * m, dtype=np.uint64)
    sib = np.empty(2 * m, dtype=np.uint8)
    node = np.empty(m * max(1, n.bit_length()), dtype=np.uint64)
    nl, nn = ctypes.c_size_t(0), ctypes.c_size_t(0)
    lib.check(lib.ms_merkle_view_ids(n, idx.ctypes.data, idx.size, leaf.ctypes.data, sib.ctypes.data, ctypes.byref(nl), node.ctypes.data, ctypes.byref(nn)))
    flags = sib[: nl.value]
    return leaf[: nl.value], np.flatnonzero(flags == 0), np.flatnonzero(flags), node[: nn.value]


def merkle_view_ids_py(n, indices):
    """the walk as the reference writes it (two queues)"""
    for i in indices:
        if i >= n:
            raise IndexError(f"leaf index {i} out of bounds ({n})")         # Error::LeafIndexOutOfBounds
    leaf_ids, initial, sibling = [], [], []
    node_queue = deque()
    leaf_queue = deque(sorted(set(int(i) for i in indices)))
    while leaf_queue:
        index = leaf_queue.popleft()
        initial.append(len(leaf_ids)); leaf_ids.append(index)
        node_queue.append((n + index) >> 1)
        if leaf_queue and (index ^ 1) == leaf_queue[0]:
            initial.append(len(leaf_ids)); leaf_ids.append(leaf_queue.popleft())
            continue
        sibling.append(len(leaf_ids)); leaf_ids.append(index ^ 1)
    node_ids = []
    while node_queue:
        index = node_queue.popleft()
        if index > 2:
            node_queue.append(index >> 1)
        if node_queue and (index ^ 1) == node_queue[0]:
            node_queue.popleft()
            continue
        node_ids.append(index ^ 1)
    return leaf_ids, initial, sibling, node_ids


class MerkleTree:
    """`MatrixMerkleTreeImpl<H>` (src/merkle.rs:296-361): `from_matrix` hashes the rows and builds the node
    array on device; `root` is nodes[1] (src/merkle.rs:145-147).  H is selected by `hash`:
      "sha256"  Sha256HashFn (src/hash.rs:58-100), the reference's default;
      "rpo256"  RPO-256 over Goldilocks, the GPU-friendly commitment the reference prepared kernels for
                (gpu/src/plan.rs:32-174, README.md:90): leaves by ms_rpo256_rows_field, nodes by
                gen_rpo_merkle_tree.  A digest is 4 Fp elements = 32 bytes, so proofs have the same shape."""

    def __init__(self, planner, leaves, nleaves, hash="sha256"):
        if hash not in HASHES:
            raise ValueError(f"unknown hash {hash!r} (one of {HASHES})")
        self.planner = planner
        self.leaves = leaves
        self.nleaves = nleaves
        self.hash = hash
        self.nodes = DeviceBytes(planner, nleaves * 32)
        L = planner.lib
        if hash == "sha256":
            L.check(L.ms_sha256_merkle(planner.handle, nleaves, leaves.ptr, self.nodes.ptr))
        else:
            L.check(L.ms_rpo256_merkle(planner.handle, nleaves, leaves.ptr, self.nodes.ptr))

    @classmethod
    def from_matrix(cls, matrix, hash="sha256"):
        return cls(matrix.planner, matrix.hash_rows(hash), matrix.num_rows(), hash)

    @classmethod
    def from_fri_layer(cls, evaluations, folding_factor, hash="sha256"):
        """`Matrix::from_arrays(evaluations.as_chunks::<N>())` + `M::from_matrix` (src/fri.rs:213-216):
        commit to a bit-reversed FRI layer whose rows are the cosets of N consecutive evaluations."""
        pl = evaluations.planner
        nrows = len(evaluations) // folding_factor
        leaves = DeviceBytes(pl, nrows * 32)
        if hash == "sha256":
            pl.lib.check(pl.lib.ms_sha256_rows_row_major(pl.handle, evaluations.field, nrows, folding_factor, evaluations.ptr, leaves.ptr))
        elif hash == "rpo256":
            if evaluations.field == STARK252_FP:
                raise ValueError("RPO-256 absorbs Goldilocks elements")
            words = folding_factor * FIELD_WORDS[evaluations.field]          # a row is N elements = N (or 3 N) Fp words in memory order
            pl.lib.check(pl.lib.ms_rpo256_rows_row_major(pl.handle, nrows, words, evaluations.ptr, leaves.ptr))
        else:
            raise ValueError(f"unknown hash {hash!r} (one of {HASHES})")
        return cls(pl, leaves, nrows, hash)

    def prove(self, indices):
        """`MerkleTreeImpl::prove` (src/merkle.rs:149-206): the batched opening of `indices` as the
        reference's MerkleView -> dict(nodes, initial_leaves, sibling_leaves, height), digests as
        bytes.  The walk over indices is bookkeeping; the digests are gathered on the device and
        come back in one copy."""
        return self.prove_launch(indices)()

    def prove_launch(self, indices, batch=None):
        """`prove` in two halves: the device gathers are launched now, the returned function fetches and assembles the view."""
        n = self.nleaves
        leaf_ids, initial, sibling, node_ids = _merkle_view_ids_arrays(n, indices, self.planner.lib)
        initial, sibling = initial.tolist(), sibling.tolist()
        fetch_leaves = _gather_digests_launch(self.planner, self.leaves, n, leaf_ids, batch)
        fetch_nodes = _gather_digests_launch(self.planner, self.nodes, n, node_ids, batch)

        def fetch():
            leaves, nodes = fetch_leaves(), fetch_nodes()
            return {"nodes": nodes,
                    "initial_leaves": [leaves[k] for k in initial],
                    "sibling_leaves": [leaves[k] for k in sibling],
                    "height": n.bit_length() - 1}
        return fetch

    def root(self):
        out = np.empty(32, dtype=np.uint8)
        self.planner.lib.check(self.planner.lib.ms_download(self.planner.handle, out.ctypes.data, self.nodes.ptr + 32, 32))
        return out.tobytes()

    def nodes_numpy(self):
        return self.nodes.to_numpy().reshape(self.nleaves, 32)


class Radix2EvaluationDomain:
    """ark_poly::Radix2EvaluationDomain over Goldilocks as the reference uses it
    (gpu/src/plan.rs:386-423): `new(n)` / `new_coset(n, offset)`; constants are canonical
    integers, `*_mont` the Montgomery words that cross the C ABI."""

    def __init__(self, size, offset=1, fft_field=GOLDILOCKS_FP):
        if size < 1 or size & (size - 1):
            raise ValueError("domain size must be a power of two")
        self.size = size
        self.log_size = size.bit_length() - 1
        self.fft_field = fft_field
        if fft_field == STARK252_FP:
            p, two_adicity, root = F252_P, 192, pow(F252_GENERATOR, (F252_P - 1) >> 192, F252_P)
        else:
            p, two_adicity, root = GL_P, 32, _GL_TWO_ADIC_ROOT
        self.p = p
        if self.log_size > two_adicity:
            raise ValueError("domain exceeds the two-adicity of the field")
        self.group_gen = pow(root, 1 << (two_adicity - self.log_size), p)
        self.group_gen_inv = pow(self.group_gen, -1, p)
        self.size_inv = pow(size % p, -1, p)
        self.offset = offset % p
        self.offset_inv = pow(self.offset, -1, p)

    @classmethod
    def new(cls, size, fft_field=GOLDILOCKS_FP):
        return cls(size, 1, fft_field)

    @classmethod
    def new_coset(cls, size, offset, fft_field=GOLDILOCKS_FP):
        return cls(size, offset, fft_field)

    def _limbs(self, x):
        if self.fft_field == STARK252_FP:
            return f252_to_mont_limbs(x)
        return np.array([gl_to_mont(x)], dtype=np.uint64)

    @property
    def offset_mont(self):
        return gl_to_mont(self.offset) if self.fft_field != STARK252_FP else f252_to_mont_limbs(self.offset)

    @property
    def group_gen_mont(self):
        return gl_to_mont(self.group_gen) if self.fft_field != STARK252_FP else f252_to_mont_limbs(self.group_gen)


def _ptr_array(vecs):
    if isinstance(vecs, ColumnSet):
        return vecs.arr
    return (ctypes.c_void_p * len(vecs))(*[v.ptr for v in vecs])


class ColumnSet:
    """A fixed list of equally long columns of one field with its pointer table built ONCE: what `enqueue` / `enqueue_to` take when the same
    columns are transformed again and again (building the table of 512 pointers and checking 512 lengths in the interpreter costs more than
    the launch that transforms 512 columns of 2^12 points).  The C++ mirror passes a std::vector of pointers and has no such cost."""

    def __init__(self, columns):
        self.columns = list(columns)
        if not self.columns:
            raise ValueError("an empty set of columns")
        self.n, self.field = len(self.columns[0]), self.columns[0].field
        if any(len(c) != self.n or c.field != self.field for c in self.columns):
            raise ValueError("all columns of a set must have the same length and field")
        self.arr = (ctypes.c_void_p * len(self.columns))(*[c.ptr for c in self.columns])

    def __len__(self):
        return len(self.columns)

    def __iter__(self):
        return iter(self.columns)


class _FftBase:
    MIN_SIZE = 1  # the reference asserts >= 2048 (gpu/src/plan.rs:248,294); this backend has no lower bound
    _inverse = 0

    def __init__(self, domain, field=GOLDILOCKS_FP, planner=None):
        self.planner = planner or get_planner()
        self.domain = domain
        self.field = field
        if (field == STARK252_FP) != (domain.fft_field == STARK252_FP):
            raise ValueError("domain and column fields do not match")
        off = domain._limbs(domain.offset)
        gen = domain._limbs(domain.group_gen)
        h = ctypes.c_void_p()
        L = self.planner.lib
        L.check(L.ms_ntt_plan_create(self.planner.handle, field, domain.log_size, self._inverse,
                                     off.ctypes.data, gen.ctypes.data, ctypes.byref(h)))
        self.handle = h
        self._keep = []
        self.planner._plans.add(self)

    @classmethod
    def from_domain(cls, domain, field=GOLDILOCKS_FP, planner=None):
        return cls(domain, field, planner)

    def encode(self, column):
        """`encode(&mut [F])` (gpu/src/plan.rs:254-263 / 300-309): queue one column, in place."""
        if len(column) != self.domain.size:
            raise ValueError(f"column has {len(column)} elements, domain {self.domain.size}")  # plan.rs:257 assert_eq!
        if column.field != self.field:
            raise ValueError("column field differs from the plan's")
        self._keep.append(column)
        self.planner.lib.check(self.planner.lib.ms_ntt_encode(self.handle, column.ptr))

    def execute(self):
        """`execute(self)` (gpu/src/plan.rs:229-232): run everything queued and block."""
        self.planner.lib.check(self.planner.lib.ms_ntt_execute(self.handle))
        self._keep = []

    def enqueue(self, columns):
        """Non-blocking launch of a batch, in place: `encode` of every column + `execute` without the wait."""
        self._check(columns)
        arr = _ptr_array(columns)
        self.planner.lib.check(self.planner.lib.ms_ntt_enqueue(self.handle, arr, len(columns)))

    def _check(self, columns):
        if isinstance(columns, ColumnSet):
            if columns.n != self.domain.size or columns.field != self.field:
                raise ValueError(f"columns of {columns.n} elements of field {columns.field}, the plan {self.domain.size} of field {self.field}")
            return
        for c in columns:
            if len(c) != self.domain.size or c.field != self.field:
                raise ValueError(f"column has {len(c)} elements of field {c.field}, the plan {self.domain.size} of field {self.field}")  # plan.rs:257 assert_eq!

    def enqueue_to(self, src_columns, dst_columns):
        """Non-blocking, out of place: dst[c] = transform(src[c]), src untouched -- `clone()` + transform without the copy (ms_ntt_enqueue_to)."""
        assert len(src_columns) == len(dst_columns)
        self._check(src_columns)
        self._check(dst_columns)
        self.planner.lib.check(self.planner.lib.ms_ntt_enqueue_to(self.handle, _ptr_array(src_columns), _ptr_array(dst_columns), len(src_columns)))

    def close(self):
        if self.handle:
            self.planner.lib.ms_ntt_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GpuFft(_FftBase):
    """`GpuFft::from(domain)` (gpu/src/plan.rs:236-279)."""
    _inverse = 0


class GpuIfft(_FftBase):
    """`GpuIfft::from(domain)` (gpu/src/plan.rs:282-325)."""
    _inverse = 1


class Matrix:
    """`Matrix<F>(Vec<GpuVec<F>>)` (src/matrix.rs:26): column-major, one GpuVec per column.

    The transforms (`interpolate`, `evaluate`, `bit_reversed_evaluate`, `into_*`, `bit_reverse_rows`, `lde`) ENQUEUE on the planner's stream and
    return: everything that consumes their result is ordered behind them on the same stream, and whatever brings bytes to the host
    (`to_numpy`, `MerkleTree.root`, the gathers' fetch) waits.  The reference's methods block (`GpuFft::execute` waits for its command buffer,
    gpu/src/plan.rs:378-386); a caller that wants that calls `planner.sync()` -- the prover does not: a wait after each of its eight transforms
    was 25 us of idle device each (scripts/prove_gaps.py)."""

    def __init__(self, columns):
        self.columns = list(columns)
        # Matrix::new (src/matrix.rs:32-38) asserts that every column has the length of the first; so do the kernels, which
        # take one row count for the whole matrix
        if any(len(c) != len(self.columns[0]) or c.field != self.columns[0].field for c in self.columns[1:]):
            raise ValueError("all columns of a matrix must have the same length and field")

    @classmethod
    def from_numpy(cls, planner, cols, field=GOLDILOCKS_FP):
        return cls([GpuVec.from_numpy(planner, c, field) for c in cols])

    def num_rows(self):
        return len(self.columns[0]) if self.columns else 0

    def num_cols(self):
        return len(self.columns)

    @property
    def field(self):
        return self.columns[0].field

    @property
    def planner(self):
        return self.columns[0].planner

    def clone(self):
        return Matrix([c.clone() for c in self.columns])

    def to_numpy(self):
        return [c.to_numpy() for c in self.columns]

    # src/matrix.rs:102-116 (into_polynomials_gpu) ------------------------------------
    def into_polynomials(self, domain):
        ifft = GpuIfft(domain, self.field, self.planner)
        ifft.enqueue(self.columns)                       # (encode + execute without the wait: see the class comment)
        ifft.close()
        return self

    def _transformed(self, cls, domain):
        """the columns' transform in NEW columns, this matrix untouched: what `self.clone().into_...` gives, without the device copy"""
        outs = [GpuVec(self.planner, len(c), self.field) for c in self.columns]
        plan = cls(domain, self.field, self.planner)
        plan.enqueue_to(self.columns, outs)
        plan.close()                                     # (a handle on the context's cached plan: closing it does not wait)
        return Matrix(outs)

    def interpolate(self, domain):        # src/matrix.rs:155-163 (`self.clone().into_polynomials(domain)`)
        return self._transformed(GpuIfft, domain)

    # src/matrix.rs:193-208 (into_evaluations_gpu): column.resize(domain.size(), 0) then fft
    def into_evaluations(self, domain, bit_reversed=False):
        """`Matrix::into_evaluations(domain)` (src/matrix.rs:193-208): columns shorter than the domain are
        coefficient vectors to be zero-extended (`column.resize(domain.size(), F::zero())`, :201) -- done on
        the device, for blow-ups 4..16 without materialising the padding."""
        n = self.num_rows()
        if n > domain.size:
            raise ValueError("column longer than the evaluation domain")
        if n == domain.size and not bit_reversed:
            fft = GpuFft(domain, self.field, self.planner)
            fft.enqueue(self.columns)
            fft.close()
            return self
        L = self.planner.lib
        outs = self.columns if n == domain.size else [GpuVec(self.planner, domain.size, self.field) for _ in self.columns]
        off = _offset_words(self.field, domain.offset)
        L.check(L.ms_evaluate(self.planner.handle, self.field, n.bit_length() - 1, domain.log_size, off.ctypes.data,
                              _ptr_array(self.columns), _ptr_array(outs), len(outs), 1 if bit_reversed else 0))
        self.columns = outs
        return self

    def _evaluated(self, domain, bit_reversed):
        """evaluate / bit_reversed_evaluate: `self.clone().into_(bit_reversed_)evaluations(domain)`, never copying: columns shorter than
        the domain go through ms_evaluate into new columns (their own storage is only read), columns of the domain's size through the
        out-of-place transform."""
        n = self.num_rows()
        if n > domain.size:
            raise ValueError("column longer than the evaluation domain")
        if n == domain.size and not bit_reversed:
            return self._transformed(GpuFft, domain)
        L = self.planner.lib
        outs = [GpuVec(self.planner, domain.size, self.field) for _ in self.columns]
        off = _offset_words(self.field, domain.offset)
        L.check(L.ms_evaluate(self.planner.handle, self.field, n.bit_length() - 1, domain.log_size, off.ctypes.data,
                              _ptr_array(self.columns), _ptr_array(outs), len(outs), 1 if bit_reversed else 0))
        return Matrix(outs)

    def evaluate(self, domain):           # src/matrix.rs:237-243
        return self._evaluated(domain, False)

    def bit_reverse_rows(self):           # src/matrix.rs:352-354
        L = self.planner.lib
        n = self.num_rows()
        arr = _ptr_array(self.columns)
        L.check(L.ms_bit_reverse(self.planner.handle, self.field, n.bit_length() - 1, arr, len(self.columns)))
        return self

    def into_bit_reversed_evaluations(self, domain):   # src/matrix.rs:225-234 (the bit reversal is fused into the last pass)
        return self.into_evaluations(domain, bit_reversed=True)

    @classmethod
    def from_chunks(cls, poly, k):
        """`composition_poly.chunks(k)` spread over k columns (src/prover.rs:113-121): column c holds
        coefficients c, c + k, c + 2k, ... of `poly` (a GpuVec)."""
        pl = poly.planner
        if k < 1 or len(poly) % k:
            raise ValueError(f"{len(poly)} coefficients do not split into {k} columns")
        n_out = len(poly) // k
        outs = [GpuVec(pl, n_out, poly.field) for _ in range(k)]
        pl.lib.check(pl.lib.ms_deinterleave(pl.handle, poly.field, n_out, k, poly.ptr, _ptr_array(outs)))
        return cls(outs)

    def bit_reversed_evaluate(self, domain):           # src/matrix.rs:245-251
        return self._evaluated(domain, True)

    def get_rows(self, positions):
        """`Matrix::get_row` for every queried position (src/trace.rs:139-152): numpy u64 array
        [len(positions), num_cols * words], Montgomery words, gathered on the device."""
        return self.get_rows_launch(positions)()

    def get_rows_launch(self, positions, batch=None):
        """Launches the gather and returns the function that fetches its result (callers start every gather of a phase first)."""
        pl, L = self.planner, self.planner.lib
        pos = np.asarray(positions, dtype=np.uint64)
        words = self.num_cols() * FIELD_WORDS[self.field]
        if len(pos) == 0:
            return lambda: np.zeros((0, words), dtype=np.uint64)
        ptr, read, keep = _gather_slot(pl, len(pos) * words * 8, batch)
        L.check(L.ms_gather_rows(pl.handle, self.field, self.num_rows(), _ptr_array(self.columns), self.num_cols(), pos.ctypes.data, len(pos), ptr))
        return lambda _keep=keep: np.array(read()[: len(pos) * words * 8]).view(np.uint64).reshape(len(pos), words)

    def hash_rows(self, hash="sha256"):
        """`hash_rows::<F, H>` (src/merkle.rs:412-436, src/matrix.rs:254-280): one digest per row ->
        DeviceBytes of num_rows x 32.  H = Sha256HashFn ("sha256") or RPO-256 ("rpo256", Goldilocks columns)."""
        pl = self.planner
        n = self.num_rows()
        leaves = DeviceBytes(pl, n * 32)
        if hash == "sha256":
            pl.lib.check(pl.lib.ms_sha256_rows(pl.handle, self.field, n, _ptr_array(self.columns), len(self.columns), leaves.ptr))
        elif hash == "rpo256":
            pl.lib.check(pl.lib.ms_rpo256_rows_field(pl.handle, self.field, n, _ptr_array(self.columns), len(self.columns), leaves.ptr))
        else:
            raise ValueError(f"unknown hash {hash!r} (one of {HASHES})")
        return leaves

    def lde(self, blowup, offset=GL_GENERATOR, bit_reversed=True):
        """Fused `interpolate(trace_domain)` + `bit_reversed_evaluate(lde_domain)`
        (src/prover.rs:50-51) in one call; returns a new Matrix, self is preserved."""
        L = self.planner.lib
        n = self.num_rows()
        if n & (n - 1) or blowup < 1 or blowup & (blowup - 1):
            raise ValueError("the number of rows and the blow-up factor must be powers of two")
        log_n, log_b = n.bit_length() - 1, blowup.bit_length() - 1
        outs = [GpuVec(self.planner, n * blowup, self.field) for _ in self.columns]
        off = _offset_words(self.field, offset)
        L.check(L.ms_lde(self.planner.handle, self.field, log_n, log_b, off.ctypes.data,
                         _ptr_array(self.columns), _ptr_array(outs), len(outs), 1 if bit_reversed else 0))
        return Matrix(outs)


def _offset_words(field, offset):
    """A domain offset (canonical int of the FFT field) as the Montgomery words the C ABI takes."""
    if field == STARK252_FP:
        return np.ascontiguousarray(f252_to_mont_limbs(offset % F252_P), dtype=np.uint64)
    return np.array([gl_to_mont(offset % GL_P)], dtype=np.uint64)


def scan_affine(a, b, init, inclusive=False):
    """The sequential loops that build extension columns (examples/brainfuck/trace.rs:108-289):
    state = init; for every row: out[row] = state; state = a[row] * state + b[row].  `a` or `b` may
    be None (all ones / all zeros: a running sum / a running product).  a, b: GpuVec of one field
    (Goldilocks Fp or Fq3); init: numpy u64 Montgomery words of one element."""
    ref = a if a is not None else b
    pl, L = ref.planner, ref.planner.lib
    out = GpuVec(pl, len(ref), ref.field)
    init = np.ascontiguousarray(init, dtype=np.uint64)
    L.check(L.ms_scan_affine(pl.handle, ref.field, len(ref), a.ptr if a is not None else None, b.ptr if b is not None else None,
                             init.ctypes.data, int(bool(inclusive)), out.ptr))
    return out


def running_product(factors, init):
    """ext[row] = init * prod(factors[:row])  (the permutation columns, trace.rs:131-145)."""
    return scan_affine(factors, None, init)


class Queries:
    """`Queries::new` (src/trace.rs:113-157): the rows of the base / extension / composition LDEs at the
    query positions and the batched Merkle openings of the three trees, all gathered on the device."""

    def __init__(self, base_trace_lde, extension_trace_lde, composition_trace_lde, base_tree, extension_tree, composition_tree, positions,
                 batch=None):
        positions = [int(p) for p in positions]
        none = lambda: None
        deferred = batch is not None                                        # the caller's batch: it fetches when its other gathers are launched too
        batch = batch if batch is not None else GatherBatch(base_trace_lde.planner)
        launched = [base_tree.prove_launch(positions, batch),               # every gather is in flight before the first download
                    extension_tree.prove_launch(positions, batch) if extension_tree is not None else none,
                    composition_tree.prove_launch(positions, batch),
                    base_trace_lde.get_rows_launch(positions, batch),
                    extension_trace_lde.get_rows_launch(positions, batch) if extension_trace_lde is not None else none,
                    composition_trace_lde.get_rows_launch(positions, batch)]
        self._launched = launched
        if not deferred:
            self.fetch()

    def fetch(self):
        """downloads (once per batch) and assembles the six members; called by the constructor's users that passed their own batch"""
        launched = self._launched
        (self.base_trace_proof, self.extension_trace_proof, self.composition_trace_proof, self.base_trace_values,
         self.extension_trace_values, self.composition_trace_values) = [f() for f in launched]
        return self


def apply_drp(evals, alpha, folding_factor, domain_offset=1):
    """`apply_drp(evals, domain_offset, alpha, folding_factor)` (src/fri.rs:526-567): `evals`
    is a GpuVec in bit-reversed order; returns the next layer's evaluations (bit-reversed).
    `alpha`: numpy u64 limbs (Montgomery) of one element of the column's field."""
    pl = evals.planner
    n = len(evals)
    out = GpuVec(pl, n // folding_factor, evals.field)
    al = np.ascontiguousarray(alpha, dtype=np.uint64).ravel()
    assert al.size == FIELD_WORDS[evals.field]
    off = _offset_words(evals.field, domain_offset)
    pl.lib.check(pl.lib.ms_fri_fold(pl.handle, evals.field, n.bit_length() - 1, folding_factor, al.ctypes.data,
                                    off.ctypes.data, evals.ptr, out.ptr))
    return out


class GpuRpo256ColumnMajor:
    """`GpuRpo256ColumnMajor::new(n, requires_padding)` / `update(col)` / `finish()` (gpu/src/plan.rs:32-107):
    row-wise RPO-256 digests of equally long Fp columns.  `finish()` returns a GpuVec of n x 4 elements."""
    RATE = 8

    def __init__(self, n, requires_padding=None, planner=None):
        self.n = n
        self.planner = planner or get_planner()
        self.cols = []
        self.requires_padding = requires_padding

    def update(self, col):
        if len(col) != self.n or col.field != GOLDILOCKS_FP:
            raise ValueError("column of the wrong length or field")
        self.cols.append(col)

    def finish(self):
        if not self.cols:
            raise ValueError("the zero-length input is not allowed")            # plan.rs:72
        if self.requires_padding is not None and self.requires_padding != (len(self.cols) % self.RATE != 0):
            raise ValueError("requires_padding does not match the number of columns absorbed")
        pl = self.planner
        out = GpuVec(pl, self.n * 4, GOLDILOCKS_FP)
        pl.lib.check(pl.lib.ms_rpo256_rows(pl.handle, self.n, _ptr_array(self.cols), len(self.cols), out.ptr))
        return out


class GpuRpo256RowMajor:
    """`GpuRpo256RowMajor` (gpu/src/plan.rs:109-148): rows of 8 Fp elements, one absorb per update."""

    def __init__(self, n, requires_padding=False, planner=None):
        self.n, self.planner, self.rows = n, planner or get_planner(), None

    def update(self, rows):
        if len(rows) != self.n * 8:
            raise ValueError("expected n rows of 8 elements")
        self.rows = rows

    def finish(self):
        if self.rows is None:
            raise ValueError("the zero-length input is not allowed")            # plan.rs:141-146 panic!()
        pl = self.planner
        out = GpuVec(pl, self.n * 4, GOLDILOCKS_FP)
        pl.lib.check(pl.lib.ms_rpo256_rows_row_major(pl.handle, self.n, 8, self.rows.ptr, out.ptr))
        return out


def gen_rpo_merkle_tree(leaves):
    """`gen_rpo_merkle_tree(leaves: &[[F; 4]])` (gpu/src/plan.rs:150-174) -> GpuVec of n x 4 node elements."""
    pl = leaves.planner
    n = len(leaves) // 4
    nodes = GpuVec(pl, n * 4, GOLDILOCKS_FP)
    pl.lib.check(pl.lib.ms_rpo256_merkle(pl.handle, n, leaves.ptr, nodes.ptr))
    return nodes


def grind_proof_of_work(planner, seed, proof_of_work_bits, max_nonce=(1 << 40)):
    """`PublicCoin::grind_proof_of_work(bits)` (src/random.rs:48-55): the smallest nonce >= 1 whose
    SHA-256(seed || nonce_be) has `bits` leading zero bits.  seed: 32 bytes."""
    seed = bytes(seed)
    assert len(seed) == 32
    out = ctypes.c_uint64(0)
    buf = ctypes.create_string_buffer(seed, 32)
    planner.lib.check(planner.lib.ms_sha256_pow_grind(planner.handle, buf, proof_of_work_bits, max_nonce, ctypes.byref(out)))
    return out.value
