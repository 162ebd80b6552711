"""Multi-GPU commitment of a column-sharded trace; new work, the reference is
single-device (one metal::Device, gpu/src/plan.rs:465-468).

One process per GPU.  What the reference's prover does on one device (src/prover.rs:50-52):

    lde  = trace.interpolate(trace_domain).bit_reversed_evaluate(lde_domain)
    tree = MerkleTree::from_matrix(&lde)

shards as follows:
  1. columns are independent (Matrix = Vec of columns, src/matrix.rs:26): rank g owns columns
     {c : c mod G == g} and runs the fused LDE on them -- no communication;
  2. a Merkle leaf hashes one element of EVERY column of a row (src/merkle.rs:428-431), so one
     exchange step turns column shards into row shards: rank r receives rows
     [r*N/G, (r+1)*N/G) of every column.  ms_cols_to_rows_alltoall sends the row blocks straight out
     of the LDE columns as RCCL send/recv pairs in one group (every pair of GPUs talks directly, all
     xGMI links carry payload at once);
  3. each rank hashes its rows and builds the Merkle subtree over them: its root is node G + r of
     the single-device tree (nodes[k] has children 2k, 2k+1, src/merkle.rs:145-147);
  4. ms_allgather_digests of the G subtree roots (32 B each) and log2(G) more hash levels give every
     rank the same root as `MerkleTree::from_matrix` on one device, byte for byte.

Everything on the data path is the C ABI (include/ministark_hip.h, "multi-GPU exchange"); this file
only sequences the calls.  The communicator's 128-byte id reaches the ranks through whatever the
host launcher offers -- `RcclComm.from_torch_distributed` uses the torch.distributed store, nothing
else of torch is involved.
"""
import ctypes

import numpy as np

from .api import FIELD_WORDS, GOLDILOCKS_FP, GL_GENERATOR, DeviceBytes, GpuVec, Matrix, MerkleTree, _ptr_array, _merkle_view_ids_arrays

COMM_ID_BYTES = 128


def owned_columns(total_cols, rank, world):
    return list(range(rank, total_cols, world))


class P2POp(ctypes.Structure):                        # ms_p2p_op
    _fields_ = [("kind", ctypes.c_uint32), ("peer", ctypes.c_uint32), ("d_ptr", ctypes.c_void_p), ("bytes", ctypes.c_uint64)]


class XchgOp(ctypes.Structure):                       # ms_xchg_op (include/ministark_hip.h)
    _fields_ = [("kind", ctypes.c_uint32), ("peer", ctypes.c_uint32), ("src_col", ctypes.c_uint32), ("dst_col", ctypes.c_uint32),
                ("src_offset", ctypes.c_uint64), ("bytes", ctypes.c_uint64)]


XCHG_SEND, XCHG_RECV, XCHG_COPY = 0, 1, 2


def exchange_schedule(lib, world, rank, total_cols, blk_bytes):
    """ms_cols_to_rows_schedule: the point-to-point operations ms_cols_to_rows_alltoall issues on `rank`, in order."""
    my_ncols = len(owned_columns(total_cols, rank, world))
    count = ctypes.c_size_t(0)
    lib.check(lib.ms_cols_to_rows_schedule(world, rank, my_ncols, total_cols, blk_bytes, None, 0, ctypes.byref(count)))
    ops = (XchgOp * max(1, count.value))()
    lib.check(lib.ms_cols_to_rows_schedule(world, rank, my_ncols, total_cols, blk_bytes, ops, count.value, ctypes.byref(count)))
    return [ops[k] for k in range(count.value)]


class RcclComm:
    """The exchange steps of the sharded commitment on one rank: ms_comm_init / ms_cols_to_rows_alltoall /
    ms_allgather_digests over RCCL.  `planner` must be this rank's own context (its own GPU)."""

    def __init__(self, planner, rank, world, unique_id):
        if world < 1 or world & (world - 1):
            raise ValueError("world size must be a power of two (Merkle subtrees)")
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("the communicator id is 128 bytes")
        self.planner, self.rank, self.world = planner, rank, world
        buf = ctypes.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        planner.lib.check(planner.lib.ms_comm_init(planner.handle, world, rank, buf))

    @staticmethod
    def unique_id(lib):
        """ncclGetUniqueId; called by ONE rank, the bytes go to the others over a host channel."""
        buf = ctypes.create_string_buffer(COMM_ID_BYTES)
        lib.check(lib.ms_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_torch_distributed(cls, planner, group=None):
        """Rank / world size / id exchange from an initialised torch.distributed process group (any backend)."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id(planner.lib) if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        return cls(planner, rank, world, box[0])

    def cols_to_rows(self, my_cols, total_cols, nrows, field=GOLDILOCKS_FP):
        """my_cols: this rank's columns of the whole domain (GpuVecs of `nrows` elements of `field`).  -> one GpuVec per
        column of the matrix, holding this rank's rows.  The geometry is given by the caller, never inferred from my_cols:
        a rank that owns no column (fewer columns than ranks) must post receives of the same size its peers send."""
        pl, L = self.planner, self.planner.lib
        if nrows % self.world:
            raise ValueError("rows do not split over the ranks")
        for c in my_cols:
            if len(c) != nrows or c.field != field:
                raise ValueError("column of the wrong length or field")
        if self.world == 1:                                     # one rank owns every column and every row: the shard IS the columns
            if len(my_cols) != total_cols:
                raise ValueError(f"rank 0 of 1 owns {total_cols} of {total_cols} columns, {len(my_cols)} given")
            return list(my_cols)
        shard = [GpuVec(pl, nrows // self.world, field) for _ in range(total_cols)]
        L.check(L.ms_cols_to_rows_alltoall(pl.handle, field, nrows, _ptr_array(my_cols), len(my_cols), total_cols, _ptr_array(shard)))
        return shard

    def p2p(self, ops):
        """ops: [(XCHG_SEND | XCHG_RECV, peer, device pointer, bytes)] -> one RCCL group (ms_p2p_batch)."""
        if not ops:
            return
        arr = (P2POp * len(ops))(*[P2POp(k, peer, ptr, nbytes) for k, peer, ptr, nbytes in ops])
        L = self.planner.lib
        L.check(L.ms_p2p_batch(self.planner.handle, arr, len(ops)))

    def allgather_digests(self, my_digest_ptr):
        out = DeviceBytes(self.planner, 32 * self.world)
        L = self.planner.lib
        L.check(L.ms_allgather_digests(self.planner.handle, my_digest_ptr, out.ptr))
        return out

    def close(self):
        if self.planner is not None and self.planner.handle:
            self.planner.lib.ms_comm_destroy(self.planner.handle)
        self.planner = None


def lde_commit_sharded(planner, comm, local_cols, total_cols, log_n, log_blowup, offset=GL_GENERATOR, field=GOLDILOCKS_FP,
                       hash="sha256"):
    """local_cols: this rank's trace columns (numpy u64 Montgomery words, or GpuVecs), in the order of
    owned_columns(total_cols, rank, world).  `comm` provides rank, world, cols_to_rows and allgather_digests
    (RcclComm on GPUs).  Returns (root_bytes, my_row_shard): my_row_shard is a list over ALL columns of
    GpuVecs holding this rank's rows of the bit-reversed LDE."""
    rank, world = comm.rank, comm.world
    N = 1 << (log_n + log_blowup)
    if N % world or N // world < 2:
        raise ValueError("LDE domain too small for this many ranks")
    mine = owned_columns(total_cols, rank, world)
    if len(local_cols) != len(mine):
        raise ValueError(f"rank {rank} of {world} owns {len(mine)} of {total_cols} columns, {len(local_cols)} given")
    V = FIELD_WORDS[field]

    # 1. local fused LDE (interpolate + bit-reversed coset evaluation), no communication
    vecs = [c if isinstance(c, GpuVec) else GpuVec.from_numpy(planner, np.asarray(c, dtype=np.uint64), field) for c in local_cols]
    for v in vecs:
        if len(v) != (1 << log_n) or v.words != (1 << log_n) * V:
            raise ValueError("column of the wrong length or field")
    lde = Matrix(vecs).lde(1 << log_blowup, offset, True).columns if vecs else []

    # 2. column shards -> row shards
    shard = comm.cols_to_rows(lde, total_cols, N, field) if total_cols else []

    # 3. hash my rows, build my subtree
    tree = MerkleTree.from_matrix(Matrix(shard), hash)
    if world == 1:
        return tree.root(), shard

    # 4. all-gather the subtree roots (nodes[1] of every rank), finish the top log2(G) levels with the same kernels
    roots = comm.allgather_digests(tree.nodes.ptr + 32)
    top = MerkleTree(planner, roots, world, hash)
    return top.root(), shard


def eval_constraints_sharded(prog, planner, comm, challenges, hints, lde_step, domain_offset, n, base_shard, ext_shard=(), n_lde=None):
    """Constraint evaluation on the ROW shards of the committed (bit-reversed) LDE that ms_cols_to_rows_alltoall leaves on
    the ranks.  n = trace_len * lde_step points of the constraint-evaluation domain (lde_step = the AIR's ce_blowup_factor,
    src/prover.rs:97-107); n_lde >= n: the LDE domain the shards belong to (rank r holds positions [r n_lde/G, (r+1) n_lde/G)).
    The first n positions of the bit-reversed LDE are the constraint-evaluation coset in its own bit-reversed order, so the
    evaluation rows live on the first G_ce = n G / n_lde ranks.

    Returns (GpuVec, first_position, count): this rank's slice of the bit-reversed evaluation vector, or None on a rank
    that holds no constraint-evaluation row.  Three cases:
      * n <= n_lde / G: rank 0 holds the whole coset as a prefix of its shard -- the ordinary evaluator, no communication;
      * G_ce divides lde_step: positions R = r rows + R' are the natural indices i = G_ce i' + rho, rho = bitrev(r): the coset
        (h w^rho)<w^G_ce>, and a rotation by lde_step * offset rows moves i' by (lde_step / G_ce) * offset and leaves rho
        alone -- a self-contained problem of size n / G_ce with offset h w^rho and step lde_step / G_ce, no communication;
      * otherwise (BASELINE configs[4]: blow-up 4, ce_blowup_factor 1, 8 GPUs -> G_ce = 2, lde_step = 1) a rotated row lives
        on another rank.  In the bit-reversed layout the natural-order neighbour of EVERY row is on the other rank (rows that
        differ in the low bits of i differ in the high bits of the position), so the "halo" of SURVEY.md 8(e) is a whole
        shard per peer: the G_ce ranks exchange their shards of every column (ms_p2p_batch, (G_ce - 1) rows columns s bytes
        into each of them), every one of them then holds the whole coset and evaluates it; the slice it owns is returned."""
    from . import expr as E
    from .api import GL_P, Radix2EvaluationDomain
    G, r = comm.world, comm.rank
    n_lde = n if n_lde is None else n_lde
    if n_lde % G or n_lde < n:
        raise ValueError("the LDE domain must split over the ranks and contain the constraint-evaluation domain")
    rows = n_lde // G
    base_shard, ext_shard = list(base_shard), list(ext_shard)
    if n <= rows:                                   # the whole coset is a prefix of rank 0's shard
        if r != 0:
            return None
        return E.eval(prog, planner, challenges, hints, lde_step, domain_offset, n, base_shard, ext_shard, bit_reversed=True), 0, n
    G_ce = n // rows
    if r >= G_ce:
        return None
    if lde_step % G_ce == 0:
        g = G_ce.bit_length() - 1
        rho = int(format(r, f"0{g}b")[::-1], 2) if g else 0
        w = Radix2EvaluationDomain(n).group_gen
        shard_offset = (domain_offset * pow(w, rho, GL_P)) % GL_P
        out = E.eval(prog, planner, challenges, hints, lde_step // G_ce, shard_offset, rows, base_shard, ext_shard, bit_reversed=True)
        return out, r * rows, rows
    # shards of every column -> the whole coset on each of the G_ce ranks
    L = planner.lib
    full, ops = [], []
    for col in base_shard + ext_shard:
        words = FIELD_WORDS[col.field]
        if len(col) != rows:
            raise ValueError("shard column of the wrong length")
        whole = GpuVec(planner, n, col.field)
        blk = rows * words * 8
        L.check(L.ms_copy(planner.handle, whole.ptr + r * blk, col.ptr, blk))
        for peer in range(G_ce):
            if peer != r:
                ops.append((XCHG_SEND, peer, col.ptr, blk))
                ops.append((XCHG_RECV, peer, whole.ptr + peer * blk, blk))
        full.append(whole)
    comm.p2p(ops)
    res = E.eval(prog, planner, challenges, hints, lde_step, domain_offset, n, full[:len(base_shard)], full[len(base_shard):], bit_reversed=True)
    mine = GpuVec(planner, rows, res.field)
    wq = FIELD_WORDS[res.field] * 8
    L.check(L.ms_copy(planner.handle, mine.ptr, res.ptr + r * rows * wq, rows * wq))
    return mine, r * rows, rows


# =====================================================================================================================
# The whole prover on row shards (BASELINE configs[4]: "full prover.rs ... columns sharded across 8 x MI355X")
# =====================================================================================================================
# After the base-trace commitment every later phase of default_prove (src/prover.rs:57-174) works on ROWS of the committed
# LDEs, so the column -> row exchange of the commitment is the only bulk movement of the trace:
#
#   phase (prover.rs)                    placement                                               moved per rank (C5, G = 8)
#   1 interpolate + LDE  (:50-51)        column shards (rank c mod G), no communication          --
#     commit             (:52-55)        cols_to_rows, hash own rows, subtree, 32-byte all-gather 7/8 of its LDE columns (112 MiB)
#   2 constraint evaluation (:97-107)    row shards (eval_constraints_sharded)                   0, or the CE coset's shards
#   3 composition trace (:111-124)       evaluations -> rank 0: iNTT + split; column c -> rank c mod G: LDE;   <= 32 MiB + 112 MiB
#                                        cols_to_rows, subtree, all-gather
#   4 out-of-domain evaluations (:137-146)  each polynomial by its owner (Horner), values gathered   a few hundred bytes
#   5 DEEP composition + its LDE (:149-152) ms_deep_rows on the rank's rows of both committed LDEs  0
#   6 FRI layers (fri.rs:179-231)        every layer stays sharded by rows: subtree + all-gather per commitment, fold in place   32 B per layer
#                                        (ms_fri_fold_rows); a layer with fewer than two leaves per rank is collected on rank 0
#   7 remainder, grinding, openings      rank 0; rows and digests of the openings come from their owners   kilobytes
#
# Every value is the one the single-device prover computes (tests/test_distributed.py compares roots, remainder, nonce and every
# opening at 2, 4 and 8 ranks): sharding changes where a row is hashed, folded or composed, not what is computed.

def _sharded_root(planner, comm, tree, hash):
    """(root of the whole tree, the top-levels tree over the G subtree roots or None at G = 1)"""
    if comm.world == 1:
        return tree.root(), None
    top = MerkleTree(planner, comm.allgather_digests(tree.nodes.ptr + 32), comm.world, hash)
    return top.root(), top


def _collect(planner, comm, buf, nbytes_of, root=0):
    """Device buffers of nbytes_of[rank] bytes on every rank -> on `root` one device buffer holding them in rank order (None elsewhere): a new
    DeviceBytes, or -- with one rank, where there is nothing to collect -- the caller's own `buf` (a DeviceBytes or a GpuVec), not a copy."""
    r, G = comm.rank, comm.world
    if G == 1 and buf is not None:                              # nothing to collect: the caller's buffer is the result
        return buf
    offs = np.concatenate([[0], np.cumsum(nbytes_of)]).astype(np.int64)
    out, ops = None, []
    if r == root:
        out = DeviceBytes(planner, max(8, int(offs[-1])))
        if nbytes_of[r]:
            planner.lib.check(planner.lib.ms_copy(planner.handle, out.ptr + int(offs[r]), buf.ptr, int(nbytes_of[r])))
        ops = [(XCHG_RECV, peer, out.ptr + int(offs[peer]), int(nbytes_of[peer])) for peer in range(G) if peer != root and nbytes_of[peer]]
    elif nbytes_of[r]:
        ops = [(XCHG_SEND, root, buf.ptr, int(nbytes_of[r]))]
    comm.p2p(ops)
    return out


def _allgather_words(planner, comm, words):
    """numpy u64 array (same length on every rank) -> [G, len] array on every rank (small host data through the device exchange)."""
    r, G = comm.rank, comm.world
    words = np.ascontiguousarray(words, dtype=np.uint64)
    if G == 1:
        return words.reshape(1, -1).copy()
    nb = words.size * 8
    buf = DeviceBytes(planner, max(8, nb * G))
    L = planner.lib
    if nb:
        L.check(L.ms_upload(planner.handle, buf.ptr + r * nb, words.ctypes.data, nb))
        ops = []
        for peer in range(G):
            if peer != r:
                ops += [(XCHG_SEND, peer, buf.ptr + r * nb, nb), (XCHG_RECV, peer, buf.ptr + peer * nb, nb)]
        comm.p2p(ops)
    return buf.to_numpy()[: nb * G].view(np.uint64).reshape(G, -1).copy()


class OpeningBatch:
    """The openings of a proof as ONE collective: every request (a Merkle view of a sharded tree, rows of a row-sharded matrix or FRI layer)
    registers what each rank contributes; `execute` lets every rank gather its contributions into one device buffer, collects the buffers on
    `root` in one exchange and one download, and returns the assembled results there (None elsewhere).  The sizes are functions of the
    positions alone, so every rank computes every rank's byte counts without talking."""

    EAGER_BYTES = 1 << 20                                       # one pooled block: the openings of a proof are a few hundred KiB

    def __init__(self, planner, comm, root=0):
        self.planner, self.comm, self.root, self.reqs = planner, comm, root, []
        self.arena, self.used, self.eager, self.launched = DeviceBytes(planner, self.EAGER_BYTES), 0, True, 0

    def add(self, sizes, run, parse):
        """sizes[q]: bytes rank q contributes; run(ptr): this rank's gathers into ptr .. ptr + sizes[rank]; parse(chunks): chunks[q] = rank q's bytes.
        The gathers are LAUNCHED HERE, behind the previous request's, while the host walks the next request's index lists (they all land
        in one arena in request order).  Once a request does not fit, it and every later one wait for `execute`, which moves what was
        gathered into a buffer of the full size (one device copy) and launches the rest behind it: every `run` is called exactly once."""
        sizes = list(sizes)
        mine = sizes[self.comm.rank]
        if self.eager and self.used + mine <= self.EAGER_BYTES:
            if mine:
                run(self.arena.ptr + self.used)
            self.used += mine
            self.launched = len(self.reqs) + 1
        else:
            self.eager = False
        self.reqs.append((sizes, run, parse))
        return len(self.reqs) - 1

    def execute(self):
        pl, comm, G, r = self.planner, self.comm, self.comm.world, self.comm.rank
        totals = [sum(sz[q] for sz, _, _ in self.reqs) for q in range(G)]
        arena = self.arena
        if not self.eager:
            arena = DeviceBytes(pl, max(8, totals[r]))
            if self.used:
                pl.lib.check(pl.lib.ms_copy(pl.handle, arena.ptr, self.arena.ptr, self.used))
            off = self.used
            for sz, run, _ in self.reqs[self.launched:]:
                if sz[r]:
                    run(arena.ptr + off)
                    off += sz[r]
        got = _collect(pl, comm, arena, totals, self.root)
        if r != self.root:
            return [None] * len(self.reqs)
        base = np.concatenate([[0], np.cumsum(totals)]).astype(np.int64)
        raw = got.to_numpy(int(base[-1]))                         # what was gathered, not the arena's capacity
        cur = [int(b) for b in base[:-1]]
        out = []
        for sz, _, parse in self.reqs:
            chunks = []
            for q in range(G):
                chunks.append(raw[cur[q]:cur[q] + sz[q]])
                cur[q] += sz[q]
            out.append(parse(chunks))
        return out


class ShardedTree:
    """A Merkle tree whose leaves are spread over the ranks in row blocks: rank r holds the subtree over leaves [r n / G, (r + 1) n / G)
    (node G + r of the whole tree) and every rank the top levels (nodes 1 .. G - 1)."""

    def __init__(self, planner, comm, local_tree, hash="sha256"):
        self.planner, self.comm, self.local, self.hash = planner, comm, local_tree, hash
        self.nleaves = local_tree.nleaves * comm.world
        self._root, self.top = _sharded_root(planner, comm, local_tree, hash)

    def root(self):
        return self._root

    def prove(self, indices, root=0):
        """`MerkleTreeImpl::prove` (src/merkle.rs:149-206) -> the MerkleView of the whole tree on `root` (None elsewhere): every digest
        is fetched by the rank that holds it and collected."""
        batch = OpeningBatch(self.planner, self.comm, root)
        self.request(batch, indices)
        return batch.execute()[0]

    def request(self, batch, indices):
        """registers this tree's view of `indices` with an OpeningBatch; -> the request's index in the batch's results"""
        pl, G, r = self.planner, self.comm.world, self.comm.rank
        n, per = self.nleaves, self.local.nleaves
        leaf_ids, initial, sibling, node_ids = _merkle_view_ids_arrays(n, [int(i) for i in indices], pl.lib)
        leaf_ids, node_ids = leaf_ids.astype(np.int64), node_ids.astype(np.int64)
        nl = len(leaf_ids)
        # where each digest lives: owner rank, kind (0 = leaf of the owner's subtree, 1 = node of it, 2 = replicated top level), local id
        lvl = np.int64(1) << (np.frexp(node_ids.astype(np.float64))[1].astype(np.int64) - 1)        # the level's first node (ids < 2^53: exact)
        top = lvl < G
        per_lvl = np.maximum(lvl // G, 1)
        j = node_ids - lvl
        owner = np.concatenate([leaf_ids // per, np.where(top, batch.root, j // per_lvl)])
        kind = np.concatenate([np.zeros(nl, dtype=np.int64), np.where(top, 2, 1)])
        local = np.concatenate([leaf_ids % per, np.where(top, node_ids, per_lvl + j % per_lvl)]).astype(np.uint64)
        # a rank's contribution: its leaves, then its nodes, then (on `root`) the top levels, each in the order of the walk
        key = owner * 3 + kind
        order = np.argsort(key, kind="stable")
        count = np.bincount(key, minlength=3 * G).reshape(G, 3)
        sizes = (32 * count.sum(axis=1)).tolist()
        srcs = ((self.local.leaves, per), (self.local.nodes, per), (self.top.nodes if self.top else None, G))

        def run(ptr):
            off = 0
            for k in range(3):
                ids = np.ascontiguousarray(local[key == 3 * r + k])
                if len(ids):
                    src, cnt = srcs[k]
                    pl.lib.check(pl.lib.ms_gather_digests(pl.handle, cnt, src.ptr, ids.ctypes.data, len(ids), ptr + off))
                    off += 32 * len(ids)

        def parse(chunks):
            got = np.concatenate([np.asarray(c, dtype=np.uint8) for c in chunks]).reshape(-1, 32)      # (owner, kind, walk) order = `order`
            digests = np.empty_like(got)
            digests[order] = got
            digests = digests.view("V32").ravel().tolist()                                              # 32-byte `bytes`, one C loop
            leaves, nodes = digests[:nl], digests[nl:]
            return {"nodes": nodes, "initial_leaves": [leaves[k] for k in initial.tolist()], "sibling_leaves": [leaves[k] for k in sibling.tolist()],
                    "height": n.bit_length() - 1}
        return batch.add(sizes, run, parse)


def _rows_request(batch, columns, field, positions, per):
    """rows `positions` (global row numbers, any order) of a matrix whose rows are spread over the ranks in blocks of `per` -> numpy
    [len(positions), words] in the order given"""
    pl, G, r = batch.planner, batch.comm.world, batch.comm.rank
    positions = [int(p) for p in positions]
    words = len(columns) * FIELD_WORDS[field]
    sizes = [sum(1 for p in positions if p // per == q) * words * 8 for q in range(G)]

    def run(ptr):
        pos = np.asarray([p - r * per for p in positions if p // per == r], dtype=np.uint64)
        pl.lib.check(pl.lib.ms_gather_rows(pl.handle, field, per, _ptr_array(columns), len(columns), pos.ctypes.data, len(pos), ptr))

    def parse(chunks):
        rows = [np.array(c).view(np.uint64).reshape(-1, words) for c in chunks]
        nxt = [0] * G
        out = np.zeros((len(positions), words), dtype=np.uint64)
        for k, p in enumerate(positions):
            q = p // per
            out[k] = rows[q][nxt[q]]
            nxt[q] += 1
        return out
    return batch.add(sizes, run, parse)


def _fri_rows_request(batch, layer_shard, folding, positions, per):
    """rows `positions` of Matrix::from_arrays(evaluations.as_chunks::<N>()) (src/fri.rs:213-215) of a row-sharded layer: `per` rows of
    `folding` evaluations on each rank.  A row is `folding` consecutive words = folding / 4 32-byte records of the shard, fetched with the
    digest gather (as pipeline.fri_layer_rows_launch does on one device); folding factor 2: picked on the host."""
    pl, G, r = batch.planner, batch.comm.world, batch.comm.rank
    rec = folding // 4
    sizes = [sum(1 for p in positions if p // per == q) * folding * 8 for q in range(G)]

    def run(ptr):
        mine = [p - r * per for p in positions if p // per == r]
        if folding % 4:
            picked = np.ascontiguousarray(layer_shard.to_numpy().reshape(-1, folding)[mine])
            pl.lib.check(pl.lib.ms_upload(pl.handle, ptr, picked.ctypes.data, picked.nbytes))
        else:
            ids = np.asarray([p * rec + k for p in mine for k in range(rec)], dtype=np.uint64)
            pl.lib.check(pl.lib.ms_gather_digests(pl.handle, per * rec, layer_shard.ptr, ids.ctypes.data, len(ids), ptr))

    def parse(chunks):
        rows = [np.array(c).view(np.uint64).reshape(-1, folding) for c in chunks]
        nxt = [0] * G
        out = np.zeros((len(positions), folding), dtype=np.uint64)
        for k, p in enumerate(positions):
            q = p // per
            out[k] = rows[q][nxt[q]]
            nxt[q] += 1
        return out
    return batch.add(sizes, run, parse)


def prove_sharded(planner, comm, local_cols, total_cols, log_rows, comp_expr, draws, blowup=4, folding=8, max_remainder_coeffs=64,
                  grinding_bits=8, hash="sha256", ce_blowup=None, phases_ms=None):
    """`pipeline.prove_phases` with the trace's columns spread over the ranks (local_cols: this rank's columns, owned_columns order;
    Fq = Fp AIRs over Goldilocks).  Returns on every rank dict(base_root, composition_root, fri_roots) and on rank 0 also ood,
    remainder_coeffs, nonce, queries (the six members of api.Queries as a dict) and fri_openings -- the values of the single-device
    prover.  See the placement table above.  phases_ms: a dict that receives this rank's wall time per phase (a device sync each)."""
    import time
    from . import expr as E
    from .api import Radix2EvaluationDomain, apply_drp, gl_to_mont, grind_proof_of_work, _offset_words
    from .composer import DeepPolyComposer
    from .pipeline import _lowered, fold_positions
    pl, L, G, r = planner, planner.lib, comm.world, comm.rank
    n_t = 1 << log_rows
    N = n_t * blowup
    ce_blowup = blowup if ce_blowup is None else ce_blowup
    n_ce = n_t * ce_blowup
    if N % G or N // G < 2 * folding:
        raise ValueError("LDE domain too small for this many ranks")
    rows = N // G
    mine = owned_columns(total_cols, r, G)
    if len(local_cols) != len(mine):
        raise ValueError(f"rank {r} of {G} owns {len(mine)} of {total_cols} columns, {len(local_cols)} given")
    trace_dom, lde_dom, ce_dom = Radix2EvaluationDomain(n_t), Radix2EvaluationDomain(N, 7), Radix2EvaluationDomain(n_ce, 7)
    prog = _lowered(comp_expr, total_cols)
    ch = np.array([gl_to_mont(c) for c in draws.challenges], dtype=np.uint64).reshape(-1, 1)
    hints = np.array([gl_to_mont(c) for c in draws.hints], dtype=np.uint64).reshape(-1, 1)
    out = {}
    t_lap = [time.perf_counter()]

    def lap(name):
        if phases_ms is not None:
            pl.sync()
            now = time.perf_counter()
            phases_ms[name] = phases_ms.get(name, 0.0) + (now - t_lap[0]) * 1e3
            t_lap[0] = now

    # 1. base trace: column shards -> LDE -> row shards -> commitment
    vecs = [c if isinstance(c, GpuVec) else GpuVec.from_numpy(pl, np.asarray(c, dtype=np.uint64)) for c in local_cols]
    base_polys = Matrix(vecs).interpolate(trace_dom) if vecs else None
    lde_local = base_polys.bit_reversed_evaluate(lde_dom).columns if vecs else []
    base_shard = comm.cols_to_rows(lde_local, total_cols, N)
    del lde_local
    tree_b = ShardedTree(pl, comm, MerkleTree.from_matrix(Matrix(base_shard), hash), hash)
    out["base_root"] = tree_b.root()
    lap("base trace: interpolate + LDE + exchange + commit")

    # 2. constraint evaluation on the row shards
    got = eval_constraints_sharded(prog, pl, comm, ch, hints, ce_blowup, 7, n_ce, base_shard, n_lde=N)
    lap("constraint evaluation")

    # 3. composition trace: evaluations -> rank 0 (iNTT, split) -> column owners (LDE) -> row shards -> commitment
    holders = 1 if n_ce <= rows else n_ce // rows
    nb = [(min(rows, n_ce) * 8 if q < holders else 0) for q in range(G)]
    evals_all = _collect(pl, comm, got[0] if got is not None else None, nb, 0)
    comp_cols_local = []
    comp_owned = owned_columns(ce_blowup, r, G)
    if r == 0:
        if G == 1 and isinstance(evals_all, GpuVec) and len(evals_all) == n_ce:
            ev = evals_all                                      # the evaluator's own output column
        else:
            ev = GpuVec(pl, n_ce)
            L.check(L.ms_copy(pl.handle, ev.ptr, evals_all.ptr, n_ce * 8))
        comp_poly = Matrix([ev]).bit_reverse_rows().into_polynomials(ce_dom).columns[0]
        comp_all = Matrix.from_chunks(comp_poly, ce_blowup).columns
        ops = [(XCHG_SEND, c % G, comp_all[c].ptr, n_t * 8) for c in range(ce_blowup) if c % G != 0]
        comm.p2p(ops)
        comp_cols_local = [comp_all[c] for c in comp_owned]
    else:
        comp_cols_local = [GpuVec(pl, n_t) for _ in comp_owned]
        comm.p2p([(XCHG_RECV, 0, v.ptr, n_t * 8) for v in comp_cols_local])
    comp_polys = Matrix(comp_cols_local) if comp_cols_local else None
    comp_lde_local = comp_polys.bit_reversed_evaluate(lde_dom).columns if comp_polys is not None else []
    comp_shard = comm.cols_to_rows(comp_lde_local, ce_blowup, N)
    del comp_lde_local
    tree_c = ShardedTree(pl, comm, MerkleTree.from_matrix(Matrix(comp_shard), hash), hash)
    out["composition_root"] = tree_c.root()
    lap("composition trace: gather + iNTT + split + LDE + exchange + commit")

    # 4. out-of-domain evaluations: every polynomial by its owner, the values to everybody
    args = list(draws.trace_args)
    helper = DeepPolyComposer.for_row_shards(args, n_t, draws.z, pl, total_cols, 0, ce_blowup, None)
    vals = np.zeros(len(args) + ce_blowup, dtype=np.uint64)
    q = [(k, mine.index(c), helper._point(o)) for k, (c, o) in enumerate(args) if c in mine] if base_polys is not None else []
    z_n = pow(draws.z, ce_blowup, (1 << 64) - (1 << 32) + 1)
    parts = []                                                 # one launch and one download for the rank's trace and composition-trace polynomials
    if base_polys is not None:
        parts.append((base_polys, GOLDILOCKS_FP, [(lc, p) for _, lc, p in q]))
    if comp_polys is not None:
        parts.append((comp_polys, GOLDILOCKS_FP, [(lc, z_n) for lc in range(len(comp_owned))]))
    res = helper._horner(parts)
    if base_polys is not None:
        for (k, _, _), v in zip(q, res[0]):
            vals[k] = v
    if comp_polys is not None:
        for lc, v in enumerate(res[-1]):
            vals[len(args) + comp_owned[lc]] = v
    allv = _allgather_words(pl, comm, vals)
    execution = [int(allv[c % G][k]) for k, (c, _) in enumerate(args)]
    composition = [int(allv[c % G][len(args) + c]) for c in range(ce_blowup)]
    out["ood"] = (execution, composition)

    # 5. the DEEP composition polynomial's LDE = the first FRI layer, on this rank's rows of both committed LDEs
    composer = DeepPolyComposer.for_row_shards(args, n_t, draws.z, pl, total_cols, 0, ce_blowup, (execution, composition))
    cur = composer.into_deep_evaluations(draws.deep, Matrix(base_shard), None, Matrix(comp_shard), N, first=r * rows)
    lap("DEEP: OOD evaluations + composition on the LDE rows")

    # 6. FRI: every layer sharded by rows while a rank holds at least two leaves of it
    n, sharded = N, True
    roots, layers = [], []                                       # layers: (evaluations, tree, sharded?, size)
    for alpha in draws.fri_alphas:
        if sharded and (n // G) // folding < 2:
            sharded = False
            if G > 1:
                whole = _collect(pl, comm, cur, [n // G * 8] * G, 0)
                cur = None
                if r == 0:
                    cur = GpuVec(pl, n)
                    L.check(L.ms_copy(pl.handle, cur.ptr, whole.ptr, n * 8))
        al = np.array([gl_to_mont(alpha)], dtype=np.uint64)
        if sharded:
            tree = ShardedTree(pl, comm, MerkleTree.from_fri_layer(cur, folding, hash), hash)
            roots.append(tree.root())
            layers.append((cur, tree, True, n))
            nch = n // G // folding
            nxt = GpuVec(pl, nch)
            L.check(L.ms_fri_fold_rows(pl.handle, GOLDILOCKS_FP, n.bit_length() - 1, folding, al.ctypes.data, _offset_words(GOLDILOCKS_FP, 1).ctypes.data,
                                       r * nch, nch, cur.ptr, nxt.ptr))
            cur = nxt
        elif r == 0:
            tree = MerkleTree.from_fri_layer(cur, folding, hash)
            roots.append(tree.root())
            layers.append((cur, tree, False, n))
            cur = apply_drp(cur, al, folding, 1)
        n //= folding
    if sharded and G > 1:
        whole = _collect(pl, comm, cur, [n // G * 8] * G, 0)
        cur = None
        if r == 0:
            cur = GpuVec(pl, n)
            L.check(L.ms_copy(pl.handle, cur.ptr, whole.ptr, n * 8))
    # the roots of layers committed after the switch to rank 0 reach the other ranks as 32-byte words
    have = np.zeros(4 * len(draws.fri_alphas), dtype=np.uint64)
    if r == 0:
        for k, rt in enumerate(roots):
            have[4 * k:4 * k + 4] = np.frombuffer(rt, dtype=np.uint64)
    allr = _allgather_words(pl, comm, have)[0]
    out["fri_roots"] = [allr[4 * k:4 * k + 4].tobytes() for k in range(len(draws.fri_alphas))]
    lap("FRI layers (commit + fold)")
    # 7. remainder, grinding (rank 0), openings (collective)
    if r == 0:
        rem = Matrix([cur.clone()]).bit_reverse_rows().into_polynomials(Radix2EvaluationDomain(n)).columns[0]
        out["remainder_coeffs"] = rem.to_numpy()[: max(n // blowup, 1)]
        out["nonce"] = grind_proof_of_work(pl, roots[-1] if roots else out["composition_root"], grinding_bits)
    lap("remainder + proof of work")
    positions = [int(p) for p in draws.positions]
    if G == 1:
        # one rank holds every row and every digest: the openings are the single-device ones (no owners to compute, nothing to collect) --
        # the same gathers into one buffer and one download that pipeline.prove_phases issues
        from .api import GatherBatch, Queries
        from .pipeline import fri_layer_rows_launch
        gb = GatherBatch(pl)
        qs = Queries(Matrix(base_shard), None, Matrix(comp_shard), tree_b.local, None, tree_c.local, positions, gb)
        pos, launched = sorted(set(positions)), []
        for lay, tree, _, _size in layers:
            pos = fold_positions(pos, folding)
            local_tree = tree.local if isinstance(tree, ShardedTree) else tree
            launched.append((pos, fri_layer_rows_launch(lay, folding, pos, gb), local_tree.prove_launch(pos, gb)))
        lap("openings: index walks + requests")
        gb.fetch()
        lap("openings: gathers + exchange + download")
        qs.fetch()
        out["queries"] = {"base_trace_proof": qs.base_trace_proof, "composition_trace_proof": qs.composition_trace_proof,
                          "base_trace_values": qs.base_trace_values, "composition_trace_values": qs.composition_trace_values,
                          "extension_trace_proof": None, "extension_trace_values": None}
        out["fri_openings"] = [{"positions": p_, "rows": rows(), "proof": proof()} for p_, rows, proof in launched]
        lap("openings: assembly")
        return out
    batch = OpeningBatch(pl, comm)                              # every opening of the proof: one exchange, one download
    kq = {"base_trace_proof": tree_b.request(batch, positions), "composition_trace_proof": tree_c.request(batch, positions),
          "base_trace_values": _rows_request(batch, base_shard, GOLDILOCKS_FP, positions, rows),
          "composition_trace_values": _rows_request(batch, comp_shard, GOLDILOCKS_FP, positions, rows)}
    pos, kfri, local_fri = sorted(set(positions)), [], []
    for k in range(len(draws.fri_alphas)):
        pos = fold_positions(pos, folding)
        if k < len(layers) and layers[k][2]:                     # a sharded layer: rows of `folding` evaluations, owners by row block
            lay, tree, _, size = layers[k]
            kfri.append((pos, _fri_rows_request(batch, lay, folding, pos, size // G // folding), tree.request(batch, pos)))
        else:
            kfri.append((pos, None, None))
            if r == 0:                                           # a layer that was collected on rank 0: the single-device gathers
                from .pipeline import fri_layer_rows_launch
                lay, tree, _, size = layers[k]
                local_fri.append((k, fri_layer_rows_launch(lay, folding, pos), tree.prove_launch(pos)))
    lap("openings: index walks + requests")
    res = batch.execute()
    lap("openings: gathers + exchange + download")
    q, openings = None, None
    if r == 0:
        q = {name: res[k] for name, k in kq.items()}
        q["extension_trace_proof"] = q["extension_trace_values"] = None
        openings = [{"positions": p, "rows": res[kr], "proof": res[kp]} if kr is not None else None for p, kr, kp in kfri]
        for k, rows_f, proof_f in local_fri:
            openings[k] = {"positions": kfri[k][0], "rows": rows_f(), "proof": proof_f()}
    if r == 0:
        out["queries"], out["fri_openings"] = q, openings
    lap("openings: assembly")
    return out
