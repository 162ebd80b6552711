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

from .api import FIELD_WORDS, GOLDILOCKS_FP, GL_GENERATOR, DeviceBytes, GpuVec, Matrix, MerkleTree, _ptr_array

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
