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


def eval_constraints_sharded(prog, planner, comm, challenges, hints, lde_step, domain_offset, n, base_shard, ext_shard=()):
    """Constraint evaluation on this rank's ROW shard of the committed (bit-reversed) LDE, without communication,
    for lde_step a multiple of the number of ranks G (blow-up >= G, e.g. 8 GPUs and blow-up 8 or 16).

    After ms_cols_to_rows_alltoall rank r holds positions R = r N/G + R' of every column, i.e. the natural indices
    i = bitrev(R) = G i' + rho with rho = bitrev_g(r), i' = bitrev(R').  Those are the points x_i = (h w^rho) (w^G)^i' -- a
    coset of the subgroup of order N/G -- and a rotation by lde_step * offset rows moves i' by (lde_step / G) * offset
    and leaves rho alone: the shard is a self-contained evaluation problem of size N/G with offset h w^rho and step
    lde_step / G (`eval_cpu::eval`'s arguments, src/eval_cpu.rs:33-42), on which the ordinary evaluator runs.
    Concatenating the ranks' results in rank order gives the bit-reversed evaluation vector of the whole domain.
    (When G does not divide lde_step a rotated row lives on another rank; that exchange is not implemented.)"""
    from . import expr as E
    from .api import GL_P, Radix2EvaluationDomain
    G, r = comm.world, comm.rank
    if lde_step % G:
        raise ValueError(f"row-sharded evaluation needs lde_step ({lde_step}) to be a multiple of the number of ranks ({G})")
    g = G.bit_length() - 1
    rho = int(format(r, f"0{g}b")[::-1], 2) if g else 0
    w = Radix2EvaluationDomain(n).group_gen
    shard_offset = (domain_offset * pow(w, rho, GL_P)) % GL_P
    return E.eval(prog, planner, challenges, hints, lde_step // G, shard_offset, n // G, list(base_shard), list(ext_shard), bit_reversed=True)
