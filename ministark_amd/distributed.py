"""Multi-GPU commitment of a column-sharded trace; new work, the reference is
single-device (one metal::Device, gpu/src/plan.rs:465-468).

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests).  What the reference's prover does on one device (src/prover.rs:50-52):

    lde  = trace.interpolate(trace_domain).bit_reversed_evaluate(lde_domain)
    tree = MerkleTree::from_matrix(&lde)

shards as follows:
  1. columns are independent (Matrix = Vec of columns, src/matrix.rs:26): rank g owns columns
     {c : c mod G == g} and runs the fused LDE on them -- no communication;
  2. a Merkle leaf hashes one element of EVERY column of a row (src/merkle.rs:428-431), so one
     exchange step turns column shards into row shards: rank r receives rows
     [r*N/G, (r+1)*N/G) of every column (point-to-point sends, every pair of GPUs talks directly,
     so all xGMI links carry payload at once);
  3. each rank hashes its rows and builds the Merkle subtree over them: its root is node G + r of
     the single-device tree (nodes[k] has children 2k, 2k+1, src/merkle.rs:145-147);
  4. an all-gather of the G subtree roots (32 B each) and log2(G) more hash levels give every
     rank the same root as `MerkleTree::from_matrix` on one device, byte for byte.
torch is plumbing only (device buffers that RCCL can see, the process group); every transform and
hash runs in libministark_hip.so.
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from .api import FIELD_WORDS, GOLDILOCKS_FP, GL_GENERATOR, DeviceBytes, GpuVec, Matrix, MerkleTree, gl_to_mont


def owned_columns(total_cols, rank, world):
    return list(range(rank, total_cols, world))


class _TensorVec(GpuVec):
    """A GpuVec living inside a torch tensor (so that torch.distributed can move it)."""

    def __init__(self, planner, tensor, field):
        self.tensor = tensor
        super().__init__(planner, tensor.numel() // FIELD_WORDS[field], field, ptr=tensor.data_ptr(), owner=False)


def lde_commit_sharded(planner, local_cols, total_cols, log_n, log_blowup, offset=GL_GENERATOR, field=GOLDILOCKS_FP,
                       group=None, device=None):
    """local_cols: this rank's trace columns (numpy u64, Montgomery words), in the order of
    owned_columns(total_cols, rank, world).  Returns (root_bytes, my_row_shard) where my_row_shard
    is a list over ALL columns of GpuVecs holding this rank's rows of the bit-reversed LDE."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world & (world - 1):
        raise ValueError("world size must be a power of two (Merkle subtrees)")
    V = FIELD_WORDS[field]
    N = 1 << (log_n + log_blowup)
    if N % world or N // world < 2:
        raise ValueError("LDE domain too small for this many ranks")
    rows = N // world
    mine = owned_columns(total_cols, rank, world)
    assert len(local_cols) == len(mine)
    dev = device if device is not None else torch.device("cpu")
    i64 = torch.int64                                      # torch has no u64 arithmetic; raw 8-byte words

    # 1. local fused LDE, bit-reversed, into one tensor [n_local, N*V]
    lde = torch.empty((max(len(mine), 1), N * V), dtype=i64, device=dev)
    if mine:
        m = Matrix([GpuVec.from_numpy(planner, c, field) for c in local_cols])
        outs = [_TensorVec(planner, lde[j], field) for j in range(len(mine))]
        off = ctypes.c_uint64(gl_to_mont(offset))
        L = planner.lib
        VP = ctypes.c_void_p
        L.check(L.ms_lde(planner.handle, field, log_n, log_blowup, ctypes.byref(off),
                         (VP * len(mine))(*[c.ptr for c in m.columns]), (VP * len(mine))(*[o.ptr for o in outs]), len(mine), 1))
        planner.sync()

    # 2. column shards -> row shards
    shard = torch.empty((total_cols, rows * V), dtype=i64, device=dev)
    ops, keep = [], []
    for peer in range(world):
        theirs = owned_columns(total_cols, peer, world)
        if peer == rank:
            for j, c in enumerate(mine):
                shard[c].copy_(lde[j, rank * rows * V:(rank + 1) * rows * V])
            continue
        if mine:
            snd = lde[: len(mine), peer * rows * V:(peer + 1) * rows * V].contiguous()
            keep.append(snd)
            ops.append(dist.P2POp(dist.isend, snd, peer, group))
        if theirs:
            rcv = torch.empty((len(theirs), rows * V), dtype=i64, device=dev)
            keep.append((rcv, theirs))
            ops.append(dist.P2POp(dist.irecv, rcv, peer, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for item in keep:
        if isinstance(item, tuple):
            rcv, theirs = item
            for j, c in enumerate(theirs):
                shard[c].copy_(rcv[j])
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)

    # 3. hash my rows, build my subtree
    cols = [_TensorVec(planner, shard[c], field) for c in range(total_cols)]
    tree = MerkleTree.from_matrix(Matrix(cols))
    my_root = np.frombuffer(tree.root(), dtype=np.uint8).copy()

    # 4. all-gather the subtree roots, finish the top log2(G) levels (same kernels)
    if world == 1:
        return tree.root(), cols
    roots = torch.empty((world, 32), dtype=torch.uint8, device=dev)
    mine_t = torch.from_numpy(my_root).to(dev)
    dist.all_gather_into_tensor(roots.view(-1), mine_t, group=group) if dev.type == "cuda" else \
        dist.all_gather(list(roots.unbind(0)), mine_t, group=group)
    top_leaves = DeviceBytes(planner, world * 32)
    host_roots = roots.cpu().numpy().copy()
    planner.lib.check(planner.lib.ms_upload(planner.handle, top_leaves.ptr, host_roots.ctypes.data, world * 32))
    top = MerkleTree(planner, top_leaves, world)
    return top.root(), cols
