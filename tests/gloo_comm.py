"""TEST INFRASTRUCTURE: the exchange steps of ministark_amd.distributed over torch.distributed "gloo" on the
execution-model simulator's host memory -- the same interface as RcclComm (rank, world, cols_to_rows,
allgather_digests), so that lde_commit_sharded's sequencing, column ownership and subtree / top-level hashing
run with world_size > 1 in the GPU-less container.  The product path is RcclComm (C ABI, RCCL)."""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from ministark_amd.api import FIELD_WORDS, GOLDILOCKS_FP, DeviceBytes, GpuVec


def _host_view(ptr, words):
    """int64 torch view of `words` 8-byte words at a simulator "device" (= host) address."""
    arr = np.ctypeslib.as_array((ctypes.c_int64 * words).from_address(ptr))
    return torch.from_numpy(arr)


class GlooComm:
    def __init__(self, planner, group=None):
        self.planner, self.group = planner, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def cols_to_rows(self, my_cols, total_cols, nrows, field=GOLDILOCKS_FP):
        """The same arguments as RcclComm.cols_to_rows, and the same operations: the schedule comes from the library's
        ms_cols_to_rows_schedule (what ms_cols_to_rows_alltoall executes over RCCL), issued here over gloo."""
        from ministark_amd.distributed import XCHG_COPY, XCHG_RECV, XCHG_SEND, exchange_schedule
        pl = self.planner
        pl.sync()
        V = FIELD_WORDS[field]
        blk = nrows // self.world * V                                  # words of one rank's rows of one column
        shard = [GpuVec(pl, nrows // self.world, field) for _ in range(total_cols)]
        mine = [_host_view(c.ptr, nrows * V) for c in my_cols]
        dst = [_host_view(s.ptr, blk) for s in shard]
        ops = []
        for op in exchange_schedule(pl.lib, self.world, self.rank, total_cols, blk * 8):
            assert op.bytes == blk * 8 and op.src_offset % 8 == 0
            w0 = op.src_offset // 8
            if op.kind == XCHG_SEND:
                ops.append(dist.P2POp(dist.isend, mine[op.src_col][w0:w0 + blk], op.peer, self.group))
            elif op.kind == XCHG_RECV:
                ops.append(dist.P2POp(dist.irecv, dst[op.dst_col], op.peer, self.group))
            elif op.kind == XCHG_COPY:
                dst[op.dst_col].copy_(mine[op.src_col][w0:w0 + blk])
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return shard

    def p2p(self, ops):
        """RcclComm.p2p over gloo: [(kind, peer, simulator "device" pointer, bytes)] as one batch of isend / irecv."""
        from ministark_amd.distributed import XCHG_SEND
        self.planner.sync()
        batch = []
        for kind, peer, ptr, nbytes in ops:
            assert nbytes % 8 == 0
            view = _host_view(ptr, nbytes // 8)
            batch.append(dist.P2POp(dist.isend if kind == XCHG_SEND else dist.irecv, view, peer, self.group))
        if batch:
            for w in dist.batch_isend_irecv(batch):
                w.wait()

    def allgather_digests(self, my_digest_ptr):
        self.planner.sync()
        mine = _host_view(my_digest_ptr, 4).clone()
        parts = [torch.empty(4, dtype=torch.int64) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        out = DeviceBytes(self.planner, 32 * self.world)
        _host_view(out.ptr, 4 * self.world).copy_(torch.cat(parts))
        return out
