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

    def cols_to_rows(self, my_cols, total_cols):
        pl = self.planner
        pl.sync()
        field = my_cols[0].field if my_cols else GOLDILOCKS_FP
        V = FIELD_WORDS[field]
        nrows = len(my_cols[0]) if my_cols else 0
        # every rank must agree on the geometry even if it owns no column
        meta = torch.tensor([nrows, field], dtype=torch.int64)
        dist.all_reduce(meta, op=dist.ReduceOp.MAX, group=self.group)
        nrows, field = int(meta[0]), int(meta[1])
        V = FIELD_WORDS[field]
        blk = nrows // self.world * V
        shard = [GpuVec(pl, nrows // self.world, field) for _ in range(total_cols)]
        mine = [_host_view(c.ptr, nrows * V) for c in my_cols]
        dst = [_host_view(s.ptr, blk) for s in shard]
        ops = []
        for peer in range(self.world):
            if peer == self.rank:
                for j, c in enumerate(range(self.rank, total_cols, self.world)):
                    dst[c].copy_(mine[j][peer * blk:(peer + 1) * blk])
                continue
            for j in range(len(mine)):
                ops.append(dist.P2POp(dist.isend, mine[j][peer * blk:(peer + 1) * blk], peer, self.group))
            for c in range(peer, total_cols, self.world):
                ops.append(dist.P2POp(dist.irecv, dst[c], peer, self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return shard

    def allgather_digests(self, my_digest_ptr):
        self.planner.sync()
        mine = _host_view(my_digest_ptr, 4).clone()
        parts = [torch.empty(4, dtype=torch.int64) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        out = DeviceBytes(self.planner, 32 * self.world)
        _host_view(out.ptr, 4 * self.world).copy_(torch.cat(parts))
        return out
