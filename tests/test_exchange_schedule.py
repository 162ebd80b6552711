"""The point-to-point schedule of ms_cols_to_rows_alltoall (ministark_hip.cpp), executed on the
host: ms_cols_to_rows_schedule returns, for every rank, the operations the entry point issues over RCCL -- (send, peer, my
column, byte offset), (receive, peer, shard column), (local copy) -- and this test replays the schedules of ALL ranks of a
2-, 4- and 8-rank world against each other with NCCL's matching rule (the k-th send from a to b meets the k-th receive at b
from a) on numpy buffers.  Every rank must end up with its rows of every column: the offsets that will run on 8 GPUs have run
here.  tests/gloo_comm.py issues the same schedule over gloo in the multi-process tests."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ministark_amd.distributed import XCHG_COPY, XCHG_RECV, XCHG_SEND, exchange_schedule, owned_columns   # noqa: E402
from tests import backends   # noqa: E402


def _replay(lib, world, total_cols, nrows, elem_bytes):
    rng = np.random.default_rng(world * 1000 + total_cols)
    cols = [rng.integers(0, 256, size=nrows * elem_bytes, dtype=np.uint8) for _ in range(total_cols)]        # column c, whole domain
    blk = nrows // world * elem_bytes
    my = {r: [cols[c] for c in owned_columns(total_cols, r, world)] for r in range(world)}
    shard = {r: [np.full(blk, 0xEE, dtype=np.uint8) for _ in range(total_cols)] for r in range(world)}
    sched = {r: exchange_schedule(lib, world, r, total_cols, blk) for r in range(world)}
    sends, recvs = {}, {}
    for r in range(world):
        for op in sched[r]:
            assert op.bytes == blk
            if op.kind == XCHG_SEND:
                assert op.peer != r and op.src_offset + op.bytes <= nrows * elem_bytes
                sends.setdefault((r, op.peer), []).append(my[r][op.src_col][op.src_offset:op.src_offset + op.bytes])
            elif op.kind == XCHG_RECV:
                assert op.peer != r
                recvs.setdefault((op.peer, r), []).append(shard[r][op.dst_col])
            else:
                assert op.kind == XCHG_COPY and op.peer == r
                shard[r][op.dst_col][:] = my[r][op.src_col][op.src_offset:op.src_offset + op.bytes]
    assert set(sends) == set(recvs)
    for pair, ss in sends.items():
        rr = recvs[pair]
        assert len(ss) == len(rr), f"{len(ss)} sends against {len(rr)} receives between ranks {pair}"
        for src, dst in zip(ss, rr):                         # matched in issue order
            dst[:] = src
    for r in range(world):
        for c in range(total_cols):
            assert np.array_equal(shard[r][c], cols[c][r * blk:(r + 1) * blk]), f"rank {r} column {c}"
    # every byte that crosses a link: (G - 1) / G of every rank's own columns
    sent = sum(len(v) for v in sends.values()) * blk
    assert sent == total_cols * nrows * elem_bytes * (world - 1) // world


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("total_cols", [1, 3, 8, 13, 32])
def test_schedules_of_all_ranks_compose(world, total_cols):
    lib = backends.planner("emu").lib              # the schedule is a pure function of the C ABI: any build of the library has it
    for elem_bytes in (8, 24):
        _replay(lib, world, total_cols, 64, elem_bytes)


def test_schedule_rejects_a_wrong_column_count():
    lib = backends.planner("emu").lib
    import ctypes
    count = ctypes.c_size_t(0)
    assert lib.ms_cols_to_rows_schedule(4, 1, 3, 5, 64, None, 0, ctypes.byref(count)) != 0      # rank 1 of 4 owns column 1 only
    assert lib.ms_cols_to_rows_schedule(4, 1, 1, 5, 64, None, 0, ctypes.byref(count)) == 0 and count.value == 8      # 3 sends, 2 + 1 + 1 receives, 1 copy
