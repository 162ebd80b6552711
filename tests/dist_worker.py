"""Worker for the world_size-2 tests: python tests/dist_worker.py <rank> <world> <port> <backend_kind> <outfile>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world, port, kind, outfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("nccl" if kind == "hip" else "gloo", rank=rank, world_size=world)
    from oracle import cref
    from tests import backends
    from ministark_amd import GOLDILOCKS_FP, GOLDILOCKS_FQ3
    from ministark_amd.distributed import RcclComm, lde_commit_sharded, owned_columns
    if kind == "emu":
        from tests.gloo_comm import GlooComm
        pl = backends.planner("emu")
        comm = GlooComm(pl)
    elif kind == "emu-rccl":                            # the PRODUCT's RcclComm / ms_comm_* / ms_cols_to_rows_alltoall / ms_p2p_batch /
        pl = backends.planner("emu")                    # ms_allgather_digests on the simulator, NCCL stood in for by tests/emu/fake_rccl.cpp
        assert os.environ.get("MS_RCCL_LIB"), "the launcher sets MS_RCCL_LIB"
        comm = RcclComm.from_torch_distributed(pl)      # the communicator id travels over the gloo group
    else:                                               # one GPU per rank: the planner lives on LOCAL_RANK's device
        local = int(os.environ.get("LOCAL_RANK", rank))
        torch.cuda.set_device(local)
        pl = backends.planner("hip", local)
        comm = RcclComm.from_torch_distributed(pl)
    results = []
    for field, V, total_cols, log_n, log_b in ((GOLDILOCKS_FP, 1, 5, 6, 2), (GOLDILOCKS_FQ3, 3, 3, 5, 3), (GOLDILOCKS_FP, 1, 1, 7, 1)):
        allc = [cref.random_elements((1 << log_n) * V, 1000 + c) for c in range(total_cols)]
        mine = [allc[c] for c in owned_columns(total_cols, rank, world)]
        root, shard = lde_commit_sharded(pl, comm, mine, total_cols, log_n, log_b, 7, field)
        results.append(root.hex())
    # row-sharded constraint evaluation on the exchanged shards, every shape of eval_constraints_sharded: (log_n, log_b, ce)
    #   ce a multiple of the ranks that hold evaluation rows -> no communication; otherwise the shard exchange (ms_p2p_batch);
    #   a coset smaller than one shard -> rank 0 alone.  Each rank checks its slice of the single-device result.
    from ministark_amd import GpuVec, Matrix
    from ministark_amd import expr as E
    from ministark_amd.distributed import eval_constraints_sharded
    ok = True
    for log_n, log_b, ce in ((6, 2, 4), (6, 2, 1), (6, 2, 2), (6, 3, 8), (7, 4, 1)):
        ncols = 3
        n_t, n_lde, n_ce = 1 << log_n, 1 << (log_n + log_b), (1 << log_n) * ce
        if n_lde % world or n_lde // world < 2:
            continue
        allc = [cref.random_elements(n_t, 2000 + c) for c in range(ncols)]
        mine = [allc[c] for c in owned_columns(ncols, rank, world)]
        lde_local = Matrix([GpuVec.from_numpy(pl, c) for c in mine]).lde(1 << log_b, 7, True).columns if mine else []
        shard = comm.cols_to_rows(lde_local, ncols, n_lde)
        x = E.X()
        t = [lambda o=0, k=k: E.Trace(k, o) for k in range(ncols)]
        expr = ((t[0](1) - t[1]() * t[2](-1)) * (x - E.Constant(3)) / (x ** n_t - 1) * (E.Challenge(0) * x ** 3 + E.Challenge(1))
                + E.Periodic([1, 2, 3, 4]) * t[1](2) + x * t[2]())
        prog = E.compile_expr(expr, ncols, False)
        ch = cref.random_elements(2, 9).reshape(2, 1)
        got = eval_constraints_sharded(prog, pl, comm, ch, ch[:1], ce, 7, n_ce, shard, n_lde=n_lde)
        # single device: the constraint-evaluation coset = the first n_ce rows of the bit-reversed LDE
        full_cols = [GpuVec.from_numpy(pl, cref.lde(c, log_n, log_b, 1, 7, True)[:n_ce].copy()) for c in allc]
        want = E.eval(prog, pl, ch, ch[:1], ce, 7, n_ce, full_cols, bit_reversed=True).to_numpy()
        rows = n_lde // world
        holds = rank * rows < n_ce
        if got is None:
            ok = ok and not holds
        else:
            vec, first, count = got
            ok = ok and holds and first == (0 if n_ce <= rows else rank * rows) and np.array_equal(vec.to_numpy(), want[first:first + count])
    results.append("eval_ok" if ok else "eval_MISMATCH")
    with open(outfile, "w") as f:
        f.write("\n".join(results))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
