"""Worker of the multi-rank prover tests: python tests/dist_prove_worker.py <rank> <world> <port> <emu | emu-rccl> <outfile>
Every rank runs distributed.prove_sharded on its columns; rank 0 also runs the single-device pipeline.prove_phases on the whole trace
and compares every output."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world, port, kind, outfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import backends
    from ministark_amd import GOLDILOCKS_FP, Matrix, pipeline
    from ministark_amd.distributed import RcclComm, owned_columns, prove_sharded
    pl = backends.planner("emu")
    if os.environ.get("MS_TEST_EAGER_BYTES"):                 # the openings' batch outgrows its first arena (OpeningBatch.execute's second leg)
        from ministark_amd import distributed
        distributed.OpeningBatch.EAGER_BYTES = int(os.environ["MS_TEST_EAGER_BYTES"])
    if kind == "emu":
        from tests.gloo_comm import GlooComm
        comm = GlooComm(pl)
    else:
        assert os.environ.get("MS_RCCL_LIB"), "the launcher sets MS_RCCL_LIB"
        comm = RcclComm.from_torch_distributed(pl)
    P = (1 << 64) - (1 << 32) + 1
    results = []
    # (log_rows, columns, blow-up, folding, max remainder coefficients, ce_blowup): BASELINE configs[4]'s shape in small (blow-up 4, the
    # fib AIR's ce_blowup 1: the constraint-evaluation coset lives on G / 4 ranks), and the additive AIR with ce_blowup = blow-up
    for log_rows, ncols, blowup, folding, max_rem, air in ((8, 8, 4, 8, 4, "fib"), (7, 5, 8, 4, 4, "additive"), (9, 8, 4, 2, 64, "fib")):
        n_t = 1 << log_rows
        if (n_t * blowup) // world < 2 * folding:
            continue
        rng = np.random.default_rng(100 + log_rows)
        cols = [rng.integers(0, P, size=n_t, dtype=np.uint64) for _ in range(ncols)]
        if air == "additive":
            comp, ce, nch = pipeline.additive_constraints(n_t, ncols, blowup)
        else:
            comp, ce, nch = pipeline.fib_constraints(n_t, ncols)
        draws = pipeline.Draws(0xC5, ncols, nch, ce, 12, n_t * blowup, pipeline.fri_num_layers(n_t * blowup, blowup, folding, max_rem))
        mine = [cols[c] for c in owned_columns(ncols, rank, world)]
        got = prove_sharded(pl, comm, mine, ncols, log_rows, comp, draws, blowup, folding, max_rem, 6, ce_blowup=ce)
        ok = True
        if rank == 0:
            want = pipeline.prove_phases(pl, Matrix.from_numpy(pl, cols, GOLDILOCKS_FP), comp, draws, blowup, folding, max_rem, 6, ce_blowup=ce)
            ok = (got["base_root"] == want["base_root"] and got["composition_root"] == want["composition_root"]
                  and got["fri_roots"] == want["fri_roots"] and got["nonce"] == want["nonce"]
                  and np.array_equal(got["remainder_coeffs"], want["remainder_coeffs"])
                  and list(got["ood"][0]) == list(want["ood"][0]) and list(got["ood"][1]) == list(want["ood"][1]))
            wq = want["queries"]
            for name in ("base_trace_proof", "composition_trace_proof"):
                ok = ok and got["queries"][name] == getattr(wq, name)
            for name in ("base_trace_values", "composition_trace_values"):
                ok = ok and np.array_equal(got["queries"][name], getattr(wq, name))
            ok = ok and len(got["fri_openings"]) == len(want["fri_openings"])
            for a, b in zip(got["fri_openings"], want["fri_openings"]):
                ok = ok and a["positions"] == b["positions"] and np.array_equal(a["rows"], b["rows"]) and a["proof"] == b["proof"]
        else:
            ok = len(got["fri_roots"]) == len(draws.fri_alphas) and len(got["base_root"]) == 32
        results.append(f"{air}:{'ok' if ok else 'MISMATCH'}:{got['base_root'].hex()[:16]}:{b''.join(got['fri_roots']).hex()[:16]}")
    with open(outfile, "w") as f:
        f.write("\n".join(results))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
