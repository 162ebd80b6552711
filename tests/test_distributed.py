"""Column-sharded LDE + row-sharded Merkle commitment over 2 ranks (gloo, CPU, kernels under the
simulator): every rank must end with the root a single device computes for the whole matrix."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import cref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _expected():
    out = []
    for V, total_cols, log_n, log_b in ((1, 5, 6, 2), (3, 3, 5, 3), (1, 1, 7, 1)):
        cols = [cref.lde(cref.random_elements((1 << log_n) * V, 1000 + c), log_n, log_b, V, 7, True) for c in range(total_cols)]
        out.append(cref.sha256_merkle(cref.sha256_rows(cols, V))[1].tobytes().hex())
    return out


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_commit_matches_single_device_gloo(tmp_path, world):
    port = str(29500 + (os.getpid() % 400) + world)
    procs, files = [], []
    for r in range(world):
        f = str(tmp_path / f"r{r}.txt")
        files.append(f)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r), str(world), port, "emu", f],
                                      cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()[-2000:]
    want = _expected() + ["eval_ok"]                 # + the row-sharded constraint evaluation check of the worker
    for f in files:
        assert open(f).read().split("\n") == want


@pytest.mark.parametrize("world", [2, 4, 8])
def test_product_exchange_entry_points_with_more_than_one_rank(tmp_path, world):
    """The same worker, but through the product's own communicator code: RcclComm -> ms_comm_init, ms_cols_to_rows_alltoall (the
    schedule-driven send / receive group and the local copies), ms_p2p_batch (the shard exchange of the row-sharded evaluator) and
    ms_allgather_digests, built into the simulator library and bound with dlsym exactly as on a GPU box -- only the nine NCCL entry
    points themselves are a stand-in (tests/emu/fake_rccl.cpp: byte FIFOs between the processes, NCCL's matching and group rules)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    fake = build_emu.build_fake_rccl()
    port = str(30100 + (os.getpid() % 400) + world)
    procs, files = [], []
    for r in range(world):
        f = str(tmp_path / f"r{r}.txt")
        files.append(f)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r), str(world), port, "emu-rccl", f],
                                      cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, MS_RCCL_LIB=fake)))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()[-2000:]
    want = _expected() + ["eval_ok"]
    for f in files:
        assert open(f).read().split("\n") == want


def _run_prove_workers(tmp_path, world, kind, env):
    port = str(30700 + (os.getpid() % 400) + world + (50 if kind == "emu" else 0))
    procs, files = [], []
    for r in range(world):
        f = str(tmp_path / f"p{r}.txt")
        files.append(f)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_prove_worker.py"), str(r), str(world), port, kind, f],
                                      cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env))
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out.decode()[-3000:]
    lines = [open(f).read().split("\n") for f in files]
    assert all(len(l) == 3 for l in lines)
    for l in lines:
        assert l == lines[0]                                    # every rank ends with the same roots
        assert all(":ok:" in entry for entry in l), l


@pytest.mark.parametrize("world", [2, 4, 8])
def test_whole_prover_on_row_shards_gloo(tmp_path, world):
    """distributed.prove_sharded -- every phase of default_prove after the base commitment on row shards (composition-trace commitment,
    out-of-domain evaluations by the owners, DEEP composition on the rows of both committed LDEs, every FRI layer folded and committed
    where its rows are, openings served by the owners) -- against pipeline.prove_phases on one device: roots, OOD values, FRI roots,
    remainder, nonce, every Merkle view and every opened row, for the fib AIR (ce_blowup 1 < blow-up 4: BASELINE configs[4]'s shape) and
    an AIR with ce_blowup = blow-up, folding factors 8, 4 and 2."""
    _run_prove_workers(tmp_path, world, "emu", dict(os.environ))


@pytest.mark.parametrize("eager_bytes", [8, 2048])
def test_openings_that_outgrow_the_first_arena_gloo(tmp_path, eager_bytes):
    """OpeningBatch with an arena of 8 bytes (nothing fits: every gather waits for `execute`) and of 2 KiB (the first requests are gathered
    at once, the rest behind them in the full-size buffer): each request's gathers run exactly once, and every Merkle view and opened row
    still equals the single-device prover's (the second leg had no test)."""
    _run_prove_workers(tmp_path, 2, "emu", dict(os.environ, MS_TEST_EAGER_BYTES=str(eager_bytes)))


@pytest.mark.parametrize("world", [2, 8])
def test_whole_prover_on_row_shards_product_exchange(tmp_path, world):
    """the same through the product's communicator entry points (ms_cols_to_rows_alltoall, ms_p2p_batch, ms_allgather_digests) over
    tests/emu/fake_rccl.cpp"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    _run_prove_workers(tmp_path, world, "emu-rccl", dict(os.environ, MS_RCCL_LIB=build_emu.build_fake_rccl()))


def _bench_env():
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    return dict(os.environ, MS_BENCH_LIB=build_emu.build(), MS_BENCH_DIST_BACKEND="gloo", MS_RCCL_LIB=build_emu.build_fake_rccl())


@pytest.mark.parametrize("world", [2, 4])
def test_bench_gpus_n_starts_its_own_ranks(world):
    """`python bench.py --gpus N` with no launcher around it (the driver's command): the process starts N ranks itself, the ranks join
    one process group, run the column-sharded LDE + commitment through the product's communicator entry points (here on the simulator
    build over the shared-memory NCCL stand-in) and rank 0's line reports n_gpus = N and the root of the one-process run."""
    import json
    env = _bench_env()
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "lde-commit", "--log-rows", "8", "--total-cols", "8", "--steps", "2"]
    lines = {}
    for n in (1, world):
        r = subprocess.run(cmd + ["--gpus", str(n)], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        out = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
        assert len(out) == 1, r.stdout.decode()[-2000:]
        lines[n] = json.loads(out[0])
    assert lines[world]["n_gpus"] == world and lines[world]["sharded_lde_commit"]["n_gpus"] == world
    assert lines[1]["n_gpus"] == 1
    assert lines[world]["sharded_lde_commit"]["root"] == lines[1]["sharded_lde_commit"]["root"] is not None
    pn, p1 = lines[world]["sharded_lde_commit"]["prove"], lines[1]["sharded_lde_commit"]["prove"]       # the whole prover on the same ranks
    assert "error" not in pn and pn["n_gpus"] == world and pn["base_root"] == p1["base_root"] and pn["fri_root_last"] == p1["fri_root_last"]


def test_bench_refuses_a_world_that_is_not_gpus():
    """--gpus 4 inside a 1-rank environment must not print an n_gpus = 1 line (`--gpus` was parsed and never read)."""
    env = dict(_bench_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--mode", "lde-commit", "--log-rows", "8", "--total-cols", "4"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 2 and not r.stdout.strip() and b"--gpus 4" in r.stderr


@pytest.mark.gpu
def test_rccl_entry_points_world1_hip():
    """The RCCL leg on the single-GPU lease, through the C ABI without torch: ncclGetUniqueId / ncclCommInitRank,
    the send/recv group (degenerate: the own block is a device copy) and ncclAllGather really execute."""
    from tests import backends
    from ministark_amd import GOLDILOCKS_FQ3, GpuVec, Matrix, MerkleTree
    from ministark_amd.distributed import RcclComm, lde_commit_sharded
    pl = backends.planner("hip")
    comm = RcclComm(pl, 0, 1, RcclComm.unique_id(pl.lib))
    try:
        cols = [cref.random_elements((1 << 10) * 3, 70 + c) for c in range(3)]
        vecs = [GpuVec.from_numpy(pl, c, GOLDILOCKS_FQ3) for c in cols]
        shard = comm.cols_to_rows(vecs, 3, 1 << 10, GOLDILOCKS_FQ3)          # one rank: the shard IS the columns (no copy)
        assert all(s is v for s, v in zip(shard, vecs))
        # the entry point itself with one rank (RcclComm no longer calls it then): the own block is a device copy
        from ministark_amd.api import _ptr_array
        out = [GpuVec(pl, 1 << 10, GOLDILOCKS_FQ3) for _ in range(3)]
        pl.lib.check(pl.lib.ms_cols_to_rows_alltoall(pl.handle, GOLDILOCKS_FQ3, 1 << 10, _ptr_array(vecs), 3, 3, _ptr_array(out)))
        assert all(np.array_equal(o.to_numpy(), c) for o, c in zip(out, cols))
        tree = MerkleTree.from_matrix(Matrix(shard))
        assert comm.allgather_digests(tree.nodes.ptr + 32).to_numpy().tobytes() == tree.root()
        root, _ = lde_commit_sharded(pl, comm, cols, 3, 10, 3, 7, GOLDILOCKS_FQ3)
        want = cref.sha256_merkle(cref.sha256_rows([cref.lde(c, 10, 3, 3, 7, True) for c in cols], 3))[1].tobytes()
        assert root == want
        with pytest.raises(Exception):
            comm.cols_to_rows(vecs[:2], 3, 1 << 10, GOLDILOCKS_FQ3)          # a rank of a 1-rank world owns all 3 columns
    finally:
        comm.close()


@pytest.mark.gpu
def test_sharded_commit_world1_nccl_hip(tmp_path):
    """The same worker the gloo tests run, with backend "nccl" (= RCCL) and the product library on GPU 0."""
    port = str(29900 + (os.getpid() % 90))
    f = str(tmp_path / "r0.txt")
    env = dict(os.environ, LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), "0", "1", port, "hip", f],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, env=env)
    assert p.returncode == 0, p.stdout.decode()[-2000:]
    assert open(f).read().split("\n") == _expected() + ["eval_ok"]


def test_whole_prover_with_one_rank_equals_the_single_device_prover_emu():
    """N = 1 of the sharded prover (what `bench.py` times next to `prove`): nothing is exchanged, the shards alias the columns, the
    openings take the single-device path -- and every output equals pipeline.prove_phases'."""
    from tests import backends
    from ministark_amd import GOLDILOCKS_FP, Matrix, pipeline
    from ministark_amd.distributed import prove_sharded

    class OneRank:                                           # the communicator surface prove_sharded uses, for a world of one
        rank, world = 0, 1

        def cols_to_rows(self, my_cols, total_cols, nrows, field=GOLDILOCKS_FP):
            assert len(my_cols) == total_cols
            return list(my_cols)

        def p2p(self, ops):
            assert not ops

        def allgather_digests(self, ptr):
            raise AssertionError("no all-gather with one rank")
    pl = backends.planner("emu")
    log_rows, ncols, blowup, folding, max_rem = 9, 8, 4, 8, 16
    n_t = 1 << log_rows
    cols = [cref.random_elements(n_t, 300 + c) for c in range(ncols)]
    comp, ce, nch = pipeline.fib_constraints(n_t, ncols)
    draws = pipeline.Draws(0xC5, ncols, nch, ce, 12, n_t * blowup, pipeline.fri_num_layers(n_t * blowup, blowup, folding, max_rem))
    got = prove_sharded(pl, OneRank(), cols, ncols, log_rows, comp, draws, blowup, folding, max_rem, 6, ce_blowup=ce)
    want = pipeline.prove_phases(pl, Matrix.from_numpy(pl, cols, GOLDILOCKS_FP), comp, draws, blowup, folding, max_rem, 6, ce_blowup=ce)
    assert got["base_root"] == want["base_root"] and got["composition_root"] == want["composition_root"]
    assert got["fri_roots"] == want["fri_roots"] and got["nonce"] == want["nonce"]
    assert np.array_equal(got["remainder_coeffs"], want["remainder_coeffs"])
    assert list(got["ood"][0]) == list(want["ood"][0]) and list(got["ood"][1]) == list(want["ood"][1])
    wq = want["queries"]
    for name in ("base_trace_proof", "composition_trace_proof"):
        assert got["queries"][name] == getattr(wq, name)
    for name in ("base_trace_values", "composition_trace_values"):
        assert np.array_equal(got["queries"][name], getattr(wq, name))
    assert len(got["fri_openings"]) == len(want["fri_openings"]) > 0
    for a, b in zip(got["fri_openings"], want["fri_openings"]):
        assert a["positions"] == b["positions"] and np.array_equal(a["rows"], b["rows"]) and a["proof"] == b["proof"]
