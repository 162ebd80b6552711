#!/usr/bin/env python3
"""bench_objects.py -- the SECONDARY objects of bench.py's record (everything that is not the headline transform): configs[2] `lde_commit`,
the prover's own LDE `lde_2_24`, configs[3] `constraint_eval` (i)-(iii), configs[4] `prove` (+ the C++ host mirror's run and the one-rank
sharded prover), `cold_start`, the C2 sweep, the reference's criterion harness, the sharded LDE + commitment / prover of any N, and the
machinery that measures every object's HBM traffic and vector-ALU use in child runs under rocprofv3 --pmc.  bench.py keeps the contract: the
timed region of the headline, its roofline, the CPU baseline, the one JSON line.  (Split out of bench.py in round 6: 1 300 lines in one file were
hard to audit.)  Nothing here is timed inside bench.py's timed region."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
BENCH_PY = os.path.join(ROOT, "bench.py")          # the child runs (--pmc-child, --cold-child) re-enter through bench.py's command line
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
P_GOLDILOCKS = (1 << 64) - (1 << 32) + 1


def sharded_lde_commit(pl, comm, steps, warmup, log_rows=22, total_cols=32, log_blowup=2, barrier=lambda: None):
    """configs[4]: column-sharded LDE + row-sharded commitment through the C ABI (ministark_amd/distributed.py).
    Fixed total work; returns the dict for the JSON line (times are max over ranks where `reduce_max` is given)."""
    import numpy as np
    from ministark_amd import GOLDILOCKS_FP, GpuVec, Matrix, MerkleTree
    from ministark_amd.distributed import owned_columns
    rank, world = comm.rank, comm.world
    n = 1 << log_rows
    N = n << log_blowup
    mine = owned_columns(total_cols, rank, world)
    P = (1 << 64) - (1 << 32) + 1
    # column c holds the same values whichever rank owns it: the root on the line is the same for every N
    trace = Matrix([GpuVec.from_numpy(pl, np.random.default_rng(0xC50000 + c).integers(0, P, size=n, dtype=np.uint64)) for c in mine])
    t_lde = t_x = t_c = 0.0
    root = None
    for it in range(warmup + steps):
        barrier()
        t0 = time.perf_counter()
        lde = trace.lde(1 << log_blowup, 7, True).columns
        pl.sync()
        t1 = time.perf_counter()
        shard = comm.cols_to_rows(lde, total_cols, N)
        pl.sync()
        barrier()
        t2 = time.perf_counter()
        tree = MerkleTree.from_matrix(Matrix(shard))
        if world > 1:
            roots = comm.allgather_digests(tree.nodes.ptr + 32)
            root = MerkleTree(pl, roots, world).root()
        else:
            root = tree.root()
        pl.sync()
        t3 = time.perf_counter()
        if it >= warmup:
            t_lde += t1 - t0; t_x += t2 - t1; t_c += t3 - t2
        del lde, shard, tree
    k = max(steps, 1)
    lde_bytes = float(total_cols) * (n * 8 + N * 8)                       # n s + beta n s per column
    return {"workload": f"2^{log_rows} rows x {total_cols} columns, blow-up {1 << log_blowup}, SHA-256 commitment; columns c mod N on rank c, rows r N/G.. after the exchange",
            "scaling": "strong", "n_gpus": world, "columns_this_rank": len(mine),
            "lde_ms": t_lde / k * 1e3, "exchange_ms": t_x / k * 1e3, "commit_ms": t_c / k * 1e3,
            "lde_algorithmic_bytes": lde_bytes, "exchange_bytes_sent_per_rank": float(len(mine)) * N * 8 * (world - 1) / world,
            "root": root.hex() if root else None}


def sharded_prove(pl, comm, steps, log_rows=22, total_cols=8, barrier=lambda: None):
    """configs[4] as north_star words it -- "full prover.rs on a 2^22-row trace, columns sharded across the GPUs": distributed.prove_sharded
    (every phase after the base commitment on row shards).  Fixed total work; wall time per proof = max over ranks (the caller reduces)."""
    import numpy as np
    from ministark_amd import GpuVec, pipeline
    from ministark_amd.distributed import owned_columns, prove_sharded
    blowup, folding = 4, 8
    n_t = 1 << log_rows
    P = (1 << 64) - (1 << 32) + 1
    mine = owned_columns(total_cols, comm.rank, comm.world)
    vecs = [GpuVec.from_numpy(pl, np.random.default_rng(0xF1B0000 + c).integers(0, P, size=n_t, dtype=np.uint64)) for c in mine]   # column c is the same for every N
    comp, ce, nch = pipeline.fib_constraints(n_t, total_cols)
    draws = pipeline.Draws(0xC5, total_cols, nch, ce, 32, n_t * blowup, pipeline.fri_num_layers(n_t * blowup, blowup, folding, 64))
    res, phases, walls = None, {}, []
    for it in range(1 + steps):
        barrier()
        ph = {}
        t0 = time.perf_counter()
        res = prove_sharded(pl, comm, vecs, total_cols, log_rows, comp, draws, blowup, folding, 64, 8, ce_blowup=ce, phases_ms=ph)
        pl.sync()
        if it:
            walls.append((time.perf_counter() - t0) * 1e3)
            for k, v in ph.items():
                phases[k] = phases.get(k, 0.0) + v / steps
    return {"workload": f"2^{log_rows} rows x {total_cols} columns, the reference's fib AIR, ProofOptions::new(32, 4, 8, 8, 64); columns c mod N on rank c, every "
                        "later phase on row shards (ministark_amd/distributed.py prove_sharded)", "scaling": "strong", "n_gpus": comm.world,
            "prove_ms": sorted(walls)[len(walls) // 2], "phases_ms_this_rank": {k: round(v, 3) for k, v in phases.items()},
            "base_root": res["base_root"].hex(), "fri_root_last": res["fri_roots"][-1].hex() if res["fri_roots"] else None}


class _stdout_to_stderr:
    """RCCL prints a version banner on the C stdout at communicator creation (flushed at exit when stdout is a file):
    the contract is ONE JSON line on stdout, so file descriptor 1 points at stderr while RCCL is in use."""

    def __enter__(self):
        import ctypes
        self.libc = ctypes.CDLL(None)
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        self.libc.fflush(None)
        os.dup2(self.saved, 1)
        os.close(self.saved)




# ---- the bounding resources of the OTHER roofline objects, measured in the run ----------------------------------------------------------
# One child process per counter pass runs every object's workload once warm and once between two marker launches (a k_fill over
# _PMC_MARK words: no workload launches that grid); the parent cuts the counter rows at the markers.  Three passes: FETCH_SIZE,
# WRITE_SIZE (separate, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; FETCH_SIZE doubled on gfx950, units of 1024 B) and the SQ
# counters.  SQ_INSTS_VALU counts wave instructions of the whole chip; a vector instruction holds its SIMD's issue slot for one quad cycle
# (SQ_ACTIVE_INST_VALU = SQ_INSTS_VALU in quad cycles, profiles/r04_ntt_sq_counters.csv), so the chip issues at most
# 1024 SIMDs x 2.4 GHz / 4 = 6.144e11 wave instructions per second; SQ_BUSY_CYCLES is summed over the 32 shader engines.
_PMC_MARK = 4242
_PMC_SQ = ("SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE")
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 4
_PMC_OBJECTS = ("lde_commit", "lde_2_24", "constraint_eval.fib_air_fp", "constraint_eval.mixed_17fp_9fq3", "constraint_eval.fib_air_fp252", "prove")


def _pmc_child_main(pl):
    """--pmc-child: every object's workload, warm once, then once between markers (order = _PMC_OBJECTS)."""
    import ctypes
    from ministark_amd import GOLDILOCKS_FP, GpuVec
    mark = GpuVec(pl, _PMC_MARK, GOLDILOCKS_FP)
    one = np.array([1], dtype=np.uint64)
    seen = []

    def marker():
        pl.lib.check(pl.lib.ms_fill(pl.handle, GOLDILOCKS_FP, _PMC_MARK, mark.ptr, one.ctypes.data))

    def pmc(name, run):
        run(); pl.sync()
        marker()
        run(); pl.sync()
        marker()
        pl.sync()
        seen.append(name)
    bench_lde_commit(pl, False, pmc=pmc)
    bench_lde_2_24(pl, pmc=pmc)
    bench_constraint_eval(pl, False, pmc=pmc)
    bench_prove(pl, False, pmc=pmc)
    if tuple(seen) != _PMC_OBJECTS:
        raise SystemExit(f"pmc child: objects {seen}")


def _measure_objects():
    """-> {object: {kernel: {counter: sum over the launches of ONE run of the workload}}} or None (no rocprofv3, being profiled, a pass failed)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None or os.environ.get("MS_BENCH_NO_PMC") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None
    res = {name: {} for name in _PMC_OBJECTS}
    mark_grid = ((_PMC_MARK + 255) // 256) * 256
    work = tempfile.mkdtemp(prefix="ms_pmc_obj_", dir="/tmp")
    try:
        for counters in (("FETCH_SIZE",), ("WRITE_SIZE",), _PMC_SQ):
            out = os.path.join(work, counters[0])
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", *counters, "-d", out, "-o", "t", "--output-format", "csv", "--", sys.executable,
                   BENCH_PY, "--pmc-child"]
            env = dict(os.environ, MS_BENCH_NO_PMC="1", TMPDIR="/tmp")
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = pr.wait(timeout=240)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(pr.pid, signal.SIGKILL)
                pr.wait()
                return None
            hits = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if rc != 0 or not hits:
                return None
            rows = list(csv.DictReader(open(hits[0])))
            rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
            window, inside, last_id = -1, False, None
            for r in rows:
                name = r["Kernel_Name"]
                if "k_fill" in name and int(float(r["Grid_Size"])) == mark_grid:
                    if r.get("Dispatch_Id") != last_id:             # one marker launch has a row per counter
                        last_id = r.get("Dispatch_Id")
                        inside = not inside
                        if inside:
                            window += 1
                    continue
                if not inside or not 0 <= window < len(_PMC_OBJECTS):
                    continue
                k = name.split("(")[0].replace("void ", "")
                d = res[_PMC_OBJECTS[window]].setdefault(k, {})
                d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            if window != len(_PMC_OBJECTS) - 1 or inside:
                return None
        return res
    except Exception:                                            # noqa: BLE001 -- an extra: the objects then say why their fields are null
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


def _attach_resources(roof, counters, kernel_us_total, elements, only=None, per="element"):
    """Fill a roofline object from the measured counters: HBM traffic next to the algorithmic bytes, and the vector ALU next to HBM
    ("report VALU utilisation and instruction counts next to GB/s").  only: substrings of the kernel names that belong
    to this object (None = every launch of the workload)."""
    if counters is None:
        roof["traffic_source"] = "not measured (rocprofv3 absent, the run itself profiled, or a counter pass failed)"
        return roof
    sel = {k: v for k, v in counters.items() if only is None or any(t in k for t in only)}
    tot = lambda c: sum(v.get(c, 0.0) for v in sel.values())
    fetch, write = 2.0 * 1024 * tot("FETCH_SIZE"), 1024.0 * tot("WRITE_SIZE")
    insts, active, busy = tot("SQ_INSTS_VALU"), tot("SQ_ACTIVE_INST_VALU"), tot("SQ_BUSY_CYCLES")
    roof["traffic"] = fetch + write
    roof["traffic_fetch_bytes"], roof["traffic_write_bytes"] = fetch, write
    roof["traffic_over_algorithmic"] = round((fetch + write) / roof["algorithmic_bytes"], 2) if roof.get("algorithmic_bytes") else None
    roof["traffic_source"] = "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (three child runs of every object's workload, cut at marker launches)"
    t = kernel_us_total * 1e-6
    roof["hbm_frac_of_traffic"] = round((fetch + write) / t / 1e9 / HBM_PEAK_GBS, 4) if t > 0 else None
    roof["valu_insts_per_" + per] = round(insts * 64.0 / elements, 1) if elements else None
    roof["valu_busy"] = round((active * 4.0 / 1024) / (busy / 32.0), 3) if busy else None
    roof["frac_of_valu_issue_peak"] = round(insts / t / VALU_ISSUE_PEAK, 4) if t > 0 else None
    roof["bound_measured"] = ("valu" if (roof["frac_of_valu_issue_peak"] or 0) > (roof["hbm_frac_of_traffic"] or 0) else "hbm")
    roof["counters_by_kernel"] = {
        k[-60:]: {"valu_insts_per_" + per: round(v.get("SQ_INSTS_VALU", 0.0) * 64.0 / elements, 1) if elements else None,
                  "valu_quad_cycles_per_inst": round(v["SQ_ACTIVE_INST_VALU"] / v["SQ_INSTS_VALU"], 3) if v.get("SQ_INSTS_VALU") else None,
                  "valu_busy": round((v.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / 1024) / (v["SQ_BUSY_CYCLES"] / 32.0), 3) if v.get("SQ_BUSY_CYCLES") else None,
                  "fetch_bytes": 2.0 * 1024 * v.get("FETCH_SIZE", 0.0), "write_bytes": 1024.0 * v.get("WRITE_SIZE", 0.0)}
        for k, v in sorted(sel.items())}
    return roof


def _attach_all(obj, pmc):
    """Every roofline object of the line carries a note ("_res") of what its counters are; they are measured AFTER all timed work (a
    counter session leaves the device in the profiler's clock state for a while: measured first, the objects timed after it ran up to
    three times slower) and attached here."""
    if isinstance(obj, dict):
        note = obj.pop("_res", None)
        if note is not None:
            name, us, elements, only, per = note
            _attach_resources(obj, None if pmc is None else pmc.get(name), us, elements, only=only, per=per)
        for v in list(obj.values()):
            _attach_all(v, pmc)


def _profiled(pl, fn, reps, after_wall=None):
    """-> (wall seconds per call, {kernel: microseconds per call}).  The wall clock is taken WITHOUT the per-launch
    hipEvents (a pair per kernel, ~150 launches per prover run, costs 15-20 % of the wall time); the kernel times come
    from a second set of runs with them."""
    fn(); pl.sync()                                            # plans, pool, specialised kernels
    fn(); pl.sync()                                            # clocks, allocator
    walls = []
    for _ in range(reps):                                      # every call timed on its own (synced), the MEDIAN is reported:
        t0 = time.perf_counter()                               # the host side of a box is noisy (10.3 .. 11.5 ms for the same proof)
        fn()
        pl.sync()
        walls.append(time.perf_counter() - t0)
    wall = sorted(walls)[len(walls) // 2]
    if after_wall:
        after_wall()
    pl.profile(True)
    for _ in range(reps):
        fn()
    pl.sync()
    prof = pl.profile_read()
    pl.profile(False)
    return wall, {k: round(v["total_us"] / reps, 1) for k, v in sorted(prof.items())}


def bench_c2_sweep(pl):
    """configs[1] over its whole range at the column counts a prover has: forward coset NTT and inverse coset NTT, wall time per
    column over one enqueue of all columns (10 repetitions), with the fraction of the HBM roofline (16 bytes per point)."""
    from ministark_amd import GOLDILOCKS_FP, ColumnSet, GpuFft, GpuIfft, GpuVec, Radix2EvaluationDomain
    rng = np.random.default_rng(5)
    out = {"workload": "configs[1] sweep: forward / inverse coset NTT (offset 7), Fp, in place, per column (the columns' pointer table built once: ColumnSet)", "peak_GBps": HBM_PEAK_GBS, "sizes": {}}
    for log_n, ncol in ((12, 512), (13, 512), (14, 256), (15, 256), (16, 128), (17, 64), (18, 64), (19, 64), (20, 32), (21, 32), (22, 16), (23, 16), (24, 8)):
        n = 1 << log_n
        cols = ColumnSet([GpuVec.from_numpy(pl, rng.integers(0, P_GOLDILOCKS, size=n, dtype=np.uint64), GOLDILOCKS_FP) for _ in range(ncol)])
        row = {"columns": ncol}
        for name, cls in (("forward", GpuFft), ("inverse", GpuIfft)):
            plan = cls(Radix2EvaluationDomain(n, 7), GOLDILOCKS_FP, pl)
            t_end = time.perf_counter() + 0.25
            while time.perf_counter() < t_end:
                plan.enqueue(cols)
                pl.sync()
            reps = 10
            t0 = time.perf_counter()
            for _ in range(reps):
                plan.enqueue(cols)
            pl.sync()
            us = (time.perf_counter() - t0) / reps / ncol * 1e6
            row[name + "_us_per_column"] = round(us, 2)
            row[name + "_hbm_frac"] = round(2.0 * n * 8 / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            plan.close()
        for c in cols:
            c.free()
        out["sizes"][f"2^{log_n}"] = row
    return out


def bench_reference_harness(pl):
    """The reference's own criterion harness (gpu/benches/fft.rs:18-74): sizes 2048 / 4096 / 32768 / 262144, ONE column, every iteration
    builds the plan (`GpuFft::from(domain)`), encodes the column and executes (a device synchronisation) -- a LATENCY figure, over the
    64-bit and the 252-bit field, subgroup and coset, forward and inverse.  The column is device-resident here (the reference's is in
    Apple's unified memory: no copy either).  Microseconds per iteration, median of 30."""
    from ministark_amd import GOLDILOCKS_FP, STARK252_FP, GpuFft, GpuIfft, GpuVec, Radix2EvaluationDomain
    rng = np.random.default_rng(18)
    out = {"workload": "gpu/benches/fft.rs: plan + encode + execute of one resident column per iteration (latency)", "unit": "us per iteration", "sizes": {}}
    for n in (2048, 4096, 32768, 262144):
        row = {}
        for fname, field, words in (("fp64", GOLDILOCKS_FP, 1), ("fp252", STARK252_FP, 4)):
            if words == 1:
                col = GpuVec.from_numpy(pl, rng.integers(0, P_GOLDILOCKS, size=n, dtype=np.uint64), field)
            else:
                a = rng.integers(0, 1 << 63, size=4 * n, dtype=np.uint64)
                a[3::4] >>= np.uint64(4)
                col = GpuVec.from_numpy(pl, a, field)
            gen = 7 if words == 1 else 3
            for vname, cls, dom in (("GpuFft", GpuFft, Radix2EvaluationDomain(n, 1, field)), ("GpuFft (coset)", GpuFft, Radix2EvaluationDomain(n, gen, field)),
                                    ("GpuIfft", GpuIfft, Radix2EvaluationDomain(n, 1, field)), ("GpuIfft (coset)", GpuIfft, Radix2EvaluationDomain(n, gen, field))):
                ts = []
                for it in range(34):
                    t0 = time.perf_counter()
                    plan = cls(dom, field, pl)
                    plan.encode(col)
                    plan.execute()
                    ts.append(time.perf_counter() - t0)
                    plan.close()
                row[f"{fname} {vname}"] = round(sorted(ts[4:])[15] * 1e6, 1)
            col.free()
        out["sizes"][str(n)] = row
    return out


def bench_lde_commit(pl, with_cpu, pmc=None):
    """configs[2] (C3): 2^20 rows x 32 columns, blow-up 8, coset NTT + Merkle commit on one GPU."""
    import numpy as np
    from ministark_amd import GOLDILOCKS_FP, Matrix, MerkleTree
    log_n, log_b, ncols = 20, 3, 32
    n, N = 1 << log_n, 1 << (log_n + log_b)
    rng = np.random.default_rng(3)
    P = (1 << 64) - (1 << 32) + 1
    host = [rng.integers(0, P, size=n, dtype=np.uint64) for _ in range(ncols)]
    trace = Matrix.from_numpy(pl, host, GOLDILOCKS_FP)
    state = {}

    def run():
        state.clear()
        lde = trace.lde(1 << log_b, 7, True)
        state["root"] = MerkleTree.from_matrix(lde).root()
    if pmc is not None:
        pmc("lde_commit", run)
        for c in trace.columns:
            c.free()
        return None
    wall, k = _profiled(pl, run, 3)
    lde_us = sum(v for name, v in k.items() if name.startswith(("ntt", "lde2")))     # iNTT passes + the two passes per coset
    lde_bytes = float(ncols) * (n * 8 + N * 8)
    hash_bytes = float(N) * ncols * 8 + 32.0 * N + 96.0 * N
    out = {"workload": "configs[2]: 2^20 rows x 32 columns (Fp), blow-up 8: interpolate + coset LDE (bit-reversed) + SHA-256 rows + Merkle tree",
           "wall_ms": round(wall * 1e3, 3), "kernel_us": k, "lde_kernel_ms": round(lde_us / 1e3, 3),
           "roofline": {"bound": "hbm", "kernel": "LDE passes (ntt_pass* + lde2_pass_*)", "algorithmic_bytes": lde_bytes,
                        "achieved": round(lde_bytes / (lde_us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(lde_bytes / (lde_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None},
           "commit": {"bound": "integer ALU (SHA-256)", "algorithmic_bytes": hash_bytes,
                      "compressions_per_s": round((N * (ncols * 8 // 64 + 1) + 2 * N) / (sum(v for nm, v in k.items() if nm.startswith("sha256")) * 1e-6), 0)},
           "root": state["root"].hex()}
    out["roofline"]["_res"] = ("lde_commit", lde_us, float(ncols) * N, ("msntt", "mslde2"), "output_point")
    out["commit"]["_res"] = ("lde_commit", sum(v for nm, v in k.items() if nm.startswith("sha256")), float(N) * (ncols * 8 // 64 + 1) + 2.0 * N, ("mssha",), "compression")
    if with_cpu:
        from oracle import cref
        t0 = time.perf_counter()
        cols = [cref.lde(c, log_n, log_b, 1, 7, True) for c in host]
        t1 = time.perf_counter()
        root = cref.sha256_merkle(cref.sha256_rows(cols, 1))[1].tobytes()
        t2 = time.perf_counter()
        out["cpu_baseline"] = {"value": round((t2 - t0) * 1e3, 1), "unit": "ms", "cores": cref.num_threads(), "kind": "port",
                               "lde_ms": round((t1 - t0) * 1e3, 1), "commit_ms": round((t2 - t1) * 1e3, 1), "root_matches": root == state["root"],
                               "sample": "the whole configs[2] matrix once, oracle/c (C/OpenMP restatement, not the reference binary)"}
    return out


def bench_lde_2_24(pl, pmc=None):
    """The LDE the prover of configs[4] runs (src/prover.rs:50-51, src/matrix.rs:245): 2^22 rows x 8 columns, blow-up 4 -> 2^24-point
    bit-reversed evaluations, in the order the prover asks for (natural in, bit-reversed out).  Since round 4 the coset transforms are
    two passes each (lde2_kernels.h, rows of 16384 words); the iNTT in front of them is the three-pass 2^22-point plan."""
    import numpy as np
    from ministark_amd import GOLDILOCKS_FP, Matrix
    log_n, log_b, ncols = 22, 2, 8
    n, N = 1 << log_n, 1 << (log_n + log_b)
    rng = np.random.default_rng(11)
    P = (1 << 64) - (1 << 32) + 1
    trace = Matrix.from_numpy(pl, [rng.integers(0, P, size=n, dtype=np.uint64) for _ in range(ncols)], GOLDILOCKS_FP)
    keep = {}

    def run():
        keep.clear()
        keep["lde"] = trace.lde(1 << log_b, 7, True)
    if pmc is not None:
        pmc("lde_2_24", run)
        keep.clear()
        for c in trace.columns:
            c.free()
        return None
    wall, k = _profiled(pl, run, 5)
    us = sum(k.values())
    alg = float(ncols) * (n * 8 + N * 8)                        # n s + beta n s per column
    moved = float(ncols) * (3 * 2 * n * 8 + (n * 8 + N * 8) + 2 * N * 8)    # what the passes read + write when nothing is re-read from cache
    for c in trace.columns:
        c.free()
    out = {"workload": "2^22 rows x 8 columns (Fp), blow-up 4: interpolate + bit-reversed coset evaluation on the 2^24-point domain (configs[4]'s base-trace LDE)",
           "wall_ms": round(wall * 1e3, 3), "kernel_us": k, "us_per_column": round(us / ncols, 1),
           "roofline": {"bound": "hbm", "kernel": "ntt_pass1-3 (iNTT, 2^22 points) + lde2_pass_a + lde2_pass_b", "algorithmic_bytes": alg,
                        "achieved": round(alg / (us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                        "bytes_the_passes_move": moved, "moved_over_algorithmic": round(moved / alg, 2)}}
    out["roofline"]["_res"] = ("lde_2_24", us, float(ncols) * N, None, "output_point")
    return out


def bench_prove(pl, with_cpu, pmc=None):
    """configs[4] on one GPU = BASELINE's "end-to-end prove time": ministark_amd/pipeline.py, 2^22 rows x 8 columns."""
    import numpy as np
    from ministark_amd import GOLDILOCKS_FP, Matrix, pipeline
    log_t, blowup, folding, ncols = 22, 4, 8, 8
    n_t = 1 << log_t
    rng = np.random.default_rng(5)
    P = (1 << 64) - (1 << 32) + 1
    trace = Matrix.from_numpy(pl, [rng.integers(0, P, size=n_t, dtype=np.uint64) for _ in range(ncols)], GOLDILOCKS_FP)
    comp, ce, nch = pipeline.fib_constraints(n_t, ncols)        # FibAirConfig::constraints (examples/fib/main.rs:73-140): ce_blowup_factor 1
    draws = pipeline.Draws(0xC5, ncols, nch, ce, 32, n_t * blowup, pipeline.fri_num_layers(n_t * blowup, blowup, folding, 64))
    res = {}

    timed = [False]                                             # the wall clock is taken without the waits at the phase boundaries

    def run():
        res.clear()
        res.update(pipeline.prove_phases(pl, trace, comp, draws, blowup, folding, 64, 8, ce_blowup=ce, time_phases=timed[0]))
    if pmc is not None:
        pmc("prove", run)
        res.clear()
        for c in trace.columns:
            c.free()
        return None
    phases = {}
    def phases_of_one_run():                                    # with the waits, without events
        timed[0] = True
        run(); pl.sync()
        phases.update(res["phases_ms"])
        timed[0] = False
    wall, k = _profiled(pl, run, 5, after_wall=phases_of_one_run)
    n_lde, n_ce = n_t * blowup, n_t * ce
    # algorithmic bytes per SURVEY.md 8(d): LDEs n s + beta n s per column, in-place transforms 2 n s, row hashing n cols s + 32 n,
    # trees 96 n, constraint evaluation sum of columns + result (on the n ce points of the constraint-evaluation domain), FRI
    # layers n s + n s / ff
    alg = (ncols * (n_t * 8 + n_lde * 8) + (n_lde * ncols * 8 + 128 * n_lde) + (ncols + 1) * n_ce * 8 + 2 * n_ce * 8
           + ce * (n_t * 8 + n_lde * 8) + (n_lde * ce * 8 + 128 * n_lde) + (ncols + ce + 1) * n_t * 8 + (n_t * 8 + n_lde * 8)
           + sum((n_lde >> (3 * i)) * 8 * (1 + 1 / 8) + 128 * (n_lde >> (3 * i + 3)) for i in range(len(draws.fri_alphas))))
    kernel_ms = sum(k.values()) / 1e3
    out = {"workload": "configs[4] on one GPU: 2^22 rows x 8 columns (Fp, Fq = Fp), the reference's fib AIR (examples/fib/main.rs:73-140: 17 constraints, ce_blowup_factor 1), ProofOptions::new(32, 4, 8, 8, 64): every data-parallel phase of default_prove, fixed challenges in place of the channel; prove_ms = median wall time of runs that wait only where the proof "
                       "does (roots, out-of-domain values, the opened rows), phases_ms from one more run that also waits at every phase boundary",
           "prove_ms": round(wall * 1e3, 3), "kernel_ms": round(kernel_ms, 3), "phases_ms": phases, "kernel_us": k,
           "roofline": {"bound": "hbm (NTT / evaluation / FRI) + integer ALU (SHA-256)", "algorithmic_bytes": float(alg),
                        "achieved": round(alg / (kernel_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(alg / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None},
           "base_root": res["base_root"].hex(), "nonce": res["nonce"]}
    out["roofline"]["_res"] = ("prove", kernel_ms * 1e3, float(n_t) * ncols, None, "trace_cell")
    try:
        # the same proof through distributed.prove_sharded over a ONE-rank communicator, the two provers taking turns on the same trace
        # (the sharded_lde_commit object is timed minutes earlier in the run, on other data): what N = 1 of the multi-GPU path costs
        from ministark_amd.distributed import RcclComm, prove_sharded
        with _stdout_to_stderr():
            comm = RcclComm(pl, 0, 1, RcclComm.unique_id(pl.lib))
            try:
                a_ms, b_ms, same = [], [], True
                for it in range(6):
                    pl.sync()
                    t0 = time.perf_counter()
                    run()
                    pl.sync()
                    t1 = time.perf_counter()
                    sh = prove_sharded(pl, comm, list(trace.columns), ncols, log_t, comp, draws, blowup, folding, 64, 8, ce_blowup=ce)
                    pl.sync()
                    t2 = time.perf_counter()
                    same = same and sh["base_root"] == res["base_root"] and sh["nonce"] == res["nonce"]
                    if it:
                        a_ms.append((t1 - t0) * 1e3)
                        b_ms.append((t2 - t1) * 1e3)
            finally:
                comm.close()
        a, b = sorted(a_ms)[len(a_ms) // 2], sorted(b_ms)[len(b_ms) // 2]
        out["one_rank_sharded_interleaved"] = {"prove_ms": round(a, 3), "prove_sharded_ms": round(b, 3), "ratio": round(b / a, 4), "same_root_and_nonce": bool(same),
                                               "how": "5 timed rounds of pipeline.prove_phases then distributed.prove_sharded (world size 1) on the same trace, medians"}
    except Exception as e:                                       # noqa: BLE001 -- for information; the figures above stand
        out["one_rank_sharded_interleaved"] = {"error": f"{type(e).__name__}: {e}"}
    for c in trace.columns:
        c.free()
    out["native_host"] = _native_prove(log_t)
    if with_cpu:
        from oracle import cref
        from oracle.prover_chain import c5_oracle_chain as _c5_oracle_chain
        cols = [cref.random_elements(n_t, 77 + c) for c in range(ncols)]                    # the same size: the whole chain once
        t0 = time.perf_counter()
        _c5_oracle_chain(cols, log_t, blowup, folding, draws, comp, ce)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(dt * 1e3, 1), "unit": "ms", "cores": cref.num_threads(), "kind": "port",
                               "sample": f"the same chain once at the same size (2^{log_t} rows x {ncols} columns), oracle/c (C/OpenMP restatement, not the reference binary) + numpy glue"}
    return out


def _native_prove(log_rows, reps=5):
    """The same chain driven by the C++ host mirror instead of Python + ctypes: examples/fib_prover.cpp (a VALID fib trace,
    the same AIR / options; its own process and context).  Wall time per proof, median of `reps`."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "fib_prover")
    so = os.path.join(ROOT, "ministark_amd", "libministark_hip.so")
    try:
        if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(so), os.path.getmtime(os.path.join(ROOT, "examples", "fib_prover.cpp"))):
            os.makedirs(os.path.dirname(exe), exist_ok=True)
            subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "examples", "fib_prover.cpp"), "-o", exe, so,
                                   "-Wl,-rpath," + os.path.dirname(so), "-Wl,-rpath,/opt/rocm/lib"], timeout=300)
        r = subprocess.run([exe, str(log_rows), str(reps)], capture_output=True, text=True, timeout=300)
        rows = re.findall(r"rep \d+: base LDE\+commit ([\d.]+) \| evaluation ([\d.]+) \| composition ([\d.]+) \| DEEP ([\d.]+) \| FRI ([\d.]+) \| PoW\+queries\+openings ([\d.]+) \| total ([\d.]+) ms", r.stdout)
        if r.returncode != 0 or not rows or "fib prover pipeline ok" not in r.stdout:
            return {"error": (r.stdout + r.stderr)[-400:]}
        rows = sorted(([float(v) for v in row] for row in rows), key=lambda row: row[-1])
        med = rows[len(rows) // 2]
        names = ("base trace: interpolate + LDE + commit", "constraint evaluation", "composition trace: iNTT + split + LDE + commit",
                 "DEEP: OOD evaluations + composition + LDE", "FRI layers (commit + fold) + remainder", "proof of work + queries + FRI openings")
        return {"program": "examples/fib_prover.cpp over ministark_amd/csrc/host/*.hpp (C++ host mirror), valid fib trace of the same shape",
                "prove_ms": med[-1], "best_ms": rows[0][-1], "repetitions": len(rows), "phases_ms": dict(zip(names, med[:-1]))}
    except Exception as e:                                   # noqa: BLE001 -- an extra; the Python-driven number stands
        return {"error": f"{type(e).__name__}: {e}"}


def bench_constraint_eval(pl, with_cpu, pmc=None):
    """configs[3] (C4): constraint composition evaluation on 2^23 points, three AIRs:
    (i) the reference's fib AIR, 8 Fp columns; (ii) 17 Fp + 9 Fq3 columns (the brainfuck shape); (iii) the fib AIR over the
    252-bit field.  Algorithmic bytes = sum over columns of n s_col + n s_Fq for the result (x is generated on the fly)."""
    import numpy as np
    from ministark_amd import GOLDILOCKS_FP, GOLDILOCKS_FQ3, STARK252_FP, GpuVec, expr as E, pipeline
    log_n = 23
    n = 1 << log_n
    rng = np.random.default_rng(23)
    P = (1 << 64) - (1 << 32) + 1
    out = {"workload": "configs[3]: one fused evaluation of the composition constraint over 2^23 points of the coset 7<w>"}

    def gl_cols(k, V=1):
        return [rng.integers(0, P, size=n * V, dtype=np.uint64) for _ in range(k)]

    def f252_cols(k):
        cols = [rng.integers(0, 1 << 63, size=4 * n, dtype=np.uint64) for _ in range(k)]
        for c in cols:
            c[3::4] >>= np.uint64(4)                       # canonical residues below 2^251 < p
        return cols
    cases = []
    comp, _, nch = pipeline.fib_constraints(n)                                   # lde_step = ce_blowup_factor = 1 (src/prover.rs:103)
    cases.append(("fib_air_fp", "(i) FibAirConfig::constraints (examples/fib/main.rs:73-140), 8 Fp columns, Fq = Fp, lde_step 1", comp, 1, 7, GOLDILOCKS_FP, False,
                  gl_cols(8), [], rng.integers(1, P, size=(nch, 1), dtype=np.uint64), 8 * 8 + 8, "goldilocks", log_n))
    comp, nch = pipeline.mixed_air_constraints()
    cases.append(("mixed_17fp_9fq3", "(ii) 17 Fp + 9 Fq3 columns (examples/brainfuck/air.rs:26-27 shape), lde_step 2", comp, 2, 7, GOLDILOCKS_FP, True,
                  gl_cols(17), gl_cols(9, 3), rng.integers(1, P, size=(nch, 3), dtype=np.uint64), 17 * 8 + 9 * 24 + 24, "goldilocks", log_n))
    comp, _, nch = pipeline.fib_constraints(n >> 2, 8, STARK252_FP)
    cases.append(("fib_air_fp252", "(iii) the fib AIR over the 252-bit field (src/eval_gpu.rs:1054-1082), 8 columns, lde_step 4", comp, 4, 3, STARK252_FP, False,
                  f252_cols(8), [], rng.integers(0, 1 << 59, size=(nch, 4), dtype=np.uint64), 8 * 32 + 32, "f252", log_n))        # (the whole domain since round 6: 11 s of host time, and every output word compared)
    for key, what, comp, lde_step, offset, field, fq_ext, base, ext, ch, bytes_per_point, oracle_field, cpu_log in cases:
        prog = E.compile_expr(comp, len(base), fq_ext, field)
        dbase = [GpuVec.from_numpy(pl, c, field) for c in base]
        dext = [GpuVec.from_numpy(pl, c, GOLDILOCKS_FQ3) for c in ext]
        res = {}

        def run():
            res["out"] = E.eval(prog, pl, ch, ch[:1], lde_step, offset, n, dbase, dext)
        if pmc is not None:
            pmc("constraint_eval." + key, run)
            del dbase, dext, res
            continue
        wall, k = _profiled(pl, run, 5)
        us = sum(k.values())
        alg = float(bytes_per_point) * n
        obj = {"workload": what, "instructions": len(prog.instrs), "wall_ms": round(wall * 1e3, 3), "kernel_us": k,
               "roofline": {"bound": "hbm" if key == "fib_air_fp" else "integer ALU (extension-field / 252-bit products) over an HBM stream",
                            "algorithmic_bytes": alg, "achieved": round(alg / (us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None}}
        obj["roofline"]["_res"] = ("constraint_eval." + key, us, float(n), None, "point")
        if with_cpu:
            from oracle import cref
            m = 1 << cpu_log                                   # bounded sample: the first 2^cpu_log points of the same columns
            Vb = 4 if oracle_field == "f252" else 1
            t0 = time.perf_counter()
            want = cref.eval_expr(comp, cpu_log, lde_step, offset, [c[:m * Vb] for c in base], [c[:3 * m] for c in ext], ch, ch[:1], fq_ext,
                                  **({"field": "f252"} if oracle_field == "f252" else {}))
            dt = time.perf_counter() - t0
            obj["cpu_baseline"] = {"value": round(dt * 1e3, 1), "unit": "ms", "cores": cref.num_threads(), "kind": "port",
                                   "points": m, "us_per_point": round(dt * 1e6 / m, 4),
                                   "sample": f"oracle_eval_expr (eval_cpu::eval restated: 512-point chunks, batch inversion) on 2^{cpu_log} points"
                                             + (" = the whole domain" if cpu_log == log_n else f" (1/{1 << (log_n - cpu_log)} of the domain, same columns' prefix, trace_len scaled with it)")}
            if cpu_log == log_n:
                obj["cpu_baseline"]["matches_device"] = bool(np.array_equal(res["out"].to_numpy(), want))
        out[key] = obj
        del dbase, dext, res
    return None if pmc is not None else out


def _cold_child_main():
    """--cold-child: what a process that proves ONCE pays (the reference's usage: examples/fib/main.rs:227-243).  Fresh process, nothing
    created yet: context -> (trace upload, not counted) -> first proof (plans, twiddle uploads, kernel code loading, the constraint
    kernels: hiprtc or the on-disk cache) -> second proof (warm).  Then the three configs[3] programs on 2^16 points each, for their
    compilation / cache-load cost alone.  Prints one JSON object on stdout."""
    t_proc = time.perf_counter()
    from ministark_amd import GOLDILOCKS_FP, GOLDILOCKS_FQ3, STARK252_FP, GpuVec, Matrix, Planner, expr as E, pipeline
    t_imp = time.perf_counter()
    pl = Planner(int(os.environ.get("LOCAL_RANK", "0")))
    pl.sync()
    t_ctx = time.perf_counter()
    log_t, blowup, folding, ncols = 22, 4, 8, 8
    n_t = 1 << log_t
    rng = np.random.default_rng(5)
    host = [rng.integers(0, P_GOLDILOCKS, size=n_t, dtype=np.uint64) for _ in range(ncols)]
    comp, ce, nch = pipeline.fib_constraints(n_t, ncols)
    draws = pipeline.Draws(0xC5, ncols, nch, ce, 32, n_t * blowup, pipeline.fri_num_layers(n_t * blowup, blowup, folding, 64))
    t_up0 = time.perf_counter()
    trace = Matrix.from_numpy(pl, host, GOLDILOCKS_FP)
    pl.sync()
    t_up1 = time.perf_counter()
    times, roots = [], []
    for _ in range(3):
        t0 = time.perf_counter()
        res = pipeline.prove_phases(pl, trace, comp, draws, blowup, folding, 64, 8, ce_blowup=ce)
        pl.sync()
        times.append((time.perf_counter() - t0) * 1e3)
        roots.append(res["base_root"].hex())
        if len(times) == 1:
            first_phases, first_jit = dict(res["phases_ms"]), pl.jit_stats()
    out = {"import_ms": round((t_imp - t_proc) * 1e3, 1), "context_ms": round((t_ctx - t_imp) * 1e3, 1), "trace_upload_ms": round((t_up1 - t_up0) * 1e3, 1),
           "first_prove_ms": round(times[0], 2), "second_prove_ms": round(times[1], 2), "third_prove_ms": round(times[2], 2),
           "cold_prove_ms": round((t_ctx - t_imp) * 1e3 + times[0], 2), "first_prove_phases_ms": first_phases, "first_prove_jit": first_jit,
           "same_root": len(set(roots)) == 1, "base_root": roots[0]}
    for c in trace.columns:
        c.free()
    # the three constraint programs of configs[3] on a small domain: their first evaluation in this process, by itself
    n = 1 << 16
    P = P_GOLDILOCKS
    jit = {}
    cases = []
    comp, _, nch = pipeline.fib_constraints(n)
    cases.append(("fib_air_fp", comp, 1, 7, GOLDILOCKS_FP, False, [rng.integers(0, P, size=n, dtype=np.uint64) for _ in range(8)], [],
                  rng.integers(1, P, size=(nch, 1), dtype=np.uint64)))
    comp, nch = pipeline.mixed_air_constraints()
    cases.append(("mixed_17fp_9fq3", comp, 2, 7, GOLDILOCKS_FP, True, [rng.integers(0, P, size=n, dtype=np.uint64) for _ in range(17)],
                  [rng.integers(0, P, size=3 * n, dtype=np.uint64) for _ in range(9)], rng.integers(1, P, size=(nch, 3), dtype=np.uint64)))
    comp, _, nch = pipeline.fib_constraints(n >> 2, 8, STARK252_FP)
    f252 = []
    for _ in range(8):
        c = rng.integers(0, 1 << 63, size=4 * n, dtype=np.uint64)
        c[3::4] >>= np.uint64(4)
        f252.append(c)
    cases.append(("fib_air_fp252", comp, 4, 3, STARK252_FP, False, f252, [], rng.integers(0, 1 << 59, size=(nch, 4), dtype=np.uint64)))
    for key, comp, lde_step, offset, field, fq_ext, base, ext, ch in cases:
        prog = E.compile_expr(comp, len(base), fq_ext, field)
        dbase = [GpuVec.from_numpy(pl, c, field) for c in base]
        dext = [GpuVec.from_numpy(pl, c, GOLDILOCKS_FQ3) for c in ext]
        pl.sync()
        b = pl.jit_stats()
        t0 = time.perf_counter()
        E.eval(prog, pl, ch, ch[:1], lde_step, offset, n, dbase, dext)
        pl.sync()
        first = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        E.eval(prog, pl, ch, ch[:1], lde_step, offset, n, dbase, dext)
        pl.sync()
        a = pl.jit_stats()
        jit[key] = {"first_eval_ms": round(first, 2), "second_eval_ms": round((time.perf_counter() - t0) * 1e3, 2),
                    "compile_ms": round(a["compile_ms"] - b["compile_ms"], 2), "load_ms": round(a["load_ms"] - b["load_ms"], 2),
                    "kernels_compiled": a["kernels_compiled"] - b["kernels_compiled"], "kernels_from_disk": a["kernels_from_disk"] - b["kernels_from_disk"],
                    "compile_failures": a["compile_failures"] - b["compile_failures"]}
    if not (jit["fib_air_fp"]["kernels_compiled"] or jit["fib_air_fp"]["kernels_from_disk"]):
        # the same AIR as the proof above: its kernel is already in this context's table -- what it cost is the first proof's record
        jit["fib_air_fp"].update({k: (round(first_jit[k], 2) if isinstance(first_jit[k], float) else first_jit[k]) for k in ("compile_ms", "load_ms", "kernels_compiled", "kernels_from_disk")},
                                 note="compiled / loaded during the first proof (same program)")
    out["constraint_programs"] = jit
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()


def bench_cold_start():
    """Two fresh processes (--cold-child), the first with an EMPTY on-disk kernel cache, the second with the cache the first one filled:
    the cost of a first-ever proof on a machine, and of the first proof of every later process.  Bar: the reference pays zero run-time
    compilation (gpu/src/plan.rs:30)."""
    import shutil
    import subprocess
    import tempfile
    work = tempfile.mkdtemp(prefix="ms_jit_cold_", dir="/tmp")
    try:
        runs = []
        for _ in range(2):
            # the compiler's own cache (comgr, ~/.cache/comgr) is switched off in BOTH processes: the first must really compile,
            # the second must owe what it saves to the library's cache alone
            env = dict(os.environ, MS_JIT_CACHE=os.path.join(work, "cache"), AMD_COMGR_CACHE="0")
            r = subprocess.run([sys.executable, BENCH_PY, "--cold-child"], env=env, capture_output=True, text=True, timeout=600)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                return {"error": (r.stdout + r.stderr)[-600:]}
            runs.append(json.loads(lines[-1]))
        empty, cached = runs
        entries = [f for f in os.listdir(os.path.join(work, "cache")) if f.endswith(".co")]
        return {"workload": "a fresh process: context + first proof of configs[4]'s shape (2^22 rows x 8 columns, fib AIR); then the configs[3] programs on 2^16 points",
                "cold_prove_ms": empty["cold_prove_ms"], "cold_prove_cached_ms": cached["cold_prove_ms"], "warm_prove_ms": min(cached["second_prove_ms"], cached["third_prove_ms"]),
                "jit_compile_ms": {k: v["compile_ms"] for k, v in empty["constraint_programs"].items()},
                "jit_cached_load_ms": {k: v["load_ms"] for k, v in cached["constraint_programs"].items()},
                "same_root_both_processes": empty["base_root"] == cached["base_root"] and empty["same_root"] and cached["same_root"],
                "cache_entries": len(entries), "cache_bytes": sum(os.path.getsize(os.path.join(work, "cache", f)) for f in entries),
                "empty_cache_process": empty, "cached_process": cached}
    except Exception as e:                                       # noqa: BLE001 -- an extra
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        shutil.rmtree(work, ignore_errors=True)
