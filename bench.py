#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

metric   : algorithmic GB/s (and field-ops/s) of the 2^24-point Goldilocks forward NTT
workload : configs[1] -- a batch of COLS columns of 2^24 canonical-uniform Goldilocks
           elements (Montgomery words), coset offset 7, transformed in place, resident in
           HBM before the timed region.  A "step" = one transform of every column.
value    : whole-job algorithmic bytes (2 * n * 8 per column: one compulsory read + one
           write) / wall time of the K timed steps, max over ranks.
N > 1    : one process per GPU, columns are independent => each rank owns its own COLS
           columns (weak scaling), no data-path collective; the only exchange is the timing
           barrier / max-reduce.  Launch: either `python -m torch.distributed.run
           --nproc-per-node N ... bench.py --gpus N` (RANK / LOCAL_RANK / WORLD_SIZE /
           MASTER_* from the environment), or plain `python bench.py --gpus N`: without
           WORLD_SIZE in the environment the process starts the N ranks itself (127.0.0.1,
           a free port) and relays rank 0's JSON line.  `n_gpus` on the line is the number
           of ranks that actually joined the process group; --gpus != that number is an
           error (exit code 2), never a silently smaller run.

Extra objects on the JSON line:
  roofline     : the transform against the HBM roofline.  achieved = algorithmic bytes of
                 one column transform / summed average duration of its kernel launches,
                 measured live with hipEvents on the library's stream; `kernels` lists
                 each launch (avg_us, its own bytes moved) for comparison with
                 profiles/*kernel_stats*.  `traffic` = HBM bytes per transform, MEASURED IN THE
                 RUN at N = 1 (two short child runs under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE: _measure_traffic) and
                 otherwise taken from the summary of the same passes under profiles/ (--traffic-json); `traffic_source` says which.
  cpu_baseline : oracle/c (C/OpenMP restatement of the reference CPU path, kind "port")
                 timed on this box's host cores on one 2^24 column, same data.
  lde_2_24     : the prover's own transform order at configs[4]'s size -- interpolate + bit-reversed coset evaluation of 2^22-row columns
                 on the 2^24-point domain (two passes per coset since round 4), with its own roofline.      [N = 1 only]
  lde_commit   : configs[2] (C3): 2^20 rows x 32 columns, blow-up 8, fused LDE + SHA-256 row hashing + Merkle tree,
                 with its own roofline (LDE kernels against HBM; algorithmic bytes n s + beta n s per column,
                 SURVEY.md 8(d)) and cpu_baseline (oracle/c on the same matrix).                       [N = 1 only]
  constraint_eval : configs[3] (C4): one fused evaluation of the composition constraint over 2^23 points for (i) the reference's
                 fib AIR on 8 Fp columns, (ii) 17 Fp + 9 Fq3 columns, (iii) the fib AIR over the 252-bit field: kernel time,
                 algorithmic bytes (columns read once + result), fraction of the HBM roofline, and
                 oracle_eval_expr (the restated eval_cpu::eval) timed on the host cores.                [N = 1 only]
  prove        : the second half of BASELINE's metric, "end-to-end prove time": every data-parallel phase of
                 default_prove on configs[4]'s shape (2^22 rows x 8 columns, ProofOptions::new(32, 4, 8, 8, 64)),
                 device-resident, fixed challenges in place of the channel (ministark_amd/pipeline.py); phases,
                 per-kernel time, and the oracle chain timed once at the same size.                   [N = 1 only]
                 `one_rank_sharded_interleaved`: the same proof through distributed.prove_sharded over a one-rank
                 communicator, the two provers taking turns on the same trace (what N = 1 of the multi-GPU path costs).
  sharded_lde_commit : configs[4]'s multi-GPU step for any N (also N = 1, where RCCL runs with one rank): a
                 2^22-row x 32-column trace, blow-up 4, columns sharded over the ranks -> LDE (no communication) ->
                 ms_cols_to_rows_alltoall -> row hashing + subtree -> ms_allgather_digests -> top levels.  The total
                 work is fixed ("strong"): the driver's N = 1, 2, 4, 8 runs give the scaling curve of the north
                 star's "column-sharded LDE"; `lde_ms` is the phase its >= 6x target refers to.
  --mode lde-commit runs only that last measurement and prints it as the line's `value` (GB/s algorithmic).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bench_objects import (HBM_PEAK_GBS, P_GOLDILOCKS, VALU_ISSUE_PEAK, _attach_all, _cold_child_main, _measure_objects, _pmc_child_main,  # noqa: E402
                           _stdout_to_stderr, bench_c2_sweep, bench_cold_start, bench_constraint_eval, bench_lde_2_24, bench_lde_commit,
                           bench_prove, bench_reference_harness, sharded_lde_commit, sharded_prove)

LOG_N = 24


_LINE_OUT = None


def _claim_stdout():
    """From here on file descriptor 1 IS stderr for everything in this process -- torch's own RCCL prints the same banner at its
    first collective, C stdio flushes at exit -- and the one JSON line goes out through a private duplicate of the real stdout."""
    global _LINE_OUT
    if _LINE_OUT is None:
        sys.stdout.flush()
        _LINE_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


LINE_LIMIT = 4096          # the driver recovers the line from a bounded tail of stdout: the printed line stays below this, always


def _g(o, *path):
    """o[path[0]][path[1]]... or None."""
    for k in path:
        if not isinstance(o, dict) or k not in o:
            return None
        o = o[k]
    return o


def compact_line(d):
    """The ONE line printed on stdout, built from the full record `d` (which goes to bench_detail.json): the contract's keys, the
    dominant kernel's `roofline`, `cpu_baseline`, and one scalar per secondary object.  Pure: tests/test_bench_line.py builds it from
    a recorded detail file and checks the size and the keys."""
    r = d.get("roofline") or {}
    line = {k: d.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                  "vs_baseline", "dtype", "data")}
    cfg = d.get("config") or {}
    line["config"] = {k: cfg[k] for k in ("workload", "columns_per_gpu", "log_n", "parallelism") if k in cfg}
    if r:
        line["roofline"] = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_transform",
                                                  "us_per_transform_events", "valu_insts_per_element", "frac_of_valu_issue_peak")}
        line["roofline"]["kernel"] = r.get("kernel")
        if r.get("traffic") and r.get("algorithmic_bytes_per_transform"):
            line["roofline"]["traffic_over_algorithmic"] = round(r["traffic"] / r["algorithmic_bytes_per_transform"], 2)
    if isinstance(d.get("cpu_baseline"), dict):
        line["cpu_baseline"] = {k: d["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "kind", "ms_per_transform", "sample")}
    sec = {
        "field_ops_per_s": d.get("field_ops_per_s"),
        "us_per_transform": d.get("us_per_transform"),
        "single_column_us": _g(d, "variants", "single_column_repeated_us_per_transform"),
        "inverse_coset_us": _g(d, "variants", "inverse_coset_us_per_transform"),
        "lde_commit_frac": _g(d, "lde_commit", "roofline", "frac"),
        "lde_commit_traffic_x": _g(d, "lde_commit", "roofline", "traffic_over_algorithmic"),
        "lde_commit_wall_ms": _g(d, "lde_commit", "wall_ms"),
        "lde_2_24_frac": _g(d, "lde_2_24", "roofline", "frac"),
        "lde_2_24_traffic_x": _g(d, "lde_2_24", "roofline", "traffic_over_algorithmic"),
        "lde_fq3_4x2_20_b8_ms": _g(d, "lde_fq3", "kernel_ms"),
        "c4_i_frac": _g(d, "constraint_eval", "fib_air_fp", "roofline", "frac"),
        "c4_ii_frac": _g(d, "constraint_eval", "mixed_17fp_9fq3", "roofline", "frac"),
        "c4_iii_frac": _g(d, "constraint_eval", "fib_air_fp252", "roofline", "frac"),
        "c4_iii_traffic_x": _g(d, "constraint_eval", "fib_air_fp252", "roofline", "traffic_over_algorithmic"),
        "c4_all_outputs_equal_oracle": (all(_g(d, "constraint_eval", k, "cpu_baseline", "matches_device") is True for k in ("fib_air_fp", "mixed_17fp_9fq3", "fib_air_fp252"))
                                        if _g(d, "constraint_eval", "fib_air_fp", "cpu_baseline") else None),
        "lde_commit_root_equals_oracle": _g(d, "lde_commit", "cpu_baseline", "root_matches"),
        "prove_ms": _g(d, "prove", "prove_ms"),
        "prove_kernel_ms": _g(d, "prove", "kernel_ms"),
        "prove_native_ms": _g(d, "prove", "native_host", "prove_ms"),
        "prove_cpu_baseline_ms": _g(d, "prove", "cpu_baseline", "value"),
        "cold_prove_ms": _g(d, "cold_start", "cold_prove_ms"),
        "cold_prove_cached_ms": _g(d, "cold_start", "cold_prove_cached_ms"),
        "jit_compile_ms": _g(d, "cold_start", "jit_compile_ms"),
        "jit_cached_load_ms": _g(d, "cold_start", "jit_cached_load_ms"),
        "sharded_lde_ms": _g(d, "sharded_lde_commit", "lde_ms"),
        "sharded_exchange_ms": _g(d, "sharded_lde_commit", "exchange_ms"),
        "sharded_commit_ms": _g(d, "sharded_lde_commit", "commit_ms"),
        "sharded_prove_ms": _g(d, "sharded_lde_commit", "prove", "prove_ms"),
        "sharded_root": (_g(d, "sharded_lde_commit", "root") or "")[:16] or None,
        "sharded_error": _g(d, "sharded_lde_commit", "error"),
        "multi_gpu": d.get("multi_gpu_note"),
        "detail": d.get("detail_file"),
    }
    line.update({k: v for k, v in sec.items() if v is not None})
    if isinstance(d.get("sharded_lde_commit"), dict) and d.get("mode") == "lde-commit":     # that mode's own object, without its prose
        slim = lambda o: {k: (slim(v) if isinstance(v, dict) else v) for k, v in o.items() if k not in ("workload", "phases_ms_this_rank")}
        line["sharded_lde_commit"] = slim(d["sharded_lde_commit"])
    # the bound is a promise: shed the optional scalars, last first, until the line fits
    optional = [k for k in sec if k in line]
    while len(json.dumps(line)) >= LINE_LIMIT - 1 and optional:
        line.pop(optional.pop())
    return line


def _write_detail(obj):
    """The full record (every object, per-kernel counters, sweeps) next to bench.py and, on a gpurun box, under gpurun_out/ so that it
    comes back; profiles/rNN_bench_detail.json is a copy of it.  -> the path written (relative), or None."""
    wrote = None
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if d != ROOT and not os.path.isdir(d):
                continue
            with open(os.path.join(d, "bench_detail.json"), "w") as f:
                json.dump(obj, f, indent=1)
            wrote = wrote or os.path.relpath(os.path.join(d, "bench_detail.json"), ROOT)
        except OSError:
            pass
    return wrote


def _emit(obj):
    """Rank 0, once: the full record to bench_detail.json, its summary (< LINE_LIMIT bytes) as the one JSON line on stdout."""
    obj["detail_file"] = _write_detail(obj)
    line = json.dumps(compact_line(obj))
    assert len(line) < LINE_LIMIT, len(line)
    out = _LINE_OUT if _LINE_OUT is not None else sys.stdout
    out.write(line + "\n")
    out.flush()


def _measure_traffic(log_n):
    """HBM bytes of ONE forward coset transform, measured now: two short child runs of this script under `rocprofv3 --kernel-trace --pmc`
    (FETCH_SIZE, then WRITE_SIZE: separate passes, no tracing domains, as /opt/skills/guides/MI355X_MICROARCH.md prescribes), reduced per
    launch and per column like scripts/summarise_profiles.py does (FETCH_SIZE doubled: gfx950 tallies a coalesced stream at 64 B per request;
    units of 1024 B).  -> (bytes, per-kernel dict) or None when rocprofv3 is absent, this process is itself being profiled, or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None or os.environ.get("MS_BENCH_NO_PMC") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None
    per = {}
    work = tempfile.mkdtemp(prefix="ms_pmc_", dir="/tmp")
    try:
        for counters in (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES")):
            out = os.path.join(work, counters[0])
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", *counters, "-d", out, "-o", "t", "--output-format", "csv", "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--cols", "2", "--log-n", str(log_n), "--no-cpu-baseline", "--no-extras", "--settle", "0.3"]
            env = dict(os.environ, MS_BENCH_NO_PMC="1", TMPDIR="/tmp")
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = pr.wait(timeout=120)
            except subprocess.TimeoutExpired:                    # the profiler AND the run under it: the whole process group it leads
                import signal
                os.killpg(pr.pid, signal.SIGKILL)
                pr.wait()
                return None
            hits = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if rc != 0 or not hits:
                return None
            for row in csv.DictReader(open(hits[0])):
                name = row["Kernel_Name"]
                # the three launches of the forward coset transform (the variants of the child run are off with --no-extras ... they are not:
                # subgroup / inverse plans run too and are told apart by their template arguments, as in scripts/summarise_profiles.py)
                counter = row["Counter_Name"]
                if "msntt2" not in name or counter not in counters:
                    continue
                if not any(t in name for t in ("ntt2_first_pass<true, false, true, 16", "ntt2_mid_pass<true, false, false, 0", "ntt2_mid_pass<true, false, true, 0",
                                               "ntt2_first_pass<false, false, true, 16", "ntt2_mid_pass<false, false, false, 0", "ntt2_mid_pass<false, false, true, 0")):
                    continue
                cols = max(1.0, float(row["Grid_Size"]) / (((1 << log_n) // 16384) * 512))
                per.setdefault(name, {}).setdefault(counter, []).append(float(row["Counter_Value"]) / cols)
        total, insts, kernels = 0.0, 0.0, {}
        mean = lambda v: sum(v) / len(v)
        for name, c in per.items():
            if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
                return None
            fb = 2 * 1024 * mean(c["FETCH_SIZE"])
            wb = 1024 * mean(c["WRITE_SIZE"])
            k = kernels[name.split("(")[0].replace("void ", "")] = {"fetch_bytes_per_column": round(fb), "write_bytes_per_column": round(wb)}
            if "SQ_INSTS_VALU" in c:                             # wave instructions of one column's launch -> per element (64 lanes)
                k["valu_insts_per_element"] = round(mean(c["SQ_INSTS_VALU"]) * 64.0 / (1 << log_n), 1)
                insts += mean(c["SQ_INSTS_VALU"])
                if c.get("SQ_BUSY_CYCLES") and mean(c["SQ_BUSY_CYCLES"]):
                    k["valu_busy"] = round((mean(c["SQ_ACTIVE_INST_VALU"]) * 4.0 / 1024) / (mean(c["SQ_BUSY_CYCLES"]) / 32.0), 3)
            total += fb + wb
        return (total, kernels, insts or None) if len(per) == 3 else None
    except Exception:                                            # noqa: BLE001 -- an extra: the line falls back to the tracked summary
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


def _spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one per GPU) as children of this process with the
    environment torch.distributed.run would give them, pass rank 0's stdout (the JSON line) through, fail if any rank fails."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's line is read on a thread while all ranks are watched: a rank that dies at start-up would otherwise leave rank 0
    # waiting in its rendezvous (and this process in read()) until torch's own timeout
    import threading
    box = {}
    reader = threading.Thread(target=lambda: box.__setitem__("out", procs[0].stdout.read().decode()), daemon=True)
    reader.start()
    deadline = time.time() + 1800
    while reader.is_alive() and time.time() < deadline:
        reader.join(timeout=0.5)
        if any(pr.poll() not in (None, 0) for pr in procs):     # a failed rank: the others cannot finish a collective with it
            time.sleep(2.0)                                        # (let a rank that is already printing finish)
            break
    if reader.is_alive():                                      # rank 0 never closed its line: a failed rank or the deadline
        for pr in procs:                                       # SIGTERM first: rank 0 prints what it has measured (on_term), then the hard stop
            if pr.poll() is None:
                pr.terminate()
        reader.join(timeout=10.0)
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    reader.join(timeout=5.0)
    out0 = box.get("out", "")
    rcs = []
    for r, pr in enumerate(procs):
        try:
            rcs.append(pr.wait(timeout=60))
        except subprocess.TimeoutExpired:
            pr.kill()
            rcs.append(-9)
    if any(rcs):
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        print(f"bench.py: ranks exited with {rcs}", file=sys.stderr)
        sys.stdout.write(out0)
        return max(1, max(abs(c) for c in rcs))
    sys.stdout.write(out0)
    sys.stdout.flush()
    return 0


def _latest_traffic_json():
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_ntt_traffic.json")))
    return hits[-1] if hits else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cols", type=int, default=8, help="columns of 2^24 per rank per step")
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--settle", type=float, default=1.0, help="seconds of untimed transforms before the warm-up steps (clock ramp)")
    ap.add_argument("--traffic-json", default=_latest_traffic_json(), help="JSON file with PMC-derived HBM bytes per transform (default: the newest profiles/rNN_ntt_traffic.json)")
    ap.add_argument("--mode", choices=["ntt", "lde-commit"], default="ntt", help="lde-commit: only the column-sharded LDE + commitment (any N)")
    ap.add_argument("--no-extras", action="store_true", help="skip the lde_commit / prove / sharded objects")
    ap.add_argument("--pmc-child", action="store_true", help="internal: every object's workload between marker launches, under rocprofv3 --pmc")
    ap.add_argument("--cold-child", action="store_true", help="internal: a fresh process's first proof (bench_cold_start)")
    ap.add_argument("--log-rows", type=int, default=22, help="--mode lde-commit: rows of the trace (configs[4]: 2^22)")
    ap.add_argument("--total-cols", type=int, default=32, help="--mode lde-commit: columns of the trace, sharded over the ranks")
    args = ap.parse_args()

    if args.cold_child:
        _cold_child_main()
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(_spawn_ranks(args.gpus))
    _claim_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as {args.gpus} GPUs", file=sys.stderr)
        sys.exit(2)
    # The launcher's own CPU test (tests/test_distributed.py) points these at the g++ simulator build of the library and at gloo;
    # unset -- always, outside that test -- the product library on the rank's GPU and RCCL (torch's "nccl") are used.
    lib_path, backend = os.environ.get("MS_BENCH_LIB"), os.environ.get("MS_BENCH_DIST_BACKEND", "nccl")
    on_gpu = backend == "nccl"
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_mod
        if on_gpu:
            torch.cuda.set_device(local_rank)
            dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist_mod.init_process_group(backend)
        dist = dist_mod
        if dist.get_world_size() != args.gpus:
            print(f"bench.py: {dist.get_world_size()} ranks joined, --gpus {args.gpus}", file=sys.stderr)
            sys.exit(2)
        world = dist.get_world_size()

    from ministark_amd import GOLDILOCKS_FP, GpuFft, GpuVec, Planner, Radix2EvaluationDomain
    from ministark_amd.distributed import RcclComm

    log_n = args.log_n
    n = 1 << log_n
    if lib_path:
        from ministark_amd import _lib
        pl = Planner(local_rank if on_gpu else 0, _lib.Lib(lib_path))     # the simulator has one device
    else:
        pl = Planner(local_rank)
    if args.pmc_child:                                           # under rocprofv3 --pmc (see _measure_objects): no line, no timing
        _pmc_child_main(pl)
        pl.sync()
        return

    def reduce_max(x):
        if dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def dist_barrier():
        if dist is not None:
            import torch
            dist.barrier()
            if on_gpu:
                torch.cuda.synchronize()

    def run_sharded():
        with _stdout_to_stderr():
            comm = RcclComm.from_torch_distributed(pl) if dist is not None else RcclComm(pl, 0, 1, RcclComm.unique_id(pl.lib))
            try:
                r = sharded_lde_commit(pl, comm, max(2, min(args.steps, 5)), 1, log_rows=args.log_rows, total_cols=args.total_cols, barrier=dist_barrier)
                if args.total_cols == 8 or args.mode == "ntt":           # the whole prover on the same communicator (fib AIR: 8 columns)
                    try:
                        pr = sharded_prove(pl, comm, 5, log_rows=args.log_rows, barrier=dist_barrier)
                    except Exception as e:                                # noqa: BLE001 -- recorded, the commitment figures stand
                        pr = {"error": f"{type(e).__name__}: {e}", "n_gpus": world}
                else:
                    pr = None
            finally:
                comm.close()
        if pr is not None:
            # every rank takes part in the reduction or none does: a rank whose prover raised must not leave the others in a collective
            failed = reduce_max(0.0 if "prove_ms" in pr else 1.0)
            if failed:
                pr = pr if "error" in pr else {"error": "the prover failed on another rank", "n_gpus": world}
            else:
                pr["prove_ms"] = round(reduce_max(pr["prove_ms"]), 3)
            r["prove"] = pr
        for key in ("lde_ms", "exchange_ms", "commit_ms"):
            r[key] = round(reduce_max(r[key]), 3)
        r["total_ms"] = round(r["lde_ms"] + r["exchange_ms"] + r["commit_ms"], 3)
        r["lde_GBps"] = round(r["lde_algorithmic_bytes"] / (r["lde_ms"] * 1e-3) / 1e9, 1)
        r["lde_hbm_frac_of_all_gpus"] = round(r["lde_GBps"] / (HBM_PEAK_GBS * world), 4)
        return r

    if args.mode == "lde-commit":
        r = run_sharded()
        if rank == 0:
            _emit({"metric": "column-sharded LDE + commitment (configs[4]), algorithmic GB/s of the LDE phase", "value": r["lde_GBps"],
                              "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["total_ms"],
                              "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                              "config": {"workload": r["workload"], "parallelism": f"columns x{world}, rows x{world} after the exchange"},
                              "mode": "lde-commit", "sharded_lde_commit": r})
        if dist is not None:
            dist.destroy_process_group()
        return
    rng = np.random.default_rng(0x6D696E69 + rank)
    P = (1 << 64) - (1 << 32) + 1
    host_cols = [rng.integers(0, P, size=n, dtype=np.uint64) for _ in range(args.cols)]
    cols = [GpuVec.from_numpy(pl, c) for c in host_cols]
    dom = Radix2EvaluationDomain.new_coset(n, 7)
    fft = GpuFft(dom, GOLDILOCKS_FP, pl)

    def barrier():
        pl.sync()
        dist_barrier()

    # settle: plans built, scratch allocated, clocks up -- before the W untimed warm-up steps the contract asks for
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < args.settle:
        fft.enqueue(cols)
        pl.sync()
    for _ in range(args.warmup):
        fft.enqueue(cols)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fft.enqueue(cols)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = reduce_max(elapsed)

    # per-kernel durations, measured live with hipEvents on the library's stream
    pl.profile(True)
    for _ in range(max(2, min(args.steps, 5))):
        fft.enqueue(cols)
    prof = pl.profile_read()
    pl.profile(False)

    # The column-sharded LDE + commitment runs on every rank (RCCL inside the library).  It is guarded: the headline
    # measurement above is complete at this point, so if the exchange raises -- or hangs: N > 1 has never run on
    # physical GPUs from this repository -- rank 0 still prints its line (with the error recorded) instead of the whole
    # run being lost to the driver's timeout.
    sharded = None
    state = {"line": None}

    def run_sharded_guarded():
        import threading

        def bail():
            if rank == 0 and state["line"] is not None:
                state["line"]["sharded_lde_commit"] = {"error": "timed out after 180 s (collective did not complete)", "n_gpus": world}
                _emit(state["line"])
            os._exit(0 if state["line"] is not None or rank != 0 else 1)
        timer = threading.Timer(float(os.environ.get("MS_BENCH_BAIL_S", "180")), bail)
        timer.daemon = True
        timer.start()
        # a launcher that loses a rank terminates the others (torch.distributed.run and _spawn_ranks both send SIGTERM): rank 0 then
        # still prints the headline it has already measured, with the reason recorded, instead of dying inside a collective
        import signal

        def on_term(signum, frame):
            if rank == 0 and state["line"] is not None:
                state["line"]["sharded_lde_commit"] = {"error": "terminated by the launcher (another rank failed) during the sharded phase", "n_gpus": world}
                _emit(state["line"])
            os._exit(0 if rank == 0 and state["line"] is not None else 143)
        try:
            signal.signal(signal.SIGTERM, on_term)
        except ValueError:                                       # not the main thread
            pass
        if os.environ.get("MS_BENCH_TEST_FAIL_RANK") == str(rank):     # tests/test_bench_multirank.py: a rank that dies -- or stalls -- before the exchange
            if os.environ.get("MS_BENCH_TEST_FAIL_MODE") == "hang":
                time.sleep(3600)
            os._exit(3)
        try:
            return run_sharded()
        except Exception as e:                                   # noqa: BLE001 -- recorded on the line, the headline stands
            return {"error": f"{type(e).__name__}: {e}", "n_gpus": world}
        finally:
            timer.cancel()

    if rank != 0:
        if not args.no_extras:
            run_sharded_guarded()
        if dist is not None:
            dist.destroy_process_group()
        return

    alg_bytes_col = 2.0 * n * 8
    total_cols = args.cols * world
    ms_per_step = elapsed / args.steps * 1e3
    value = alg_bytes_col * total_cols * args.steps / elapsed / 1e9
    field_ops = 3 * (n // 2) * log_n + n            # butterflies (mul+add+sub) + coset scale
    kernels = []
    us_per_transform = 0.0
    for name, r in sorted(prof.items()):
        # a launch covers `group` columns; normalise to one column
        cols_per_call = r["bytes_per_call"] / alg_bytes_col
        us_col = r["avg_us"] / cols_per_call
        us_per_transform += us_col
        kernels.append({"name": name, "avg_us_per_column": round(us_col, 2), "calls": r["calls"],
                        "bytes_moved_per_column": alg_bytes_col,               # one read + one write of the column per pass
                        "GBps": round(alg_bytes_col / us_col / 1e3, 1) if us_col else None,     # (the simulator's events read 0)
                        "frac_of_hbm_peak": round(alg_bytes_col / us_col / 1e3 / HBM_PEAK_GBS, 3) if us_col else None})
    achieved = alg_bytes_col / us_per_transform / 1e3 if us_per_transform else 0.0
    # HBM bytes per transform: measured now by two short PMC child runs (FETCH_SIZE, WRITE_SIZE; default single-GPU run only), else
    # from the summary of the same passes under profiles/ (scripts/collect_profiles.sh), named in `traffic_source`
    traffic, traffic_source, traffic_kernels = None, None, None      # filled in at the END of the run (counter sessions disturb what is timed after them)
    out = {
        "metric": "2^24-point Goldilocks NTT algorithmic bandwidth", "value": round(value, 2), "unit": "GB/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"configs[1]: forward coset NTT (offset 7), {args.cols} columns x 2^{log_n} Goldilocks Fp per GPU, in place, HBM-resident",
                   "columns_per_gpu": args.cols, "log_n": log_n, "parallelism": f"columns x{world}"},
        "field_ops_per_s": round(field_ops * total_cols * args.steps / elapsed, 1),
        "us_per_transform": round(elapsed / args.steps / args.cols * 1e6, 2),
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source, "traffic_kernels": traffic_kernels,
                     "kernel": "ntt2_first_pass + 2 x ntt2_mid_pass: the three radix-256 passes of one forward coset transform",
                     "algorithmic_bytes_per_transform": alg_bytes_col,
                     "us_per_transform_events": round(us_per_transform, 2), "kernels": kernels},
    }
    out["multi_gpu_note"] = ("2 / 4 / 8 ranks unmeasured on hardware from this repository (one GPU per lease); rehearsed on the simulator + fake RCCL, tests/test_bench_multirank.py"
                             if world == 1 else f"{world} ranks, one process per GPU, RCCL")
    if world == 1:
        # for information: the other transforms of configs[1] on the same columns (wall time per transform, 5 steps each)
        from ministark_amd import GpuIfft
        variants = {}
        for name, plan in (("forward_subgroup", GpuFft(Radix2EvaluationDomain(n), GOLDILOCKS_FP, pl)),
                           ("inverse_coset", GpuIfft(dom, GOLDILOCKS_FP, pl)),
                           ("inverse_subgroup", GpuIfft(Radix2EvaluationDomain(n), GOLDILOCKS_FP, pl))):
            t_s = time.perf_counter()
            while time.perf_counter() - t_s < 0.3:               # plan built, clocks back up after the idle gap
                plan.enqueue(cols)
                pl.sync()
            t1 = time.perf_counter()
            for _ in range(5):
                plan.enqueue(cols)
            pl.sync()
            variants[name + "_us_per_transform"] = round((time.perf_counter() - t1) / 5 / args.cols * 1e6, 2)
            plan.close()
        # the shape of the reference's own criterion harness (gpu/benches/fft.rs:36-43): ONE column transformed again and again
        # in place -- column + scratch = 256 MiB, the size of the Infinity Cache; next to the batch, never instead of it
        if not args.no_extras:                                   # (kept out of the --no-extras runs the rocprofv3 kernel statistics come from:
            plan = GpuFft(dom, GOLDILOCKS_FP, pl)                # one-column launches would mix into the per-launch averages)
            t_s = time.perf_counter()
            while time.perf_counter() - t_s < 0.3:
                plan.enqueue(cols[:1])
                pl.sync()
            t1 = time.perf_counter()
            for _ in range(40):
                plan.enqueue(cols[:1])
            pl.sync()
            variants["single_column_repeated_us_per_transform"] = round((time.perf_counter() - t1) / 40 * 1e6, 2)
            plan.close()
        out["variants"] = variants
    if not args.no_extras:
        state["line"] = dict(out)                                # what rank 0 prints if the exchange never returns
        sharded = run_sharded_guarded()                          # every rank takes part
    if sharded is not None:
        out["sharded_lde_commit"] = sharded
    if world == 1 and not args.no_extras:
        for c in cols:
            c.free()
        out["c2_sweep"] = bench_c2_sweep(pl)
        out["reference_criterion_harness"] = bench_reference_harness(pl)
        out["lde_commit"] = bench_lde_commit(pl, not args.no_cpu_baseline)
        out["lde_2_24"] = bench_lde_2_24(pl)
        out["constraint_eval"] = bench_constraint_eval(pl, not args.no_cpu_baseline)
        out["prove"] = bench_prove(pl, not args.no_cpu_baseline)
        out["cold_start"] = bench_cold_start()
    if not args.no_cpu_baseline and world == 1:          # the CPU baseline is timed on rank 0 at N = 1 only
        from oracle import cref
        x = host_cols[0].copy()
        cref.ntt(x[: 1 << 16], 16, 1, False, 7)          # warm the OpenMP pool
        reps, best = 0, 1e30
        t_start = time.perf_counter()
        while reps < 3 and time.perf_counter() - t_start < 20:
            a = x.copy()
            t1 = time.perf_counter()
            cref.lib().oracle_ntt(cref._p(a), log_n, 1, 0, 7)
            best = min(best, time.perf_counter() - t1)
            reps += 1
        out["cpu_baseline"] = {"value": round(alg_bytes_col / best / 1e9, 3), "unit": "GB/s", "cores": cref.num_threads(),
                               "kind": "port", "ms_per_transform": round(best * 1e3, 2),
                               "sample": f"{reps} x one 2^{log_n} column, forward coset NTT, oracle/c (C/OpenMP restatement, not the reference binary)"}
    if world == 1 and not args.no_extras:                        # counters last: nothing is timed after a profiler session
        pl.sync()
        measured = _measure_traffic(log_n)
        if measured is not None:
            traffic, traffic_kernels, wave_insts = measured
            traffic_source = "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU (three child runs of 2 columns x 2 steps), per launch and column"
            if wave_insts and us_per_transform:
                out["roofline"]["valu_insts_per_element"] = round(wave_insts * 64.0 / n, 1)
                out["roofline"]["frac_of_valu_issue_peak"] = round(wave_insts / (us_per_transform * 1e-6) / VALU_ISSUE_PEAK, 4)
        _attach_all({k: out[k] for k in ("lde_commit", "lde_2_24", "constraint_eval", "prove") if isinstance(out.get(k), dict)}, _measure_objects())
    if traffic is None and args.traffic_json and os.path.exists(args.traffic_json):
        traffic = json.load(open(args.traffic_json)).get("hbm_bytes_per_transform")
        traffic_source = os.path.relpath(args.traffic_json, ROOT)
    out["roofline"].update({"traffic": traffic, "traffic_source": traffic_source, "traffic_kernels": traffic_kernels})
    _emit(out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
