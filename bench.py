#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

metric   : algorithmic GB/s (and field-ops/s) of the 2^24-point Goldilocks forward NTT
workload : configs[1] -- a batch of COLS columns of 2^24 canonical-uniform Goldilocks
           elements (Montgomery words), coset offset 7, transformed in place, resident in
           HBM before the timed region.  A "step" = one transform of every column.
value    : whole-job algorithmic bytes (2 * n * 8 per column: one compulsory read + one
           write) / wall time of the K timed steps, max over ranks.
N > 1    : one process per GPU (torchrun), columns are independent => each rank owns its
           own COLS columns (weak scaling), no data-path collective; the only exchange is
           the timing barrier / max-reduce.

Extra objects on the JSON line:
  roofline     : the transform against the HBM roofline.  achieved = algorithmic bytes of
                 one column transform / summed average duration of its kernel launches,
                 measured live with hipEvents on the library's stream; `kernels` lists
                 each launch (avg_us, its own bytes moved) for comparison with
                 profiles/*kernel_stats*.  `traffic` = HBM bytes per transform from the
                 rocprofv3 PMC passes when bench is run with --traffic-json, else null.
  cpu_baseline : oracle/c (C/OpenMP restatement of the reference CPU path, kind "port")
                 timed on this box's host cores on one 2^24 column, same data.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N = 24
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cols", type=int, default=8, help="columns of 2^24 per rank per step")
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--settle", type=float, default=1.0, help="seconds of untimed transforms before the warm-up steps (clock ramp)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "r01_ntt_traffic.json"), help="JSON file with PMC-derived HBM bytes per transform")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_mod
        torch.cuda.set_device(local_rank)
        dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist = dist_mod

    from ministark_amd import GOLDILOCKS_FP, GpuFft, GpuVec, Planner, Radix2EvaluationDomain

    log_n = args.log_n
    n = 1 << log_n
    pl = Planner(local_rank)
    rng = np.random.default_rng(0x6D696E69 + rank)
    P = (1 << 64) - (1 << 32) + 1
    host_cols = [rng.integers(0, P, size=n, dtype=np.uint64) for _ in range(args.cols)]
    cols = [GpuVec.from_numpy(pl, c) for c in host_cols]
    dom = Radix2EvaluationDomain.new_coset(n, 7)
    fft = GpuFft(dom, GOLDILOCKS_FP, pl)

    def barrier():
        pl.sync()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

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
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel durations, measured live with hipEvents on the library's stream
    pl.profile(True)
    for _ in range(max(2, min(args.steps, 5))):
        fft.enqueue(cols)
    prof = pl.profile_read()
    pl.profile(False)

    if rank != 0:
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
                        "bytes_moved_per_column": alg_bytes_col,
                        "GBps": round(alg_bytes_col / us_col / 1e3, 1)})
    achieved = alg_bytes_col / us_per_transform / 1e3 if us_per_transform else 0.0
    traffic = None
    if args.traffic_json and os.path.exists(args.traffic_json):
        traffic = json.load(open(args.traffic_json)).get("hbm_bytes_per_transform")
    out = {
        "metric": "2^24-point Goldilocks NTT algorithmic bandwidth", "value": round(value, 2), "unit": "GB/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"configs[1]: forward coset NTT (offset 7), {args.cols} columns x 2^{log_n} Goldilocks Fp per GPU, in place, HBM-resident",
                   "columns_per_gpu": args.cols, "log_n": log_n, "parallelism": f"columns x{world}"},
        "field_ops_per_s": round(field_ops * total_cols * args.steps / elapsed, 1),
        "us_per_transform": round(elapsed / args.steps / args.cols * 1e6, 2),
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "algorithmic_bytes_per_transform": alg_bytes_col,
                     "us_per_transform_events": round(us_per_transform, 2), "kernels": kernels},
    }
    if world == 1:
        # for information: the other transforms of configs[1] on the same columns (wall time per transform, 5 steps each)
        from ministark_amd import GpuIfft
        variants = {}
        for name, plan in (("forward_subgroup", GpuFft(Radix2EvaluationDomain(n), GOLDILOCKS_FP, pl)),
                           ("inverse_coset", GpuIfft(dom, GOLDILOCKS_FP, pl)),
                           ("inverse_subgroup", GpuIfft(Radix2EvaluationDomain(n), GOLDILOCKS_FP, pl))):
            for _ in range(2):
                plan.enqueue(cols)
            pl.sync()
            t1 = time.perf_counter()
            for _ in range(5):
                plan.enqueue(cols)
            pl.sync()
            variants[name + "_us_per_transform"] = round((time.perf_counter() - t1) / 5 / args.cols * 1e6, 2)
            plan.close()
        out["variants"] = variants
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
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
