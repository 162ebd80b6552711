#!/usr/bin/env python3
"""gpurun_out/profiles_raw/ (scripts/collect_profiles.sh) -> rNN_* summaries.
    python scripts/summarise_profiles.py round_tag [--raw DIR] [--out DIR]
Run by collect_profiles.sh on the GPU box right after the measurements (--out gpurun_out/profiles_raw/summary; copy those files to
profiles/).  Every input must be NEWER than the run's RUN_STAMP file: a raw file left over from an earlier collection (gpurun merges
gpurun_out/, it does not replace it -- round 3's SQ counter summary was made from round 2's file that way) is refused, and a
missing counter file is an error, not a skipped section.
FETCH_SIZE is doubled per /opt/skills/guides/MI355X_MICROARCH.md (gfx950 tallies a coalesced stream at
64 B); counter units are KB = 1024 B; values are averaged per launch (one launch = one 2^24 column)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
RAW = os.path.join(ROOT, "gpurun_out", "profiles_raw")
PROF = os.path.join(ROOT, "profiles")
if "--raw" in args:
    i = args.index("--raw"); RAW = args[i + 1]; del args[i:i + 2]
if "--out" in args:
    i = args.index("--out"); PROF = args[i + 1]; del args[i:i + 2]
tag = args[0] if args else "r05"
os.makedirs(PROF, exist_ok=True)
STAMP = os.path.join(RAW, "RUN_STAMP")
if not os.path.exists(STAMP):
    sys.exit(f"summarise_profiles: {STAMP} missing -- not the output of scripts/collect_profiles.sh")
T0 = os.path.getmtime(STAMP)
problems = []


def fresh(path, what):
    """path if it exists and was written after the run's stamp; otherwise None and a recorded problem"""
    if path is None or not os.path.exists(path):
        problems.append(f"{what}: missing")
        return None
    if os.path.getmtime(path) < T0 - 1.0:
        problems.append(f"{what}: {os.path.relpath(path, RAW)} is older than RUN_STAMP (left over from an earlier collection) -- refused")
        return None
    return path


def one(pattern):
    hits = glob.glob(os.path.join(RAW, pattern), recursive=True)
    return fresh(hits[0] if hits else None, pattern)


def counters(path, per_column=False):
    """-> {kernel: {counter: [values per dispatch]}}; per_column divides by the columns a launch covers
    (grid = 4096 tiles x 256 threads per 2^24 column)."""
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        # work-items per 2^24 column: 4096 tiles x 256 threads (ntt_kernels.h), 1024 tiles x 512 threads (ntt2_kernels.h)
        per_col = (1024 * 512) if "msntt2" in r["Kernel_Name"] else (4096 * 256)
        cols = max(1.0, float(r["Grid_Size"]) / per_col) if per_column and "msntt" in r["Kernel_Name"] else 1.0
        out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]) / cols)
    return out


stats = one("stats/**/*kernel_stats.csv")
if stats:
    rows = [r for r in csv.DictReader(open(stats)) if "msntt" in r["Name"]]
    with open(os.path.join(PROF, f"{tag}_ntt_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_NONNUMERIC)
        w.writeheader()
        w.writerows(rows)
traffic = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python bench.py --steps 2 --warmup 1 --cols 2 --no-cpu-baseline",
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a coalesced stream's bytes); WRITE_SIZE as reported. "
                   "Units: KB = 1024 B. Normalised to one column (2^24 x 8 B); a launch covers several columns.", "kernels": {}}
fetch, write = one("pmc_fetch/**/*counter_collection.csv"), one("pmc_write/**/*counter_collection.csv")
if fetch and write:
    fc, wc = counters(fetch, True), counters(write, True)
    total = 0.0
    # the three launches of ONE forward coset transform (bench.py's other variants -- subgroup, inverse -- run in the
    # same process and must not be added in)
    # (round 4: the limb-form kernels' first template argument is the cache-policy switch STREAM)
    trio = tuple(f"{k}<{st}, {rest}" for st in ("true", "false")
                 for k, rest in (("ntt2_first_pass", "false, true, 16"), ("ntt2_mid_pass", "false, false, 0"), ("ntt2_mid_pass", "false, true, 0")))
    for k in fc:
        if "msntt" not in k or not any(t in k for t in trio):
            continue
        fb = 2 * 1024 * sum(fc[k]["FETCH_SIZE"]) / len(fc[k]["FETCH_SIZE"])      # per 2^24 column
        wb = 1024 * sum(wc[k]["WRITE_SIZE"]) / len(wc[k]["WRITE_SIZE"])
        traffic["kernels"][k] = {"fetch_bytes_per_column": fb, "write_bytes_per_column": wb}
        total += fb + wb
        shutil.copy(fetch, os.path.join(PROF, f"{tag}_ntt_pmc_fetch_size.csv"))
        shutil.copy(write, os.path.join(PROF, f"{tag}_ntt_pmc_write_size.csv"))
    traffic["hbm_bytes_per_transform"] = total
    traffic["algorithmic_bytes_per_transform"] = 2.0 * 8 * (1 << 24)
    json.dump(traffic, open(os.path.join(PROF, f"{tag}_ntt_traffic.json"), "w"), indent=1)
for name, dst in (("pmc_sq", "ntt_sq_counters"), ("pmc_sha", "sha256_sq_counters")):
    p = one(f"{name}/**/*counter_collection.csv")
    if not p:
        continue
    c = counters(p, True)
    with open(os.path.join(PROF, f"{tag}_{dst}.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches", "counter", "mean_per_dispatch (NTT passes: per 2^24 column)"])
        for k in sorted(c):
            if "msntt" in k or "mssha" in k:
                for cn, vals in sorted(c[k].items()):
                    w.writerow([k, len(vals), cn, sum(vals) / len(vals)])
for src, dst in (("c2_sweep.json", f"{tag}_c2_sweep.json"), ("c2_sweep_small.json", f"{tag}_c2_sweep_small.json"), ("lde_sq_counters.txt", f"{tag}_lde_sq_counters.txt"),
                 ("bench_lde_commit_n1.json", f"{tag}_bench_lde_commit_n1.json"),
                 ("bench.json", f"{tag}_bench_line.json"), ("bench_detail.json", f"{tag}_bench_detail.json"), ("bench_configs.jsonl", f"{tag}_bench_configs.jsonl"), ("bench_commit.json", f"{tag}_bench_commit.json")):
    p = os.path.join(RAW, src)
    if os.path.exists(p) and os.path.getsize(p) and fresh(p, src):
        shutil.copy(p, os.path.join(PROF, dst))
print(json.dumps(traffic, indent=1)[:1200])
if problems:
    print("summarise_profiles: INCOMPLETE\n  " + "\n  ".join(problems), file=sys.stderr)
    sys.exit(1)
