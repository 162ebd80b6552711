#!/usr/bin/env python3
"""Per-size, per-kernel durations of the C2 sweep from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --stats -d OUT -o c2 --output-format csv -- python scripts/c2_sweep.py --all
    python scripts/c2_kernel_table.py OUT/c2_kernel_trace.csv > profiles/rNN_c2_sweep_kernels.txt
A launch covers all columns of a size (grid.y) and n / 16384 tiles per column (grid.x / 512); the last column is the median duration
normalised to 2^24 words, the unit of the headline's per-pass figures."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    wg = int(r["Workgroup_Size_X"])
    agg[(int(r["Grid_Size_X"]) // wg, int(r["Grid_Size_Y"]), r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(f"{'points':>7} {'columns':>7} {'launches':>8} {'median us':>10} {'us / 2^24 words':>16}  kernel")
for (tiles, cols, name), d in sorted(agg.items(), key=lambda kv: (kv[0][0], kv[0][2])):
    d.sort()
    med = d[len(d) // 2] / 1e3
    log_n = (tiles * 16384).bit_length() - 1
    name = name.replace("void ", "").replace("(msntt2::Params)", "")
    print(f"   2^{log_n:<3d} {cols:7d} {len(d):8d} {med:10.1f} {med / cols / (tiles * 16384 / 2 ** 24):16.1f}  {name}")
