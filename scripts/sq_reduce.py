#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc counter_collection CSVs (one row per dispatch and counter) to per-kernel averages."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "?")
        k = k.split("(")[0][-70:]
        c = row.get("Counter_Name")
        acc[k][c] += float(row.get("Counter_Value", 0))
        cnt[k][c] += 1
for k in sorted(acc):
    n = max(cnt[k].values())
    if n < 1:
        continue
    a = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
    line = f"{k}  dispatches={n}\n   " + "  ".join(f"{c}={a[c]:.4g}" for c in sorted(a))
    wc, busy = a.get("SQ_WAVE_CYCLES"), a.get("SQ_BUSY_CYCLES")
    if wc and a.get("SQ_INSTS_VALU"):
        line += (f"\n   per wave: cycles(quad)={wc / a['SQ_WAVES']:.0f} valu={a['SQ_INSTS_VALU'] / a['SQ_WAVES']:.0f}"
                 f" active_valu={a.get('SQ_ACTIVE_INST_VALU', 0) / wc:.3f} wait_any={a.get('SQ_WAIT_ANY', 0) / wc:.3f} wait_inst={a.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}")
    print(line)
