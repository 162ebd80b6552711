#!/usr/bin/env python3
"""HBM bytes per kernel of any command, the way bench.py counts them: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate, no tracing
domains), FETCH_SIZE doubled (gfx950), units of 1 KiB; summed over the launches of each kernel.   python scripts/pmc_kernels.py -- <command ...>"""
import csv
import glob
import os
import subprocess
import sys
import tempfile

cmd = sys.argv[sys.argv.index("--") + 1:]
tot = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    out = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", out, "-o", "t", "--output-format", "csv", "--"] + cmd,
                   cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    for row in csv.DictReader(open(glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)[0])):
        if row["Counter_Name"] == counter:
            k = tot.setdefault(row["Kernel_Name"].split("(")[0].replace("void ", "")[-70:], {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "launches": 0})
            k[counter] += float(row["Counter_Value"])
            k["launches"] += counter == "FETCH_SIZE"
for name, k in sorted(tot.items()):
    print(f"{name:72s} launches {k['launches']:4d}  fetch {2 * 1024 * k['FETCH_SIZE'] / 1e6:10.1f} MB  write {1024 * k['WRITE_SIZE'] / 1e6:10.1f} MB")
