"""Where the GPU waits for the host inside one proof: gaps between consecutive kernels of a rocprofv3 --kernel-trace of
scripts/prove_once.py (the LAST proof is analysed).

    rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/gaps -o t -- python scripts/prove_once.py
    python scripts/prove_gaps.py gpurun_out/gaps

Prints the proof's span, the sum of kernel durations, the idle time, and the idle time grouped by the kernel that PRECEDES each gap (a gap after
sha256_merkle_top is a root download + the host's next launches; a gap between two FRI kernels is pure launch overhead)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    for p in ("mssha::", "msntt2::", "mslde2::", "msdeep::", "msfri::", "mseval::", "msscan::", "msstage::", "msntt::"):
        name = name.replace(p, "")
    return name.split("<")[0][:40]


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    # a proof ends with the one launch that copies the opened digests / FRI rows (copy_records32): the last proof = everything after the
    # second-to-last of them
    ends = [i for i, r in enumerate(rows) if "copy_records32" in r[2]]
    assert len(ends) >= 2, len(ends)
    start = ends[-2] + 1
    proof = rows[start:]
    span = proof[-1][1] - proof[0][0]
    busy = sum(e - s for s, e, _ in proof)
    gaps = defaultdict(lambda: [0, 0])
    big = []
    for (s0, e0, n0), (s1, e1, n1) in zip(proof, proof[1:]):
        g = max(0, s1 - e0)
        gaps[n0][0] += g
        gaps[n0][1] += 1
        big.append((g, n0, n1))
    print(f"last proof: {len(proof)} kernels, span {span / 1e6:.3f} ms, kernels {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms")
    for n, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0]):
        print(f"  after {n:40s} {c:4d} gaps  {g / 1e3:8.1f} us  (mean {g / 1e3 / c:6.1f})")
    print("largest gaps:")
    for g, a, b in sorted(big, reverse=True)[:25]:
        print(f"  {g / 1e3:7.1f} us  {a} -> {b}")


if __name__ == "__main__":
    main()
