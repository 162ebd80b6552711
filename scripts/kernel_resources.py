#!/usr/bin/env python3
"""Registers, scratch (spills), LDS and instruction counts of EVERY kernel of the product library, from the gfx950 assembly.

    python scripts/kernel_resources.py [> profiles/rNN_kernel_resources.txt]

Compiles each translation unit of ministark_amd/build.py::SOURCES with `hipcc -S --cuda-device-only` (no GPU needed) and prints
one line per kernel, sorted by scratch bytes then VGPRs.  A kernel with scratch > 0 spills.
"""
import os
import subprocess
import sys
import tempfile
from collections import Counter
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import isa_count  # noqa: E402
from ministark_amd import build as msbuild  # noqa: E402


def listing(src, tmp):
    out = os.path.join(tmp, os.path.basename(src) + ".s")
    flags = [f for f in msbuild.FLAGS if f not in ("-shared", "-fPIC")]          # the product's own flags
    cmd = [msbuild.HIPCC] + flags + ["-S", "--cuda-device-only", src, "-o", out]
    subprocess.run(cmd, check=True)
    return out


def main():
    msbuild.embed_headers()
    srcs = [os.path.join(msbuild.CSRC, s) for s in msbuild.SOURCES]
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        with ThreadPoolExecutor(len(srcs)) as ex:
            outs = list(ex.map(lambda s: listing(s, tmp), srcs))
        for src, path in zip(srcs, outs):
            md = isa_count.meta(path)
            for name, body in isa_count.kernels(path):
                if name not in md:
                    continue
                c = Counter(body)
                valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith(("v_readlane", "v_writelane", "v_readfirstlane")))
                vmem = sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_")))
                scr = sum(v for k, v in c.items() if k.startswith("scratch_"))
                lds = sum(v for k, v in c.items() if k.startswith("ds_"))
                m = md[name]
                rows.append((m.get("private_segment_fixed_size", 0), m.get("vgpr_count", 0), os.path.basename(src), name, m, valu, vmem, scr, lds))
    names = [r[3] for r in rows]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    rows = [r + (d,) for r, d in zip(rows, dem)]
    rows.sort(key=lambda r: (-r[0], -r[1], r[3]))
    print(f"# {len(rows)} kernels; columns: unit  vgpr  sgpr  vgpr-spills  sgpr-spills  scratch-bytes  lds-bytes | static VALU  global/buffer  scratch_*  ds_* instructions | kernel")
    for scratch, vg, unit, name, m, valu, vmem, scr, lds, d in rows:
        d = d.split("(")[0].replace("void ", "")
        print(f"{unit:13s} {vg:4d} {m.get('sgpr_count', 0):4d} {m.get('vgpr_spill_count', 0):5d} {m.get('sgpr_spill_count', 0):5d} {scratch:6d} {m.get('group_segment_fixed_size', 0):7d} | {valu:6d} {vmem:5d} {scr:5d} {lds:5d} | {d}")
    spilled = [r for r in rows if r[0] > 0]
    print(f"# kernels with scratch: {len(spilled)} of {len(rows)}; with scalar registers spilled into vector lanes: {sum(1 for r in rows if r[4].get('sgpr_spill_count', 0))}")


if __name__ == "__main__":
    main()
