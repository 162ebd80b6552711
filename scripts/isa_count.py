#!/usr/bin/env python3
"""Per-kernel instruction statistics of a gfx950 assembly listing (hipcc -S --cuda-device-only).

    python scripts/isa_count.py file.s [substring-of-kernel-name ...]

Prints, per kernel, the number of VALU / SALU / VMEM / LDS / other instructions, the VALU split into the fast class
(2.4 cycles per wave64: v_add/sub/and/or/xor/not/mov/lshr, profiles/r01_ubench2_instr_rates.txt) and the rest (4.2),
and the weighted issue cycles; plus vgpr / sgpr / spill / scratch from the metadata.
"""
import re
import sys
from collections import Counter

FAST = re.compile(r"^v_(add_u32|sub_u32|subrev_u32|and_b32|or_b32|xor_b32|not_b32|mov_b32|lshrrev_b32|ashrrev_i32|add_nc_u32|sub_nc_u32)(_e32|_e64)?$")


def kernels(path):
    name, body = None, []
    for line in open(path):
        s = line.strip()
        m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", s)
        if m and not s.startswith(".L") and not m.group(1).startswith("BB"):
            if name and body:
                yield name, body
            name, body = m.group(1), []
            continue
        if name is None or not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            if s.startswith(".end_amdhsa_kernel") or s.startswith(".Lfunc_end"):
                if name and body:
                    yield name, body
                name, body = None, []
            continue
        body.append(s.split()[0])
    if name and body:
        yield name, body


def meta(path):
    """Per-kernel metadata of the code object notes.  The keys of a kernel's block are sorted alphabetically
    (.group_segment_fixed_size comes BEFORE .name), so a block is collected whole and then filed under its name."""
    out = {}
    cur = None
    inside = False

    def flush():
        if cur and ".name" in cur:
            out[cur[".name"]] = {k[1:]: v for k, v in cur.items() if k != ".name"}
    for line in open(path):
        if line.startswith("amdhsa.kernels:"):
            inside = True
            continue
        if not inside:
            continue
        if re.match(r"^[A-Za-z]", line):                      # next top-level key: the kernel list is over
            flush()
            cur, inside = None, False
            continue
        if line.startswith("  - "):                            # a new kernel block
            flush()
            cur = {}
            line = "    " + line[4:]
        if cur is None:
            continue
        m = re.match(r"^    (\.\w+):\s*(\S+)\s*$", line)
        if m:
            key, val = m.group(1), m.group(2)
            if key == ".name":
                cur[key] = val
            elif key in (".vgpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size", ".group_segment_fixed_size"):
                cur[key] = int(val)
    flush()
    return out


def main():
    path = sys.argv[1]
    filt = sys.argv[2:]
    md = meta(path)
    for name, body in kernels(path):
        if filt and not any(f in name for f in filt):
            continue
        c = Counter(body)
        valu = {k: v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_readlane") and not k.startswith("v_writelane")}
        lanes = sum(v for k, v in c.items() if k.startswith(("v_readlane", "v_writelane", "v_readfirstlane")))
        nfast = sum(v for k, v in valu.items() if FAST.match(k))
        nvalu = sum(valu.values())
        salu = sum(v for k, v in c.items() if k.startswith("s_") and not k.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_load", "s_buffer_load")))
        smem = sum(v for k, v in c.items() if k.startswith(("s_load", "s_buffer_load")))
        vmem = sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_")))
        lds = sum(v for k, v in c.items() if k.startswith("ds_"))
        nops = c.get("s_nop", 0)
        m = md.get(name, {})
        print(f"{name}\n   VALU {nvalu} (fast {nfast}, slow {nvalu - nfast}; ~{nfast * 2.4 + (nvalu - nfast) * 4.2:.0f} issue cycles)  lane-moves {lanes}  SALU {salu}  SMEM {smem}  VMEM {vmem}  LDS {lds}  s_nop {nops}  total {len(body)}")
        print(f"   vgpr {m.get('vgpr_count')} sgpr {m.get('sgpr_count')} vspill {m.get('vgpr_spill_count')} sspill {m.get('sgpr_spill_count')} scratch {m.get('private_segment_fixed_size')} lds {m.get('group_segment_fixed_size')}")
        top = sorted(valu.items(), key=lambda kv: -kv[1])[:14]
        print("   " + "  ".join(f"{k}:{v}" for k, v in top))


if __name__ == "__main__":
    main()
