#!/usr/bin/env python3
"""What the reference's main crate asks of crate ministark-gpu on the hot path -> tests/golden/reference_gpu_api_uses.json.

Run in the build container (needs /root/reference); the fixture travels with the repo and tests/test_rust_shim.py checks the
HIP arm (rust/gpu/src/hip/) against it: every constructor / method / constant the callers use on `GpuFft`, `GpuIfft` and the
prelude items must exist there with the same receiver and argument shapes."""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_gpu_api_uses.json")
CALLERS = ["src/matrix.rs", "src/fri.rs", "src/prover.rs", "src/composer.rs", "src/merkle.rs"]


def main():
    uses = {"GpuFft": set(), "GpuIfft": set(), "prelude_items": set(), "sites": []}
    for rel in CALLERS:
        text = open(os.path.join(REF, rel)).read()
        for m in re.finditer(r"\b(GpuFft|GpuIfft)(?:::<[^>]*>)?::(\w+)", text):
            uses[m.group(1)].add(m.group(2))
            uses["sites"].append(f"{rel}:{text[:m.start()].count(chr(10)) + 1} {m.group(0)}")
        # methods called on the values bound from GpuFft::from / GpuIfft::from
        for m in re.finditer(r"let mut (\w+) = (GpuFft|GpuIfft)::from\(", text):
            var, ty = m.group(1), m.group(2)
            for mm in re.finditer(r"\b%s\.(\w+)\(([^)]*)\)" % re.escape(var), text[m.end():m.end() + 600]):
                uses[ty].add(mm.group(1))
                uses["sites"].append(f"{rel}:{text[:m.end() + mm.start()].count(chr(10)) + 1} {var}.{mm.group(1)}({mm.group(2).strip()})")
        for item in ("get_planner", "AddAssignStage", "FillBuffStage", "MulPowStage", "buffer_no_copy", "buffer_mut_no_copy"):
            if re.search(r"\b%s\b" % item, text):
                uses["prelude_items"].add(item)
    # signatures of the Metal arm the callers rely on
    plan = open(os.path.join(REF, "gpu/src/plan.rs")).read()
    sigs = {}
    for ty in ("GpuFft", "GpuIfft"):
        blk = plan[plan.index(f"impl<'a, F: GpuField + ark_ff::Field> {ty}<'a, F>"):]
        sigs[ty] = {name: re.search(r"pub fn %s\(([^)]*)\)" % name, blk).group(1).strip() for name in ("encode", "execute")}
        sigs[ty]["MIN_SIZE"] = re.search(r"pub const MIN_SIZE: usize = (\d+);", blk).group(1)
    out = {"GpuFft": sorted(uses["GpuFft"]), "GpuIfft": sorted(uses["GpuIfft"]), "prelude_items": sorted(uses["prelude_items"]),
           "metal_signatures": sigs, "sites": sorted(set(uses["sites"]))}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
