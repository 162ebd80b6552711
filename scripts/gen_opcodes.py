#!/usr/bin/env python3
"""The constraint program's opcodes live in ONE place -- `enum ms_eval_op` of include/ministark_hip.h -- and the three lowerings of the
expression DAG (Python, the C++ host mirror, the Rust arm) carry generated copies between `>>> opcodes` / `<<< opcodes` markers.

    python scripts/gen_opcodes.py            # rewrite the marked blocks
    python scripts/gen_opcodes.py --check    # exit 1 if a block is stale (tests/test_opcode_tables.py)
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ministark_hip.h")
RUST_USED = ("X_P", "CONST_P", "CONST_Q", "TRACE_P", "TRACE_Q", "PERIODIC_P", "NEG_P", "ADD_PP", "MUL_PP", "INV_P", "POW_P", "EMBED",
             "STORE_Q", "STORE_P")      # the Q-typed twins are base + 1 in eval_hip.rs (NEG_P + 1 ...): only the names it spells out


def opcodes():
    """[(name without the MS_OP_ prefix, value)] in value order, MS_OP_PUBLIC_COUNT checked and left out."""
    text = open(HEADER).read()
    m = re.search(r"enum ms_eval_op \{(.*?)\};", text, re.S)
    if not m:
        raise SystemExit("enum ms_eval_op not found in include/ministark_hip.h")
    ops = [(n, int(v)) for n, v in re.findall(r"MS_OP_(\w+)\s*=\s*(\d+)", m.group(1))]
    count = dict(ops).pop("PUBLIC_COUNT")
    ops = sorted((o for o in ops if o[0] != "PUBLIC_COUNT"), key=lambda o: o[1])
    if [v for _, v in ops] != list(range(count)):
        raise SystemExit("enum ms_eval_op is not 0 .. MS_OP_PUBLIC_COUNT - 1 without gaps")
    return ops


def wrap(items, indent, width=118):
    lines, cur = [], indent
    for it in items:
        if len(cur) + len(it) + 1 > width and cur.strip():
            lines.append(cur.rstrip())
            cur = indent
        cur += it + " "
    lines.append(cur.rstrip())
    return "\n".join(lines)


def blocks(ops):
    py = "(" + wrap([f"OP_{n}," for n, _ in ops], " ").lstrip()[:-1] + f") = range({len(ops)})\n"
    cpp = "enum Op : uint32_t {\n" + wrap([f"OP_{n} = {v}," for n, v in ops], "    ")[:-1] + "\n};\n"
    val = dict(ops)
    rs = "// opcodes of the constraint program (include/ministark_hip.h)\n" + "".join(f"const {n}: u32 = {val[n]};\n" for n in RUST_USED)
    return {os.path.join("ministark_amd", "expr.py"): ("#", py),
            os.path.join("ministark_amd", "csrc", "host", "expr.hpp"): ("//", cpp),
            os.path.join("rust", "src", "eval_hip.rs"): ("//", rs)}


def main():
    check = "--check" in sys.argv[1:]
    stale = []
    for rel, (cm, body) in blocks(opcodes()).items():
        path = os.path.join(ROOT, rel)
        text = open(path).read()
        pat = re.compile(r"(%s >>> opcodes[^\n]*\n)(.*?)(%s <<< opcodes)" % (re.escape(cm), re.escape(cm)), re.S)
        if not pat.search(text):
            raise SystemExit(f"{rel}: no '>>> opcodes' block")
        new = pat.sub(lambda m: m.group(1) + body + m.group(3), text, count=1)
        if new != text:
            stale.append(rel)
            if not check:
                open(path, "w").write(new)
    if check and stale:
        print("stale opcode tables (run scripts/gen_opcodes.py):", ", ".join(stale))
        return 1
    print("opcode tables " + ("up to date" if not stale else "rewritten: " + ", ".join(stale)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
