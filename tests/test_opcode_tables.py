"""The constraint program's opcodes have ONE definition (`enum ms_eval_op`, include/ministark_hip.h): the tables of the three lowerings
(ministark_amd/expr.py, csrc/host/expr.hpp, rust/src/eval_hip.rs) are generated from it and must be current, and the kernels' own enum
(csrc/eval_kernels.h: the public opcodes followed by the library's internal ones) must start with the same names in the same order."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import gen_opcodes  # noqa: E402


def test_generated_tables_are_current():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_opcodes.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_kernel_enum_starts_with_the_public_opcodes():
    ops = gen_opcodes.opcodes()
    text = open(os.path.join(ROOT, "ministark_amd", "csrc", "eval_kernels.h")).read()
    body = re.search(r"enum Op : uint32_t \{(.*?)\};", text, re.S).group(1)
    body = re.sub(r"//[^\n]*", "", body)
    names = [n.split("=")[0].strip() for n in body.split(",") if n.strip()]
    assert names[0] == "OP_X_P" and "= 0" in body.split(",")[0]
    assert names[: len(ops)] == ["OP_" + n for n, _ in ops]
    assert names[-1] == "OP_COUNT" and all(n not in [f"OP_{o}" for o, _ in ops] for n in names[len(ops):-1])


def test_python_lowering_uses_the_header_values():
    from ministark_amd import expr as E
    for name, value in gen_opcodes.opcodes():
        assert getattr(E, "OP_" + name) == value
