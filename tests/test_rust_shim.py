"""The Rust shim of INTEGRATION.md exists as source under rust/gpu/src/hip/ (the build image has no Rust toolchain):
stand-in for compiling it -- every `extern "C"` declaration of sys.rs is checked against include/ministark_hip.h
(name, arity, parameter names, and types through an independent C -> Rust type table), and the wrapper modules
declare the reference's items."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP = os.path.join(ROOT, "rust", "gpu", "src", "hip")


def _c_prototypes():
    src = open(os.path.join(ROOT, "include", "ministark_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    out = {}
    for m in re.finditer(r"\b(ms_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        params = " ".join(m.group(2).split())
        out[m.group(1)] = [] if params in ("", "void") else [p.strip() for p in params.split(",")]
    return out


def _rust_externs():
    src = open(os.path.join(HIP, "sys.rs")).read()
    block = src[src.index('extern "C" {'):]
    block = block[:block.index("\n}")]
    out = {}
    for m in re.finditer(r"pub fn (ms_[a-z0-9_]+)\((.*?)\)\s*(?:->\s*([^;]+))?;", block):
        args = [a.strip() for a in m.group(2).split(",")] if m.group(2).strip() else []
        out[m.group(1)] = ([tuple(x.strip() for x in a.split(":", 1)) for a in args], (m.group(3) or "").strip())
    return out


def _expect(ctype):
    """independent of scripts/gen_rust_sys.py: C parameter type (no name) -> Rust type, for the shapes the header uses"""
    t = " ".join(ctype.replace("*", " * ").split())
    table = {
        "int": "c_int", "unsigned": "c_uint", "long": "c_long", "size_t": "usize", "uint64_t": "u64",
        "ms_ctx *": "*mut ms_ctx", "ms_ctx * *": "*mut *mut ms_ctx", "ms_ntt_plan *": "*mut ms_ntt_plan", "ms_ntt_plan * *": "*mut *mut ms_ntt_plan",
        "void *": "*mut c_void", "const void *": "*const c_void", "void * *": "*mut *mut c_void",
        "void * const *": "*const *mut c_void", "const void * const *": "*const *const c_void",
        "const unsigned *": "*const c_uint", "const uint32_t *": "*const u32", "const uint64_t *": "*const u64", "uint64_t *": "*mut u64",
        "size_t *": "*mut usize", "char *": "*mut c_char", "int *": "*mut c_int", "ms_xchg_op *": "*mut ms_xchg_op",
    }
    return table[t]


def test_sys_rs_matches_the_header():
    c, r = _c_prototypes(), _rust_externs()
    assert sorted(c) == sorted(r), f"missing in sys.rs: {sorted(set(c) - set(r))}; extra: {sorted(set(r) - set(c))}"
    for name, params in c.items():
        rargs, _ = r[name]
        assert len(rargs) == len(params), name
        for cp, (rname, rtype) in zip(params, rargs):
            m = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)$", cp)
            ctype, cname = m.group(1).strip(), m.group(2)
            assert rname.rstrip("_") == cname, (name, cp, rname)
            assert rtype == _expect(ctype), (name, cp, rtype)


def test_sys_rs_is_what_the_generator_writes():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import gen_rust_sys
    assert open(os.path.join(HIP, "sys.rs")).read() == gen_rust_sys.render(gen_rust_sys.prototypes(open(gen_rust_sys.HEADER).read()))


def test_wrappers_declare_the_reference_items():
    stage = open(os.path.join(HIP, "stage.rs")).read()
    # the 17 element-wise stages of gpu/src/stage.rs:115-1155
    for s in ("MulIntoStage", "MulAssignStage", "MulPowStage", "AddAssignStage", "AddIntoStage", "AddIntoConstStage", "ConvertIntoStage",
              "AddAssignConstStage", "MulIntoConstStage", "MulAssignConstStage", "InverseInPlaceStage", "NegInPlaceStage", "NegIntoStage",
              "InverseIntoStage", "ExpIntoStage", "ExpInPlaceStage", "FillBuffStage"):
        assert re.search(rf"pub struct {s}<", stage), s
        assert re.search(rf"impl<[^>]*> {s}<[^>]*> \{{\s*pub fn new\(n: usize\)", stage), s
    plan = open(os.path.join(HIP, "plan.rs")).read()
    for item in ("pub struct Planner", "pub static PLANNER", "pub struct GpuFft", "pub struct GpuIfft", "pub fn encode", "pub fn execute(self)",
                 "pub struct GpuRpo256ColumnMajor", "pub struct GpuRpo256RowMajor", "pub fn gen_rpo_merkle_tree"):
        assert item in plan, item
    utils = open(os.path.join(HIP, "utils.rs")).read()
    for item in ("pub trait GpuField", "pub struct GpuVec", "pub fn bit_reverse"):
        assert item in utils, item
    # every sys:: function the wrappers call is declared
    r = _rust_externs()
    for f in (stage, plan, utils):
        for fn in re.findall(r"sys::(ms_[a-z0-9_]+)\(", f):
            assert fn in r, fn
