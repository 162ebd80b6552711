"""The Rust shim of INTEGRATION.md exists as source under rust/gpu/src/hip/ (the build image has no Rust toolchain):
stand-in for compiling it -- every `extern "C"` declaration of sys.rs is checked against include/ministark_hip.h
(name, arity, parameter names, and types through an independent C -> Rust type table), and the wrapper modules
declare the reference's items."""
import os
import re
import sys

import pytest

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
        "size_t *": "*mut usize", "char *": "*mut c_char", "unsigned char *": "*mut u8", "int *": "*mut c_int", "ms_xchg_op *": "*mut ms_xchg_op", "const ms_p2p_op *": "*const ms_p2p_op", "ms_jit_stats *": "*mut ms_jit_stats", "const size_t *": "*const usize",
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
    for item in ("pub struct Planner", "pub fn get_planner() -> &'static Planner", "impl Default for Planner", "transform!(GpuFft, false", "transform!(GpuIfft, true",
                 "pub fn encode(&mut self, buffer: &'a mut [F])", "pub fn execute(mut self)",
                 "pub struct GpuRpo256ColumnMajor", "pub struct GpuRpo256RowMajor", "pub fn gen_rpo_merkle_tree"):
        assert item in plan, item
    utils = open(os.path.join(HIP, "utils.rs")).read()
    # GpuField and the slice functions bit_reverse / bit_reverse_index are the crate's own platform-independent items
    # (gpu/src/lib.rs:20-41, gpu/src/utils.rs:4-78): the arm must use them, not redefine them
    assert "use crate::GpuField;" in utils and "pub trait GpuField" not in utils
    for item in ("pub struct DeviceVec", "pub fn field_id<F: GpuField>() -> c_int", "pub fn bit_reverse_device"):
        assert item in utils, item
    # every sys:: function the wrappers call is declared
    r = _rust_externs()
    for f in (stage, plan, utils):
        for fn in re.findall(r"sys::(ms_[a-z0-9_]+)\(", f):
            assert fn in r, fn


# ---- the callers' view (round 3): what src/matrix.rs / src/fri.rs ask of GpuFft / GpuIfft must exist in the HIP arm with the same shapes
def _shim_transform_macro():
    """The `transform!` macro body of rust/gpu/src/hip/plan.rs (both GpuFft and GpuIfft are instances of it)."""
    text = open(os.path.join(ROOT, "rust", "gpu", "src", "hip", "plan.rs")).read()
    body = text[text.index("macro_rules! transform"):text.index("transform!(GpuFft")]
    insts = re.findall(r"transform!\((\w+), (true|false)", text)
    return body, dict(insts)


def test_shim_serves_every_call_the_reference_makes():
    import json
    uses = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_gpu_api_uses.json")))
    body, insts = _shim_transform_macro()
    assert insts == {"GpuFft": "false", "GpuIfft": "true"}                     # forward / inverse
    for ty in ("GpuFft", "GpuIfft"):
        for item in uses[ty]:
            if item == "from":
                assert "impl<'a, F: GpuField> From<Radix2EvaluationDomain<F::FftField>> for $name<'a, F>" in body
            elif item == "MIN_SIZE":
                m = re.search(r"pub const MIN_SIZE: usize = (\d+);", body)
                assert m and m.group(1) == uses["metal_signatures"][ty]["MIN_SIZE"]   # the callers' CPU fall-back threshold
            else:
                m = re.search(r"pub fn %s\(([^)]*)\)" % item, body)
                assert m, f"{ty}::{item} missing in the HIP arm"
                want = uses["metal_signatures"][ty][item]
                norm = lambda sig: [re.sub(r"'\w+\s*", "", p.split(":", 1)[-1].strip()) for p in sig.replace("mut self", "self").split(",")]
                assert norm(m.group(1)) == norm(want), (ty, item, m.group(1), want)    # same receiver, same `&mut [F]` host slice
    # prelude items: planner and the stage the callers name; buffer_no_copy / buffer_mut_no_copy are Metal's unified-memory
    # wrappers, replaced by DeviceVec in the one arm that uses them (rust/patches/src_matrix.rs.patch: sum_columns_gpu)
    mod = open(os.path.join(ROOT, "rust", "gpu", "src", "hip", "mod.rs")).read()
    stage = open(os.path.join(ROOT, "rust", "gpu", "src", "hip", "stage.rs")).read()
    prelude_patch = open(os.path.join(ROOT, "rust", "patches", "gpu_src_prelude.rs.patch")).read()
    for item in uses["prelude_items"]:
        if item.startswith("buffer_"):
            assert "sum_columns_gpu" in open(os.path.join(ROOT, "rust", "patches", "src_matrix.rs.patch")).read()
            continue
        assert re.search(r"\b%s\b" % item, mod + stage), item
        assert re.search(r"\+pub use crate::hip::\w+::%s;" % item, prelude_patch), item


def test_shim_only_calls_declared_entry_points_with_the_right_arity():
    externs = _rust_externs()
    for name in ("plan.rs", "stage.rs", "utils.rs", "../../../src/eval_hip.rs", "../../../src/composer_hip.rs"):
        text = open(os.path.join(ROOT, "rust", "gpu", "src", "hip", name)).read()
        for m in re.finditer(r"sys::(ms_\w+)\(", text):
            fn = m.group(1)
            assert fn in externs, f"{name} calls {fn}, which sys.rs does not declare"
            # count top-level commas of the call
            depth, i, commas, nonempty = 1, m.end(), 0, False
            while depth:
                ch = text[i]
                if ch in "([{<" and not (ch == "<" and text[i - 1] == " "):
                    depth += ch in "([{"
                elif ch in ")]}":
                    depth -= 1
                elif ch == "," and depth == 1:
                    commas += 1
                if depth and not ch.isspace():
                    nonempty = True
                i += 1
            if text[:i - 1].rstrip().endswith(","):            # rustfmt's trailing comma
                commas -= 1
            nargs = commas + 1 if nonempty else 0
            assert nargs == len(externs[fn][0]), f"{name}: {fn} called with {nargs} arguments, declared with {len(externs[fn][0])}"


def test_evaluator_arm_uses_the_opcodes_of_the_header_and_the_signature_of_eval_cpu():
    """rust/src/eval_hip.rs (-> src/eval_hip.rs of the reference): its opcode constants are the header's, its `eval` has the
    parameter list of eval_cpu::eval (src/eval_cpu.rs:33-42; fixture: the same list, so that src/air.rs can switch with a cfg)."""
    text = open(os.path.join(ROOT, "rust", "src", "eval_hip.rs")).read()
    header = open(os.path.join(ROOT, "include", "ministark_hip.h")).read()
    doc = dict((name, int(num)) for num, name in re.findall(r"\b(\d+) ([A-Z]+_[PQ]{1,2}|EMBED)\b", header))
    for name, value in re.findall(r"const ([A-Z_]+): u32 = (\d+);", text):
        assert doc[name] == int(value), (name, value, doc.get(name))
    sig = re.search(r"pub fn eval<Fp: GpuFftField<FftField = Fp> \+ FftField, Fq: StarkExtensionOf<Fp>>\((.*?)\) -> Matrix<Fq>", text, re.S).group(1)
    params = [p.strip() for p in sig.strip().rstrip(",").split(",\n")]
    assert params == ["expr: &Expr<AlgebraicItem<FieldVariant<Fp, Fq>>>", "challenges: &[Fq]", "hints: &[Fq]", "lde_step: usize",
                      "domain_offset: Fp", "x_lde: &[Fp]", "base_trace_lde_cols: &[&[Fp]]", "extension_trace_lde_cols: Option<&[&[Fq]]>"]
    if os.path.isdir("/root/reference"):                                 # ... which is eval_cpu::eval's, word for word
        ref = open("/root/reference/src/eval_cpu.rs").read()
        rsig = re.search(r"pub fn eval<Fp: GpuFftField<FftField = Fp> \+ FftField, Fq: StarkExtensionOf<Fp>>\((.*?)\) -> Matrix<Fq>", ref, re.S).group(1)
        assert [p.strip() for p in rsig.strip().rstrip(",").split(",\n")] == params
    assert {"src_air.rs.patch", "src_lib.rs.patch", "src_composer.rs.patch"} <= set(os.listdir(os.path.join(ROOT, "rust", "patches")))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree is only present in the build container")
def test_patches_apply_to_the_reference(tmp_path):
    import shutil
    import subprocess
    pdir = os.path.join(ROOT, "rust", "patches")
    patches = sorted(f for f in os.listdir(pdir) if f.endswith(".patch"))
    assert {"src_matrix.rs.patch", "src_merkle.rs.patch", "src_fri.rs.patch", "gpu_src_lib.rs.patch", "gpu_src_prelude.rs.patch"} <= set(patches)
    for f in patches:
        text = open(os.path.join(pdir, f)).read()
        rel = re.match(r"--- a/(\S+)", text).group(1)
        dst = tmp_path / rel
        dst.parent.mkdir(parents=True, exist_ok=True)
        shutil.copy(os.path.join("/root/reference", rel), dst)
        r = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(pdir, f)], cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode == 0, f"{f}: {r.stdout}{r.stderr}"
