//! `hip` arm of the constraint evaluator: the same signature as `eval_cpu::eval` (src/eval_cpu.rs:33-42), so that
//! `AirConfig::eval_constraint` (src/air.rs:86-128) switches to it with one `cfg` line (rust/patches/src_air.rs.patch).
//!
//! `eval_gpu::eval` (src/eval_gpu.rs:46-131) walks the expression graph with `Expr::graph_eval` and a value type whose
//! `+ * - / pow` encode one Metal stage per node.  This arm walks it the same way, but its value type only RECORDS the node: the
//! graph becomes the typed register program of include/ministark_hip.h ("constraint program": { op, dst, a, b } words, an Fp and an
//! Fq register file, constants in Montgomery form), registers are assigned by a linear scan, and ONE fused kernel evaluates the
//! whole composition constraint per point (`ms_eval_program_ex`).  Typing follows eval_cpu.rs:306-428 (a node is Fp iff both operands
//! are), `x / y = x * y^-1` with `0^-1 = 0` (eval_cpu.rs:280-294, 440-442), the result is returned as Fq (eval_cpu.rs:262-275).
//! The same lowering in C++ and Python, tested against the CPU restatement of eval_cpu: ministark_amd/csrc/host/expr.hpp,
//! ministark_amd/expr.py.
//!
//! Source only (no Rust toolchain in the build image).
#![cfg(feature = "hip")]

use crate::constraints::AlgebraicItem;
use crate::eval_cpu::build_periodic_column_evals_map;
use crate::expression::Expr;
use crate::utils::FieldVariant;
use crate::utils::GpuAllocator;
use crate::Matrix;
use crate::StarkExtensionOf;
use alloc::rc::Rc;
use alloc::vec::Vec;
use ark_ff::FftField;
use ark_ff::Zero;
use core::cell::RefCell;
use core::ffi::c_void;
use core::ops::Add;
use core::ops::Div;
use core::ops::Mul;
use core::ops::Neg;
use ministark_gpu::hip::field_id;
use ministark_gpu::hip::get_planner;
use ministark_gpu::hip::sys;
use ministark_gpu::hip::DeviceVec;
use ministark_gpu::GpuFftField;
use num_traits::Pow;

// opcodes of the constraint program (include/ministark_hip.h)
const X_P: u32 = 0;
const CONST_P: u32 = 1;
const CONST_Q: u32 = 2;
const TRACE_P: u32 = 3;
const TRACE_Q: u32 = 4;
const PERIODIC_P: u32 = 5;
const NEG_P: u32 = 7;
const ADD_PP: u32 = 9;
const MUL_PP: u32 = 12;
const INV_P: u32 = 15;
const POW_P: u32 = 17;
const EMBED: u32 = 19;
const STORE_Q: u32 = 20;
const STORE_P: u32 = 21;

/// One node of the program in SSA form: operands are node indices, `q` says which register file the result lives in.
struct Node {
    op: u32,
    a: Option<usize>,
    b: Option<usize>,
    q: bool,
    imm0: u32,
    imm1: i32,
}

#[derive(Default)]
struct Builder {
    nodes: Vec<Node>,
    consts: Vec<u64>, // Montgomery limbs, as the elements lie in memory
}

impl Builder {
    fn emit(&mut self, op: u32, q: bool, a: Option<usize>, b: Option<usize>, imm0: u32, imm1: i32) -> usize {
        self.nodes.push(Node { op, a, b, q, imm0, imm1 });
        self.nodes.len() - 1
    }

    /// The element's in-memory limbs (arkworks keeps Montgomery form; the device uses the same words) -> offset in `consts`.
    fn constant<T: Copy>(&mut self, v: &T) -> u32 {
        let words = core::mem::size_of::<T>() / 8;
        let off = self.consts.len() as u32;
        let p = v as *const T as *const u64;
        for i in 0..words {
            self.consts.push(unsafe { *p.add(i) });
        }
        off
    }
}

/// What `graph_eval` computes with: a handle to a recorded node.
#[derive(Clone)]
struct Val {
    b: Rc<RefCell<Builder>>,
    id: usize,
}

impl Val {
    fn is_q(&self) -> bool {
        self.b.borrow().nodes[self.id].q
    }

    /// ADD / MUL with the operand typing of eval_cpu.rs:306-428: PP -> P, QQ -> Q, mixed -> Q with the Fq operand first.
    fn binary(self, rhs: Self, base: u32) -> Self {
        let (qa, qb) = (self.is_q(), rhs.is_q());
        let id = match (qa, qb) {
            (false, false) => self.b.borrow_mut().emit(base, false, Some(self.id), Some(rhs.id), 0, 0),
            (true, true) => self.b.borrow_mut().emit(base + 1, true, Some(self.id), Some(rhs.id), 0, 0),
            (true, false) => self.b.borrow_mut().emit(base + 2, true, Some(self.id), Some(rhs.id), 0, 0),
            (false, true) => self.b.borrow_mut().emit(base + 2, true, Some(rhs.id), Some(self.id), 0, 0),
        };
        Self { b: self.b, id }
    }

    fn unary(self, op_p: u32, imm0: u32) -> Self {
        let q = self.is_q();
        let id = self.b.borrow_mut().emit(op_p + u32::from(q), q, Some(self.id), None, imm0, 0);
        Self { b: self.b, id }
    }
}

impl Add for Val {
    type Output = Self;

    fn add(self, rhs: Self) -> Self {
        self.binary(rhs, ADD_PP)
    }
}

impl Mul for Val {
    type Output = Self;

    fn mul(self, rhs: Self) -> Self {
        self.binary(rhs, MUL_PP)
    }
}

impl Neg for Val {
    type Output = Self;

    fn neg(self) -> Self {
        self.unary(NEG_P, 0)
    }
}

impl Div for Val {
    type Output = Self;

    fn div(self, rhs: Self) -> Self {
        let inv = rhs.unary(INV_P, 0);
        self.binary(inv, MUL_PP)
    }
}

impl Pow<usize> for Val {
    type Output = Self;

    fn pow(self, exp: usize) -> Self {
        self.unary(POW_P, u32::try_from(exp).expect("exponent exceeds 32 bits"))
    }
}

/// Registers by a linear scan over the (topologically ordered) nodes; returns the instruction words.
fn assign_registers(nodes: &[Node], root: usize, fq_is_ext: bool) -> Vec<u32> {
    let n = nodes.len();
    let mut last_use = alloc::vec![usize::MAX; n];
    for (k, node) in nodes.iter().enumerate() {
        for opnd in [node.a, node.b].into_iter().flatten() {
            last_use[opnd] = k;
        }
    }
    last_use[root] = n;
    let (mut free_p, mut free_q) = (Vec::<u32>::new(), Vec::<u32>::new());
    let (mut next_p, mut next_q) = (0u32, 0u32);
    let mut reg = alloc::vec![0u32; n];
    let mut words = Vec::with_capacity(4 * (n + 1));
    for (k, node) in nodes.iter().enumerate() {
        // an operand that dies here gives its register back before the result takes one
        let (a, b) = (node.a, if node.b == node.a { None } else { node.b });
        for opnd in [a, b].into_iter().flatten() {
            if last_use[opnd] == k {
                if nodes[opnd].q { free_q.push(reg[opnd]) } else { free_p.push(reg[opnd]) }
            }
        }
        let r = if node.q {
            free_q.pop().unwrap_or_else(|| { next_q += 1; next_q - 1 })
        } else {
            free_p.pop().unwrap_or_else(|| { next_p += 1; next_p - 1 })
        };
        reg[k] = r;
        let (wa, wb) = match node.op {
            X_P => (0, 0),
            CONST_P | CONST_Q | PERIODIC_P => (node.imm0, 0),
            TRACE_P | TRACE_Q => (node.imm0, node.imm1 as u32),
            op if op == POW_P || op == POW_P + 1 => (reg[node.a.unwrap()], node.imm0),
            _ => (reg[node.a.unwrap()], node.b.map_or(0, |b| reg[b])),
        };
        words.extend_from_slice(&[node.op, r, wa, wb]);
        if last_use[k] == usize::MAX {
            if node.q { free_q.push(r) } else { free_p.push(r) }
        }
    }
    words.extend_from_slice(&[if fq_is_ext { STORE_Q } else { STORE_P }, 0, reg[root], 0]);
    assert!(next_p <= 256 && next_q <= 128, "constraint program needs too many registers (limits 256 Fp / 128 Fq)");
    words
}

#[allow(clippy::too_many_arguments)]
pub fn eval<Fp: GpuFftField<FftField = Fp> + FftField, Fq: StarkExtensionOf<Fp>>(
    expr: &Expr<AlgebraicItem<FieldVariant<Fp, Fq>>>,
    challenges: &[Fq],
    hints: &[Fq],
    lde_step: usize,
    domain_offset: Fp,
    x_lde: &[Fp],
    base_trace_lde_cols: &[&[Fp]],
    extension_trace_lde_cols: Option<&[&[Fq]]>,
) -> Matrix<Fq> {
    use AlgebraicItem::*;
    let n = x_lde.len();
    assert!(n.is_power_of_two());
    debug_assert!(x_lde[0] == domain_offset, "x_lde is lde_domain.elements() (src/prover.rs:93-96)");
    let fq_is_ext = core::mem::size_of::<Fq>() != core::mem::size_of::<Fp>();
    let num_base_columns = base_trace_lde_cols.len();
    let num_extension_columns = extension_trace_lde_cols.map_or(0, <[_]>::len);

    // periodic columns: their short evaluation tables come from the CPU path's own helper and go to the device once
    let periodic_map = build_periodic_column_evals_map(expr, domain_offset, n / lde_step, lde_step, 1);
    let mut periodic_cols = Vec::new(); // keys, in table order
    let mut periodic_dev: Vec<DeviceVec<Fp>> = Vec::new();
    for (col, evals) in &periodic_map {
        match evals {
            FieldVariant::Fp(evals) => periodic_dev.push(DeviceVec::from_slice(evals)),
            FieldVariant::Fq(_) => unimplemented!("periodic columns with extension-field coefficients"),
        }
        periodic_cols.push(*col);
    }

    // ---- the graph -> SSA nodes (graph_eval caches shared nodes: every node is recorded once)
    let builder = Rc::new(RefCell::new(Builder::default()));
    let leaf = |op: u32, q: bool, imm0: u32, imm1: i32| {
        let id = builder.borrow_mut().emit(op, q, None, None, imm0, imm1);
        Val { b: Rc::clone(&builder), id }
    };
    let fq_const = |v: &Fq| {
        let off = builder.borrow_mut().constant(v);
        leaf(if fq_is_ext { CONST_Q } else { CONST_P }, fq_is_ext, off, 0)
    };
    let root = expr.graph_eval(&mut |item| match *item {
        X => leaf(X_P, false, 0, 0),
        Constant(FieldVariant::Fp(v)) => {
            let off = builder.borrow_mut().constant(&v);
            leaf(CONST_P, false, off, 0)
        }
        Constant(FieldVariant::Fq(v)) => fq_const(&v),
        Challenge(i) => fq_const(&challenges[i]),
        Hint(i) => fq_const(&hints[i]),
        Trace(col, offset) => {
            let offset = i32::try_from(offset).unwrap();
            if col < num_base_columns {
                leaf(TRACE_P, false, col as u32, offset)
            } else if col < num_base_columns + num_extension_columns {
                leaf(if fq_is_ext { TRACE_Q } else { TRACE_P }, fq_is_ext, (col - num_base_columns) as u32, offset)
            } else {
                panic!("invalid column {col}")
            }
        }
        Periodic(col) => {
            let index = periodic_cols.iter().position(|c| *c == col).unwrap();
            leaf(PERIODIC_P, false, index as u32, 0)
        }
    });
    let mut root_id = root.id;
    if fq_is_ext && !root.is_q() {
        root_id = builder.borrow_mut().emit(EMBED, true, Some(root_id), None, 0, 0); // the result is always Fq
    }
    drop(root);
    let builder = Rc::try_unwrap(builder).ok().expect("no value outlives the walk").into_inner();
    let program = assign_registers(&builder.nodes, root_id, fq_is_ext);

    // ---- columns to the device (the callers hold host slices: src/utils.rs:438-460), one launch, the result back
    let base_dev: Vec<DeviceVec<Fp>> = base_trace_lde_cols.iter().map(|c| DeviceVec::from_slice(c)).collect();
    let ext_dev: Vec<DeviceVec<Fq>> = extension_trace_lde_cols.unwrap_or(&[]).iter().map(|c| DeviceVec::from_slice(c)).collect();
    let base_ptrs: Vec<*const c_void> = base_dev.iter().map(|c| c.device_ptr() as *const c_void).collect();
    let ext_ptrs: Vec<*const c_void> = ext_dev.iter().map(|c| c.device_ptr() as *const c_void).collect();
    let periodic_ptrs: Vec<*const c_void> = periodic_dev.iter().map(|c| c.device_ptr() as *const c_void).collect();
    let periodic_lens: Vec<u32> = periodic_dev.iter().map(|c| c.len() as u32).collect();
    let out = DeviceVec::<Fq>::with_len(n);
    // x_lde is what `lde_domain.elements()` generates (src/prover.rs:93-96): offset * w^i; the kernel derives it from the offset,
    // which also lets it hoist the zerofier inverses and x^n into short tables (a caller-supplied x array would forbid that)
    sys::check(unsafe {
        sys::ms_eval_program_ex(
            get_planner().ctx(),
            program.as_ptr(),
            (program.len() / 4) as u32,
            builder.consts.as_ptr() as *const c_void,
            builder.consts.len() as u32,
            n.trailing_zeros(),
            lde_step as u32,
            &domain_offset as *const Fp as *const c_void,
            core::ptr::null(),
            base_ptrs.as_ptr(),
            base_ptrs.len() as u32,
            ext_ptrs.as_ptr(),
            ext_ptrs.len() as u32,
            periodic_ptrs.as_ptr(),
            periodic_lens.as_ptr(),
            periodic_ptrs.len() as u32,
            field_id::<Fq>(),
            out.device_ptr(),
            0,
        )
    });
    let mut result = Vec::with_capacity_in(n, GpuAllocator);
    result.resize(n, Fq::zero());
    out.download(&mut result);
    Matrix::new(alloc::vec![result])
}
