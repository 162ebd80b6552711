//! `Planner` / `get_planner`, `GpuFft`, `GpuIfft`, the RPO front-ends and `gen_rpo_merkle_tree` for the `hip` arm
//! (reference: gpu/src/plan.rs:32-174, 236-325, 327-351, 464-469).  Same names, same signatures on host slices; the
//! `*_device` methods are the resident variants the patched callers use (rust/patches/).
use super::sys;
use super::utils::{field_id, DeviceVec};
use crate::GpuField;
use ark_poly::domain::Radix2EvaluationDomain;
use ark_poly::EvaluationDomain;
use core::ffi::c_void;
use core::marker::PhantomData;
use once_cell::sync::Lazy;

/// `Planner` (gpu/src/plan.rs:331-351): owns the device context; `command_queue` becomes the context's stream.
pub struct Planner { ctx: *mut sys::ms_ctx }
unsafe impl Send for Planner {}
unsafe impl Sync for Planner {}
impl Planner {
    pub fn new(device: i32) -> Self {
        let mut ctx = core::ptr::null_mut();
        sys::check(unsafe { sys::ms_ctx_create(device, &mut ctx) });
        Self { ctx }
    }
    pub fn ctx(&self) -> *mut sys::ms_ctx { self.ctx }
    /// `command_buffer.commit(); command_buffer.wait_until_completed()`
    pub fn sync(&self) { sys::check(unsafe { sys::ms_sync(self.ctx) }) }
    /// Specialised constraint kernels of this context: compiled / loaded from the on-disk cache / left to the interpreter.  The Metal arm has
    /// nothing to report (its shaders are a build artefact, gpu/src/plan.rs:30); here `compile_failures > 0` is worth a log line.
    pub fn jit_stats(&self) -> sys::ms_jit_stats {
        let mut st = sys::ms_jit_stats::default();
        sys::check(unsafe { sys::ms_eval_jit_stats(self.ctx, &mut st) });
        st
    }
}
impl Default for Planner {
    /// `Planner::default()` (gpu/src/plan.rs:464-468: the system default device): GPU `$MINISTARK_HIP_DEVICE`, else 0.
    fn default() -> Self {
        Planner::new(std::env::var("MINISTARK_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0))
    }
}
impl Drop for Planner {
    fn drop(&mut self) { unsafe { sys::ms_ctx_destroy(self.ctx); } }
}
static PLANNER: Lazy<Planner> = Lazy::new(Planner::default);
/// `get_planner()` (gpu/src/plan.rs:327-329)
pub fn get_planner() -> &'static Planner { &PLANNER }

fn new_plan<F: GpuField>(d: &Radix2EvaluationDomain<F::FftField>, inverse: bool) -> *mut sys::ms_ntt_plan
where F::FftField: ark_ff::FftField {
    let mut plan = core::ptr::null_mut();
    sys::check(unsafe {
        sys::ms_ntt_plan_create(get_planner().ctx(), field_id::<F>(), d.log_size_of_group, inverse as i32,
            &d.offset as *const _ as *const c_void, &d.group_gen as *const _ as *const c_void, &mut plan)
    });
    plan
}

/// a host buffer handed to `encode`: uploaded now, written back by `execute` (the Metal arm wraps the same memory with
/// `buffer_mut_no_copy`, gpu/src/plan.rs:254-263, and likewise relies on it staying alive until `execute`)
struct Pending<F> { host: *mut F, len: usize, dev: DeviceVec<F> }

macro_rules! transform {
    ($name:ident, $inverse:expr, $doc:expr) => {
        #[doc = $doc]
        pub struct $name<'a, F: GpuField> { plan: *mut sys::ms_ntt_plan, n: usize, pending: Vec<Pending<F>>, _m: PhantomData<&'a mut [F]> }
        impl<'a, F: GpuField> $name<'a, F> {
            /// the Metal arm's threshold (gpu/src/plan.rs:243): callers fall back to arkworks below it (src/fri.rs:580,602); kept
            /// so that the same calls take the same branch.  (The library itself has no lower bound.)
            pub const MIN_SIZE: usize = 2048;
            /// `encode(&mut self, buffer: &mut [F])` (gpu/src/plan.rs:254-263, 300-309)
            pub fn encode(&mut self, buffer: &'a mut [F]) {
                assert_eq!(self.n, buffer.len());
                let dev = DeviceVec::from_slice(buffer);
                sys::check(unsafe { sys::ms_ntt_encode(self.plan, dev.device_ptr()) });
                self.pending.push(Pending { host: buffer.as_mut_ptr(), len: buffer.len(), dev });
            }
            /// the same on a column that already lives in HBM: nothing crosses PCIe
            pub fn encode_device(&mut self, column: &'a mut DeviceVec<F>) {
                assert_eq!(self.n, column.len());
                sys::check(unsafe { sys::ms_ntt_encode(self.plan, column.device_ptr()) });
            }
            /// `execute(self)` (gpu/src/plan.rs:275-278, 321-324): runs every encoded transform, waits, writes host buffers back
            pub fn execute(mut self) {
                sys::check(unsafe { sys::ms_ntt_execute(self.plan) });
                for p in self.pending.drain(..) {
                    // SAFETY: `encode` borrowed the slice for 'a, which outlives self
                    p.dev.download(unsafe { core::slice::from_raw_parts_mut(p.host, p.len) });
                }
            }
        }
        impl<'a, F: GpuField> From<Radix2EvaluationDomain<F::FftField>> for $name<'a, F>
        where F::FftField: ark_ff::FftField {
            fn from(d: Radix2EvaluationDomain<F::FftField>) -> Self {
                Self { plan: new_plan::<F>(&d, $inverse), n: d.size(), pending: Vec::new(), _m: PhantomData }
            }
        }
        impl<'a, F: GpuField> Drop for $name<'a, F> {
            fn drop(&mut self) { unsafe { sys::ms_ntt_plan_destroy(self.plan); } }
        }
    };
}
transform!(GpuFft, false, "`GpuFft` (gpu/src/plan.rs:236-279): forward transform over `domain` (subgroup or coset), natural order in and out.");
transform!(GpuIfft, true, "`GpuIfft` (gpu/src/plan.rs:282-325): inverse transform including the n^-1 (and offset^-i) scaling.");

/// `prover.rs:50-51` in one call on resident columns: `interpolate(trace_domain)` + `bit_reversed_evaluate(lde_domain)`.
pub fn lde_device<F: GpuField>(columns: &[DeviceVec<F>], log_blowup: u32, offset: &F::FftField, bit_reversed: bool) -> Vec<DeviceVec<F>> {
    let n = columns[0].len();
    let outs: Vec<DeviceVec<F>> = columns.iter().map(|_| DeviceVec::with_len(n << log_blowup)).collect();
    let ins: Vec<*const c_void> = columns.iter().map(|c| c.device_ptr() as *const c_void).collect();
    let out_ptrs: Vec<*mut c_void> = outs.iter().map(|c| c.device_ptr()).collect();
    sys::check(unsafe {
        sys::ms_lde(get_planner().ctx(), field_id::<F>(), n.trailing_zeros(), log_blowup, offset as *const _ as *const c_void,
            ins.as_ptr(), out_ptrs.as_ptr(), ins.len() as u32, bit_reversed as i32)
    });
    outs
}
/// `Matrix::into_evaluations(domain)` (+ `bit_reverse_rows`) on resident coefficient columns shorter than the domain: the
/// reference's `column.resize(domain.size(), F::zero())` (src/matrix.rs:201) is implicit, nothing is padded or copied.
pub fn evaluate_device<F: GpuField>(coeffs: &[DeviceVec<F>], log_domain: u32, offset: &F::FftField, bit_reversed: bool) -> Vec<DeviceVec<F>> {
    let n = coeffs[0].len();
    let outs: Vec<DeviceVec<F>> = coeffs.iter().map(|_| DeviceVec::with_len(1usize << log_domain)).collect();
    let ins: Vec<*const c_void> = coeffs.iter().map(|c| c.device_ptr() as *const c_void).collect();
    let out_ptrs: Vec<*mut c_void> = outs.iter().map(|c| c.device_ptr()).collect();
    sys::check(unsafe {
        sys::ms_evaluate(get_planner().ctx(), field_id::<F>(), n.trailing_zeros(), log_domain, offset as *const _ as *const c_void,
            ins.as_ptr(), out_ptrs.as_ptr(), ins.len() as u32, bit_reversed as i32)
    });
    outs
}
/// `MatrixMerkleTreeImpl::from_matrix` with SHA-256 (src/merkle.rs:412-508, src/hash.rs:77-99) on resident columns:
/// (leaves, nodes) as 32-byte digests; nodes[1] is the root (src/merkle.rs:145-147).
pub fn sha256_commit_device<F: GpuField>(columns: &[DeviceVec<F>]) -> (DeviceVec<[u8; 32]>, DeviceVec<[u8; 32]>) {
    let n = columns[0].len();
    let leaves = DeviceVec::<[u8; 32]>::with_len(n);
    let nodes = DeviceVec::<[u8; 32]>::with_len(n);
    let cols: Vec<*const c_void> = columns.iter().map(|c| c.device_ptr() as *const c_void).collect();
    sys::check(unsafe { sys::ms_sha256_rows(get_planner().ctx(), field_id::<F>(), n, cols.as_ptr(), cols.len() as u32, leaves.device_ptr()) });
    sys::check(unsafe { sys::ms_sha256_merkle(get_planner().ctx(), n, leaves.device_ptr() as *const c_void, nodes.device_ptr()) });
    (leaves, nodes)
}

/// `GpuRpo256ColumnMajor` (gpu/src/plan.rs:32-107): `update(col)` per column, `finish()` -> n digests of 4 elements.
pub struct GpuRpo256ColumnMajor<'a, F: GpuField> { n: usize, requires_padding: bool, cols: Vec<*const c_void>, _m: PhantomData<&'a F> }
impl<'a, F: GpuField> GpuRpo256ColumnMajor<'a, F> {
    pub const RATE: usize = 8;
    pub fn new(n: usize, requires_padding: bool) -> Self { Self { n, requires_padding, cols: Vec::new(), _m: PhantomData } }
    pub fn update(&mut self, col: &'a DeviceVec<F>) {
        assert_eq!(col.len(), self.n);
        self.cols.push(col.device_ptr() as *const c_void);
    }
    pub fn finish(self) -> DeviceVec<F> {
        assert!(!self.cols.is_empty(), "the zero-length input is not allowed");                   // plan.rs:72
        assert_eq!(self.requires_padding, self.cols.len() % Self::RATE != 0);
        let out = DeviceVec::<F>::with_len(self.n * 4);
        sys::check(unsafe { sys::ms_rpo256_rows(get_planner().ctx(), self.n, self.cols.as_ptr(), self.cols.len() as u32, out.device_ptr()) });
        out
    }
}
/// `GpuRpo256RowMajor` (gpu/src/plan.rs:109-148): rows of 8 elements.
pub struct GpuRpo256RowMajor<'a, F: GpuField> { n: usize, rows: Option<&'a DeviceVec<F>> }
impl<'a, F: GpuField> GpuRpo256RowMajor<'a, F> {
    pub fn new(n: usize, _requires_padding: bool) -> Self { Self { n, rows: None } }
    pub fn update(&mut self, rows: &'a DeviceVec<F>) { assert_eq!(rows.len(), self.n * 8); self.rows = Some(rows); }
    pub fn finish(self) -> DeviceVec<F> {
        let rows = self.rows.expect("the zero-length input is not allowed");                     // plan.rs:141-146
        let out = DeviceVec::<F>::with_len(self.n * 4);
        sys::check(unsafe { sys::ms_rpo256_rows_row_major(get_planner().ctx(), self.n, 8, rows.device_ptr(), out.device_ptr()) });
        out
    }
}
/// `gen_rpo_merkle_tree(leaves)` (gpu/src/plan.rs:150-174)
pub fn gen_rpo_merkle_tree<F: GpuField>(leaves: &DeviceVec<F>) -> DeviceVec<F> {
    let n = leaves.len() / 4;
    let nodes = DeviceVec::<F>::with_len(n * 4);
    sys::check(unsafe { sys::ms_rpo256_merkle(get_planner().ctx(), n, leaves.device_ptr(), nodes.device_ptr()) });
    nodes
}
