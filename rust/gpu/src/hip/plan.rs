//! `Planner`, `GpuFft`, `GpuIfft`, the RPO front-ends and `gen_rpo_merkle_tree` for the `hip` arm
//! (reference: gpu/src/plan.rs:32-174, 236-325, 327-351, 464-469).
use super::sys;
use super::utils::{GpuField, GpuVec};
use ark_poly::domain::Radix2EvaluationDomain;
use core::ffi::c_void;
use core::marker::PhantomData;
use once_cell::sync::Lazy;

/// `Planner` (gpu/src/plan.rs:327-351): owns the device context; `command_queue` becomes the context's stream.
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
}
impl Drop for Planner {
    fn drop(&mut self) { unsafe { sys::ms_ctx_destroy(self.ctx); } }
}
/// `get_planner()` (gpu/src/plan.rs:464-469)
pub static PLANNER: Lazy<Planner> = Lazy::new(|| Planner::new(0));

/// `GpuFft` (gpu/src/plan.rs:236-279): forward transform over `domain`, natural order in and out,
/// `encode` per column, `execute` consumes the plan.
pub struct GpuFft<'a, F: GpuField> { plan: *mut sys::ms_ntt_plan, _m: PhantomData<&'a F> }
impl<'a, F: GpuField> GpuFft<'a, F> {
    /// the Metal arm asserts >= 2048 (plan.rs:248); this backend has no lower bound
    pub const MIN_SIZE: usize = 1;
    pub fn encode(&mut self, column: &mut GpuVec<F>) {
        sys::check(unsafe { sys::ms_ntt_encode(self.plan, column.device_ptr()) })
    }
    pub fn execute(self) {
        sys::check(unsafe { sys::ms_ntt_execute(self.plan) });
        sys::check(unsafe { sys::ms_ntt_plan_destroy(self.plan) });
        core::mem::forget(self);
    }
}
impl<'a, F: GpuField> From<Radix2EvaluationDomain<F::FftField>> for GpuFft<'a, F>
where F::FftField: ark_ff::FftField {
    fn from(d: Radix2EvaluationDomain<F::FftField>) -> Self {
        let mut plan = core::ptr::null_mut();
        sys::check(unsafe {
            sys::ms_ntt_plan_create(PLANNER.ctx(), F::FIELD_ID, d.log_size_of_group, 0,
                &d.offset as *const _ as *const c_void, &d.group_gen as *const _ as *const c_void, &mut plan)
        });
        Self { plan, _m: PhantomData }
    }
}
impl<'a, F: GpuField> Drop for GpuFft<'a, F> {
    fn drop(&mut self) { unsafe { sys::ms_ntt_plan_destroy(self.plan); } }
}

/// `GpuIfft` (gpu/src/plan.rs:282-325): inverse transform including the n^-1 (and offset^-i) scaling.
pub struct GpuIfft<'a, F: GpuField> { plan: *mut sys::ms_ntt_plan, _m: PhantomData<&'a F> }
impl<'a, F: GpuField> GpuIfft<'a, F> {
    pub const MIN_SIZE: usize = 1;
    pub fn encode(&mut self, column: &mut GpuVec<F>) {
        sys::check(unsafe { sys::ms_ntt_encode(self.plan, column.device_ptr()) })
    }
    pub fn execute(self) {
        sys::check(unsafe { sys::ms_ntt_execute(self.plan) });
        sys::check(unsafe { sys::ms_ntt_plan_destroy(self.plan) });
        core::mem::forget(self);
    }
}
impl<'a, F: GpuField> From<Radix2EvaluationDomain<F::FftField>> for GpuIfft<'a, F>
where F::FftField: ark_ff::FftField {
    fn from(d: Radix2EvaluationDomain<F::FftField>) -> Self {
        let mut plan = core::ptr::null_mut();
        sys::check(unsafe {
            sys::ms_ntt_plan_create(PLANNER.ctx(), F::FIELD_ID, d.log_size_of_group, 1,
                &d.offset as *const _ as *const c_void, &d.group_gen as *const _ as *const c_void, &mut plan)
        });
        Self { plan, _m: PhantomData }
    }
}
impl<'a, F: GpuField> Drop for GpuIfft<'a, F> {
    fn drop(&mut self) { unsafe { sys::ms_ntt_plan_destroy(self.plan); } }
}

/// `prover.rs:50-51` in one call: `interpolate(trace_domain)` + `bit_reversed_evaluate(lde_domain)`.
pub fn lde<F: GpuField>(columns: &[GpuVec<F>], log_blowup: u32, offset: &F::FftField, bit_reversed: bool) -> Vec<GpuVec<F>> {
    let n = columns[0].len();
    let outs: Vec<GpuVec<F>> = columns.iter().map(|_| GpuVec::with_len(n << log_blowup)).collect();
    let ins: Vec<*const c_void> = columns.iter().map(|c| c.device_ptr() as *const c_void).collect();
    let out_ptrs: Vec<*mut c_void> = outs.iter().map(|c| c.device_ptr()).collect();
    sys::check(unsafe {
        sys::ms_lde(PLANNER.ctx(), F::FIELD_ID, n.trailing_zeros(), log_blowup, offset as *const _ as *const c_void,
            ins.as_ptr(), out_ptrs.as_ptr(), ins.len() as u32, bit_reversed as i32)
    });
    outs
}

/// `GpuRpo256ColumnMajor` (gpu/src/plan.rs:32-107): `update(col)` per column, `finish()` -> n digests of 4 elements.
pub struct GpuRpo256ColumnMajor<'a, F: GpuField> { n: usize, requires_padding: bool, cols: Vec<*const c_void>, _m: PhantomData<&'a F> }
impl<'a, F: GpuField> GpuRpo256ColumnMajor<'a, F> {
    pub const RATE: usize = 8;
    pub fn new(n: usize, requires_padding: bool) -> Self { Self { n, requires_padding, cols: Vec::new(), _m: PhantomData } }
    pub fn update(&mut self, col: &'a GpuVec<F>) {
        assert_eq!(col.len(), self.n);
        self.cols.push(col.device_ptr() as *const c_void);
    }
    pub fn finish(self) -> GpuVec<F> {
        assert!(!self.cols.is_empty(), "the zero-length input is not allowed");                   // plan.rs:72
        assert_eq!(self.requires_padding, self.cols.len() % Self::RATE != 0);
        let out = GpuVec::<F>::with_len(self.n * 4);
        sys::check(unsafe { sys::ms_rpo256_rows(PLANNER.ctx(), self.n, self.cols.as_ptr(), self.cols.len() as u32, out.device_ptr()) });
        out
    }
}
/// `GpuRpo256RowMajor` (gpu/src/plan.rs:109-148): rows of 8 elements.
pub struct GpuRpo256RowMajor<'a, F: GpuField> { n: usize, rows: Option<&'a GpuVec<F>> }
impl<'a, F: GpuField> GpuRpo256RowMajor<'a, F> {
    pub fn new(n: usize, _requires_padding: bool) -> Self { Self { n, rows: None } }
    pub fn update(&mut self, rows: &'a GpuVec<F>) { assert_eq!(rows.len(), self.n * 8); self.rows = Some(rows); }
    pub fn finish(self) -> GpuVec<F> {
        let rows = self.rows.expect("the zero-length input is not allowed");                     // plan.rs:141-146
        let out = GpuVec::<F>::with_len(self.n * 4);
        sys::check(unsafe { sys::ms_rpo256_rows_row_major(PLANNER.ctx(), self.n, 8, rows.device_ptr(), out.device_ptr()) });
        out
    }
}
/// `gen_rpo_merkle_tree(leaves)` (gpu/src/plan.rs:150-174)
pub fn gen_rpo_merkle_tree<F: GpuField>(leaves: &GpuVec<F>) -> GpuVec<F> {
    let n = leaves.len() / 4;
    let nodes = GpuVec::<F>::with_len(n * 4);
    sys::check(unsafe { sys::ms_rpo256_merkle(PLANNER.ctx(), n, leaves.device_ptr(), nodes.device_ptr()) });
    nodes
}
