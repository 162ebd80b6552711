//! `GpuField`, `GpuVec`, `bit_reverse` for the `hip` arm (reference: gpu/src/lib.rs:20-26,
//! src/utils.rs:438-493 `GpuAllocator`/`GpuVec`, gpu/src/utils.rs:4-78).
//!
//! The one API-visible difference from the Metal arm: Apple GPUs share memory with the host
//! (`buffer_no_copy`, gpu/src/utils.rs:103-134), an MI355X does not.  A `GpuVec<F>` therefore owns a
//! device allocation; `to_vec()` / `from_slice()` are the explicit mirror, and columns are expected to
//! stay on the device between calls.
use super::plan::PLANNER;
use super::sys;
use core::ffi::{c_int, c_void};
use core::marker::PhantomData;

/// `GpuField::field_name()` (gpu/src/lib.rs:20-26) becomes a numeric id of the C ABI.
pub trait GpuField: Sized + Copy {
    type FftField;
    const FIELD_ID: c_int;
}
// gpu/src/fields.rs:37-95 (Goldilocks Fp, Fq3) and :229-264 (the 252-bit field):
//   impl GpuField for p18446744069414584321::ark::Fp  { type FftField = Self; const FIELD_ID: c_int = sys::MS_GOLDILOCKS_FP; }
//   impl GpuField for p18446744069414584321::ark::Fq3 { type FftField = Fp;   const FIELD_ID: c_int = sys::MS_GOLDILOCKS_FQ3; }
//   impl GpuField for p3618...::ark::Fp                { type FftField = Self; const FIELD_ID: c_int = sys::MS_STARK252_FP; }

/// A column of `len` elements of `F` in HBM (arkworks' in-memory representation, Montgomery limbs).
pub struct GpuVec<F: GpuField> {
    ptr: *mut c_void,
    len: usize,
    _m: PhantomData<F>,
}

impl<F: GpuField> GpuVec<F> {
    pub fn with_len(len: usize) -> Self {
        let mut ptr = core::ptr::null_mut();
        let bytes = core::cmp::max(len * core::mem::size_of::<F>(), 8);
        sys::check(unsafe { sys::ms_alloc(PLANNER.ctx(), bytes, &mut ptr) });
        Self { ptr, len, _m: PhantomData }
    }
    pub fn from_slice(values: &[F]) -> Self {
        let v = Self::with_len(values.len());
        if !values.is_empty() {
            sys::check(unsafe { sys::ms_upload(PLANNER.ctx(), v.ptr, values.as_ptr() as *const c_void, core::mem::size_of_val(values)) });
        }
        v
    }
    pub fn to_vec(&self) -> Vec<F> {
        let mut out = Vec::<F>::with_capacity(self.len);
        if self.len != 0 {
            sys::check(unsafe { sys::ms_download(PLANNER.ctx(), out.as_mut_ptr() as *mut c_void, self.ptr, self.len * core::mem::size_of::<F>()) });
        }
        unsafe { out.set_len(self.len) };
        out
    }
    pub fn len(&self) -> usize { self.len }
    pub fn is_empty(&self) -> bool { self.len == 0 }
    pub fn device_ptr(&self) -> *mut c_void { self.ptr }
}
impl<F: GpuField> Clone for GpuVec<F> {
    /// `column.clone()` inside `interpolate` / `evaluate` (src/matrix.rs:155-163, 237-243): device-to-device.
    fn clone(&self) -> Self {
        let v = Self::with_len(self.len);
        sys::check(unsafe { sys::ms_copy(PLANNER.ctx(), v.ptr, self.ptr, self.len * core::mem::size_of::<F>()) });
        v
    }
}
impl<F: GpuField> Drop for GpuVec<F> {
    fn drop(&mut self) { unsafe { sys::ms_free(PLANNER.ctx(), self.ptr); } }
}

/// `bit_reverse(&mut [F])` on GPU columns (gpu/src/utils.rs:32-78, `BitReverseGpuStage`).
pub fn bit_reverse<F: GpuField>(columns: &mut [&mut GpuVec<F>]) {
    if columns.is_empty() { return; }
    let n = columns[0].len();
    assert!(n.is_power_of_two());
    let ptrs: Vec<*mut c_void> = columns.iter().map(|c| c.device_ptr()).collect();
    sys::check(unsafe { sys::ms_bit_reverse(PLANNER.ctx(), F::FIELD_ID, n.trailing_zeros(), ptrs.as_ptr(), ptrs.len() as u32) });
    sys::check(unsafe { sys::ms_sync(PLANNER.ctx()) });
}
