//! Device memory and field ids for the `hip` arm.
//!
//! What stays exactly as in the reference: the crate's platform-independent items -- `GpuField` / `GpuFftField` /
//! `GpuAdd` / `GpuMul` / `GpuFrom` (gpu/src/lib.rs:20-41), `utils::{bit_reverse, bit_reverse_index}` (gpu/src/utils.rs:4-78,
//! CPU functions on slices) and the main crate's `GpuVec<T> = Vec<T, GpuAllocator>` (src/utils.rs:438-460: a std `Vec`,
//! global allocator on every non-Apple target).  Callers keep handing `&mut [F]` host slices to `GpuFft::encode`
//! (gpu/src/plan.rs:254), so `prover.rs`, `composer.rs`, `matrix.rs`, `fri.rs` compile unchanged against this arm.
//!
//! What is new: an MI355X does not share memory with the host (Apple GPUs do: `buffer_no_copy`,
//! gpu/src/utils.rs:103-134).  `DeviceVec<F>` is a column in HBM; the slice-taking front-ends upload into one and download
//! at `execute()`, the `*_device` front-ends and `rust/patches/` keep columns resident between calls.
use super::plan::get_planner;
use super::sys;
use crate::GpuField;
use core::ffi::{c_int, c_void};
use core::marker::PhantomData;

/// `GpuField::field_name()` (gpu/src/lib.rs:37-40, values in gpu/src/fields.rs:16,58,214,252) -> field id of the C ABI.
pub fn field_id<F: GpuField>() -> c_int {
    match F::field_name().as_str() {
        "p18446744069414584321_fp" => sys::MS_GOLDILOCKS_FP,
        "p18446744069414584321_fq3" => sys::MS_GOLDILOCKS_FQ3,
        "p3618502788666131213697322783095070105623107215331596699973092056135872020481_fp" => sys::MS_STARK252_FP,
        other => panic!("no HIP kernels for field {other}"),
    }
}

/// A column of `len` elements of `F` in HBM (arkworks' in-memory representation, Montgomery limbs, untouched).
pub struct DeviceVec<F> {
    ptr: *mut c_void,
    len: usize,
    _m: PhantomData<F>,
}
unsafe impl<F: Send> Send for DeviceVec<F> {}

impl<F> DeviceVec<F> {
    pub fn with_len(len: usize) -> Self {
        let mut ptr = core::ptr::null_mut();
        let bytes = core::cmp::max(len * core::mem::size_of::<F>(), 8);
        sys::check(unsafe { sys::ms_alloc(get_planner().ctx(), bytes, &mut ptr) });
        Self { ptr, len, _m: PhantomData }
    }
    /// host -> device (what `buffer_no_copy` is on unified memory)
    pub fn from_slice(values: &[F]) -> Self {
        let v = Self::with_len(values.len());
        v.upload(values);
        v
    }
    pub fn upload(&self, values: &[F]) {
        assert_eq!(values.len(), self.len);
        if self.len != 0 {
            sys::check(unsafe { sys::ms_upload(get_planner().ctx(), self.ptr, values.as_ptr() as *const c_void, core::mem::size_of_val(values)) });
        }
    }
    /// device -> host, into an existing slice of the same length
    pub fn download(&self, out: &mut [F]) {
        assert_eq!(out.len(), self.len);
        if self.len != 0 {
            sys::check(unsafe { sys::ms_download(get_planner().ctx(), out.as_mut_ptr() as *mut c_void, self.ptr, self.len * core::mem::size_of::<F>()) });
        }
    }
    pub fn to_vec(&self) -> Vec<F> {
        let mut out = Vec::<F>::with_capacity(self.len);
        // SAFETY: every element is written by the download below; F is plain old data (field elements)
        unsafe { out.set_len(self.len) };
        self.download(&mut out);
        out
    }
    /// `column.resize(domain.size(), F::zero())` of `into_evaluations_gpu` (src/matrix.rs:201): a longer device column whose
    /// head is this one and whose tail is zero.  (The fused `lde` / `evaluate` entry points never need it: they read the
    /// short column and treat the rest of the domain as implicit zeros.)
    pub fn resized(&self, new_len: usize, zero: &F) -> Self {
        let out = Self::with_len(new_len);
        let keep = core::cmp::min(self.len, new_len);
        sys::check(unsafe { sys::ms_copy(get_planner().ctx(), out.ptr, self.ptr, keep * core::mem::size_of::<F>()) });
        if new_len > keep {
            let tail = vec_of(zero, new_len - keep);
            let dst = unsafe { (out.ptr as *mut u8).add(keep * core::mem::size_of::<F>()) } as *mut c_void;
            sys::check(unsafe { sys::ms_upload(get_planner().ctx(), dst, tail.as_ptr() as *const c_void, core::mem::size_of_val(&tail[..])) });
        }
        out
    }
    pub fn len(&self) -> usize { self.len }
    pub fn is_empty(&self) -> bool { self.len == 0 }
    pub fn device_ptr(&self) -> *mut c_void { self.ptr }
}
fn vec_of<F>(value: &F, n: usize) -> Vec<F> {
    let mut v = Vec::with_capacity(n);
    for _ in 0..n { v.push(unsafe { core::ptr::read(value) }); }
    v
}
impl<F> Clone for DeviceVec<F> {
    /// `column.clone()` inside `interpolate` / `evaluate` (src/matrix.rs:155-163, 237-243): device-to-device.
    fn clone(&self) -> Self {
        let v = Self::with_len(self.len);
        sys::check(unsafe { sys::ms_copy(get_planner().ctx(), v.ptr, self.ptr, self.len * core::mem::size_of::<F>()) });
        v
    }
}
impl<F> Drop for DeviceVec<F> {
    fn drop(&mut self) { unsafe { sys::ms_free(get_planner().ctx(), self.ptr); } }
}

/// Bit reversal of resident columns (`BitReverseGpuStage`, gpu/src/stage.rs:280-332).  The slice function
/// `crate::utils::bit_reverse` (gpu/src/utils.rs:32-78) is CPU code and stays what host slices use.
pub fn bit_reverse_device<F: GpuField>(columns: &mut [&mut DeviceVec<F>]) {
    if columns.is_empty() { return; }
    let n = columns[0].len();
    assert!(n.is_power_of_two());
    let ptrs: Vec<*mut c_void> = columns.iter().map(|c| c.device_ptr()).collect();
    sys::check(unsafe { sys::ms_bit_reverse(get_planner().ctx(), field_id::<F>(), n.trailing_zeros(), ptrs.as_ptr(), ptrs.len() as u32) });
}
