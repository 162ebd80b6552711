//! The element-wise stages of `gpu/src/stage.rs:115-1155` for the `hip` arm: same struct names, `new(n)` and
//! `encode(...)` with the Metal command buffer replaced by the planner's stream (`get_planner().sync()` stands in for
//! `command_buffer.commit(); wait_until_completed()`).  Each is a thin delegate onto one C entry point, exactly
//! as each Metal stage is a thin delegate onto one shader of evaluation_shaders.h.metal:11-168.
//! `shift` rotates the rhs index: dst[i] (op)= rhs[(i + shift) mod n]  (stage.rs:153-173).
use super::plan::get_planner;
use super::sys;
use super::utils::{field_id, DeviceVec};
use crate::GpuField;
use core::ffi::c_void;
use core::marker::PhantomData;

/// `MulIntoStage` (gpu/src/stage.rs:115-173)
pub struct MulIntoStage<LhsF, RhsF = LhsF> { n: usize, _m: PhantomData<(LhsF, RhsF)> }
impl<LhsF: GpuField, RhsF: GpuField> MulIntoStage<LhsF, RhsF> {
    pub fn new(n: usize) -> Self { Self { n, _m: PhantomData } }
    pub fn encode(&self, dst: &mut DeviceVec<LhsF>, lhs: &DeviceVec<LhsF>, rhs: &DeviceVec<RhsF>, shift: isize) {
        sys::check(unsafe { sys::ms_binary(get_planner().ctx(), sys::MS_MUL, field_id::<LhsF>(), field_id::<RhsF>(), self.n, dst.device_ptr(), lhs.device_ptr() as *const c_void, rhs.device_ptr() as *const c_void, shift as _) })
    }
}
/// `MulAssignStage` (gpu/src/stage.rs:176-233)
pub struct MulAssignStage<LhsF, RhsF = LhsF> { n: usize, _m: PhantomData<(LhsF, RhsF)> }
impl<LhsF: GpuField, RhsF: GpuField> MulAssignStage<LhsF, RhsF> {
    pub fn new(n: usize) -> Self { Self { n, _m: PhantomData } }
    pub fn encode(&self, dst: &mut DeviceVec<LhsF>, rhs: &DeviceVec<RhsF>, shift: isize) {
        sys::check(unsafe { sys::ms_binary(get_planner().ctx(), sys::MS_MUL, field_id::<LhsF>(), field_id::<RhsF>(), self.n, dst.device_ptr(), dst.device_ptr() as *const c_void, rhs.device_ptr() as *const c_void, shift as _) })
    }
}
/// `AddAssignStage` (gpu/src/stage.rs:393-454)
pub struct AddAssignStage<LhsF, RhsF = LhsF> { n: usize, _m: PhantomData<(LhsF, RhsF)> }
impl<LhsF: GpuField, RhsF: GpuField> AddAssignStage<LhsF, RhsF> {
    pub fn new(n: usize) -> Self { Self { n, _m: PhantomData } }
    pub fn encode(&self, dst: &mut DeviceVec<LhsF>, rhs: &DeviceVec<RhsF>, shift: isize) {
        sys::check(unsafe { sys::ms_binary(get_planner().ctx(), sys::MS_ADD, field_id::<LhsF>(), field_id::<RhsF>(), self.n, dst.device_ptr(), dst.device_ptr() as *const c_void, rhs.device_ptr() as *const c_void, shift as _) })
    }
}
/// `AddIntoStage` (gpu/src/stage.rs:457-520)
pub struct AddIntoStage<LhsF, RhsF = LhsF> { n: usize, _m: PhantomData<(LhsF, RhsF)> }
impl<LhsF: GpuField, RhsF: GpuField> AddIntoStage<LhsF, RhsF> {
    pub fn new(n: usize) -> Self { Self { n, _m: PhantomData } }
    pub fn encode(&self, dst: &mut DeviceVec<LhsF>, lhs: &DeviceVec<LhsF>, rhs: &DeviceVec<RhsF>, shift: isize) {
        sys::check(unsafe { sys::ms_binary(get_planner().ctx(), sys::MS_ADD, field_id::<LhsF>(), field_id::<RhsF>(), self.n, dst.device_ptr(), lhs.device_ptr() as *const c_void, rhs.device_ptr() as *const c_void, shift as _) })
    }
}
/// `AddIntoConstStage` (gpu/src/stage.rs:523-578)
pub struct AddIntoConstStage<LhsF, RhsF = LhsF> { n: usize, _m: PhantomData<(LhsF, RhsF)> }
impl<LhsF: GpuField, RhsF: GpuField> AddIntoConstStage<LhsF, RhsF> {
    pub fn new(n: usize) -> Self { Self { n, _m: PhantomData } }
    pub fn encode(&self, dst: &mut DeviceVec<LhsF>, lhs: &DeviceVec<LhsF>, rhs: &RhsF) {
        sys::check(unsafe { sys::ms_binary_const(get_planner().ctx(), sys::MS_ADD, field_id::<LhsF>(), field_id::<RhsF>(), self.n, dst.device_ptr(), lhs.device_ptr() as *const c_void, rhs as *const RhsF as *const c_void) })
    }
}
/// `AddAssignConstStage` (gpu/src/stage.rs:637-691)
pub struct AddAssignConstStage<LhsF, RhsF = LhsF> { n: usize, _m: PhantomData<(LhsF, RhsF)> }
impl<LhsF: GpuField, RhsF: GpuField> AddAssignConstStage<LhsF, RhsF> {
    pub fn new(n: usize) -> Self { Self { n, _m: PhantomData } }
    pub fn encode(&self, dst: &mut DeviceVec<LhsF>, rhs: &RhsF) {
        sys::check(unsafe { sys::ms_binary_const(get_planner().ctx(), sys::MS_ADD, field_id::<LhsF>(), field_id::<RhsF>(), self.n, dst.device_ptr(), dst.device_ptr() as *const c_void, rhs as *const RhsF as *const c_void) })
    }
}
/// `MulIntoConstStage` (gpu/src/stage.rs:694-749)
pub struct MulIntoConstStage<LhsF, RhsF = LhsF> { n: usize, _m: PhantomData<(LhsF, RhsF)> }
impl<LhsF: GpuField, RhsF: GpuField> MulIntoConstStage<LhsF, RhsF> {
    pub fn new(n: usize) -> Self { Self { n, _m: PhantomData } }
    pub fn encode(&self, dst: &mut DeviceVec<LhsF>, lhs: &DeviceVec<LhsF>, rhs: &RhsF) {
        sys::check(unsafe { sys::ms_binary_const(get_planner().ctx(), sys::MS_MUL, field_id::<LhsF>(), field_id::<RhsF>(), self.n, dst.device_ptr(), lhs.device_ptr() as *const c_void, rhs as *const RhsF as *const c_void) })
    }
}
/// `MulAssignConstStage` (gpu/src/stage.rs:752-805)
pub struct MulAssignConstStage<LhsF, RhsF = LhsF> { n: usize, _m: PhantomData<(LhsF, RhsF)> }
impl<LhsF: GpuField, RhsF: GpuField> MulAssignConstStage<LhsF, RhsF> {
    pub fn new(n: usize) -> Self { Self { n, _m: PhantomData } }
    pub fn encode(&self, dst: &mut DeviceVec<LhsF>, rhs: &RhsF) {
        sys::check(unsafe { sys::ms_binary_const(get_planner().ctx(), sys::MS_MUL, field_id::<LhsF>(), field_id::<RhsF>(), self.n, dst.device_ptr(), dst.device_ptr() as *const c_void, rhs as *const RhsF as *const c_void) })
    }
}
/// `MulPowStage` (gpu/src/stage.rs:334-390): dst[i] *= rhs[(i + shift) mod n]^power
pub struct MulPowStage<LhsF, RhsF = LhsF> { n: usize, _m: PhantomData<(LhsF, RhsF)> }
impl<LhsF: GpuField, RhsF: GpuField> MulPowStage<LhsF, RhsF> {
    pub fn new(n: usize) -> Self { Self { n, _m: PhantomData } }
    pub fn encode(&self, dst: &mut DeviceVec<LhsF>, rhs: &DeviceVec<RhsF>, power: usize, shift: isize) {
        sys::check(unsafe { sys::ms_mul_pow(get_planner().ctx(), field_id::<LhsF>(), field_id::<RhsF>(), self.n, dst.device_ptr(), dst.device_ptr() as *const c_void, rhs.device_ptr() as *const c_void, power as u32, shift as _) })
    }
}
/// `ConvertIntoStage` (gpu/src/stage.rs:581-634): dst[i] = LhsF::from(src[i])
pub struct ConvertIntoStage<LhsF, RhsF = LhsF> { n: usize, _m: PhantomData<(LhsF, RhsF)> }
impl<LhsF: GpuField, RhsF: GpuField> ConvertIntoStage<LhsF, RhsF> {
    pub fn new(n: usize) -> Self { Self { n, _m: PhantomData } }
    pub fn encode(&self, dst: &mut DeviceVec<LhsF>, src: &DeviceVec<RhsF>) {
        sys::check(unsafe { sys::ms_convert(get_planner().ctx(), field_id::<LhsF>(), field_id::<RhsF>(), self.n, dst.device_ptr(), src.device_ptr() as *const c_void) })
    }
}
/// `FillBuffStage` (gpu/src/stage.rs:1111-1155): dst[i] = value
pub struct FillBuffStage<F> { n: usize, _m: PhantomData<F> }
impl<F: GpuField> FillBuffStage<F> {
    pub fn new(n: usize) -> Self { Self { n, _m: PhantomData } }
    pub fn encode(&self, dst: &mut DeviceVec<F>, value: F) {
        sys::check(unsafe { sys::ms_fill(get_planner().ctx(), field_id::<F>(), self.n, dst.device_ptr(), &value as *const F as *const c_void) })
    }
}
/// `InverseInPlaceStage` (gpu/src/stage.rs:808-852)
pub struct InverseInPlaceStage<F> { n: usize, _m: PhantomData<F> }
impl<F: GpuField> InverseInPlaceStage<F> {
    pub fn new(n: usize) -> Self { Self { n, _m: PhantomData } }
    pub fn encode(&self, dst: &mut DeviceVec<F>) {
        sys::check(unsafe { sys::ms_unary(get_planner().ctx(), sys::MS_INV, field_id::<F>(), self.n, dst.device_ptr(), dst.device_ptr() as *const c_void, 0) })
    }
}
/// `NegInPlaceStage` (gpu/src/stage.rs:855-896)
pub struct NegInPlaceStage<F> { n: usize, _m: PhantomData<F> }
impl<F: GpuField> NegInPlaceStage<F> {
    pub fn new(n: usize) -> Self { Self { n, _m: PhantomData } }
    pub fn encode(&self, dst: &mut DeviceVec<F>) {
        sys::check(unsafe { sys::ms_unary(get_planner().ctx(), sys::MS_NEG, field_id::<F>(), self.n, dst.device_ptr(), dst.device_ptr() as *const c_void, 0) })
    }
}
/// `NegIntoStage` (gpu/src/stage.rs:899-946)
pub struct NegIntoStage<F> { n: usize, _m: PhantomData<F> }
impl<F: GpuField> NegIntoStage<F> {
    pub fn new(n: usize) -> Self { Self { n, _m: PhantomData } }
    pub fn encode(&self, dst: &mut DeviceVec<F>, src: &DeviceVec<F>) {
        sys::check(unsafe { sys::ms_unary(get_planner().ctx(), sys::MS_NEG, field_id::<F>(), self.n, dst.device_ptr(), src.device_ptr() as *const c_void, 0) })
    }
}
/// `InverseIntoStage` (gpu/src/stage.rs:949-996)
pub struct InverseIntoStage<F> { n: usize, _m: PhantomData<F> }
impl<F: GpuField> InverseIntoStage<F> {
    pub fn new(n: usize) -> Self { Self { n, _m: PhantomData } }
    pub fn encode(&self, dst: &mut DeviceVec<F>, src: &DeviceVec<F>) {
        sys::check(unsafe { sys::ms_unary(get_planner().ctx(), sys::MS_INV, field_id::<F>(), self.n, dst.device_ptr(), src.device_ptr() as *const c_void, 0) })
    }
}
/// `ExpIntoStage` (gpu/src/stage.rs:999-1053)
pub struct ExpIntoStage<F> { n: usize, _m: PhantomData<F> }
impl<F: GpuField> ExpIntoStage<F> {
    pub fn new(n: usize) -> Self { Self { n, _m: PhantomData } }
    pub fn encode(&self, dst: &mut DeviceVec<F>, src: &DeviceVec<F>, exponent: usize) {
        sys::check(unsafe { sys::ms_unary(get_planner().ctx(), sys::MS_EXP, field_id::<F>(), self.n, dst.device_ptr(), src.device_ptr() as *const c_void, exponent as u32) })
    }
}
/// `ExpInPlaceStage` (gpu/src/stage.rs:1056-1108)
pub struct ExpInPlaceStage<F> { n: usize, _m: PhantomData<F> }
impl<F: GpuField> ExpInPlaceStage<F> {
    pub fn new(n: usize) -> Self { Self { n, _m: PhantomData } }
    pub fn encode(&self, dst: &mut DeviceVec<F>, exponent: usize) {
        sys::check(unsafe { sys::ms_unary(get_planner().ctx(), sys::MS_EXP, field_id::<F>(), self.n, dst.device_ptr(), dst.device_ptr() as *const c_void, exponent as u32) })
    }
}
// FftGpuStage / ScaleAndNormalizeGpuStage / BitReverseGpuStage / GenerateTwiddlesStage (stage.rs:37-112, 236-332,
// 1157-1210) have no counterpart here: a transform is one plan (`GpuFft` / `GpuIfft`, plan.rs) whose passes fuse the
// scaling, the twiddles come from plan-owned tables, and `utils::bit_reverse` is one call.
// Rpo256*Stage (stage.rs:1212-1500): see `GpuRpo256ColumnMajor`, `GpuRpo256RowMajor`, `gen_rpo_merkle_tree` in plan.rs.
