//! `hip` arm of crate `ministark-gpu`: the same public items as the Metal arm
//! (`gpu/src/plan.rs`, `gpu/src/stage.rs`, `gpu/src/utils.rs`), implemented on
//! `libministark_hip.so` (hand-written HIP kernels for gfx950 behind the C ABI of
//! `include/ministark_hip.h`).
//!
//! Wiring (a maintainer's diff):
//!   * `gpu/Cargo.toml`:  `[features] hip = []`
//!   * `gpu/build.rs`:    link `ministark_hip` from `$MINISTARK_HIP_LIB_DIR`
//!   * `gpu/src/lib.rs`:  `#[cfg(feature = "hip")] pub mod hip;`
//!   * `gpu/src/prelude.rs`: `#[cfg(feature = "hip")] pub use crate::hip::{get_planner, GpuFft, GpuIfft, ...};`
//!     next to the existing `#[cfg(all(target_arch = "aarch64", target_os = "macos"))]` re-exports
//!   (rust/patches/gpu_*.patch; the main crate's arms are rust/patches/src_*.patch).
//!
//! These files are source only: the build image has no Rust toolchain.  `sys.rs` is generated from the
//! header (scripts/gen_rust_sys.py) and checked against it by tests/test_rust_shim.py; the item-for-item
//! behaviour of the wrappers is what the C++ mirror (ministark_amd/csrc/host/*.hpp) and the Python mirror
//! (ministark_amd/*.py) implement and test.
#![cfg(feature = "hip")]

pub mod plan;
pub mod stage;
pub mod sys;
pub mod utils;

pub use plan::{evaluate_device, gen_rpo_merkle_tree, get_planner, lde_device, sha256_commit_device, GpuFft, GpuIfft, GpuRpo256ColumnMajor,
               GpuRpo256RowMajor, Planner};
pub use utils::{bit_reverse_device, field_id, DeviceVec};
