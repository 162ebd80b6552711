//! `hip` arm of `DeepPolyComposer` (src/composer.rs:43-188): out-of-domain evaluations by Horner on the device
//! (`ms_horner_eval`: one block-parallel pass per query instead of one sequential walk per column) and the DEEP composition
//! polynomial by `ms_deep_compose` -- the quotients (P_c(X) - P_c(z_t)) / (X - z_t), their alpha-weighted sum and the degree
//! adjustment (alpha + beta X), computed through n coset evaluations instead of divide_out_point(s)_into per column
//! (src/utils.rs:137-175) + sum_columns + the sequential adjustment loop (src/composer.rs:168-185).  Same coefficients, bit for bit
//! (tests/test_deep_parity.py checks the C ABI against both formulations; tests/test_verifier_relations.py against the verifier's
//! own recomputation at the query positions).
//!
//! The polynomial matrices are host vectors in the reference (src/utils.rs:438-460), so both calls upload them; a prover that keeps
//! its columns in `DeviceVec`s calls `sys::ms_horner_eval` / `sys::ms_deep_compose` directly on the resident columns.
//! Source only (no Rust toolchain in the build image); called from src/composer.rs by rust/patches/src_composer.rs.patch.
#![cfg(feature = "hip")]

use crate::composer::DeepCompositionCoeffs;
use crate::utils::GpuAllocator;
use crate::utils::GpuVec;
use crate::Matrix;
use crate::StarkExtensionOf;
use alloc::vec::Vec;
use ark_ff::FftField;
use ark_ff::Field;
use ark_ff::Zero;
use core::ffi::c_void;
use ministark_gpu::hip::field_id;
use ministark_gpu::hip::get_planner;
use ministark_gpu::hip::sys;
use ministark_gpu::hip::DeviceVec;
use ministark_gpu::GpuFftField;

fn upload<F: Field>(m: &Matrix<F>) -> Vec<DeviceVec<F>> {
    m.iter().map(|column| DeviceVec::from_slice(column)).collect()
}

fn pointers<F>(columns: &[DeviceVec<F>]) -> Vec<*const c_void> {
    columns.iter().map(|c| c.device_ptr() as *const c_void).collect()
}

/// z * g^offset (g^-1 for negative offsets): src/composer.rs:62-64
fn shifted<Fp: Field, Fq: Field + core::ops::Mul<Fp, Output = Fq>>(z: Fq, g: Fp, g_inv: Fp, offset: isize) -> Fq {
    let generator = if offset >= 0 { g } else { g_inv };
    z * generator.pow([offset.unsigned_abs() as u64])
}

/// `(execution_trace_evals, composition_trace_evals)` of `DeepPolyComposer::get_ood_evals`.
#[allow(clippy::too_many_arguments)]
pub fn get_ood_evals<Fp: GpuFftField<FftField = Fp> + FftField, Fq: StarkExtensionOf<Fp>>(
    trace_arguments: &[(usize, isize)],
    num_base_columns: usize,
    z: Fq,
    g: Fp,
    g_inv: Fp,
    base_trace_polys: &Matrix<Fp>,
    extension_trace_polys: Option<&Matrix<Fq>>,
    composition_trace_polys: &Matrix<Fq>,
) -> (Vec<Fq>, Vec<Fq>) {
    let n = base_trace_polys.num_rows();
    let base = upload(base_trace_polys);
    let ext = extension_trace_polys.map_or_else(Vec::new, upload);
    let comp = upload(composition_trace_polys);
    // one launch per polynomial matrix: queries (column, point) in the order of the trace arguments
    let horner = |coeff_field, columns: &[*const c_void], queries: &[(u32, Fq)]| -> Vec<Fq> {
        let mut out = alloc::vec![Fq::zero(); queries.len()];
        if !queries.is_empty() {
            let qcol: Vec<u32> = queries.iter().map(|q| q.0).collect();
            let points: Vec<Fq> = queries.iter().map(|q| q.1).collect();
            sys::check(unsafe {
                sys::ms_horner_eval(
                    get_planner().ctx(),
                    coeff_field,
                    field_id::<Fq>(),
                    n,
                    columns.as_ptr(),
                    columns.len() as u32,
                    qcol.as_ptr(),
                    points.as_ptr() as *const c_void,
                    queries.len() as u32,
                    out.as_mut_ptr() as *mut c_void,
                )
            });
        }
        out
    };
    let mut base_queries = Vec::new();
    let mut ext_queries = Vec::new();
    for &(column, offset) in trace_arguments {
        let x = shifted(z, g, g_inv, offset);
        if column < num_base_columns {
            base_queries.push((column as u32, x));
        } else {
            ext_queries.push(((column - num_base_columns) as u32, x));
        }
    }
    let mut base_evals = horner(field_id::<Fp>(), &pointers(&base), &base_queries).into_iter();
    let mut ext_evals = horner(field_id::<Fq>(), &pointers(&ext), &ext_queries).into_iter();
    let execution_trace_evals = trace_arguments
        .iter()
        .map(|&(column, _)| if column < num_base_columns { base_evals.next().unwrap() } else { ext_evals.next().unwrap() })
        .collect();
    let z_n = z.pow([composition_trace_polys.num_cols() as u64]);
    let comp_queries: Vec<(u32, Fq)> = (0..comp.len()).map(|c| (c as u32, z_n)).collect();
    let composition_trace_evals = horner(field_id::<Fq>(), &pointers(&comp), &comp_queries);
    (execution_trace_evals, composition_trace_evals)
}

/// The single column of `DeepPolyComposer::into_deep_poly`.
#[allow(clippy::too_many_arguments)]
pub fn into_deep_poly<Fp: GpuFftField<FftField = Fp> + FftField, Fq: StarkExtensionOf<Fp>>(
    trace_arguments: &[(usize, isize)],
    num_base_columns: usize,
    z: Fq,
    g: Fp,
    g_inv: Fp,
    base_trace_polys: &Matrix<Fp>,
    extension_trace_polys: Option<&Matrix<Fq>>,
    composition_trace_polys: &Matrix<Fq>,
    composition_coeffs: DeepCompositionCoeffs<Fq>,
) -> GpuVec<Fq> {
    let (execution_trace_evals, composition_trace_evals) = get_ood_evals(
        trace_arguments,
        num_base_columns,
        z,
        g,
        g_inv,
        base_trace_polys,
        extension_trace_polys,
        composition_trace_polys,
    );
    let DeepCompositionCoeffs {
        execution_trace: execution_trace_alphas,
        composition_trace: composition_trace_alphas,
        degree: (degree_alpha, degree_beta),
    } = composition_coeffs;
    let n = base_trace_polys.num_rows();
    let base = upload(base_trace_polys);
    let ext = extension_trace_polys.map_or_else(Vec::new, upload);
    let comp = upload(composition_trace_polys);
    let num_extension_columns = ext.len();

    // terms (column, point, alpha, out-of-domain value); points are shared between terms
    let mut points: Vec<Fq> = Vec::new();
    let mut point_index = |p: Fq| -> u32 {
        if let Some(i) = points.iter().position(|q| *q == p) {
            return i as u32;
        }
        points.push(p);
        (points.len() - 1) as u32
    };
    let (mut term_col, mut term_point, mut term_alpha, mut term_ood) = (Vec::new(), Vec::new(), Vec::new(), Vec::new());
    let z_n = z.pow([composition_trace_polys.num_cols() as u64]);
    for (c, (alpha, ood)) in composition_trace_alphas.iter().zip(&composition_trace_evals).enumerate() {
        term_col.push((num_base_columns + num_extension_columns + c) as u32); // the composition columns follow the extension columns
        term_point.push(point_index(z_n));
        term_alpha.push(*alpha);
        term_ood.push(*ood);
    }
    for ((&(column, offset), alpha), ood) in trace_arguments.iter().zip(&execution_trace_alphas).zip(&execution_trace_evals) {
        term_col.push(column as u32);
        term_point.push(point_index(shifted(z, g, g_inv, offset)));
        term_alpha.push(*alpha);
        term_ood.push(*ood);
    }
    // Fq = Fp AIRs: every polynomial is a base column for the kernel; otherwise extension + composition columns are Fq columns
    let fq_is_ext = core::mem::size_of::<Fq>() != core::mem::size_of::<Fp>();
    let mut base_ptrs = pointers(&base);
    let mut ext_ptrs = pointers(&ext);
    if fq_is_ext {
        ext_ptrs.extend(pointers(&comp));
    } else {
        base_ptrs.extend(ext_ptrs.drain(..));
        base_ptrs.extend(pointers(&comp));
    }
    let out = DeviceVec::<Fq>::with_len(n);
    sys::check(unsafe {
        sys::ms_deep_compose(
            get_planner().ctx(),
            field_id::<Fq>(),
            n.trailing_zeros(),
            core::ptr::null(),
            base_ptrs.as_ptr(),
            base_ptrs.len() as u32,
            ext_ptrs.as_ptr(),
            ext_ptrs.len() as u32,
            points.as_ptr() as *const c_void,
            points.len() as u32,
            term_col.as_ptr(),
            term_point.as_ptr(),
            term_alpha.as_ptr() as *const c_void,
            term_ood.as_ptr() as *const c_void,
            term_col.len() as u32,
            &degree_alpha as *const Fq as *const c_void,
            &degree_beta as *const Fq as *const c_void,
            out.device_ptr(),
        )
    });
    let mut coeffs = Vec::with_capacity_in(n, GpuAllocator);
    coeffs.resize(n, Fq::zero());
    out.download(&mut coeffs);
    coeffs
}
