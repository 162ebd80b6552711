"""ministark_amd -- MI355X (gfx950) backend for miniSTARK's gpu-poly hot path.

Product = ministark_amd/libministark_hip.so (hand-written HIP kernels behind the C ABI
in include/ministark_hip.h).  This package is the thin host mirror of the reference's
`ministark-gpu` interface on top of it.  No CPU fallback exists.
"""
from .api import (GOLDILOCKS_FP, GOLDILOCKS_FQ3, STARK252_FP, GL_GENERATOR, GL_P, ColumnSet, GpuFft, GpuIfft,  # noqa: F401
                  GpuVec, Matrix, MerkleTree, DeviceBytes, Planner, Radix2EvaluationDomain, get_planner, gl_from_mont, gl_to_mont, apply_drp,
                  F252_P, F252_GENERATOR, f252_to_mont_limbs, f252_from_mont_limbs,
                  GpuRpo256ColumnMajor, GpuRpo256RowMajor, gen_rpo_merkle_tree, grind_proof_of_work,
                  scan_affine, running_product, Queries)
