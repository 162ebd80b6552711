// Fused constraint evaluation for gfx950: replaces the reference's `eval_gpu::eval`
// (src/eval_gpu.rs:46-131: ONE full-array dispatch per expression-DAG node, after cloning
// every trace column) with ONE kernel that evaluates the whole composition-constraint
// program at each LDE point; results are bit-identical to the live CPU evaluator
// `eval_cpu::eval` (src/eval_cpu.rs:33-150), which is the parity oracle.
//
// The host lowers the `Expr<AlgebraicItem<FieldVariant<Fp,Fq>>>` DAG (src/expression.rs:33-40,
// src/constraints.rs:21-28) -- after common-subexpression sharing, as reuse_shared_nodes does
// (src/expression.rs:186-357) -- to a typed register program (see include/ministark_hip.h,
// "constraint program").  Typing follows eval_cpu.rs:306-428: a node is Fp iff both operands
// are, else Fq; x / y = x * y^-1 with 0^-1 = 0 (ark_ff::batch_inversion leaves zeros).
// One lane per LDE point; trace reads are `column[(i + lde_step*offset) mod n]`
// (eval_cpu.rs:115-134), coalesced across the wave; the program is wave-uniform so the
// interpreter's branches never diverge.  Registers live in the lane's private segment.
#pragma once
#include <hip/hip_runtime.h>
#include "gl.h"
#include "gl_dev.h"
#include "stage_kernels.h"

namespace mseval {

static constexpr int NT = 256;
static constexpr int MAXCOLS = 96;        // base + extension columns
static constexpr int MAXPERIODIC = 16;

enum Op : uint32_t {
    OP_X_P = 0, OP_CONST_P, OP_CONST_Q, OP_TRACE_P, OP_TRACE_Q, OP_PERIODIC_P, OP_PERIODIC_Q,
    OP_NEG_P, OP_NEG_Q, OP_ADD_PP, OP_ADD_QQ, OP_ADD_QP, OP_MUL_PP, OP_MUL_QQ, OP_MUL_QP,
    OP_INV_P, OP_INV_Q, OP_POW_P, OP_POW_Q, OP_EMBED, OP_STORE_Q, OP_STORE_P, OP_COUNT
};
struct Instr { uint32_t op, dst, a, b; };

struct EvalParams {
    const Instr* prog;            // device
    const uint64_t* consts;       // device, Montgomery words
    const uint64_t* base_cols[MAXCOLS];
    const uint64_t* ext_cols[MAXCOLS];
    const uint64_t* periodic[MAXPERIODIC];
    uint32_t periodic_len[MAXPERIODIC];
    uint64_t* out;                // n x Fq3 (or n x Fp when the program ends in STORE_P)
    const uint64_t* x_lde;        // optional: x values (Fp); nullptr -> generated as h*w^i
    const uint64_t* tw_lo;        // w_n^i two-level tables of the size-n forward plan (Montgomery form)
    const uint64_t* tw_hi;
    uint64_t h_mont;              // domain offset
    size_t n;
    uint32_t ninstr, lo_bits, lde_step, log_n;
    uint32_t xshift;              // the w table belongs to a domain of 2^(log_n + xshift) points
};

template <int NP, int NQ>
__global__ void __launch_bounds__(NT) eval_program(EvalParams P) {
    using F3 = msstage::Fq3T;
    using F1 = msstage::FpT;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= P.n) return;
    uint64_t rp[NP];
    gl::Fq3 rq[NQ];
    const size_t nmask = P.n - 1;
    for (uint32_t pc = 0; pc < P.ninstr; pc++) {
        const Instr I = P.prog[pc];
        switch (I.op) {
        case OP_X_P: {
            uint64_t x;
            if (P.x_lde) x = P.x_lde[i];
            else {
                const size_t e = i << P.xshift;
                x = P.tw_lo[e & ((1u << P.lo_bits) - 1)];
                if (e >> P.lo_bits) x = gld::mmul(x, P.tw_hi[e >> P.lo_bits]);
                x = gld::mmul(x, P.h_mont);
            }
            rp[I.dst] = x;
        } break;
        case OP_CONST_P: rp[I.dst] = P.consts[I.a]; break;
        case OP_CONST_Q: rq[I.dst] = {P.consts[I.a], P.consts[I.a + 1], P.consts[I.a + 2]}; break;
        case OP_TRACE_P: {
            const size_t j = (i + (size_t)((long long)(int32_t)I.b * (long long)P.lde_step)) & nmask;   // n is a power of two
            rp[I.dst] = P.base_cols[I.a][j];
        } break;
        case OP_TRACE_Q: {
            const size_t j = (i + (size_t)((long long)(int32_t)I.b * (long long)P.lde_step)) & nmask;
            const uint64_t* c = P.ext_cols[I.a] + 3 * j;
            rq[I.dst] = {c[0], c[1], c[2]};
        } break;
        case OP_PERIODIC_P: rp[I.dst] = P.periodic[I.a][i % P.periodic_len[I.a]]; break;
        case OP_PERIODIC_Q: { const uint64_t* c = P.periodic[I.a] + 3 * (i % P.periodic_len[I.a]); rq[I.dst] = {c[0], c[1], c[2]}; } break;
        case OP_NEG_P: rp[I.dst] = gl::neg(rp[I.a]); break;
        case OP_NEG_Q: rq[I.dst] = gl::neg(rq[I.a]); break;
        case OP_ADD_PP: rp[I.dst] = gl::add(rp[I.a], rp[I.b]); break;
        case OP_ADD_QQ: rq[I.dst] = gl::add(rq[I.a], rq[I.b]); break;
        case OP_ADD_QP: rq[I.dst] = msstage::Mix<F3, F1>::add(rq[I.a], rp[I.b]); break;
        case OP_MUL_PP: rp[I.dst] = gld::mmul(rp[I.a], rp[I.b]); break;
        case OP_MUL_QQ: rq[I.dst] = F3::mul(rq[I.a], rq[I.b]); break;
        case OP_MUL_QP: rq[I.dst] = msstage::Mix<F3, F1>::mul(rq[I.a], rp[I.b]); break;
        case OP_INV_P: rp[I.dst] = F1::inv(rp[I.a]); break;
        case OP_INV_Q: rq[I.dst] = F3::inv(rq[I.a]); break;
        case OP_POW_P: rp[I.dst] = msstage::powu<F1>(rp[I.a], I.b); break;
        case OP_POW_Q: rq[I.dst] = msstage::powu<F3>(rq[I.a], I.b); break;
        case OP_EMBED: rq[I.dst] = {rp[I.a], 0, 0}; break;
        case OP_STORE_Q: { uint64_t* o = P.out + 3 * i; const gl::Fq3 v = rq[I.a]; o[0] = v.c0; o[1] = v.c1; o[2] = v.c2; } break;
        case OP_STORE_P: P.out[i] = rp[I.a]; break;
        default: break;
        }
    }
}

// Fp252 instantiation (Fq = Fp = the 252-bit field, src/eval_gpu.rs:1054-1082): only the P-typed
// opcodes are legal; registers and constants are 4-limb elements (`a` of CONST_P indexes u64 words).
template <int NP>
__global__ void __launch_bounds__(NT) eval_program252(EvalParams P) {
    using F = msstage::Fp252T;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= P.n) return;
    f252::E rp[NP];
    const size_t nmask = P.n - 1;
    for (uint32_t pc = 0; pc < P.ninstr; pc++) {
        const Instr I = P.prog[pc];
        switch (I.op) {
        case OP_X_P: {
            f252::E x;
            if (P.x_lde) x = F::load(P.x_lde, i);
            else {
                const size_t e = i << P.xshift;
                x = F::load(P.tw_lo, e & ((1u << P.lo_bits) - 1));
                if (e >> P.lo_bits) x = f252::mul(x, F::load(P.tw_hi, e >> P.lo_bits));
                x = f252::mul(x, F::load(P.consts + P.h_mont, 0));      // h_mont = word index of the offset in consts
            }
            rp[I.dst] = x;
        } break;
        case OP_CONST_P: rp[I.dst] = f252::E{{P.consts[I.a], P.consts[I.a + 1], P.consts[I.a + 2], P.consts[I.a + 3]}}; break;
        case OP_TRACE_P: {
            const size_t j = (i + (size_t)((long long)(int32_t)I.b * (long long)P.lde_step)) & nmask;
            rp[I.dst] = F::load(P.base_cols[I.a], j);
        } break;
        case OP_PERIODIC_P: rp[I.dst] = F::load(P.periodic[I.a], i % P.periodic_len[I.a]); break;
        case OP_NEG_P: rp[I.dst] = f252::neg(rp[I.a]); break;
        case OP_ADD_PP: rp[I.dst] = f252::add(rp[I.a], rp[I.b]); break;
        case OP_MUL_PP: rp[I.dst] = f252::mul(rp[I.a], rp[I.b]); break;
        case OP_INV_P: rp[I.dst] = f252::inv(rp[I.a]); break;
        case OP_POW_P: rp[I.dst] = msstage::powu<F>(rp[I.a], I.b); break;
        case OP_STORE_P: F::store(P.out, i, rp[I.a]); break;
        default: break;
        }
    }
}

}  // namespace mseval
