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
#if !defined(__HIPCC_RTC__)      // hiprtc pre-includes the runtime declarations
#include <hip/hip_runtime.h>
#endif
#include "gl.h"
#include "gl_dev.h"
#include "stage_kernels.h"

namespace mseval {

static constexpr int NT = 256;
static constexpr int MAXCOLS = 96;        // base + extension columns
static constexpr int MAXPERIODIC = 64;    // caller periodic columns + hoisted short-period tables (eval_opt.h)

enum Op : uint32_t {
    OP_X_P = 0, OP_CONST_P, OP_CONST_Q, OP_TRACE_P, OP_TRACE_Q, OP_PERIODIC_P, OP_PERIODIC_Q,
    OP_NEG_P, OP_NEG_Q, OP_ADD_PP, OP_ADD_QQ, OP_ADD_QP, OP_MUL_PP, OP_MUL_QQ, OP_MUL_QP,
    OP_INV_P, OP_INV_Q, OP_POW_P, OP_POW_Q, OP_EMBED, OP_STORE_Q, OP_STORE_P,
    OP_XPOW_P,      // internal (eval_opt.h): dst = consts[a] * w_n^(b*i mod n)  ==  x^b with consts[a] = h^b
    OP_TABLE_P, OP_TABLE_Q,   // internal (eval_opt.h, split_inversions): dst = table a of `periodic` at this launch position (P: b = row offset, eval_shift.h)
    // internal (eval_regroup.h): sums of products accumulated UNREDUCED, one Montgomery reduction per sum.  dst of the first three = accumulator
    OP_ACC_ZERO,              // acc[dst] = 0
    OP_ACC_MACC,              // acc[dst] += P[a] * C, C a wave-uniform constant whose limbs / digits sit at consts[b ..) (host-prepared)
    OP_ACC_MACP,              // acc[dst] += P[a] * P[b]
    OP_ACC_RED,               // P[dst] = (acc[a]) * R^-1 mod p, canonical
    // the same over the cubic extension: ONE accumulator of three column sets (c0, c1, c2 of Fp[x] / (x^3 - 2)); dst of MACC / MACP = mode
    OP_ACCQ_ZERO,
    OP_ACCQ_MACC,             // mode bit 0: the register is Q[a] (else P[a]); bit 1: the constant at consts[b ..) is an Fq3 (limbs of C0, C1, C2, 2 C1, 2 C2) else Fp
    OP_ACCQ_MACP,             // mode 0: c0 += P[a] * P[b];  mode 1: c_i += Q[a]_i * P[b]
    OP_ACCQ_RED,              // Q[dst] = the three reduced components
    OP_COUNT
};
struct Instr { uint32_t op, dst, a, b; };

struct EvalParams {
    const Instr* prog;            // device
    const uint64_t* consts;       // device, Montgomery words
    const uint64_t* base_cols[MAXCOLS];
    const uint64_t* ext_cols[MAXCOLS];
    const uint64_t* periodic[MAXPERIODIC];
    uint32_t periodic_len[MAXPERIODIC];
    uint64_t* out;                // n x Fq3 (or n x Fp when the program ends in STORE_P); STORE with b > 0 writes table b-1 of `periodic`
    const uint64_t* x_lde;        // optional: x values (Fp); nullptr -> generated as h*w^i
    const uint64_t* tw_lo;        // w_n^i two-level tables of the size-n forward plan (Montgomery form)
    const uint64_t* tw_hi;
    uint64_t h_mont;              // domain offset
    size_t n;
    uint32_t ninstr, lo_bits, lde_step, log_n;
    uint32_t xshift;              // the w table belongs to a domain of 2^(log_n + xshift) points
    uint32_t bitrev;              // columns and output are in bit-reversed order over the 2^log_n points: position R holds point bitrev(R)
};


// ---- the operations of the program, shared by the interpreter below and by the specialised kernels
// eval_jit.h generates (one call per instruction, registers as named locals)
// position R of the launch -> index i of the evaluation point x_i = h * w^i
__device__ __forceinline__ size_t ev_point(const EvalParams& P, size_t R) {
    return (P.bitrev && P.log_n) ? (size_t)(__brevll((unsigned long long)R) >> (64 - P.log_n)) : R;
}
// where row (i + lde_step*offset) mod n of a column lives.  In bit-reversed storage consecutive positions differ in
// the HIGH bits of i, so a wave's 64 rotated rows are again 64 consecutive positions (unless a carry runs that far).
__device__ __forceinline__ size_t ev_row(const EvalParams& P, size_t i, uint32_t off) {
    const size_t j = (i + (size_t)((long long)(int32_t)off * (long long)P.lde_step)) & (P.n - 1);       // n is a power of two
    return (P.bitrev && P.log_n) ? (size_t)(__brevll((unsigned long long)j) >> (64 - P.log_n)) : j;
}
__device__ __forceinline__ uint64_t ev_wpow(const EvalParams& P, size_t e) {                // w^e from the two-level table
    uint64_t x = P.tw_lo[e & ((1u << P.lo_bits) - 1)];
    if (e >> P.lo_bits) x = gld::mmul(x, P.tw_hi[e >> P.lo_bits]);
    return x;
}
__device__ __forceinline__ uint64_t ev_x(const EvalParams& P, size_t i) {
    if (P.x_lde) return P.x_lde[i];
    return gld::mmul(ev_wpow(P, i << P.xshift), P.h_mont);
}
__device__ __forceinline__ uint64_t ev_xpow(const EvalParams& P, size_t i, uint32_t cslot, uint32_t e) {
    return gld::mmul(ev_wpow(P, (((size_t)e * i) & (P.n - 1)) << P.xshift), P.consts[cslot]);
}
__device__ __forceinline__ uint64_t ev_trace_p(const EvalParams& P, size_t i, uint32_t col, uint32_t off) { return P.base_cols[col][ev_row(P, i, off)]; }
__device__ __forceinline__ gl::Fq3 ev_trace_q(const EvalParams& P, size_t i, uint32_t col, uint32_t off) {
    const uint64_t* c = P.ext_cols[col] + 3 * ev_row(P, i, off);
    return {c[0], c[1], c[2]};
}
__device__ __forceinline__ uint64_t ev_periodic_p(const EvalParams& P, size_t i, uint32_t id) { return P.periodic[id][i % P.periodic_len[id]]; }
__device__ __forceinline__ gl::Fq3 ev_periodic_q(const EvalParams& P, size_t i, uint32_t id) {
    const uint64_t* c = P.periodic[id] + 3 * (i % P.periodic_len[id]);
    return {c[0], c[1], c[2]};
}
// full-length tables (one entry per launch position R, whatever the layout): the batch-inverted denominators
// off != 0 (eval_shift.h): the entry of the point `off` trace rows further on, wherever the layout keeps it
__device__ __forceinline__ uint64_t ev_table_p(const EvalParams& P, size_t R, size_t i, uint32_t id, uint32_t off) { return P.periodic[id][off ? ev_row(P, i, off) : R]; }
__device__ __forceinline__ gl::Fq3 ev_table_q(const EvalParams& P, size_t R, uint32_t id) {
    const uint64_t* c = P.periodic[id] + 3 * R;
    return {c[0], c[1], c[2]};
}
__device__ __forceinline__ gl::Fq3 ev_const_q(const EvalParams& P, uint32_t a) { return {P.consts[a], P.consts[a + 1], P.consts[a + 2]}; }
__device__ __forceinline__ void ev_store_p(const EvalParams& P, size_t i, uint32_t slot, uint64_t v) { (slot ? (uint64_t*)P.periodic[slot - 1] : P.out)[i] = v; }
__device__ __forceinline__ void ev_store_q(const EvalParams& P, size_t i, uint32_t slot, const gl::Fq3& v) {
    uint64_t* o = (slot ? (uint64_t*)P.periodic[slot - 1] : P.out) + 3 * i;
    o[0] = v.c0; o[1] = v.c1; o[2] = v.c2;
}
// Fp252 (Fq = Fp): elements are 4 words; h_mont is the word index of the domain offset in consts
__device__ __forceinline__ f252::E ev252_load(const uint64_t* p, size_t i) { return msstage::Fp252T::load(p, i); }
__device__ __forceinline__ f252::E ev252_wpow(const EvalParams& P, size_t e) {
    f252::E x = ev252_load(P.tw_lo, e & ((1u << P.lo_bits) - 1));
    if (e >> P.lo_bits) x = f252::mul(x, ev252_load(P.tw_hi, e >> P.lo_bits));
    return x;
}
__device__ __forceinline__ f252::E ev252_x(const EvalParams& P, size_t i) {
    if (P.x_lde) return ev252_load(P.x_lde, i);
    return f252::mul(ev252_wpow(P, i << P.xshift), ev252_load(P.consts + P.h_mont, 0));
}
__device__ __forceinline__ f252::E ev252_xpow(const EvalParams& P, size_t i, uint32_t cslot, uint32_t e) {
    return f252::mul(ev252_wpow(P, (((size_t)e * i) & (P.n - 1)) << P.xshift), ev252_load(P.consts + cslot, 0));
}
__device__ __forceinline__ f252::E ev252_const(const EvalParams& P, uint32_t a) { return f252::E{{P.consts[a], P.consts[a + 1], P.consts[a + 2], P.consts[a + 3]}}; }
__device__ __forceinline__ f252::E ev252_trace(const EvalParams& P, size_t i, uint32_t col, uint32_t off) { return ev252_load(P.base_cols[col], ev_row(P, i, off)); }
__device__ __forceinline__ f252::E ev252_periodic(const EvalParams& P, size_t i, uint32_t id) { return ev252_load(P.periodic[id], i % P.periodic_len[id]); }
__device__ __forceinline__ f252::E ev252_table(const EvalParams& P, size_t R, size_t i, uint32_t id, uint32_t off) { return ev252_load(P.periodic[id], off ? ev_row(P, i, off) : R); }
// The launch position of a lane of the 252-bit kernels (blocks of 256): the wave's 64 positions in gld::wave_elem order, so that the
// result leaves in whole contiguous stores (Fp252T::store_wave); domains below one wave keep the plain order and the plain store.
__device__ __forceinline__ size_t ev252_pos(const EvalParams& P) {
    return (size_t)blockIdx.x * 256 + (P.n >= 64 ? gld::wave_elem(threadIdx.x) : threadIdx.x);
}
__device__ __forceinline__ void ev252_store(const EvalParams& P, size_t R, uint32_t slot, const f252::E& v) {
    uint64_t* dst = slot ? (uint64_t*)P.periodic[slot - 1] : P.out;
    if (P.n >= 64) msstage::Fp252T::store_wave(dst, R, v);
    else msstage::Fp252T::store(dst, R, v);
}

// ---- rows of a column (or full-length table) that a workgroup of the specialised 252-bit kernel reads at SEVERAL row offsets, staged once in
// LDS (natural layout only).  Why: the kernel holds ~200 registers, two waves per SIMD, and reads column c at row offset 1 long before it reads
// it at offset 0; by then the lines have left the L2 (32 CUs x 8 waves x 2 KiB per load instruction = the whole 4 MiB between the two) and come
// from memory again: 4.7 GB fetched for 2.4 GB of columns on configs[3] (iii) (round 6, FETCH_SIZE).  The window of a workgroup is its 256
// launch positions plus the offsets' reach: elements (R0 + lo + s) mod n, s < W, at stg[4 s ..].
__device__ __forceinline__ void ev252_stage(uint64_t* stg, const uint64_t* g, size_t R0, int lo, unsigned W, size_t n) {
    for (unsigned s = threadIdx.x; s < W; s += 256) {
        const size_t j = (R0 + (size_t)(long long)((int)s + lo)) & (n - 1);
        const uint4* src = (const uint4*)(g + 4 * j);
        uint4* dst = (uint4*)(stg + 4 * (size_t)s);
        const uint4 a = src[0], b = src[1];
        dst[0] = a; dst[1] = b;
    }
}
__device__ __forceinline__ f252::E ev252_lds(const uint64_t* stg, unsigned s) {
    const uint4* p = (const uint4*)(stg + 4 * (size_t)s);
    const uint4 a = p[0], b = p[1];
    return f252::E{{(uint64_t)a.x | ((uint64_t)a.y << 32), (uint64_t)a.z | ((uint64_t)a.w << 32), (uint64_t)b.x | ((uint64_t)b.y << 32), (uint64_t)b.z | ((uint64_t)b.w << 32)}};
}

// ---- unreduced sums of products (eval_regroup.h) ---------------------------------------------------------------------------
// Goldilocks: the first factor is cut at 32 bits, the second into 22 / 22 / 20-bit limbs; the six partial products (54 bits) go to six
// 64-bit columns of weights 2^0, 2^22, 2^44, 2^32, 2^54, 2^76 -- six multiply-adds per term and no carry anywhere (1 024 terms fit);
// the columns are put together once (< 2^141) and reduced once: value * 2^-64 mod p, what the sum of the Montgomery products is.
static constexpr int ACC_MAX_TERMS_GL = 512, ACC_MAX_TERMS_252 = 16, NACC = 2;
struct Acc6 { uint64_t s[6]; };
__device__ __forceinline__ void acc_zero(Acc6& A) {
    #pragma unroll
    for (int k = 0; k < 6; k++) A.s[k] = 0;
}
__device__ __forceinline__ void acc_mac_limbs(Acc6& A, uint64_t c, uint32_t y0, uint32_t y1, uint32_t y2) {
    const uint32_t c0 = (uint32_t)c, c1 = (uint32_t)(c >> 32);
    A.s[0] += (uint64_t)c0 * y0; A.s[1] += (uint64_t)c0 * y1; A.s[2] += (uint64_t)c0 * y2;
    A.s[3] += (uint64_t)c1 * y0; A.s[4] += (uint64_t)c1 * y1; A.s[5] += (uint64_t)c1 * y2;
}
// a word of the constant pool at a wave-uniform slot: read through the constant address space, so that it is a scalar load wherever the
// compiler places it (the pool is written by the host before the launch and by nobody during it)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint64_t ev_cword(const uint64_t* consts, uint32_t slot) { return ((const __attribute__((address_space(4))) uint64_t*)consts)[slot]; }
#else
__device__ __forceinline__ uint64_t ev_cword(const uint64_t* consts, uint32_t slot) { return consts[slot]; }
#endif
__device__ __forceinline__ void acc_macc(Acc6& A, uint64_t v, const uint64_t* consts, uint32_t slot) {     // limbs: y0 | y1 << 32, y2
    const uint64_t w0 = ev_cword(consts, slot), w1 = ev_cword(consts, slot + 1);
    acc_mac_limbs(A, v, (uint32_t)w0, (uint32_t)(w0 >> 32), (uint32_t)w1);
}
__device__ __forceinline__ void acc_macp(Acc6& A, uint64_t v, uint64_t b) {
    acc_mac_limbs(A, v, (uint32_t)b & 0x3FFFFFu, (uint32_t)(b >> 22) & 0x3FFFFFu, (uint32_t)(b >> 44));
}
// (top 2^128 + hi 2^64 + lo) 2^-64 mod p, canonical: the Montgomery reduction of the low 128 bits (felt_u64.h.metal:165-177 as in
// gl::mont_mul) with hi brought below p first, plus top 2^64 = top (2^32 - 1)
__device__ __forceinline__ uint64_t acc_reduce_words(uint64_t lo, uint64_t hi, uint32_t top) {
    const uint64_t xl = lo, xh = hi >= gl::P ? hi - gl::P : hi;
    const uint64_t s = xl + (xl << 32);
    const uint64_t ov = s < xl;
    const uint64_t bb = s - (s >> 32) - ov;
    const uint64_t r = xh - bb;
    return gl::add((xh < bb) ? r + gl::P : r, (uint64_t)top * 0xFFFFFFFFull);
}
__device__ __forceinline__ uint64_t acc_reduce(const Acc6& A) {
    typedef unsigned __int128 u128;
    const uint64_t* S = A.s;
    const u128 low = (u128)S[0] + ((u128)S[1] << 22) + ((u128)S[2] << 44) + ((u128)S[3] << 32) + ((u128)S[4] << 54);   // < 2^119
    const u128 top = (u128)(S[5] & ((1ull << 52) - 1)) << 76;
    const u128 sum = low + top;
    const uint32_t over = (uint32_t)(S[5] >> 52) + (sum < top ? 1u : 0u);
    return acc_reduce_words((uint64_t)sum, (uint64_t)(sum >> 64), over);
}
// Fq3 = Fp[x] / (x^3 - 2): (a0 + a1 x + a2 x^2)(b0 + b1 x + b2 x^2) = (a0 b0 + 2 a1 b2 + 2 a2 b1) + (a0 b1 + a1 b0 + 2 a2 b2) x + (a0 b2 + a1 b1 + a2 b0) x^2
// (the reference's product, felt_u64.h.metal:205-231, before its Karatsuba regrouping).  Up to three partial products per column and term:
// 256 terms fit.
static constexpr int ACC_MAX_TERMS_Q = 256;
struct AccQ { Acc6 c[3]; };
__device__ __forceinline__ void acc_zero(AccQ& A) { acc_zero(A.c[0]); acc_zero(A.c[1]); acc_zero(A.c[2]); }
__device__ __forceinline__ void accq_macc_p_cp(AccQ& A, uint64_t t, const uint64_t* consts, uint32_t s) { acc_macc(A.c[0], t, consts, s); }
__device__ __forceinline__ void accq_macc_q_cp(AccQ& A, const gl::Fq3& t, const uint64_t* consts, uint32_t s) {
    acc_macc(A.c[0], t.c0, consts, s); acc_macc(A.c[1], t.c1, consts, s); acc_macc(A.c[2], t.c2, consts, s);
}
__device__ __forceinline__ void accq_macc_p_cq(AccQ& A, uint64_t t, const uint64_t* consts, uint32_t s) {
    acc_macc(A.c[0], t, consts, s); acc_macc(A.c[1], t, consts, s + 2); acc_macc(A.c[2], t, consts, s + 4);
}
__device__ __forceinline__ void accq_macc_q_cq(AccQ& A, const gl::Fq3& t, const uint64_t* consts, uint32_t s) {      // slots: C0 +0, C1 +2, C2 +4, 2 C1 +6, 2 C2 +8
    acc_macc(A.c[0], t.c0, consts, s);     acc_macc(A.c[0], t.c1, consts, s + 8); acc_macc(A.c[0], t.c2, consts, s + 6);
    acc_macc(A.c[1], t.c0, consts, s + 2); acc_macc(A.c[1], t.c1, consts, s);     acc_macc(A.c[1], t.c2, consts, s + 8);
    acc_macc(A.c[2], t.c0, consts, s + 4); acc_macc(A.c[2], t.c1, consts, s + 2); acc_macc(A.c[2], t.c2, consts, s);
}
__device__ __forceinline__ void accq_macp_p_p(AccQ& A, uint64_t a, uint64_t b) { acc_macp(A.c[0], a, b); }
__device__ __forceinline__ void accq_macp_q_p(AccQ& A, const gl::Fq3& a, uint64_t b) {
    const uint32_t y0 = (uint32_t)b & 0x3FFFFFu, y1 = (uint32_t)(b >> 22) & 0x3FFFFFu, y2 = (uint32_t)(b >> 44);
    acc_mac_limbs(A.c[0], a.c0, y0, y1, y2); acc_mac_limbs(A.c[1], a.c1, y0, y1, y2); acc_mac_limbs(A.c[2], a.c2, y0, y1, y2);
}
__device__ __forceinline__ gl::Fq3 accq_reduce(const AccQ& A) { return {acc_reduce(A.c[0]), acc_reduce(A.c[1]), acc_reduce(A.c[2])}; }
// The 252-bit field: the nineteen digit columns of f252::mul_t take up to sixteen products before its reduction runs (fp252.h)
struct Acc19 { uint64_t c[19]; };
__device__ __forceinline__ void acc_zero(Acc19& A) {
    #pragma unroll
    for (int k = 0; k < 19; k++) A.c[k] = 0;
}
__device__ __forceinline__ void acc_macc(Acc19& A, const f252::E& v, const uint64_t* consts, uint32_t slot) {   // digits: five words, two per word
    uint32_t x[9], y[9];
    f252::digits9(v, x);
    #pragma unroll
    for (int k = 0; k < 4; k++) { const uint64_t w = consts[slot + k]; y[2 * k] = (uint32_t)w; y[2 * k + 1] = (uint32_t)(w >> 32); }
    y[8] = (uint32_t)consts[slot + 4];
    f252::mac81(A.c, x, y);
}
__device__ __forceinline__ void acc_macp(Acc19& A, const f252::E& v, const f252::E& b) {
    uint32_t x[9], y[9];
    f252::digits9(v, x);
    f252::digits9(b, y);
    f252::mac81(A.c, x, y);
}
__device__ __forceinline__ f252::E acc_reduce(Acc19& A) { return f252::reduce_columns<true>(A.c); }

template <int NP, int NQ>
__global__ void __launch_bounds__(NT) eval_program(EvalParams P) {
    using F3 = msstage::Fq3T;
    using F1 = msstage::FpT;
    const size_t R = (size_t)blockIdx.x * NT + threadIdx.x;
    if (R >= P.n) return;
    const size_t i = ev_point(P, R);
    uint64_t rp[NP];
    gl::Fq3 rq[NQ];
    Acc6 acc[NACC];
    AccQ accq;
    for (uint32_t pc = 0; pc < P.ninstr; pc++) {
        const Instr I = P.prog[pc];
        switch (I.op) {
        case OP_X_P: rp[I.dst] = ev_x(P, i); break;
        case OP_CONST_P: rp[I.dst] = P.consts[I.a]; break;
        case OP_CONST_Q: rq[I.dst] = ev_const_q(P, I.a); break;
        case OP_TRACE_P: rp[I.dst] = ev_trace_p(P, i, I.a, I.b); break;
        case OP_TRACE_Q: rq[I.dst] = ev_trace_q(P, i, I.a, I.b); break;
        case OP_PERIODIC_P: rp[I.dst] = ev_periodic_p(P, i, I.a); break;
        case OP_PERIODIC_Q: rq[I.dst] = ev_periodic_q(P, i, I.a); break;
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
        case OP_STORE_Q: ev_store_q(P, R, I.b, rq[I.a]); break;
        case OP_STORE_P: ev_store_p(P, R, I.b, rp[I.a]); break;
        case OP_XPOW_P: rp[I.dst] = ev_xpow(P, i, I.a, I.b); break;
        case OP_TABLE_P: rp[I.dst] = ev_table_p(P, R, i, I.a, I.b); break;
        case OP_TABLE_Q: rq[I.dst] = ev_table_q(P, R, I.a); break;
        case OP_ACC_ZERO: acc_zero(acc[I.dst & (NACC - 1)]); break;
        case OP_ACC_MACC: acc_macc(acc[I.dst & (NACC - 1)], rp[I.a], P.consts, I.b); break;
        case OP_ACC_MACP: acc_macp(acc[I.dst & (NACC - 1)], rp[I.a], rp[I.b]); break;
        case OP_ACC_RED: rp[I.dst] = acc_reduce(acc[I.a & (NACC - 1)]); break;
        case OP_ACCQ_ZERO: acc_zero(accq); break;
        case OP_ACCQ_MACC:
            if ((I.dst & 3) == 0) accq_macc_p_cp(accq, rp[I.a], P.consts, I.b);
            else if ((I.dst & 3) == 1) accq_macc_q_cp(accq, rq[I.a], P.consts, I.b);
            else if ((I.dst & 3) == 2) accq_macc_p_cq(accq, rp[I.a], P.consts, I.b);
            else accq_macc_q_cq(accq, rq[I.a], P.consts, I.b);
            break;
        case OP_ACCQ_MACP: if (I.dst & 1) accq_macp_q_p(accq, rq[I.a], rp[I.b]); else accq_macp_p_p(accq, rp[I.a], rp[I.b]); break;
        case OP_ACCQ_RED: rq[I.dst] = accq_reduce(accq); break;
        default: break;
        }
    }
}

// Fp252 instantiation (Fq = Fp = the 252-bit field, src/eval_gpu.rs:1054-1082): only the P-typed
// opcodes are legal; registers and constants are 4-limb elements (`a` of CONST_P indexes u64 words).
template <int NP>
__global__ void __launch_bounds__(NT) eval_program252(EvalParams P) {
    using F = msstage::Fp252T;
    static_assert(NT == 256, "ev252_pos");
    const size_t R = ev252_pos(P);
    if (R >= P.n) return;
    const size_t i = ev_point(P, R);
    f252::E rp[NP];
    Acc19 acc[NACC];
    for (uint32_t pc = 0; pc < P.ninstr; pc++) {
        const Instr I = P.prog[pc];
        switch (I.op) {
        case OP_X_P: rp[I.dst] = ev252_x(P, i); break;
        case OP_CONST_P: rp[I.dst] = ev252_const(P, I.a); break;
        case OP_TRACE_P: rp[I.dst] = ev252_trace(P, i, I.a, I.b); break;
        case OP_PERIODIC_P: rp[I.dst] = ev252_periodic(P, i, I.a); break;
        case OP_NEG_P: rp[I.dst] = f252::neg(rp[I.a]); break;
        case OP_ADD_PP: rp[I.dst] = f252::add(rp[I.a], rp[I.b]); break;
        case OP_MUL_PP: rp[I.dst] = f252::mul(rp[I.a], rp[I.b]); break;
        case OP_INV_P: rp[I.dst] = f252::inv(rp[I.a]); break;
        case OP_POW_P: rp[I.dst] = msstage::powu<F>(rp[I.a], I.b); break;
        case OP_STORE_P: ev252_store(P, R, I.b, rp[I.a]); break;
        case OP_XPOW_P: rp[I.dst] = ev252_xpow(P, i, I.a, I.b); break;
        case OP_TABLE_P: rp[I.dst] = ev252_table(P, R, i, I.a, I.b); break;
        case OP_ACC_ZERO: acc_zero(acc[I.dst & (NACC - 1)]); break;
        case OP_ACC_MACC: acc_macc(acc[I.dst & (NACC - 1)], rp[I.a], P.consts, I.b); break;
        case OP_ACC_MACP: acc_macp(acc[I.dst & (NACC - 1)], rp[I.a], rp[I.b]); break;
        case OP_ACC_RED: rp[I.dst] = acc_reduce(acc[I.a & (NACC - 1)]); break;
        default: break;
        }
    }
}

// ---- batch inversion of a full-length table, in place: t[i] <- t[i]^-1, zeros stay zero (ark_ff::batch_inversion, which
// eval_cpu.rs:101-107 applies per 512-point chunk and division node).  Montgomery's trick over the K elements of a lane
// (elements tid + j * stride: coalesced): K - 1 products forward, ONE Fermat inverse, 2 (K - 1) products backward -- 3
// multiplications + 1/K of an inversion per element instead of 72 (Fp) / ~370 (Fp252) multiplications.  The inverse of a
// field element is unique and every product is canonical, so the table holds exactly what the per-point inverse produces.
__device__ __forceinline__ bool ev_is_zero(uint64_t v) { return v == 0; }
__device__ __forceinline__ bool ev_is_zero(const gl::Fq3& v) { return (v.c0 | v.c1 | v.c2) == 0; }
__device__ __forceinline__ bool ev_is_zero(const f252::E& v) { return (v.l[0] | v.l[1] | v.l[2] | v.l[3]) == 0; }
template <class F, int K>
__global__ void __launch_bounds__(NT) batch_inverse(uint64_t* t, size_t n) {
    using T = typename F::T;
    const size_t tid = (size_t)blockIdx.x * NT + threadIdx.x, stride = (size_t)gridDim.x * NT;
    T pre[K];
    T acc = F::one();
    #pragma unroll
    for (int j = 0; j < K; j++) {
        const size_t i = tid + (size_t)j * stride;
        pre[j] = acc;
        if (i < n) { const T v = F::load(t, i); if (!ev_is_zero(v)) acc = F::mul(acc, v); }
    }
    T inv = F::inv(acc);
    #pragma unroll
    for (int j = K - 1; j >= 0; j--) {
        const size_t i = tid + (size_t)j * stride;
        if (i < n) {
            const T v = F::load(t, i);
            if (!ev_is_zero(v)) { F::store(t, i, F::mul(inv, pre[j])); inv = F::mul(inv, v); }
        }
    }
}

// Two levels for fields whose inverse is very expensive (the 252-bit field: ~92 000 instructions): the products of the lanes'
// K elements go to a scratch array of n / K entries, that array is inverted by batch_inverse (one Fermat inverse per K * K
// elements), and a second sweep turns each lane's inverted product back into the K inverses.
// (252-bit field: the lanes of a wave take their 64 elements in gld::wave_elem order and store with store_wave -- whole contiguous stores;
// the launches cover n exactly, a multiple of 64 K)
template <class F, int K>
__global__ void __launch_bounds__(NT) batch_inverse_up(const uint64_t* t, size_t n, uint64_t* prod) {
    using T = typename F::T;
    const size_t tid = (size_t)blockIdx.x * NT + (F::V == 4 ? gld::wave_elem(threadIdx.x) : threadIdx.x), stride = (size_t)gridDim.x * NT;
    T acc = F::one();
    #pragma unroll
    for (int j = 0; j < K; j++) {
        const size_t i = tid + (size_t)j * stride;
        if (i < n) { const T v = F::load(t, i); if (!ev_is_zero(v)) acc = F::mul(acc, v); }
    }
    if constexpr (F::V == 4) F::store_wave(prod, tid, acc);
    else F::store(prod, tid, acc);                            // never zero
}
// compile-time loop: f(ic<0>) .. f(ic<N - 1>) (UP) or the reverse.  `#pragma unroll` is not honoured around bodies as large as a 252-bit
// product; the prefix products pre[] then live in scratch memory -- and scratch stores reach HBM: batch_inverse_down wrote 537 MB and
// read 797 MB for a 268 MB table (round 6, WRITE_SIZE / FETCH_SIZE).
template <int N> struct bi_ic { static constexpr int value = N; };
template <int N, bool UP, class Fn>
__device__ __forceinline__ void bi_static_for(Fn&& f) {
    if constexpr (N > 0) {
        if constexpr (UP) { bi_static_for<N - 1, UP>(f); f(bi_ic<N - 1>{}); }
        else { f(bi_ic<N - 1>{}); bi_static_for<N - 1, UP>(f); }
    }
}
template <class F, int K>
__global__ void __launch_bounds__(NT) batch_inverse_down(uint64_t* t, size_t n, const uint64_t* prod_inv) {
    using T = typename F::T;
    const size_t tid = (size_t)blockIdx.x * NT + (F::V == 4 ? gld::wave_elem(threadIdx.x) : threadIdx.x), stride = (size_t)gridDim.x * NT;
    T pre[K];
    T acc = F::one();
    bi_static_for<K, true>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const size_t i = tid + (size_t)j * stride;
        pre[j] = acc;
        if (i < n) { const T v = F::load(t, i); if (!ev_is_zero(v)) acc = F::mul(acc, v); }
    });
    T inv = F::load(prod_inv, tid);
    bi_static_for<K, false>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const size_t i = tid + (size_t)j * stride;
        if (i < n) {
            const T v = F::load(t, i);
            if constexpr (F::V == 4) {
                // every lane stores (a zero entry stores the zero it read): the wave writes its 64 elements together (store_wave)
                const bool z = ev_is_zero(v);
                const T o = z ? v : F::mul(inv, pre[j]);
                F::store_wave(t, i, o);
                if (!z) inv = F::mul(inv, v);
            } else if (!ev_is_zero(v)) { F::store(t, i, F::mul(inv, pre[j])); inv = F::mul(inv, v); }
        }
    });
}

// ---- the denominators X + c of a Goldilocks program: table AND inversion in one pass (round 5) ---------------------------------------
// t[R] = (x_R + c)^-1 at every launch position R, 0^-1 = 0: what the denominators' program followed by k_batch_inverse leaves, without the
// table of x_R + c going to memory and back.  A lane owns the positions tid + j stride, stride = n / K; their points are point(tid) plus
// j stride (natural order) or plus rev(j) (bit-reversed storage: the top bits of a position are the low bits of its point), so x_j is x_0
// times one of K wave-uniform factors the host supplies (XcParams::m; exact products of canonical values: the words ev_x would give).
struct XcParams { uint64_t m[16]; uint64_t c; };
template <int K>
__global__ void __launch_bounds__(NT) batch_inverse_x_plus_c(EvalParams P, uint64_t* t, XcParams X) {
    static_assert(K <= 16, "XcParams holds 16 factors");
    using F = msstage::FpT;
    const size_t tid = (size_t)blockIdx.x * NT + threadIdx.x, stride = (size_t)gridDim.x * NT;      // stride * K == P.n
    const uint64_t x0 = ev_x(P, ev_point(P, tid));
    uint64_t val[K], pre[K];
    uint64_t acc = F::one();
    #pragma unroll
    for (int j = 0; j < K; j++) {
        pre[j] = acc;
        val[j] = gl::add(j ? gld::mmul(x0, X.m[j]) : x0, X.c);
        if (val[j] != 0) acc = gld::mmul(acc, val[j]);
    }
    uint64_t inv = F::inv(acc);
    #pragma unroll
    for (int j = K - 1; j >= 0; j--) {
        const size_t i = tid + (size_t)j * stride;
        if (val[j] == 0) t[i] = 0;
        else { t[i] = gld::mmul(inv, pre[j]); inv = gld::mmul(inv, val[j]); }
    }
}

}  // namespace mseval
