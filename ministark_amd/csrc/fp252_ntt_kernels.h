// Multi-pass NTT over the 252-bit StarkWare field (round 2): 2^11 <= n <= 2^30 in two or three passes of
// radix <= 1024, each on a 2048-element tile held in LDS (64 KiB, two workgroups per CU).  Replaces the
// bit-reverse + 9 LDS stages + two-stages-per-launch sequence of fp252_kernels.h for those sizes and adds what
// that sequence lacked: zero-extended input read straight from the short coefficient column, and the
// bit-reversed store fused into the last pass (gpu/src/plan.rs:378-462 with the fp252 instantiations of
// fft_shaders.h.metal:108-119; Matrix::into_evaluations / bit_reversed_evaluate, src/matrix.rs:193-251).
//
// The field is compute-bound (a Montgomery product is ~235 VALU instructions against 64 bytes moved per
// butterfly), so the design minimises instructions and launches, not bytes:
//   * decimation in time inside a tile, on LAZY residues: X' = X + T, Y' = X + (2p - T) with T = w Y < 2p straight
//     out of the product (no conditional subtraction anywhere); the bound grows by 2p per stage and a radix-1024
//     tile ends below 24p < 2^256.  One exact reduction per element at the very end of the transform;
//   * two stages per LDS round trip (4 elements per lane), first round without its three trivial products;
//   * n = R0 R1 (R2): pass q transforms digit q of the index in place (rows at stride S_q = n / (R_0..R_q), 2048 / R_q
//     consecutive elements per row), multiplies by the inter-pass factor w_n^(k_q l n/(R_q S_q)) and leaves lazy
//     values (< 2p) in memory.  The last pass owns whole rows of R consecutive elements for 2048 / R adjacent k_0 and
//     writes either the natural order (runs of adjacent k_0) or the bit-reversed order (whole rows, contiguous);
//   * all columns of a call in one launch (grid.y).
#pragma once
#include <hip/hip_runtime.h>
#include "fp252.h"
#include "ntt_kernels.h"     // MAXC

namespace ms252 {

static constexpr int TILE_LOG = 11;
static constexpr int TILE_ELEMS = 1 << TILE_LOG;       // 2048 elements * 32 B = 64 KiB
static constexpr int NT2 = 512;

struct alignas(16) U2 { uint64_t x, y; };

struct PassParams {
    const uint64_t* src[msntt::MAXC];
    uint64_t* dst[msntt::MAXC];
    const uint64_t* twr;        // w_R^e, e < R/2: the butterflies inside a tile
    const uint64_t* tw_lo;      // w_n^e two-level (inter-pass factors)
    const uint64_t* tw_hi;
    const uint64_t* sc_lo;      // scale powers c g^i (forward coset: input; inverse: output)
    const uint64_t* sc_hi;
    unsigned log_n, lo_bits;
    unsigned log_r, log_c;      // tile = 2^log_r rows x 2^log_c elements, log_r + log_c = TILE_LOG
    unsigned log_s;             // strided pass: row stride 2^log_s elements
    unsigned log_tw;            // strided pass: factor exponent (k l) << log_tw
    unsigned valid_rows;        // first pass: rows >= valid_rows are implicit zeros
    unsigned log_r0, log_r1;    // last pass: sizes of the earlier digits (log_r1 = 0 with two passes)
    int scale_in, scale_out, bitrev_out;
};
static_assert(sizeof(PassParams) <= msntt::MAX_KERNARG_BYTES, "kernel-argument block (ntt_kernels.h: MAX_KERNARG_BYTES)");

__device__ __forceinline__ f252::E ldg2(const uint64_t* p, size_t i) {
    const U2* q = (const U2*)p + 2 * i;
    const U2 a = q[0], b = q[1];
    return {{a.x, a.y, b.x, b.y}};
}
__device__ __forceinline__ void stg2(uint64_t* p, size_t i, const f252::E& v) {
    U2* q = (U2*)p + 2 * i;
    q[0] = U2{v.l[0], v.l[1]};
    q[1] = U2{v.l[2], v.l[3]};
}
// LDS: two planes of 16-byte halves (lanes that walk consecutive elements then walk consecutive 16-byte words); the
// XOR of the low four index bits with bits 7..10 spreads the bit-reversed scatter of a load (consecutive lanes differ in
// the TOP bits of the slot) over the banks and only permutes runs of 16 consecutive slots among themselves
__device__ __forceinline__ unsigned slot(unsigned idx) { return idx ^ ((idx >> 7) & 15u); }
__device__ __forceinline__ f252::E lds_ld(const U2 (*lds)[TILE_ELEMS], unsigned idx) {
    const unsigned s = slot(idx);
    const U2 a = lds[0][s], b = lds[1][s];
    return {{a.x, a.y, b.x, b.y}};
}
__device__ __forceinline__ void lds_st(U2 (*lds)[TILE_ELEMS], unsigned idx, const f252::E& v) {
    const unsigned s = slot(idx);
    lds[0][s] = U2{v.l[0], v.l[1]};
    lds[1][s] = U2{v.l[2], v.l[3]};
}
__device__ __forceinline__ unsigned brev_bits(unsigned x, unsigned bits) { return bits ? __brev(x) >> (32 - bits) : 0; }

// X' = X + T, Y' = X + (K p - T) for T < K p: the bound of both results is bound(X) + K p
template <int K>
__device__ __forceinline__ void bfly(f252::E& a, f252::E& b) {
    const f252::E m = f252::kp_minus<K>(b);
    const f252::E s = f252::add_lazy(a, b);
    b = f252::add_lazy(a, m);
    a = s;
}

// DIT stages 1..log_r over the tile (virtual index v, column c at slot (v << log_c) | c), the input in bit-reversed
// virtual order, the output natural.  Inputs < 2p; after the first round (no products on three of the four inputs:
// twiddles 1, 1, 1, w_4) the values are < 8p, every later stage adds 2p: < 8p + 2p (log_r - 2) <= 24p < 2^256.
template <int NTH>
__device__ __forceinline__ void tile_dit(U2 (*lds)[TILE_ELEMS], unsigned log_r, unsigned log_c, const uint64_t* twr) {
    const unsigned cmask = (1u << log_c) - 1;
    for (unsigned s = 1; s <= log_r; s += 2) {
        const unsigned half = 1u << (s - 1);
        if (s < log_r) {
            // stages s and s + 1 on the four elements v0 + {0, 1, 2, 3} half
            for (unsigned t = threadIdx.x; t < (unsigned)TILE_ELEMS / 4; t += NTH) {
                const unsigned c = t & cmask, g = t >> log_c, i = g & (half - 1), v0 = ((g >> (s - 1)) << (s + 1)) + i;
                const unsigned i0 = (v0 << log_c) | c, st = half << log_c;
                f252::E x0 = lds_ld(lds, i0), x1 = lds_ld(lds, i0 + st), x2 = lds_ld(lds, i0 + 2 * st), x3 = lds_ld(lds, i0 + 3 * st);
                if (s == 1) {
                    bfly<2>(x0, x1);                                                       // stage 1: w_2^0 = 1, inputs < 2p
                    bfly<2>(x2, x3);
                    x3 = f252::mul_t<false>(x3, ldg2(twr, (size_t)1 << (log_r - 2)));      // stage 2: w_4^1
                    bfly<4>(x0, x2);                                                       // w_4^0 = 1: T = x2 < 4p
                    bfly<2>(x1, x3);
                } else {
                    const f252::E w1 = ldg2(twr, (size_t)i << (log_r - s));                // w_(2^s)^i
                    x1 = f252::mul_t<false>(x1, w1);
                    x3 = f252::mul_t<false>(x3, w1);
                    bfly<2>(x0, x1);
                    bfly<2>(x2, x3);
                    x2 = f252::mul_t<false>(x2, ldg2(twr, (size_t)i << (log_r - s - 1)));              // w_(2^(s+1))^i
                    x3 = f252::mul_t<false>(x3, ldg2(twr, (size_t)(i + half) << (log_r - s - 1)));     // w_(2^(s+1))^(i + half)
                    bfly<2>(x0, x2);
                    bfly<2>(x1, x3);
                }
                lds_st(lds, i0, x0); lds_st(lds, i0 + st, x1); lds_st(lds, i0 + 2 * st, x2); lds_st(lds, i0 + 3 * st, x3);
            }
        } else {
            // log_r odd: the last stage alone, two butterflies per lane
            for (unsigned t = threadIdx.x; t < (unsigned)TILE_ELEMS / 2; t += NTH) {
                const unsigned c = t & cmask, g = t >> log_c, i = g & (half - 1), v0 = ((g >> (s - 1)) << s) + i;
                const unsigned i0 = (v0 << log_c) | c, st = half << log_c;
                f252::E x0 = lds_ld(lds, i0), x1 = lds_ld(lds, i0 + st);
                if (s > 1) x1 = f252::mul_t<false>(x1, ldg2(twr, (size_t)i << (log_r - s)));
                bfly<2>(x0, x1);
                lds_st(lds, i0, x0); lds_st(lds, i0 + st, x1);
            }
        }
        __syncthreads();
    }
}

// canonical table power (two-level above 2^lo_bits entries)
__device__ __forceinline__ f252::E pow_tab(const uint64_t* lo, const uint64_t* hi, unsigned lo_bits, size_t e) {
    f252::E r = ldg2(lo, e & (((size_t)1 << lo_bits) - 1));
    if (e >> lo_bits) r = f252::mul(r, ldg2(hi, e >> lo_bits));
    return r;
}

// ---- pass q < last: rows at stride 2^log_s, 2^log_c consecutive elements per row, in place (same addresses) ------------
// grid = (n / 2048, columns)
template <int NTH, bool FIRST>
__global__ void __launch_bounds__(NTH, NTH / 128) ntt252_strided_pass(PassParams P) {
    __shared__ U2 lds[2][TILE_ELEMS];
    const uint64_t* __restrict__ src = P.src[blockIdx.y];
    uint64_t* __restrict__ dst = P.dst[blockIdx.y];
    const unsigned log_c = P.log_c, log_r = P.log_r, cmask = (1u << log_c) - 1;
    const unsigned lt_bits = P.log_s - log_c;
    const size_t U = blockIdx.x >> lt_bits, l0 = (size_t)(blockIdx.x & ((1u << lt_bits) - 1)) << log_c;
    const size_t base = ((U << log_r) << P.log_s) + l0;
    for (unsigned e = threadIdx.x; e < (unsigned)TILE_ELEMS; e += NTH) {
        const unsigned row = e >> log_c, c = e & cmask;
        f252::E x = f252::zero();
        if (!FIRST || row < P.valid_rows) {
            const size_t j = base + ((size_t)row << P.log_s) + c;
            x = ldg2(src, j);
            if (FIRST && P.scale_in) x = f252::mul_t<false>(x, pow_tab(P.sc_lo, P.sc_hi, P.lo_bits, j));
        }
        lds_st(lds, (brev_bits(row, log_r) << log_c) | c, x);
    }
    __syncthreads();
    tile_dit<NTH>(lds, log_r, log_c, P.twr);
    for (unsigned e = threadIdx.x; e < (unsigned)TILE_ELEMS; e += NTH) {
        const unsigned k = e >> log_c, c = e & cmask;
        const f252::E x = lds_ld(lds, e);
        const size_t ex = ((size_t)k * (l0 + c)) << P.log_tw;                              // < n
        stg2(dst, base + ((size_t)k << P.log_s) + c, f252::mul_t<false>(x, pow_tab(P.tw_lo, P.tw_hi, P.lo_bits, ex)));
    }
}

// ---- last pass: rows of 2^log_r consecutive elements, 2^log_c rows with adjacent k_0 per tile ---------------------------
// address of row (k_0, mid) = k_0 S_0 + mid R, S_0 = n / R_0; output index k = k_0 + R_0 (mid + R_1 k_last)
template <int NTH>
__global__ void __launch_bounds__(NTH, NTH / 128) ntt252_last_pass(PassParams P) {
    __shared__ U2 lds[2][TILE_ELEMS];
    const uint64_t* __restrict__ src = P.src[blockIdx.y];
    uint64_t* __restrict__ dst = P.dst[blockIdx.y];
    const unsigned log_c = P.log_c, log_r = P.log_r, cmask = (1u << log_c) - 1, rmask = (1u << log_r) - 1;
    const unsigned log_s0 = P.log_n - P.log_r0, kt_bits = P.log_r0 - log_c;
    const size_t k0b = (size_t)(blockIdx.x & ((1u << kt_bits) - 1)) << log_c, mid = blockIdx.x >> kt_bits;
    for (unsigned e = threadIdx.x; e < (unsigned)TILE_ELEMS; e += NTH) {
        const unsigned pos = e & rmask, c = e >> log_r;
        const f252::E x = ldg2(src, ((k0b + c) << log_s0) + (mid << log_r) + pos);
        lds_st(lds, (brev_bits(pos, log_r) << log_c) | c, x);
    }
    __syncthreads();
    tile_dit<NTH>(lds, log_r, log_c, P.twr);
    for (unsigned e = threadIdx.x; e < (unsigned)TILE_ELEMS; e += NTH) {
        unsigned k, c;
        size_t out;
        if (P.bitrev_out) {
            const unsigned pos = e & rmask;
            c = e >> log_r; k = brev_bits(pos, log_r);
            out = ((size_t)brev_bits((unsigned)(k0b + c), P.log_r0) << log_s0) + ((size_t)brev_bits((unsigned)mid, P.log_r1) << log_r) + pos;
        } else {
            c = e & cmask; k = e >> log_c;
            out = (k0b + c) + (mid << P.log_r0) + ((size_t)k << (P.log_r0 + P.log_r1));
        }
        f252::E x = lds_ld(lds, (k << log_c) | c);
        if (P.scale_out) {
            const size_t kk = (k0b + c) + (mid << P.log_r0) + ((size_t)k << (P.log_r0 + P.log_r1));
            x = f252::mul(x, pow_tab(P.sc_lo, P.sc_hi, P.lo_bits, kk));
        } else x = f252::reduce_lazy(x);
        stg2(dst, out, x);
    }
}

}  // namespace ms252
