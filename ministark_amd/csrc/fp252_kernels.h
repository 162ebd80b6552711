// NTT / iNTT over the 252-bit StarkWare field for gfx950 (a3 + the fp252 instantiations of
// gpu/src/metal/fft_shaders.h.metal:108-119 and gpu/tests/shaders.rs:69-91).
// The field is compute-bound (one Montgomery product = 20 64x64 multiplies), so the structure
// is deliberately plain: bit-reverse, then radix-2 DIT stages -- the first nine inside LDS on
// 512-element chunks, the rest two stages per launch in registers (ntt252_stages) -- with the coset /
// normalisation scale fused into the first / last kernel.  Twiddles come from a one-level table up to
// 2^21 points (a two-level lookup costs a second ~440-instruction product per butterfly).
#pragma once
#include <hip/hip_runtime.h>
#include "fp252.h"

namespace ms252 {

static constexpr int NT = 256;
static constexpr int CHUNK_LOG = 9;                    // 512 elements * 32 B = 16 KiB of LDS

struct Params {
    uint64_t* col;
    const uint64_t* tw_lo;     // w^i, i < 2^lo_bits        (Montgomery, 4 limbs each)
    const uint64_t* tw_hi;     // w^(i << lo_bits)
    const uint64_t* sc_lo;     // scale powers c*g^i (forward coset: g = h; inverse: c = 1/n, g = 1/h)
    const uint64_t* sc_hi;
    unsigned log_n, lo_bits, stage;
    int scale_in, scale_out;
};
__device__ __forceinline__ f252::E ld(const uint64_t* p, size_t i) { return {{p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]}}; }
__device__ __forceinline__ void st(uint64_t* p, size_t i, const f252::E& x) { p[4 * i] = x.l[0]; p[4 * i + 1] = x.l[1]; p[4 * i + 2] = x.l[2]; p[4 * i + 3] = x.l[3]; }
__device__ __forceinline__ f252::E pow2l(const uint64_t* lo, const uint64_t* hi, unsigned lo_bits, size_t e) {
    f252::E r = ld(lo, e & ((1u << lo_bits) - 1));
    if (e >> lo_bits) r = f252::mul(r, ld(hi, e >> lo_bits));
    return r;
}

// stages 1..min(CHUNK_LOG, log_n) of a DIT transform whose input is already bit-reversed
static __global__ void __launch_bounds__(NT) ntt252_local(Params P) {
    __shared__ uint64_t lds[4 << CHUNK_LOG];
    const unsigned clog = P.log_n < (unsigned)CHUNK_LOG ? P.log_n : (unsigned)CHUNK_LOG;
    const size_t chunk = (size_t)1 << clog, base = (size_t)blockIdx.x * chunk;
    for (unsigned q = threadIdx.x; q < chunk; q += NT) {
        f252::E x = ld(P.col, base + q);
        if (P.scale_in) {
            // element at bit-reversed position base+q is coefficient index rev(base+q)
            const size_t j = P.log_n ? (size_t)(__brevll((unsigned long long)(base + q)) >> (64 - P.log_n)) : 0;
            x = f252::mul(x, pow2l(P.sc_lo, P.sc_hi, P.lo_bits, j));
        }
        st(lds, q, x);
    }
    __syncthreads();
    for (unsigned s = 1; s <= clog; s++) {
        const unsigned half = 1u << (s - 1);
        for (unsigned q = threadIdx.x; q < chunk / 2; q += NT) {
            const unsigned i = q & (half - 1), lo = ((q >> (s - 1)) << s) + i, hi = lo + half;
            const f252::E u = ld(lds, lo);
            const f252::E t = f252::mul(ld(lds, hi), pow2l(P.tw_lo, P.tw_hi, P.lo_bits, (size_t)i << (P.log_n - s)));
            st(lds, lo, f252::add(u, t));
            st(lds, hi, f252::sub(u, t));
        }
        __syncthreads();
    }
    const bool last = P.log_n <= (unsigned)CHUNK_LOG;
    for (unsigned q = threadIdx.x; q < chunk; q += NT) {
        f252::E x = ld(lds, q);
        if (last && P.scale_out) x = f252::mul(x, pow2l(P.sc_lo, P.sc_hi, P.lo_bits, base + q));
        st(P.col, base + q, x);
    }
}
// R consecutive global DIT stages (stage+1 .. stage+R) in one pass: lane = one group of 2^R elements
// spaced 2^stage apart (consecutive lanes take consecutive low indices: 32-byte loads, coalesced).
// A single global stage moves 64 bytes per butterfly and is HBM-bound (measured 3.9 TB/s); two stages per
// pass halve the traffic.  More than two were measured slower: 2^R elements of 8 dwords plus the
// product's temporaries leave 2-3 waves per SIMD, too few to keep the loads in flight.
static constexpr unsigned MAX_FUSED_STAGES = 2;
template <int R>
__global__ void __launch_bounds__(NT) ntt252_stages(Params P) {
    constexpr int G = 1 << R;
    const size_t gid = (size_t)blockIdx.x * NT + threadIdx.x;
    const size_t n = (size_t)1 << P.log_n;
    if (gid >= (n >> R)) return;
    const unsigned s0 = P.stage;                               // stages s0+1 .. s0+R
    const size_t i0 = gid & (((size_t)1 << s0) - 1), base = ((gid >> s0) << (s0 + R)) + i0;
    f252::E x[G];
    #pragma clang loop unroll(full)
    for (int j = 0; j < G; j++) x[j] = ld(P.col, base + ((size_t)j << s0));
    #pragma clang loop unroll(full)
    for (int r = 1; r <= R; r++) {
        const int half = 1 << (r - 1);
        #pragma clang loop unroll(full)
        for (int q = 0; q < G / 2; q++) {
            const int k = q & (half - 1), lo = ((q >> (r - 1)) << r) + k, hi = lo + half;
            const size_t i = i0 + ((size_t)k << s0);           // index inside the half-block of stage s0+r
            const f252::E t = f252::mul(x[hi], pow2l(P.tw_lo, P.tw_hi, P.lo_bits, i << (P.log_n - s0 - r)));
            const f252::E u = x[lo];
            x[lo] = f252::add(u, t);
            x[hi] = f252::sub(u, t);
        }
    }
    const bool last = s0 + R == P.log_n;
    #pragma clang loop unroll(full)
    for (int j = 0; j < G; j++) {
        const size_t e = base + ((size_t)j << s0);
        if (last && P.scale_out) x[j] = f252::mul(x[j], pow2l(P.sc_lo, P.sc_hi, P.lo_bits, e));
        st(P.col, e, x[j]);
    }
}

// FRI degree-respecting projection over Fp252: apply_drp (src/fri.rs:526-567) collapsed exactly as in
// fri_kernels.h -- out[c] = sum_k (alpha / x_i)^k * A_k with A = the size-ff inverse DFT of the ff
// contiguous (bit-reversed) evaluations of chunk c, i = bitrev(c).  The field has no power-of-two
// roots, so the small DFT is a plain radix-2 DIT with the ff/2 twiddles handed over by the host.
struct Fold252Params {
    const uint64_t* src;
    uint64_t* dst;
    const uint64_t* tw_lo;      // w_n^(-i) two-level table of the size-n inverse plan
    const uint64_t* tw_hi;
    unsigned lo_bits, log_m;    // log_m = log2(number of chunks)
    uint64_t hinv[4];           // h^-1
    uint64_t alpha[4];
    uint64_t zinv[8][4];        // (w_n^(-n/ff))^k, k < ff/2
};
template <int FF>
__global__ void __launch_bounds__(NT, 2) fri_fold252(Fold252Params P) {      // FF elements of 8 dwords live in registers
    constexpr int LOGF = FF == 2 ? 1 : FF == 4 ? 2 : FF == 8 ? 3 : 4;
    const size_t c = (size_t)blockIdx.x * NT + threadIdx.x;
    if (c >> P.log_m) return;
    const size_t i = P.log_m ? (size_t)(__brevll((unsigned long long)c) >> (64 - P.log_m)) : 0;
    const f252::E xinv = f252::mul(pow2l(P.tw_lo, P.tw_hi, P.lo_bits, i), f252::E{{P.hinv[0], P.hinv[1], P.hinv[2], P.hinv[3]}});
    f252::E A[FF];
    #pragma clang loop unroll(full)
    for (int q = 0; q < FF; q++) A[q] = ld(P.src, c * FF + q);
    #pragma clang loop unroll(full)
    for (int s = 1; s <= LOGF; s++) {
        constexpr int dummy = 0; (void)dummy;
        const int half = 1 << (s - 1);
        #pragma clang loop unroll(full)
        for (int q = 0; q < FF / 2; q++) {
            const int k = q & (half - 1), lo = ((q >> (s - 1)) << s) + k, hi = lo + half;
            const int e = k << (LOGF - s);
            const f252::E t = e ? f252::mul(A[hi], f252::E{{P.zinv[e][0], P.zinv[e][1], P.zinv[e][2], P.zinv[e][3]}}) : A[hi];
            const f252::E u = A[lo];
            A[lo] = f252::add(u, t);
            A[hi] = f252::sub(u, t);
        }
    }
    const f252::E beta = f252::mul(f252::E{{P.alpha[0], P.alpha[1], P.alpha[2], P.alpha[3]}}, xinv);
    f252::E acc = A[FF - 1];
    #pragma clang loop unroll(full)
    for (int k = FF - 2; k >= 0; k--) acc = f252::add(f252::mul(acc, beta), A[k]);
    st(P.dst, c, acc);
}

}  // namespace ms252
