// FRI degree-respecting projection for gfx950: the reference's `apply_drp`
// (src/fri.rs:526-567, called from build_layer src/fri.rs:199-231) as ONE streaming kernel.
//
// The reference does: bit_reverse -> iNTT(coset(n,h)) -> coeffs *= ff -> per chunk of ff
// coefficients dot (1, a, .., a^(ff-1)) -> NTT(coset(n/ff, h^ff)) -> bit_reverse, i.e. two
// full transforms and two permutation passes per layer.  Algebraically (exact in the field, so
// bit-identical): with x_i = h w^i and z = w^(n/ff),
//     out[i] = sum_{k<ff} (a / x_i)^k * A_k,   A_k = sum_{j<ff} z^(-jk) f(x_i z^j)
// (the ff of "coeffs *= ff" cancels the 1/ff of the size-ff interpolation).  In the committed
// bit-reversed layout the ff evaluations of coset i sit contiguously at chunk c = bitrev(i), in
// bit-reversed j order, and out[c] is the folded value in the next layer's bit-reversed order.
// So: one lane per chunk, a size-ff inverse butterfly network in registers (power-of-two
// twiddles), Horner in (a/x_i).  Reads n elements, writes n/ff.  HBM-bound for Fp.
#pragma once
#include <hip/hip_runtime.h>
#include "gl.h"
#include "gl_dev.h"
#include "stage_kernels.h"

namespace msfri {

static constexpr int NT = 256;

struct FoldParams {
    const uint64_t* src;
    uint64_t* dst;
    const uint64_t* tw_lo;     // w_n^(-i) two-level table (Montgomery form) of the size-n inverse plan
    const uint64_t* tw_hi;
    unsigned lo_bits;
    unsigned log_m;            // bits 0-7: log2(n / ff) = number of chunks; bits 8+: log2(table size / n)
    uint64_t hinv;             // h^-1, Montgomery form
    uint64_t alpha[3];         // Montgomery form; Fp: alpha[0]
    size_t c0, count;          // this call folds chunks [c0, c0 + count) of the layer; src / dst point at chunk c0 (ms_fri_fold: 0, all)
};

template <int FF, int V>
__global__ void __launch_bounds__(NT) fri_fold(FoldParams P) {
    const size_t c = (size_t)blockIdx.x * NT + threadIdx.x;
    const unsigned log_m = P.log_m & 255, tshift = P.log_m >> 8;
    if (c >= P.count) return;
    // x_i^-1 = h^-1 w_n^-i,  i = bitrev(c0 + c);  the table holds powers of w_N^-1 with N = n << tshift
    const size_t i = log_m ? (size_t)(__brevll((unsigned long long)(P.c0 + c)) >> (64 - log_m)) : 0;
    const size_t e = i << tshift;
    uint64_t xinv = P.tw_lo[e & ((1u << P.lo_bits) - 1)];
    if (e >> P.lo_bits) xinv = gld::mmul(xinv, P.tw_hi[e >> P.lo_bits]);
    xinv = gld::mmul(xinv, P.hinv);
    uint64_t A[V][FF];
    const uint64_t* __restrict__ in = P.src + c * FF * V;
    #pragma unroll
    for (int q = 0; q < FF; q++) {
        #pragma unroll
        for (int v = 0; v < V; v++) A[v][q] = in[q * V + v];
    }
    #pragma unroll
    for (int v = 0; v < V; v++) {
        gld::dft_lazy<FF, true, true>(A[v]);          // inputs are stored in bit-reversed j order
        #pragma unroll
        for (int q = 0; q < FF; q++) A[v][q] = gld::canon(A[v][q]);
    }
    if constexpr (V == 1) {
        const uint64_t beta = gld::mmul(P.alpha[0], xinv);
        uint64_t acc = A[0][FF - 1];
        #pragma unroll
        for (int k = FF - 2; k >= 0; k--) acc = gl::add(gld::mmul(acc, beta), A[0][k]);
        P.dst[c] = acc;
    } else {
        using F3 = msstage::Fq3T;
        const gl::Fq3 beta = {gld::mmul(P.alpha[0], xinv), gld::mmul(P.alpha[1], xinv), gld::mmul(P.alpha[2], xinv)};
        gl::Fq3 acc = {A[0][FF - 1], A[1][FF - 1], A[2][FF - 1]};
        #pragma unroll
        for (int k = FF - 2; k >= 0; k--) acc = gl::add(F3::mul(acc, beta), gl::Fq3{A[0][k], A[1][k], A[2][k]});
        P.dst[3 * c] = acc.c0; P.dst[3 * c + 1] = acc.c1; P.dst[3 * c + 2] = acc.c2;
    }
}

}  // namespace msfri
