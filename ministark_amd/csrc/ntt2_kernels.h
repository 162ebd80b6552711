// Radix-256 NTT passes on 24-bit limbs (round 2) -- drop-in replacements for ntt_first_pass and
// ntt_mid_pass<16,...> of ntt_kernels.h: same source / destination layouts, same twiddle conventions,
// bit-identical results; only the inside of a pass is different.
//
// What changed and why (measurements: profiles/r02_ubench4_*, r02_ubench5_*):
//   * the pass is VALU-bound, so the two radix-16 networks of a pass run in the redundant limb form of
//     gl_limb.h (adds without carries, power-of-two twiddles as limb rotations) and the twiddle that
//     follows a network is eight v_mad_u64_u32 against four pre-shifted copies of the factor:
//     162 -> ~115 cycles per element and network instead of 227;
//   * those copies must not cost vector loads or registers, so the work is laid out such that every
//     general twiddle except pass 1's inter-pass factor is WAVE-UNIFORM and arrives through scalar loads:
//     a workgroup is 8 waves, lanes run along 64 consecutive words (512-byte runs, the granule at which
//     strided HBM access reaches the copy rate), each lane owns two radix-16 networks (h = 0, 1) and the
//     row digit b = wave + 8 h is uniform per (wave, h).  w_256^(a' b) then depends on a register index
//     and uniform values only; the per-tile factor of a middle pass w_U^k likewise (k = a' + 16 b' with
//     a' uniform after the exchange): both come from host-built tables of {w, w 2^24, w 2^48, w 2^72};
//   * the exchange between the two networks keeps every word in its lane (wave index <-> register index),
//     so LDS is addressed linearly and conflict-free; it runs in two rounds through 64 KiB, two workgroups
//     per CU.  Pass 1 alone re-maps lanes (word, then low output digit) for its digit-reversed stores.
//
// Tile = 256 rows x 64 words = 16384 words per 512-thread workgroup, 32 words per lane.
#pragma once
#include <hip/hip_runtime.h>
#include "gl.h"
#include "gl_dev.h"
#include "gl_limb.h"
#include "ntt_kernels.h"

namespace msntt2 {

using msntt::MAXC;
using msntt::DigitField;
static constexpr int NT = 512;          // threads per workgroup (8 waves)
static constexpr int TW = 64;           // words per tile row (one per lane)
static constexpr int TILE = 256 * TW;   // words per tile
static constexpr int XPITCH = 68;       // pass 1 exchange: row pitch in words (conflict-free transposed reads)

struct Params {
    const uint64_t* src[MAXC];
    uint64_t* dst[MAXC];
    // plain (non-Montgomery) tables of 4 pre-shifted copies: t[4 e + i] = w^e * 2^(24 i) mod p
    const uint64_t* wr4;       // w_256^e, e < 256                      (between the two networks)
    const uint64_t* twu4;      // middle pass: [U][k], w_n^((rev(U) k) << log_s)   (after the second network)
    const uint64_t* sc4;       // last pass, SCALE 1: the constant n^-1 (4 copies)
    const uint64_t* g_plain;   // pass 1 coset: g^j1 plain, j1 < 256
    // Montgomery-form tables of the round-1 kernels (per-lane twiddles of pass 1, scale walk of the last pass)
    const uint64_t* tw_lo;
    const uint64_t* tw_hi;
    const uint64_t* aux_lo;
    const uint64_t* aux_hi;
    unsigned log_n, V, valid_rows, lo_bits, log_s, nfields;
    DigitField fields[3];
};

__device__ __forceinline__ uint64_t tw_pow(const Params& P, uint64_t e) {
    uint64_t lo = P.tw_lo[e & ((1u << P.lo_bits) - 1)];
    uint64_t hi_i = e >> P.lo_bits;
    return hi_i ? gld::mmul(lo, P.tw_hi[hi_i]) : lo;
}
__device__ __forceinline__ uint64_t aux_pow(const Params& P, uint64_t e) {
    uint64_t lo = P.aux_lo[e & ((1u << P.lo_bits) - 1)];
    uint64_t hi_i = e >> P.lo_bits;
    return hi_i ? gld::mmul(lo, P.aux_hi[hi_i]) : lo;
}
__device__ __forceinline__ unsigned digit_rev(const Params& P, unsigned x) {
    unsigned r = 0;
    for (unsigned f = 0; f < P.nfields; f++)
        r |= ((x >> P.fields[f].in_shift) & P.fields[f].mask) << P.fields[f].out_shift;
    return r;
}
// four copies of a twiddle at a wave-uniform table slot (the compiler turns this into s_load_dwordx8)
// The tables are read-only for the lifetime of a plan: reading them through the constant address space lets the
// compiler keep the scalar loads even after the kernel has issued global stores.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) uint64_t* cptr_t;
#else
typedef const uint64_t* cptr_t;
#endif
__device__ __forceinline__ glimb::W4 w4_at(const uint64_t* t, unsigned slot) {
    cptr_t p = (cptr_t)(t + 4 * (size_t)slot);
    return glimb::w4_from(p[0], p[1], p[2], p[3]);
}

// Materialise a value here.  Without it LLVM sinks the products of the second half of a network (needed only
// after the next barrier) below the other network, i.e. keeps 4 limbs + 8 twiddle words alive instead of 2 words.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint64_t pin(uint64_t x) { asm volatile("" : "+v"(x)); return x; }
#else
MS_HD uint64_t pin(uint64_t x) { return x; }
#endif

// first network of a pass: 16 loaded words (rows 16 a + b) -> w_256^(a' b) * DFT16, as weak 64-bit residues
template <bool INV, int NA>
__device__ __forceinline__ void net1(uint64_t* x, const Params& P, unsigned b, bool coset) {
    glimb::L4 v[16];
    if (coset) {
        #pragma unroll
        for (int a = 0; a < NA; a++) v[a] = glimb::mul_to_limbs(x[a], ((cptr_t)P.g_plain)[16 * a + b]);   // uniform: scalar load
    } else {
        #pragma unroll
        for (int a = 0; a < NA; a++) v[a] = glimb::from_u64(x[a]);
    }
    if constexpr (NA == 16) glimb::dft<16, INV>(v);
    else glimb::dft16_pruned<NA, INV>(v);
    __builtin_amdgcn_sched_barrier(0);          // no twiddle (scalar) loads hoisted above the network: they would only be spilled
    #pragma unroll
    for (int c = 0; c < 16; c++) {
        x[c] = pin(glimb::mul_fold(v[c], w4_at(P.wr4, (b * c) & 255)));
        if ((c & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // keep the accumulators of at most 4 elements live
    }
}

// ---- middle / last pass, radix 256 ---------------------------------------------------------------------
// grid = (n V / 16384, columns); rows at stride sw = 2^log_s V words (>= 64), tile = 64 consecutive words.
// SCALE (last pass): 0 none, 1 the constant in sc4 (n^-1 of an inverse transform on the subgroup).
template <bool INV, bool LAST, int SCALE>
__global__ void __launch_bounds__(NT, 4) ntt2_mid_pass(Params P) {
    __shared__ uint64_t xch[16 * 8 * TW];                    // 64 KiB: [b][a' - 8 round][lane]
    const uint64_t* __restrict__ src = P.src[blockIdx.y];
    uint64_t* __restrict__ dst = P.dst[blockIdx.y];
    const unsigned lane = threadIdx.x & 63;
    const unsigned w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t sw = ((size_t)1 << P.log_s) * P.V;
    const unsigned tiles_per_u = (unsigned)(sw / TW);
    const unsigned U = blockIdx.x / tiles_per_u;
    const size_t base = (size_t)U * 256 * sw + (size_t)(blockIdx.x % tiles_per_u) * TW + lane;

    // Register budget: 128 per lane at two workgroups per CU, and a network in limb form holds 64.
    // Addresses walk by the uniform stride 16 sw (one 64-bit add per access; 32 precomputed row offsets would
    // sit in scalar registers for the whole kernel).  The second half of the lane's words is loaded only after
    // the first network has run, and the first half of each network's results goes to LDS at once: what waits
    // in registers beside a network in flight is then 16 words, not 32.
    const size_t step = 16 * sw;
    uint64_t x[2][16];
    #pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint64_t* p = src + base + (size_t)(w + 8 * h) * sw;
        #pragma unroll
        for (int a = 0; a < 16; a++) { x[h][a] = *p; p += step; }
        net1<INV, 16>(x[h], P, w + 8 * h, false);
        #pragma unroll
        for (int j = 0; j < 8; j++) xch[((w + 8 * h) * 8 + j) * TW + lane] = x[h][j];
        __builtin_amdgcn_sched_barrier(0);
    }

    // exchange: (wave, h) = b, register a'  ->  (wave, h) = a', register b; the lane keeps its word.
    // Round r moves the registers a' in [8r, 8r + 8) and is followed at once by the second network of h = r.
    #pragma unroll
    for (int r = 0; r < 2; r++) {
        if (r) {
            __syncthreads();
            #pragma unroll
            for (int h = 0; h < 2; h++)
                #pragma unroll
                for (int j = 0; j < 8; j++) xch[((w + 8 * h) * 8 + j) * TW + lane] = x[h][8 + j];
        }
        __syncthreads();
        uint64_t y[16];
        #pragma unroll
        for (int b = 0; b < 16; b++) y[b] = xch[(b * 8 + w) * TW + lane];
        const unsigned ap = w + 8 * r;                        // a'
        glimb::W4 wn[4];                                      // twiddles one group ahead of their use (scalar loads from a 2 MiB table)
        if constexpr (!LAST) {
            #pragma unroll
            for (int j = 0; j < 4; j++) wn[j] = w4_at(P.twu4, U * 256 + ap + 16 * j);
        }
        glimb::L4 v[16];
        #pragma unroll
        for (int b = 0; b < 16; b++) v[b] = glimb::from_u64(y[b]);
        glimb::dft<16, INV>(v);
        size_t pos = base + (size_t)ap * sw;
        #pragma unroll
        for (int g = 0; g < 4; g++) {
            glimb::W4 wc[4];
            if constexpr (!LAST) {
                #pragma unroll
                for (int j = 0; j < 4; j++) wc[j] = wn[j];
                if (g < 3) {
                    #pragma unroll
                    for (int j = 0; j < 4; j++) wn[j] = w4_at(P.twu4, U * 256 + ap + 16 * (4 * (g + 1) + j));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            #pragma unroll
            for (int j = 0; j < 4; j++, pos += step) {
                const int d = 4 * g + j;
                uint64_t val;
                if constexpr (!LAST) val = glimb::mul_fold<true>(v[d], wc[j]);
                else if constexpr (SCALE == 1) val = glimb::mul_fold<true>(v[d], w4_at(P.sc4, 0));
                else val = glimb::to_canon(v[d]);
                dst[pos] = val;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- pass 1 -----------------------------------------------------------------------------------------------
// grid = (n V / 16384, columns): rows j1 = 16 a + b of n V / 256 words, tile = 64 consecutive words of j'.
// The exchange re-maps the lanes: readers are lane = (c3, tl), wave = th with word t = 8 th + tl and, in round r,
// the low output digit a' = c3 + 8 r -- one second network per lane and round, every lane busy in both rounds.
// A wave then stores 8 runs of 8 consecutive k1 at the digit-reversed position of its 8 words (the two rounds
// fill the two halves of each 128-byte line), the layout ntt_first_pass writes.
template <bool INV, bool COSET, int NA>
__global__ void __launch_bounds__(NT, 4) ntt2_first_pass(Params P) {
    __shared__ uint64_t xch[16 * 8 * XPITCH];                // [b][a' - 8 round][word], pitch 68 words
    const uint64_t* __restrict__ src = P.src[blockIdx.y];
    uint64_t* __restrict__ dst = P.dst[blockIdx.y];
    const unsigned V = P.V;
    const unsigned lane = threadIdx.x & 63;
    const unsigned w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t row_words = ((size_t)1 << (P.log_n - 8)) * V;
    const size_t w0 = (size_t)blockIdx.x * TW;

    uint64_t x[2][16];
    auto load_half = [&](int h) {
        const uint64_t* p = src + (size_t)(w + 8 * h) * row_words + w0 + lane;
        #pragma unroll
        for (int a = 0; a < NA; a++) {
            if constexpr (NA < 16) x[h][a] = *p;
            else {                                            // rows >= valid_rows are implicit zeros (uniform select, no branch)
                const bool in = 16u * a + w + 8 * h < P.valid_rows;
                const uint64_t val = *(in ? p : src);
                x[h][a] = in ? val : 0;
            }
            p += 16 * row_words;
        }
    };
#ifdef MS_NTT2_EARLY
    load_half(0); load_half(1);
#endif
    #pragma unroll
    for (int h = 0; h < 2; h++) {
#ifndef MS_NTT2_EARLY
        load_half(h);
#endif
        net1<INV, NA>(x[h], P, w + 8 * h, COSET);
        #pragma unroll
        for (int j = 0; j < 8; j++) xch[((w + 8 * h) * 8 + j) * XPITCH + lane] = x[h][j];
        __builtin_amdgcn_sched_barrier(0);
    }

    const unsigned c3 = lane & 7, tl = lane >> 3;
    #pragma unroll
    for (int r = 0; r < 2; r++) {
        if (r) {
            __syncthreads();
            #pragma unroll
            for (int h = 0; h < 2; h++)
                #pragma unroll
                for (int j = 0; j < 8; j++) xch[((w + 8 * h) * 8 + j) * XPITCH + lane] = x[h][8 + j];
        }
        __syncthreads();
        const unsigned ap = c3 + 8 * r;                       // a' = low digit of k1
        glimb::L4 v[16];
        #pragma unroll
        for (int b = 0; b < 16; b++) v[b] = glimb::from_u64(xch[(b * 8 + c3) * XPITCH + 8 * w + tl]);
        glimb::dft<16, INV>(v);
        uint64_t z[16];
        #pragma unroll
        for (int d = 0; d < 16; d++) {
            z[d] = pin(glimb::to_weak(v[d]));
            if ((d & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        // inter-pass twiddle (h w_n^k1)^j' for k1 = a' + 16 d:  A * B^d   (j' k1 < n: no wrap); everything that is
        // per lane is derived here, after the network, so that it does not occupy registers during it
        const size_t wd = w0 + 8 * w + tl;                    // this lane's word after the exchange
        const unsigned jp = (unsigned)(wd / V), vv = (unsigned)(wd % V);
        uint64_t* q = dst + ((size_t)digit_rev(P, jp) << 8) * V + vv + (size_t)ap * V;
        const uint64_t B = tw_pow(P, (uint64_t)jp * 16);
        uint64_t tw = tw_pow(P, (uint64_t)jp * ap);
        if constexpr (COSET) tw = gld::mmul(tw, aux_pow(P, jp));
        #pragma unroll
        for (int d = 0; d < 16; d++, q += 16 * V) {
            *q = gld::mmul(z[d], tw);
            if (d < 15) tw = gld::mmul(tw, B);
            if ((d & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

}  // namespace msntt2
