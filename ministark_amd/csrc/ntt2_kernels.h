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

// Tile index of a workgroup.  scripts/ntt_pass_bench.hip re-defines it (e.g. blockIdx.x & 3) to time a pass on cache-resident tiles,
// i.e. its arithmetic, exchange and issue alone.
#ifndef NTT2_BX
#define NTT2_BX blockIdx.x
#endif
// The tile loads / stores of the radix-256 passes.  Round 4: every word of a pass is read once and written once, so both carry the
// non-temporal hint (global_load/store ... nt): 164 -> 154 us per 2^24 column for the three passes (profiles/r04_pass_bench2_nt.txt;
// each hint alone, or on one pass alone, gives a third of it).  scripts/ntt_pass_bench2.hip re-defines the macros per pass (second /
// third argument: 1, 2 or 3 = which pass of a three-pass plan the access belongs to).
namespace msntt2 {
#if defined(__HIP_DEVICE_COMPILE__)
template <class T> __device__ __forceinline__ T nt_load(const T* p) { return __builtin_nontemporal_load(p); }
template <class T> __device__ __forceinline__ void nt_store(T* p, const T& v) {
    if constexpr (sizeof(T) == 16) {
        typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
        v2u64 t; __builtin_memcpy(&t, &v, 16); __builtin_nontemporal_store(t, (v2u64*)p);
    } else __builtin_nontemporal_store(v, p);
}
#else
template <class T> MS_HD T nt_load(const T* p) { return *p; }
template <class T> MS_HD void nt_store(T* p, const T& v) { *p = v; }
#endif
}
namespace msntt2 { using gld::lane_xor1; }      // gl_dev.h: the word of the neighbouring lane (one DPP move per half)
// STREAM is the kernels' first template parameter: the launcher sets it when data + scratch of a launch exceed the 256 MiB Infinity
// Cache (ms_ntt.cpp).  Batches that fit stay on the default policy: the next pass finds them in the cache, and the hint costs
// 2^17 x 64 columns 1.37 -> 1.54 us per column (profiles/r04_c2_sweep_nt_always.json).
#ifndef NTT2_LD
#define NTT2_LD(p, pass) (STREAM ? msntt2::nt_load(p) : *(p))
#endif
#ifndef NTT2_ST
#define NTT2_ST(p, v, pass) do { if constexpr (STREAM) msntt2::nt_store(p, v); else *(p) = (v); } while (0)
#endif

namespace msntt2 {

using msntt::MAXC;
using msntt::DigitField;
static constexpr int NT = 512;          // threads per workgroup (8 waves)
static constexpr int TW = 64;           // words per tile row (one per lane)
static constexpr int TILE = 256 * TW;   // words per tile
struct alignas(16) Pair { uint64_t x, y; };   // two adjacent words, one 16-byte store
static constexpr int XPITCH = 68;       // pass 1 exchange: row pitch in words (conflict-free transposed reads)

struct Params {
    const uint64_t* src[MAXC];
    uint64_t* dst[MAXC];
    // plain (non-Montgomery) tables of 4 pre-shifted copies: t[4 e + i] = w^e * 2^(24 i) mod p
    const uint64_t* wr4;       // w_256^e, e < 256                      (between the two networks)
    const uint64_t* twu4;      // middle pass: [U][k], w_n^((rev(U) k) << log_s)   (after the second network)
    const uint64_t* sc4;       // last pass, SCALE 1: the constant n^-1 (4 copies)
    const uint64_t* scu4;      // last pass, SCALE 2: [k] h^-(k 2^log_s), k < 256 (4 copies): the row part of the coset scale
    const uint64_t* g4;        // pass 1 coset: g^j1, j1 < 256 (4 copies; the product on the loads uses three)
    // uniform inter-pass factor of a three-pass plan (UNI kernels, see ntt2_first_pass):
    const uint64_t* tin4;      // pass 1: [j2][b][a'] w_256^(a' b) w_n^(a' R3 j2)            (between the two networks)
    const uint64_t* tout4;     // pass 1: [j2][b'] h^(R3 j2) w_n^(16 b' R3 j2)                (after the second network)
    // Montgomery-form tables of the round-1 kernels (per-lane twiddles of pass 1, scale walk of the last pass)
    const uint64_t* tw_lo;
    const uint64_t* tw_hi;
    const uint64_t* aux_lo;
    const uint64_t* aux_hi;
    unsigned log_n, V, valid_rows, lo_bits, log_s, nfields, r3;   // r3 = log2 of the last radix (UNI kernels)
    unsigned xcd_map;          // pass 1: take tiles in the XCD-aware order of first_pass_tile (0 = workgroup b takes tile b)
    DigitField fields[3];
};
static_assert(sizeof(Params) <= msntt::MAX_KERNARG_BYTES, "kernel-argument block (ntt_kernels.h: MAX_KERNARG_BYTES)");

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
#if defined(__HIP_DEVICE_COMPILE__)
// One 32-byte scalar load (s_load_dwordx8).  Written as a vector load on purpose: from four 64-bit loads the compiler
// sometimes builds four overlapping s_load_dwordx4 (16 scalar registers per factor), which spills scalar registers into
// vector lanes -- 472 v_readlane / v_writelane per lane and tile in lde2_strided_pass before this.
__device__ __forceinline__ glimb::W4 w4_at(const uint64_t* t, unsigned slot) {
    typedef uint32_t u32x8 __attribute__((ext_vector_type(8), aligned(8)));
    const u32x8 v = *(const __attribute__((address_space(4))) u32x8*)(t + 4 * (size_t)slot);
    glimb::W4 r;
    r.lo[0] = v[0]; r.hi[0] = v[1]; r.lo[1] = v[2]; r.hi[1] = v[3];
    r.lo[2] = v[4]; r.hi[2] = v[5]; r.lo[3] = v[6]; r.hi[3] = v[7];
    return r;
}
#else
MS_HD glimb::W4 w4_at(const uint64_t* t, unsigned slot) {
    const uint64_t* p = t + 4 * (size_t)slot;
    return glimb::w4_from(p[0], p[1], p[2], p[3]);
}
#endif

// Four consecutive table slots (128 bytes) as two s_load_dwordx16.  Consecutive w4_at calls are not left alone by the
// compiler: it widens EACH of them to a dwordx16 that also fetches its neighbour (16 scalar registers per factor).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void w4x4_at(const uint64_t* t, unsigned slot, glimb::W4* out) {
    typedef uint32_t u32x16 __attribute__((ext_vector_type(16), aligned(8)));
    const __attribute__((address_space(4))) u32x16* p = (const __attribute__((address_space(4))) u32x16*)(t + 4 * (size_t)slot);
    const u32x16 a = p[0], b = p[1];
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        out[0].lo[i] = a[2 * i]; out[0].hi[i] = a[2 * i + 1]; out[1].lo[i] = a[8 + 2 * i]; out[1].hi[i] = a[9 + 2 * i];
        out[2].lo[i] = b[2 * i]; out[2].hi[i] = b[2 * i + 1]; out[3].lo[i] = b[8 + 2 * i]; out[3].hi[i] = b[9 + 2 * i];
    }
}
#else
MS_HD void w4x4_at(const uint64_t* t, unsigned slot, glimb::W4* out) {
    for (int i = 0; i < 4; i++) out[i] = w4_at(t, slot + i);
}
#endif

// Materialise a value here.  Without it LLVM sinks the products of the second half of a network (needed only
// after the next barrier) below the other network, i.e. keeps 4 limbs + 8 twiddle words alive instead of 2 words.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint64_t pin(uint64_t x) { asm volatile("" : "+v"(x)); return x; }
#else
MS_HD uint64_t pin(uint64_t x) { return x; }
#endif
// the same for a wave-uniform value (stays in a scalar register)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ unsigned spin(unsigned x) { asm volatile("" : "+s"(x)); return x; }
#else
MS_HD unsigned spin(unsigned x) { return x; }
#endif

using gld::wave_lockstep;

// run index g of a middle pass (log_s = 8) -> block U of R rows and the run q of 64 words inside a row of 256 V words.  PERM kernels
// are V = 1 (four runs per row): shifts; otherwise one uniform division by the run count 4 V
template <bool PERM>
__device__ __forceinline__ void run_of(unsigned g, unsigned runs_per_u, unsigned& U, unsigned& q) {
    if constexpr (PERM) { U = g >> 2; q = g & 3; (void)runs_per_u; }
    else { U = g / runs_per_u; q = g - U * runs_per_u; }
}

// hide a wave-uniform value's origin from the optimiser (so that what is derived from it is recomputed, not kept in registers)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void opaque(unsigned& v) { asm volatile("" : "+s"(v)); }
#else
MS_HD void opaque(unsigned&) {}
#endif

// three of the four copies of a table slot, for a product on the way into a network (wave-uniform: scalar registers)
__device__ __forceinline__ glimb::Q3 q3_at(const uint64_t* t, unsigned slot) {
    const glimb::W4 w = w4_at(t, slot);
    glimb::Q3 r;
    #pragma unroll
    for (int i = 0; i < 3; i++) { r.lo[i] = w.lo[i]; r.hi[i] = w.hi[i]; }
    return r;
}

// first network of a pass: 16 loaded words (rows 16 a + b) -> w_256^(a' b) * DFT16, as weak 64-bit residues
//   IN   0: words as they are; 1: times the wave-uniform g4[16 a + b] (coset, pass 1); 2: times the per-lane q
//        (both as glimb::mul3_to_limbs: three pre-shifted copies of the factor, no reduction between product and network)
//   UNI  the factor after the network comes from tin4 at slot tslot + a' (pass 1 of a three-pass plan) instead of wr4
template <bool INV, int NA, int IN, bool UNI = false>
__device__ __forceinline__ void net1(uint64_t* x, const Params& P, unsigned b, const glimb::Q3& q = glimb::Q3{}, unsigned tslot = 0) {
    glimb::L4 v[16];
    if constexpr (IN == 1) {
        #pragma unroll
        for (int a0 = 0; a0 < NA; a0 += 4) {           // four scales (24 scalar registers) at a time
            glimb::Q3 g[4];
            #pragma unroll
            for (int j = 0; j < 4 && a0 + j < NA; j++) g[j] = q3_at(P.g4, 16 * (a0 + j) + b);
            __builtin_amdgcn_sched_barrier(0);
            #pragma unroll
            for (int j = 0; j < 4 && a0 + j < NA; j++) v[a0 + j] = glimb::mul3_to_limbs(x[a0 + j], g[j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        #pragma unroll
        for (int a = 0; a < NA; a++) {
            if constexpr (IN == 2) v[a] = glimb::mul3_to_limbs(x[a], q);
            else v[a] = glimb::from_u64(x[a]);
        }
    }
    auto tw = [&](int c) { return UNI ? w4_at(P.tin4, tslot + c) : w4_at(P.wr4, (b * c) & 255); };
    glimb::W4 wn[4];                            // the first group's factors are requested before the network (32 SGPRs: 60.8 -> 55.8 us in pass 1)
    #pragma unroll
    for (int j = 0; j < 4; j++) wn[j] = tw(j);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NA == 16) glimb::dft<16, INV>(v);
    else glimb::dft16_pruned<NA, INV>(v);
    __builtin_amdgcn_sched_barrier(0);          // no further twiddle (scalar) loads hoisted above the network: they would only be spilled
    #pragma unroll
    for (int g = 0; g < 4; g++) {
        glimb::W4 wc[4];
        #pragma unroll
        for (int j = 0; j < 4; j++) wc[j] = wn[j];
        if (g < 3) {
            #pragma unroll
            for (int j = 0; j < 4; j++) wn[j] = tw(4 * (g + 1) + j);
        }
        __builtin_amdgcn_sched_barrier(0);
        #pragma unroll
        for (int j = 0; j < 4; j++) x[4 * g + j] = pin(glimb::mul_fold_co(v[4 * g + j], wc[j]));     // the accumulators of at most 4 elements live
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- middle / last pass, radix 256 ---------------------------------------------------------------------
// grid = (n V / 16384, columns); rows at stride sw = 2^log_s V words (>= 64), tile = 64 consecutive words.
// SCALE (last pass): 0 none, 1 the constant in sc4 (n^-1 of an inverse transform on the subgroup), 2 the inverse coset
//   transform's n^-1 h^-pos, pos = k 2^log_s + low: n^-1 h^-low is one value per lane and tile (it does not depend on the
//   row), applied on the loads like LOADQ's factor; h^-(k 2^log_s) is wave-uniform per output row (scu4).
// LOADQ (pass 2 of a three-pass plan whose pass 1 is the UNI kernel): every word is multiplied on the way in by
//   w_n^(k1 j3), k1 = the lane's position in the row, j3 = this tile's block -- the part of the inter-pass factor
//   (h w_n^k1)^(R3 j2 + j3) that pass 1 cannot apply with wave-uniform operands.  It is one value per lane and tile,
//   so the 128-bit product replaces the plain conversion to limbs (no running product; h^j3 sits in twu4).
// PERM (with LOADQ, V = 1): pass 1 left every 256-word row in the order its stores like best (see
//   ntt2_first_pass<.., PERM>): k1 = a' + 16 d sits at 64 (d >> 2) + 32 ((d >> 1) & 1) + 16 (a' >> 3) + 2 (a' & 7) + (d & 1),
//   a permutation INSIDE each run of 64 words.  The 64 words k1 = 64 q + lane of this tile are therefore this tile's own four
//   128-byte lines, read once in a permuted lane order; the stores go to the natural positions, so the permutation ends here
//   and the pass may run in place (every word of the tile is read before the first one is stored).
template <bool STREAM, bool INV, bool LAST, int SCALE, bool LOADQ = false, bool PERM = false>
__global__ void __launch_bounds__(NT, 4) ntt2_mid_pass(Params P) {
    __shared__ uint64_t xch[16 * 8 * TW];                    // 64 KiB: [b][a' - 8 round][lane]
    const uint64_t* __restrict__ src = P.src[blockIdx.y];
    uint64_t* __restrict__ dst = P.dst[blockIdx.y];
    const unsigned lane = threadIdx.x & 63;
    const unsigned w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t sw = ((size_t)1 << P.log_s) * P.V;
    const unsigned tiles_per_u = (unsigned)(sw / TW);
    const unsigned U = NTT2_BX / tiles_per_u;
    const size_t base = (size_t)U * 256 * sw + (size_t)(NTT2_BX % tiles_per_u) * TW + lane;
    size_t rbase = base;
    if constexpr (PERM) {       // natural k1 = 64 q + lane sits at 64 q + pi(lane): inside this tile's own 64 words, so dst may be src
        const unsigned q = NTT2_BX % tiles_per_u;
        rbase = (size_t)U * 256 * sw + 64 * q + ((lane >> 5) & 1) * 32 + ((lane >> 3) & 1) * 16 + (lane & 7) * 2 + ((lane >> 4) & 1);
    }

    // Register budget: 128 per lane at two workgroups per CU, and a network in limb form holds 64.
    // Addresses walk by the uniform stride 16 sw (one 64-bit add per access; 32 precomputed row offsets would
    // sit in scalar registers for the whole kernel).  The first half of each network's results goes to LDS at once,
    // the second half as soon as round 0 of the exchange has been read (before the second network runs).
    const size_t step = 16 * sw;
    uint64_t x[2][16];
    auto load_half = [&](int h) {
        const uint64_t* p = src + rbase + (size_t)(w + 8 * h) * sw;
        #pragma unroll
        for (int a = 0; a < 16; a++) { x[h][a] = NTT2_LD(p, LAST ? 3 : 2); p += step; }
    };
    // Both halves of the tile are requested up front (the registers allow it since the second half of the first networks'
    // results leaves for LDS before the second network runs).
    uint64_t qlo = 0, qhi = 0;
    if constexpr (SCALE == 2) {
        static_assert(SCALE != 2 || (LAST && !LOADQ), "the coset scale belongs to the last pass");
        const uint64_t e = (NTT2_BX * (uint64_t)TW + lane) / P.V;      // the element this lane's words belong to (U = 0)
        qlo = P.aux_lo[e & ((1u << P.lo_bits) - 1)];
        qhi = P.aux_hi[e >> P.lo_bits];
    }
    if constexpr (LOADQ) {                                   // the two table words of w_n^(k1 j3) first: they come back before the
        const unsigned k1 = ((NTT2_BX % tiles_per_u) * TW + lane) / P.V;              // tile's words and are combined meanwhile
        const uint64_t e = (uint64_t)k1 * digit_rev(P, U);
        qlo = P.tw_lo[e & ((1u << P.lo_bits) - 1)];
        qhi = P.tw_hi[e >> P.lo_bits];
    }
    load_half(0); load_half(1);
    glimb::Q3 qpl{};
    if constexpr (LOADQ || SCALE == 2) {                     // three plain copies q 2^(24 i): the data keeps its own Montgomery factor
        const uint64_t qm = gld::mmul(qlo, qhi);
        qpl = glimb::q3_from(gld::mmul(qm, 1), gld::mmul(qm, (uint64_t)1 << 24), gld::mmul(qm, (uint64_t)1 << 48));
    }
    #pragma unroll
    for (int h = 0; h < 2; h++) {
        net1<INV, 16, (LOADQ || SCALE == 2) ? 2 : 0>(x[h], P, w + 8 * h, qpl);
        #pragma unroll
        for (int j = 0; j < 8; j++) xch[((w + 8 * h) * 8 + j) * TW + lane] = x[h][j];
        __builtin_amdgcn_sched_barrier(0);
    }

    // exchange: (wave, h) = b, register a'  ->  (wave, h) = a', register b; the lane keeps its word.
    // Round r moves the registers a' in [8r, 8r + 8) and is followed by the second network of h = r.
    #pragma unroll
    for (int r = 0; r < 2; r++) {
        __syncthreads();
        uint64_t y[16];
        #pragma unroll
        for (int b = 0; b < 16; b++) y[b] = xch[(b * 8 + w) * TW + lane];
        if (r == 0) {
            // the second half of the first networks' results moves to LDS as soon as everybody has read round 0 -- BEFORE
            // this round's network, so that the 32 registers it occupied are free while the network runs
            __syncthreads();
            #pragma unroll
            for (int h = 0; h < 2; h++)
                #pragma unroll
                for (int j = 0; j < 8; j++) xch[((w + 8 * h) * 8 + j) * TW + lane] = x[h][8 + j];
        }
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
                if constexpr (!LAST) val = glimb::mul_fold_co<false>(v[d], wc[j]);       // a weak residue: every pass accepts any 64-bit representative
                else if constexpr (SCALE == 1) val = glimb::mul_fold_co<true>(v[d], w4_at(P.sc4, 0));
                else if constexpr (SCALE == 2) val = glimb::mul_fold_co<true>(v[d], w4_at(P.scu4, ap + 16 * d));
                else val = glimb::to_canon(v[d]);
                NTT2_ST(dst + pos, val, LAST ? 3 : 2);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- middle pass of a three-pass plan (256, R, 256) with R = 2^LOGR <= 16: no exchange at all --------------------------
// Columns of 2^17..2^20 points.  Rows of sw = 256 V words (log_s = 8), a block U = R rows.  A lane owns one word of a 64-word
// run in all R rows of a block, so the whole radix-R network sits in its registers: the pass is the product on the loads
// (LOADQ's per-lane factor, see ntt2_mid_pass), the network, and the wave-uniform factor w_U^k2 (twu4[U][k2], with h^j3 of a
// coset transform) on the stores.  A workgroup takes 256 / R runs (32 words per lane, 16384 per workgroup, as everywhere).
// PERM: the rows are in the order ntt2_first_pass<.., PERM> leaves them; the natural order is restored here, in place.
template <bool STREAM, bool INV, int LOGR, bool PERM>
__global__ void __launch_bounds__(NT, 4) ntt2_small_mid_pass(Params P) {
    constexpr int R = 1 << LOGR, NNET = 32 / R;               // networks per lane
    const uint64_t* __restrict__ src = P.src[blockIdx.y];
    uint64_t* __restrict__ dst = P.dst[blockIdx.y];
    const unsigned lane = threadIdx.x & 63;
    const unsigned w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t sw = ((size_t)1 << P.log_s) * P.V;
    const unsigned runs_per_u = (unsigned)(sw / TW);          // runs of 64 words per row
    const unsigned g0 = NTT2_BX * (256 / R) + w * NNET;    // this wave's first run (runs count through the blocks U)
    const unsigned pl = PERM ? ((lane >> 5) & 1) * 32 + ((lane >> 3) & 1) * 16 + (lane & 7) * 2 + ((lane >> 4) & 1) : lane;

    // every word of the lane is requested up front; the factors' table words four networks at a time, one group ahead
    constexpr int CH = NNET < 4 ? NNET : 4, NCH = NNET / CH;
    uint64_t x[NNET][R];
    uint64_t qlo[CH], qhi[CH];
    auto factor_words = [&](int c) {
        #pragma unroll
        for (int i = 0; i < CH; i++) {
            unsigned U, q; run_of<PERM>(g0 + c * CH + i, runs_per_u, U, q);
            const unsigned k1 = (q * TW + lane) / P.V;
            const uint64_t e = (uint64_t)k1 * digit_rev(P, U);
            qlo[i] = P.tw_lo[e & ((1u << P.lo_bits) - 1)];
            qhi[i] = P.tw_hi[e >> P.lo_bits];
        }
    };
    factor_words(0);
    #pragma unroll
    for (int i = 0; i < NNET; i++) {
        unsigned U, q; run_of<PERM>(g0 + i, runs_per_u, U, q);
        const uint64_t* p = src + (size_t)U * R * sw + (size_t)q * TW + pl;
        #pragma unroll
        for (int a = 0; a < R; a++) { x[i][a] = NTT2_LD(p, 2); p += sw; }
    }
    if constexpr (PERM) wave_lockstep();      // in place: a lane's stores land on words that other lanes of its wave have read
    #pragma unroll
    for (int c = 0; c < NCH; c++) {
        uint64_t qm[CH];
        #pragma unroll
        for (int i = 0; i < CH; i++) qm[i] = gld::mmul(qlo[i], qhi[i]);
        if (c + 1 < NCH) factor_words(c + 1);
        #pragma unroll
        for (int ii = 0; ii < CH; ii++) {
            const int i = c * CH + ii;
            unsigned U, q; run_of<PERM>(g0 + i, runs_per_u, U, q);
            const glimb::Q3 qpl = glimb::q3_from(gld::mmul(qm[ii], 1), gld::mmul(qm[ii], (uint64_t)1 << 24), gld::mmul(qm[ii], (uint64_t)1 << 48));
            glimb::L4 v[R];
            #pragma unroll
            for (int a = 0; a < R; a++) v[a] = glimb::mul3_to_limbs(x[i][a], qpl);
            glimb::dft<R, INV>(v);
            uint64_t* o = dst + (size_t)U * R * sw + (size_t)q * TW + lane;
            #pragma unroll
            for (int k0 = 0; k0 < R; k0 += 4) {               // four factors (32 scalar registers) at a time
                glimb::W4 wc[4];
                #pragma unroll
                for (int j = 0; j < 4 && k0 + j < R; j++) wc[j] = w4_at(P.twu4, U * R + k0 + j);
                __builtin_amdgcn_sched_barrier(0);
                #pragma unroll
                for (int j = 0; j < 4 && k0 + j < R; j++, o += sw) NTT2_ST(o, glimb::mul_fold_co<false>(v[k0 + j], wc[j]), 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// ---- middle pass of a three-pass plan (256, R, 256) with R = 16 T2, T2 = 2, 4, 8 (columns of 2^21..2^23 points) ------------
// Rows of sw = 256 V words, a block U = R rows, row j2 = a T2 + b.  A workgroup takes 16 / T2 runs of 64 words; a slot
// (wave, h) = (run s, b) holds the 16 rows a of its b in registers: first network over a (ntt2_mid_pass's, with LOADQ's
// per-lane factor on the loads), times w_R^(a' b) (wave-uniform: wr4 holds the powers of w_R for this pass), exchange through LDS in which every word
// stays in its lane and slot (s, c) collects the outputs a' = c mod T2 of all b, radix-T2 networks over b, times the
// wave-uniform w_U^k2 (twu4[U][k2], k2 = a' + 16 b') on the stores.  PERM as in ntt2_mid_pass (in place).
template <bool STREAM, bool INV, int LOGT2, bool PERM>
__global__ void __launch_bounds__(NT, 4) ntt2_mid_pass_r(Params P) {
    constexpr int T2 = 1 << LOGT2, R = 16 * T2, NM = 8 / T2;  // NM networks of radix T2 per slot and round
    static_assert(LOGT2 >= 1 && LOGT2 <= 3, "R = 32, 64 or 128");
    __shared__ uint64_t xch[16 * 8 * TW];                    // 64 KiB: [slot][a' - 8 round][lane]
    const uint64_t* __restrict__ src = P.src[blockIdx.y];
    uint64_t* __restrict__ dst = P.dst[blockIdx.y];
    const unsigned lane = threadIdx.x & 63;
    const unsigned w_in = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned w = w_in;
    const size_t sw = ((size_t)1 << P.log_s) * P.V;
    const unsigned runs_per_u = (unsigned)(sw / TW);
    const unsigned pl = PERM ? ((lane >> 5) & 1) * 32 + ((lane >> 3) & 1) * 16 + (lane & 7) * 2 + ((lane >> 4) & 1) : lane;

    uint64_t x[2][16];
    uint64_t qlo[2], qhi[2];
    #pragma unroll
    for (int h = 0; h < 2; h++) {                            // the factors' table words first (see ntt2_mid_pass)
        const unsigned g = NTT2_BX * (16 / T2) + (w + 8 * h) / T2;
        unsigned U, q; run_of<PERM>(g, runs_per_u, U, q);
        const unsigned k1 = (q * TW + lane) / P.V;
        const uint64_t e = (uint64_t)k1 * digit_rev(P, U);
        qlo[h] = P.tw_lo[e & ((1u << P.lo_bits) - 1)];
        qhi[h] = P.tw_hi[e >> P.lo_bits];
    }
    #pragma unroll
    for (int h = 0; h < 2; h++) {
        const unsigned g = NTT2_BX * (16 / T2) + (w + 8 * h) / T2, b = (w + 8 * h) % T2;
        unsigned U, q; run_of<PERM>(g, runs_per_u, U, q);
        const uint64_t* p = src + (size_t)U * R * sw + (size_t)b * sw + (size_t)q * TW + pl;
        #pragma unroll
        for (int a = 0; a < 16; a++) { x[h][a] = NTT2_LD(p, 2); p += T2 * sw; }
    }
    #pragma unroll
    for (int h = 0; h < 2; h++) {
        const unsigned b = (w + 8 * h) % T2;
        const uint64_t qm = gld::mmul(qlo[h], qhi[h]);
        const glimb::Q3 qpl = glimb::q3_from(gld::mmul(qm, 1), gld::mmul(qm, (uint64_t)1 << 24), gld::mmul(qm, (uint64_t)1 << 48));
        net1<INV, 16, 2>(x[h], P, b, qpl);                    // the factor after the network: wr4 = w_R^e here, slot a' b < R
        #pragma unroll
        for (int j = 0; j < 8; j++) xch[((w + 8 * h) * 8 + j) * TW + lane] = x[h][j];
        __builtin_amdgcn_sched_barrier(0);
    }
    #pragma unroll
    for (int r = 0; r < 2; r++) {
        __syncthreads();
        // (the slot's scalars are derived afresh in every round instead of being kept alive across the networks above, where
        // 64 scalar registers hold twiddles: they would be spilled into vector lanes)
        unsigned w = w_in;
        opaque(w);
        uint64_t y[2][NM][T2];
        #pragma unroll
        for (int h = 0; h < 2; h++) {
            const unsigned s = (w + 8 * h) / T2, c = (w + 8 * h) % T2;
            #pragma unroll
            for (int m = 0; m < NM; m++)
                #pragma unroll
                for (int b = 0; b < T2; b++) y[h][m][b] = xch[((s * T2 + b) * 8 + c + T2 * m) * TW + lane];
        }
        if (r == 0) {                                         // second half of the first networks' results (as ntt2_mid_pass)
            __syncthreads();
            #pragma unroll
            for (int h = 0; h < 2; h++)
                #pragma unroll
                for (int j = 0; j < 8; j++) xch[((w + 8 * h) * 8 + j) * TW + lane] = x[h][8 + j];
        }
        #pragma unroll
        for (int h = 0; h < 2; h++) {
            const unsigned g = NTT2_BX * (16 / T2) + (w + 8 * h) / T2, c = (w + 8 * h) % T2;
            unsigned U, q; run_of<PERM>(g, runs_per_u, U, q);
            uint64_t* const o = dst + (size_t)U * R * sw + (size_t)q * TW + lane;
            #pragma unroll
            for (int m = 0; m < NM; m++) {
                const unsigned ap = 8 * r + c + T2 * m;        // a'
                glimb::L4 v[T2];
                #pragma unroll
                for (int b = 0; b < T2; b++) v[b] = glimb::from_u64(y[h][m][b]);
                glimb::dft<T2, INV>(v);
                #pragma unroll
                for (int b0 = 0; b0 < T2; b0 += 4) {          // at most four factors (32 scalar registers) at a time
                    glimb::W4 wc[4];
                    #pragma unroll
                    for (int j = 0; j < 4 && b0 + j < T2; j++) wc[j] = w4_at(P.twu4, U * R + ap + 16 * (b0 + j));
                    __builtin_amdgcn_sched_barrier(0);
                    #pragma unroll
                    for (int j = 0; j < 4 && b0 + j < T2; j++) NTT2_ST(o + (size_t)(ap + 16 * (b0 + j)) * sw, glimb::mul_fold_co<false>(v[b0 + j], wc[j]), 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
}

// ---- last pass with the bit-reversed store fused (the LDE's forward transform, Fp columns) -----------------------
// Same tile and arithmetic as ntt2_mid_pass<false, true, 0>; the output X[k], k = k3 2^log_s + low, goes to position
// rev(low) 256 + rev8(k3): for every word of the tile a run of 256 consecutive words.  A round of the exchange yields
// the k3 = a' + 16 d with a' in [8r, 8r + 8), i.e. in the run the eight 16-word chunks (2 rev3(a' & 7) + r): the results
// take a third trip through LDS ([word][chunk][rev4(d)], pitch 129: conflict-free both ways) so that 16 consecutive lanes
// store one whole 128-byte line.  The buffer is shared with the exchange, so here the second half of the first networks'
// results waits in registers until round 0 has stored (a few spilled registers, as before round 2b).
static constexpr int BR_PITCH = 129;
template <bool STREAM>
__global__ void __launch_bounds__(NT, 4) ntt2_last_pass_bitrev(Params P) {
    __shared__ uint64_t xch[64 * BR_PITCH];                  // >= 16 * 8 * TW words of the exchange
    const uint64_t* __restrict__ src = P.src[blockIdx.y];
    uint64_t* __restrict__ dst = P.dst[blockIdx.y];
    const unsigned lane = threadIdx.x & 63;
    const unsigned w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t sw = (size_t)1 << P.log_s;                  // V = 1; the last pass has a single block U = 0
    const size_t lo0 = (size_t)NTT2_BX * TW;
    const size_t base = lo0 + lane;
    const size_t step = 16 * sw;
    uint64_t x[2][16];
    #pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint64_t* p = src + base + (size_t)(w + 8 * h) * sw;
        #pragma unroll
        for (int a = 0; a < 16; a++) { x[h][a] = NTT2_LD(p, 3); p += step; }
        net1<false, 16, 0>(x[h], P, w + 8 * h);
        #pragma unroll
        for (int j = 0; j < 8; j++) xch[((w + 8 * h) * 8 + j) * TW + lane] = x[h][j];
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned ch = __brev(w) >> 29;                     // rev3(a' & 7): this wave's chunk of every run
    #pragma unroll
    for (int r = 0; r < 2; r++) {
        if (r) {
            __syncthreads();                                  // round 0's stores have read the buffer
            #pragma unroll
            for (int h = 0; h < 2; h++)
                #pragma unroll
                for (int j = 0; j < 8; j++) xch[((w + 8 * h) * 8 + j) * TW + lane] = x[h][8 + j];
        }
        __syncthreads();
        glimb::L4 v[16];
        #pragma unroll
        for (int b = 0; b < 16; b++) v[b] = glimb::from_u64(xch[(b * 8 + w) * TW + lane]);
        glimb::dft<16, false>(v);
        __syncthreads();                                      // everybody has read the exchange
        #pragma unroll
        for (int d = 0; d < 16; d++) {
            constexpr int R4[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
            xch[lane * BR_PITCH + ch * 16 + R4[d]] = pin(glimb::to_canon(v[d]));
            if ((d & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        #pragma unroll
        for (int it = 0; it < 16; it++) {
            const unsigned idx = it * NT + threadIdx.x;       // (word t, chunk c, i): 16 lanes = one 128-byte line
            const unsigned i = idx & 15, c = (idx >> 4) & 7, t = idx >> 7;
            const size_t e_rev = P.log_s ? (size_t)(__brevll((unsigned long long)(lo0 + t)) >> (64 - P.log_s)) : 0;
            NTT2_ST(dst + ((e_rev << 8) + c * 32 + r * 16 + i), (uint64_t)xch[t * BR_PITCH + c * 16 + i], 3);
        }
    }
}

// ---- pass 1 -----------------------------------------------------------------------------------------------
// grid = (n V / 16384, columns): rows j1 = 16 a + b of n V / 256 words, tile = 64 consecutive words of j'.
// The exchange re-maps the lanes: readers are lane = (c3, tl), wave = th with word t = 8 th + tl and, in round r,
// the low output digit a' = c3 + 8 r -- one second network per lane and round, every lane busy in both rounds.
// A wave then stores 8 runs of 8 consecutive k1 at the digit-reversed position of its 8 words (the two rounds
// fill the two halves of each 128-byte line), the layout ntt_first_pass writes.
// UNI (three-pass plans; their last radix R3 is 256 -- it has to be >= 64 so that a tile of 64 words has one j2 = j' / R3): the inter-pass
// factor (h w_n^k1)^j', j' = R3 j2 + j3, k1 = a' + 16 b', is split into
//     w_n^(a' R3 j2)            merged into the factor between the two networks (a' is a register index there),
//     h^(R3 j2) w_n^(16 b' R3 j2)  after the second network (b' is the register index),
//     (h w_n^k1)^j3             left to pass 2 (ntt2_mid_pass<.., LOADQ>), where it is one value per lane and tile,
// all three wave-uniform or per-lane constants: no running product, no Montgomery multiplications in this pass.
// PERM (UNI, V = 1): the 64-byte pieces of the layout above (8 lanes x 8 B per store, the other half of each line a round
// later) cost pass 1 about 9 us per 2^24 column against whole-line stores (profiles/r02_ubench6_*).  With PERM a row of
// 256 k1 is stored in the order  64 (d >> 2) + 32 ((d >> 1) & 1) + 16 (a' >> 3) + 2 (a' & 7) + (d & 1)  (k1 = a' + 16 d):
// a lane's outputs d, d + 1 are adjacent (one 16-byte store), the 8 lanes a' & 7 fill one 128-byte line per store, and
// every run of 64 natural k1 stays inside its own 64 words; pass 2 reads the rows back in this order
// (ntt2_mid_pass<.., PERM>) and writes the natural one -- in place, since the permutation never leaves a pass-2 tile.
// (Pass 1 itself cannot run in place without a rendezvous of the four tiles that share a 512 KiB slab of the column; that was
// built and measured in round 3 -- 72 against 56 us per column, profiles/r03_ntt3_inplace.txt -- and dropped.)
// Which tile workgroup b of pass 1 takes.  Workgroups are dealt to the 8 XCDs round-robin (b % 8; observed, not promised -- this is an
// ordering for speed, any order is correct).  Tile (j2, g) -- the g-th run of 64 words of block j2 of a row, four per block when the last
// radix is 256 -- writes the 2 KiB runs j2 of output rows 64 g .. 64 g + 63: with b -> tile b the runs that are neighbours in memory
// (j2, j2 + 1 of one g) are written by different XCDs at different times; here an XCD keeps one g and walks consecutive j2, so what its
// L2 writes back is contiguous.  Bare access pattern 53.6 -> 49.8 us per 2^24 column (profiles/r04_ubench9_tile_maps.txt, map 5; the
// in-place passes 2 and 3 are best left in launch order: 49.3 -> 52.2 under the same re-ordering).
__device__ __forceinline__ unsigned first_pass_tile(unsigned b, unsigned ntiles) {
    const unsigned x = b & 7, i = b >> 3, g = x & 3, j2 = (x >> 2) * (ntiles >> 3) + i;
    return j2 * 4 + g;
}
template <bool STREAM, bool INV, bool COSET, int NA, bool UNI = false, bool PERM = false>
__global__ void __launch_bounds__(NT, 4) ntt2_first_pass(Params P) {
    __shared__ uint64_t xch[16 * 8 * XPITCH];                // [b][a' - 8 round][word], pitch 68 words
    const uint64_t* __restrict__ src = P.src[blockIdx.y];
    uint64_t* __restrict__ dst = P.dst[blockIdx.y];
    const unsigned V = P.V;
    const unsigned lane = threadIdx.x & 63;
    const unsigned w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t row_words = ((size_t)1 << (P.log_n - 8)) * V;
    const unsigned bx = P.xcd_map ? first_pass_tile(NTT2_BX, gridDim.x) : NTT2_BX;
    const size_t w0 = (size_t)bx * TW;
    const unsigned j2 = UNI ? (unsigned)((w0 / V) >> P.r3) : 0;       // uniform: the block of R3 V words this tile lies in

    uint64_t x[2][16];
    auto load_half = [&](int h) {
        const uint64_t* p = src + (size_t)(w + 8 * h) * row_words + w0 + lane;
        #pragma unroll
        for (int a = 0; a < NA; a++) {
            if constexpr (NA < 16) x[h][a] = NTT2_LD(p, 1);
            else {                                            // rows >= valid_rows are implicit zeros (uniform select, no branch)
                const bool in = 16u * a + w + 8 * h < P.valid_rows;
                const uint64_t val = NTT2_LD(in ? p : src, 1);
                x[h][a] = in ? val : 0;
            }
            p += 16 * row_words;
        }
    };
    load_half(0); load_half(1);          // both halves in flight before the first network (no spills since round 2b)
    #pragma unroll
    for (int h = 0; h < 2; h++) {
        net1<INV, NA, COSET ? 1 : 0, UNI>(x[h], P, w + 8 * h, glimb::Q3{}, (j2 * 16 + w + 8 * h) * 16);
        #pragma unroll
        for (int j = 0; j < 8; j++) xch[((w + 8 * h) * 8 + j) * XPITCH + lane] = x[h][j];
        __builtin_amdgcn_sched_barrier(0);
    }

    const unsigned c3 = lane & 7, tl = lane >> 3;
    #pragma unroll
    for (int r = 0; r < 2; r++) {
        __syncthreads();
        const unsigned ap = c3 + 8 * r;                       // a' = low digit of k1
        uint64_t y[16];
        #pragma unroll
        for (int b = 0; b < 16; b++) y[b] = xch[(b * 8 + c3) * XPITCH + 8 * w + tl];
        if (r == 0) {                                         // second half to LDS before the network (see ntt2_mid_pass)
            __syncthreads();
            #pragma unroll
            for (int h = 0; h < 2; h++)
                #pragma unroll
                for (int j = 0; j < 8; j++) xch[((w + 8 * h) * 8 + j) * XPITCH + lane] = x[h][8 + j];
        }
        glimb::L4 v[16];
        #pragma unroll
        for (int b = 0; b < 16; b++) v[b] = glimb::from_u64(y[b]);
        glimb::dft<16, INV>(v);
        if constexpr (PERM) {
            const unsigned jp = (unsigned)(w0 + 8 * w + tl);   // this lane's word after the exchange (V = 1)
            Pair* q2 = (Pair*)(dst + ((size_t)digit_rev(P, jp) << 8) + r * 16 + c3 * 2);
            glimb::W4 wn[4];
            #pragma unroll
            for (int j = 0; j < 4; j++) wn[j] = w4_at(P.tout4, j2 * 16 + j);
            #pragma unroll
            for (int g = 0; g < 4; g++) {
                glimb::W4 wc[4];
                #pragma unroll
                for (int j = 0; j < 4; j++) wc[j] = wn[j];
                if (g < 3) {
                    #pragma unroll
                    for (int j = 0; j < 4; j++) wn[j] = w4_at(P.tout4, j2 * 16 + 4 * (g + 1) + j);
                }
                __builtin_amdgcn_sched_barrier(0);
                #pragma unroll
                for (int j = 0; j < 4; j += 2, q2 += 16)     // d = 4 g + j: position 64 g + 32 (j >> 1) + 16 r + 2 c3 + (d & 1)
                    NTT2_ST(q2, (Pair{glimb::mul_fold_co(v[4 * g + j], wc[j]), glimb::mul_fold_co(v[4 * g + j + 1], wc[j + 1])}), 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (UNI) {
            const size_t wd = w0 + 8 * w + tl;                // this lane's word after the exchange
            const unsigned jp = (unsigned)(wd / V), vv = (unsigned)(wd % V);
            uint64_t* q = dst + ((size_t)digit_rev(P, jp) << 8) * V + vv + (size_t)ap * V;
            glimb::W4 wn[4];                                  // scalar loads one group ahead of their use
            #pragma unroll
            for (int j = 0; j < 4; j++) wn[j] = w4_at(P.tout4, j2 * 16 + j);
            #pragma unroll
            for (int g = 0; g < 4; g++) {
                glimb::W4 wc[4];
                #pragma unroll
                for (int j = 0; j < 4; j++) wc[j] = wn[j];
                if (g < 3) {
                    #pragma unroll
                    for (int j = 0; j < 4; j++) wn[j] = w4_at(P.tout4, j2 * 16 + 4 * (g + 1) + j);
                }
                __builtin_amdgcn_sched_barrier(0);
                #pragma unroll
                for (int j = 0; j < 4; j++, q += 16 * V) NTT2_ST(q, glimb::mul_fold_co(v[4 * g + j], wc[j]), 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            uint64_t z[16];
            #pragma unroll
            for (int d = 0; d < 16; d++) {
                z[d] = pin(glimb::to_weak(v[d]));
                if ((d & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            // inter-pass twiddle (h w_n^k1)^j' for k1 = a' + 16 d:  A * B^d   (j' k1 < n: no wrap); everything that is
            // per lane is derived here, after the network, so that it does not occupy registers during it
            const size_t wd = w0 + 8 * w + tl;                // this lane's word after the exchange
            const unsigned jp = (unsigned)(wd / V), vv = (unsigned)(wd % V);
            uint64_t* q = dst + ((size_t)digit_rev(P, jp) << 8) * V + vv + (size_t)ap * V;
            const uint64_t B = tw_pow(P, (uint64_t)jp * 16);
            uint64_t tw = tw_pow(P, (uint64_t)jp * ap);
            if constexpr (COSET) tw = gld::mmul(tw, aux_pow(P, jp));
            #pragma unroll
            for (int d = 0; d < 16; d++, q += 16 * V) {
                NTT2_ST(q, gld::mmul(z[d], tw), 1);
                if (d < 15) tw = gld::mmul(tw, B);
                if ((d & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

}  // namespace msntt2
