// Two-pass coset transforms for the LDE of columns of n = 256 L points, L = 256 T, T in {2, 4, 8, 16, 32, 64}
// (2^17 <= n <= 2^22), Goldilocks Fp -- src/matrix.rs:142-251 / src/prover.rs:50-51 for blow-up beta:
//
//   the evaluations on the coset h<w_N>, N = beta n, are beta transforms of size n: E_j[k] = sum_i c_i (h w_N^j)^i w_n^(i k),
//   j < beta, and in the committed bit-reversed order E_j is the CONTIGUOUS block rev(j) n .. rev(j) n + n, itself in
//   bit-reversed order of k.  So instead of three passes over the N-point column (ntt2_kernels.h: 8 + 64 + 4 x 64 MiB per
//   2^20-row column at beta = 8) each block is produced by two:
//
//   pass A (lde2_strided_pass)  i = i1 L + i0: radix 256 over i1 (rows at stride L, tile = 64 consecutive i0), input scaled by
//                               (G^L)^i1 (wave-uniform), output k1 at row k1 (in place layout) times (G w_n^k1)^i0, G = h w_N^j;
//   pass B (lde2_rows_pass)     radix L over i0 on whole contiguous rows: 256 x T with a 16 x 16 x T register / LDS
//                               decomposition, output k = k1 + 256 k0 written to row rev8(k1) at rev(k0): whole rows of
//                               L consecutive words again, no scattered store anywhere.  T = 32, 64 (round 4: rows of 8192 /
//                               16384 words, the 2^23 / 2^24-point domains of a 2^21 / 2^22-row trace at blow-up 4): the radix-T
//                               part is 16 x T1 (T1 = T / 16) with one more exchange, its factor w_T^(t1 s0) wave-uniform.
//
// Arithmetic is ntt2_kernels.h's (limb form, wave-uniform twiddles through scalar loads); the two per-lane twiddles are
// pass A's inter-pass factor (running product, as in ntt2_first_pass) and pass B's w_L^(k t) between its radix-256 and
// radix-T stages, which comes from a 32 KiB table read with coalesced loads.
#pragma once
#include <hip/hip_runtime.h>
#include "ntt2_kernels.h"

namespace mslde2 {

using msntt::MAXC;
using msntt2::NT;
using msntt2::TW;
using msntt2::cptr_t;
using msntt2::pin;
using msntt2::w4_at;
using msntt2::w4x4_at;

struct Params {
    const uint64_t* src[MAXC];
    uint64_t* dst[MAXC];
    const uint64_t* wr4;       // w_256^e, 4 plain copies each (the n-point plan's table)
    const uint64_t* gpl;       // pass A: [j][i1] (G_j^L)^i1, 256 per coset, 4 plain copies each (the product on the loads uses three)
    const uint64_t* aux;       // pass A: [j][i0] G_j^i0 Montgomery, L per coset
    const uint64_t* tw_lo;     // w_N^e two-level (Montgomery), lo_bits low bits
    const uint64_t* tw_hi;
    const uint64_t* t2;        // pass B: [k][t] w_L^(k t), 256 x T, 4 plain copies each (per-lane factor of a limb-form product)
    // UNI (T >= 4): pass A's inter-pass factor (G w_n^k1)^i0, i0 = 64 i0h + t, split as in ntt2_first_pass<.., UNI>:
    const uint64_t* tin4;      // pass A: [i0h][b][a'] w_256^(a' b) w_n^(a' 64 i0h), 4 plain copies     (between the networks)
    const uint64_t* tout4;     // pass A: [j][i0h][b'] G_j^(64 i0h) w_n^(16 b' 64 i0h), 4 plain copies   (after the second network)
                               // pass B applies (G_j w_n^k1)^t, one value per lane and (wave, half), on its loads
    const uint64_t* c3;        // pass B, T >= 32: [t1][s0] w_T^(t1 s0), t1 < T / 16, s0 < 16, 4 plain copies (between radix 16 and radix T / 16)
    unsigned log_n, log_b, lo_bits;   // n = 2^log_n points per coset, beta = 2^log_b cosets
    // INVERSE transform through the forward kernels (one coset, natural order; round 6): sum_j x_j w^(-j k) = sum_j x_((n - j) mod n) w^(j k), so
    // pass A reads its column backwards (rev) and the forward tables of the offset-1 plan do the rest; the n^-1 h^-k of an inverse (coset)
    // transform multiplies the natural-order output of pass B: oscale[k] (Montgomery, n words) or, on the subgroup, the constant oscale_c.
    const uint64_t* oscale;
    uint64_t oscale_c;
    unsigned rev, oscale_mode;        // oscale_mode: 0 none, 1 the constant, 2 the table
};
static_assert(sizeof(Params) <= msntt::MAX_KERNARG_BYTES, "kernel-argument block (ntt_kernels.h: MAX_KERNARG_BYTES)");

__device__ __forceinline__ uint64_t twn_pow(const Params& P, uint64_t e) {      // w_n^e = w_N^(beta e), e < n
    e <<= P.log_b;
    const uint64_t lo = P.tw_lo[e & ((1u << P.lo_bits) - 1)];
    const uint64_t hi_i = e >> P.lo_bits;
    return hi_i ? gld::mmul(lo, P.tw_hi[hi_i]) : lo;
}

// first network of a pass on 16 loaded words (rows 16 a + b): input scale, DFT16, times w_256^(a' b)
//   IN 0: none; 1: the wave-uniform gpl_j[16 a + b] (pass A); 2: the per-lane q (pass B under UNI)   [glimb::mul3_to_limbs]
//   UNI: the factor after the network comes from tin4 at slot tslot + a' instead of wr4
template <int IN, bool UNI = false>
__device__ __forceinline__ void net1(uint64_t* x, const Params& P, unsigned b, const uint64_t* gpl_j, const glimb::Q3& q = glimb::Q3{}, unsigned tslot = 0) {
    glimb::L4 v[16];
    if constexpr (IN == 1) {
        #pragma unroll
        for (int a0 = 0; a0 < 16; a0 += 4) {            // four input scales (24 scalar registers) at a time
            glimb::Q3 g[4];
            #pragma unroll
            for (int e = 0; e < 4; e++) g[e] = msntt2::q3_at(gpl_j, 16 * (a0 + e) + b);
            __builtin_amdgcn_sched_barrier(0);
            #pragma unroll
            for (int e = 0; e < 4; e++) v[a0 + e] = glimb::mul3_to_limbs(x[a0 + e], g[e]);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        #pragma unroll
        for (int a = 0; a < 16; a++) {
            if constexpr (IN == 2) v[a] = glimb::mul3_to_limbs(x[a], q);
            else v[a] = glimb::from_u64(x[a]);
        }
    }
    auto tw4 = [&](int c0, glimb::W4* o) {      // four factors: consecutive slots under UNI (two wide scalar loads)
        if constexpr (UNI) w4x4_at(P.tin4, tslot + c0, o);
        else {
            #pragma unroll
            for (int j = 0; j < 4; j++) o[j] = w4_at(P.wr4, (b * (c0 + j)) & 255);
        }
    };
    glimb::W4 wn[4];                            // the first group's factors are requested before the network (as msntt2::net1)
    tw4(0, wn);
    __builtin_amdgcn_sched_barrier(0);
    glimb::dft<16, false>(v);
    __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
    for (int g = 0; g < 4; g++) {
        glimb::W4 wc[4];
        #pragma unroll
        for (int j = 0; j < 4; j++) wc[j] = wn[j];
        if (g < 3) tw4(4 * (g + 1), wn);
        __builtin_amdgcn_sched_barrier(0);
        #pragma unroll
        for (int j = 0; j < 4; j++) x[4 * g + j] = pin(glimb::mul_fold_co(v[4 * g + j], wc[j]));
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- pass A ----------------------------------------------------------------------------------------------------------
// grid = (L / 64, beta, columns): coefficients src[col][i1 L + i0] -> dst[col][j n + k1 L + i0].  The cosets of a column are
// adjacent in dispatch order, so its coefficients (read beta times) come from L2 / the Infinity Cache after the first read.
// V = 3 (Fq3 columns, three interleaved words per element): the grid's y extent is 3 beta -- one workgroup per (tile, coset, word plane c);
// it reads the words 3 i + c of the coefficients (24-byte stride: each line is read by the three planes' workgroups back to back on
// one XCD, as the cosets are) and writes plane c of the PLANAR scratch column (dst + c N): the same arithmetic, tables and twiddles as
// for an Fp column, since all three words of an element share its index i.
template <bool STREAM, bool UNI, int V = 1, bool REV = false>      // REV: the column backwards (Params::rev), an inverse transform's pass A
__global__ void __launch_bounds__(NT, 4) lde2_strided_pass(Params P) {
    static_assert(!REV || V == 1, "the backwards read is an Fp plan");
    __shared__ uint64_t xch[16 * 8 * TW];                    // 64 KiB: [b][a' - 8 round][lane]
    const uint64_t* __restrict__ src = P.src[blockIdx.z];
    // Which (tile, coset) this workgroup takes.  A tile of coefficients is read by all beta cosets; in plain grid order (tile fastest) the
    // beta readers of a tile are a whole coset apart and, for 2^22-row columns (32 MiB = all eight L2s), the tile has left the cache by
    // then: 2.8 x the input fetched at beta = 4 (round 5, FETCH_SIZE).  Workgroups are dealt to the XCDs round-robin (linear id % 8;
    // observed, as for first_pass_tile -- only speed depends on it), so the beta cosets of a tile take CONSECUTIVE slots of one XCD:
    // the first one misses, the others find the lines in that XCD's L2 (or merge with the miss in flight).
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y, slot = lin >> 3;
    unsigned j, bx, plane = 0;
    if constexpr (V == 1) {
        j = slot & ((1u << P.log_b) - 1);
        bx = ((slot >> P.log_b) << 3) | (lin & 7);
    } else {
        const unsigned per = (unsigned)V << P.log_b, q = slot / per, jc = slot - q * per;     // wave-uniform: scalar unit
        j = jc / V; plane = jc - j * V;
        bx = (q << 3) | (lin & 7);
    }
    const size_t n = (size_t)1 << P.log_n, L = n >> 8;
    uint64_t* __restrict__ dst = P.dst[blockIdx.z] + (size_t)plane * (n << P.log_b) + (size_t)j * n;
    const unsigned lane = threadIdx.x & 63;
    const unsigned w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t i0 = (size_t)bx * TW + lane;
    const uint64_t* gpl_j = P.gpl + (size_t)j * 256 * 4;

    const size_t step = 16 * L;
    uint64_t x[2][16];
    #pragma unroll
    for (int h = 0; h < 2; h++) {                            // both halves requested before the first network
        if constexpr (REV) {                                  // element (n - idx) mod n: the column backwards, index 0 stays
            size_t idx = i0 + (size_t)(w + 8 * h) * L;
            #pragma unroll
            for (int a = 0; a < 16; a++) { x[h][a] = src[(n - idx) & (n - 1)]; idx += step; }
            continue;
        }
        const uint64_t* p = src + (i0 + (size_t)(w + 8 * h) * L) * V + plane;
        #pragma unroll
        for (int a = 0; a < 16; a++) { x[h][a] = *p; p += step * V; }
    }
    #pragma unroll
    for (int h = 0; h < 2; h++) {
        net1<1, UNI>(x[h], P, w + 8 * h, gpl_j, glimb::Q3{}, (bx * 16 + w + 8 * h) * 16);
        #pragma unroll
        for (int q = 0; q < 8; q++) xch[((w + 8 * h) * 8 + q) * TW + lane] = x[h][q];
        __builtin_amdgcn_sched_barrier(0);
    }
    #pragma unroll
    for (int r = 0; r < 2; r++) {
        __syncthreads();
        const unsigned ap = w + 8 * r;                        // a' = low digit of k1
        uint64_t y[16];
        #pragma unroll
        for (int b = 0; b < 16; b++) y[b] = xch[(b * 8 + w) * TW + lane];
        if (r == 0) {                                         // second half to LDS before the network (as ntt2_mid_pass)
            __syncthreads();
            #pragma unroll
            for (int h = 0; h < 2; h++)
                #pragma unroll
                for (int q = 0; q < 8; q++) xch[((w + 8 * h) * 8 + q) * TW + lane] = x[h][8 + q];
        }
        glimb::L4 v[16];
        #pragma unroll
        for (int b = 0; b < 16; b++) v[b] = glimb::from_u64(y[b]);
        glimb::dft<16, false>(v);
        uint64_t* q = dst + (size_t)ap * L + i0;
        if constexpr (UNI) {
            const unsigned slot0 = ((j * (unsigned)(L >> 6)) + bx) * 16;
            #pragma unroll
            for (int g = 0; g < 4; g++) {                     // four factors at a time (more of them in flight spill scalar registers)
                glimb::W4 wc[4];
                w4x4_at(P.tout4, slot0 + 4 * g, wc);
                __builtin_amdgcn_sched_barrier(0);
                #pragma unroll
                for (int e = 0; e < 4; e++, q += step) NTT2_ST(q, glimb::mul_fold_co(v[4 * g + e], wc[e]), 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            uint64_t z[16];
            #pragma unroll
            for (int d = 0; d < 16; d++) {
                z[d] = pin(glimb::to_weak(v[d]));
                if ((d & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            // (G w_n^k1)^i0 for k1 = a' + 16 d: A B^d with A = G^i0 w_n^(i0 a'), B = w_n^(16 i0)   (i0 k1 < n: no wrap)
            const uint64_t B = twn_pow(P, (uint64_t)i0 * 16);
            uint64_t tw = gld::mmul(twn_pow(P, (uint64_t)i0 * ap), P.aux[(size_t)j * L + i0]);
            #pragma unroll
            for (int d = 0; d < 16; d++, q += step) {
                NTT2_ST(q, gld::mmul(z[d], tw), 1);
                if (d < 15) tw = gld::mmul(tw, B);
                if ((d & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// ---- pass B ----------------------------------------------------------------------------------------------------------
// Rows of L = 256 T consecutive words, i0 = a (16 T) + b T + t.  A workgroup takes 64 / T rows: lane = (row select, t),
// the radix-256 part over (a, b) is ntt2_mid_pass's (first network over a, uniform w_256^(a' b), exchange, second network
// over b), then w_L^(k t) per lane, an exchange that gathers the T values t of a (row, k) pair into one lane, the radix-T
// network, and a last trip through LDS so that the stores are runs of consecutive words.
// grid = (256 T / 64, columns, beta)          [64 / T rows per workgroup, 256 rows per coset]
static constexpr int X2P = 65;                               // pitch of the second exchange (words): conflict-free both ways
// NATURAL (one coset): the output X[k1 + 256 k0] goes to its natural position -- the forward transform of a 2^18-point column in two
// passes (T = 4; ms_ntt.cpp routes GpuFft there): a workgroup's 64 / T consecutive rows k1 give runs of 64 / T words per k0.
// V = 3 (Fq3 columns; T <= 16): the source is the PLANAR scratch of lde2_strided_pass<.., 3>, the destination the interleaved column.  The
// lane groups that are rows for an Fp column (rs = lane >> log T, 64 / T of them) become (row, word plane): rs = 4 row + plane, plane 3
// idle -- 16 / T rows x 3 planes per workgroup, three quarters of the lanes at work -- so that the three words of every element a
// workgroup produces meet in its LDS and leave as whole runs of 48 T consecutive words: no partial line is ever stored.
template <bool STREAM, int T, bool UNI, bool NATURAL = false, int V = 1, int OSCALE = 0>     // OSCALE (NATURAL): 1 the constant, 2 the table of Params
__global__ void __launch_bounds__(NT, 4) lde2_rows_pass(Params P) {
    static_assert(OSCALE == 0 || NATURAL, "the output scale of an inverse transform belongs to the natural-order stores");
    static_assert(!NATURAL || T <= 4, "natural-order stores: runs of 64 / T words, T = 2 or 4");
    static_assert(V == 1 || (V == 3 && T <= 16 && !NATURAL), "Fq3 columns: rows of at most 4096 elements, bit-reversed order");
    constexpr int LOGT = T == 64 ? 6 : T == 32 ? 5 : T == 16 ? 4 : T == 8 ? 3 : T == 4 ? 2 : T == 2 ? 1 : 0;
    constexpr int RSEL = 64 / T;                             // rows per workgroup
    constexpr int T1 = T > 16 ? T / 16 : 1, LOGT1 = T1 == 4 ? 2 : T1 == 2 ? 1 : 0;     // T >= 32: radix T = 16 x T1
    constexpr int ITEMP = 16 * T1 + 1;                       // T >= 32: pitch (words) of a (k, row) item in the third exchange
    constexpr int X3WORDS = 8192 + 8192 / 16;               // third exchange: the 8192 words of a round, one pad word per 16 (V = 3: + 11 per plane,
                                                             // inside the slots of the idle fourth plane)
    constexpr int XCHW = X3WORDS > 128 * X2P ? X3WORDS : 128 * X2P;
    // The second half of the first network's outputs (x[h][8..15], 32 registers) waits in registers through all of round 0; NSP of those
    // words per lane wait in the tail of the LDS buffer instead (80 KiB per workgroup: still two per CU) -- with them in registers the
    // compiler kept 5 .. 19 registers in scratch memory, and scratch stores reach HBM.
    constexpr int NSP = 3, XWORDS = XCHW + NSP * NT;
    static_assert(XWORDS * 8 <= 80 * 1024, "two workgroups per CU");
    __shared__ uint64_t xch[XWORDS];
    const unsigned j = blockIdx.z;
    const size_t n = (size_t)1 << P.log_n;
    constexpr size_t L = 256 * T;
    const unsigned lane = threadIdx.x & 63;
    const unsigned w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned rs = lane >> LOGT, t = lane & (T - 1);
    constexpr int RW = V == 3 ? RSEL / 4 : RSEL;             // rows of the transform per workgroup
    // the row (within the workgroup) and the word plane of lane group g; the idle fourth plane reads plane 0 again and stores nothing
    auto row_of = [](unsigned g) { return V == 3 ? g >> 2 : g; };
    const unsigned row0 = blockIdx.x * RW, my_row = row0 + row_of(rs);
    const unsigned my_plane = V == 3 ? ((rs & 3) == 3 ? 0 : (rs & 3)) : 0;
    const uint64_t* __restrict__ src = P.src[blockIdx.y] + (size_t)my_plane * (n << P.log_b) + (size_t)j * n + (size_t)my_row * L + t;
    // Fp columns, T <= 16: the same address as a wave-uniform base + a 32-bit lane offset in words (scalar-base loads; counted: 4165 -> 4127 instructions
    // and no register left in scratch at T = 16; at T = 32 / 64 the 64-bit form is the shorter one)
    const uint64_t* __restrict__ sbase = P.src[blockIdx.y] + (size_t)j * n + (size_t)row0 * L;
    const unsigned soff = (V == 3 ? my_plane * (unsigned)(n << P.log_b) : 0u) + row_of(rs) * (unsigned)L + t;
    const unsigned jr = P.log_b ? __brev(j) >> (32 - P.log_b) : 0;     // block of coset j in the bit-reversed order
    uint64_t* __restrict__ dst = P.dst[blockIdx.y] + (size_t)jr * n * V;

    // UNI: the part of pass A's factor that is per lane there, (G_j w_n^k1)^(i0 & 63), is one value per lane and half here
    // (k1 = this lane's row; i0 & 63 = ((b & (64 / T - 1)) T + t, b = w + 8 h): the 128-bit product replaces the conversion
    uint64_t qm[2] = {0, 0};                                 // the factor in Montgomery form; its three plain copies are made per half
    if constexpr (UNI) {
        static_assert(!UNI || T >= 4, "i0 & 63 must not reach the register digit a");
        #pragma unroll
        for (int h = 0; h < 2; h++) {
            const unsigned tl = ((w + 8 * h) & (64 / T - 1)) * T + t, k1 = my_row;
            qm[h] = gld::mmul(twn_pow(P, (uint64_t)k1 * tl), P.aux[(size_t)j * L + tl]);
        }
    }
    uint64_t x[2][16];
    #pragma unroll
    for (int h = 0; h < 2; h++) {
        if constexpr (T <= 16 && V == 1) {
            const unsigned po = soff + (w + 8 * h) * T;
            #pragma unroll
            for (int a = 0; a < 16; a++) x[h][a] = NTT2_LD(sbase + (po + a * 16 * T), 2);
        } else {
            const uint64_t* p = src + (size_t)(w + 8 * h) * T;
            #pragma unroll
            for (int a = 0; a < 16; a++) { x[h][a] = NTT2_LD(p, 2); p += 16 * T; }
        }
        glimb::Q3 qh{};
        if constexpr (UNI) qh = glimb::q3_from(gld::mmul(qm[h], 1), gld::mmul(qm[h], (uint64_t)1 << 24), gld::mmul(qm[h], (uint64_t)1 << 48));
        net1<UNI ? 2 : 0>(x[h], P, w + 8 * h, nullptr, qh);
        #pragma unroll
        for (int q = 0; q < 8; q++) xch[((w + 8 * h) * 8 + q) * TW + lane] = x[h][q];
        if (h == 1) {
            #pragma unroll
            for (int k = 0; k < NSP; k++) xch[XCHW + k * NT + threadIdx.x] = x[1][16 - NSP + k];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    #pragma unroll
    for (int r = 0; r < 2; r++) {
        if (r) {
            __syncthreads();                                  // the stores of round 0 have read the buffer
            #pragma unroll
            for (int h = 0; h < 2; h++)
                #pragma unroll
                for (int q = 0; q < 8; q++) xch[((w + 8 * h) * 8 + q) * TW + lane] = (h == 1 && q >= 8 - NSP) ? xch[XCHW + (q - (8 - NSP)) * NT + threadIdx.x] : x[h][8 + q];
        }
        __syncthreads();
        const unsigned ap = w + 8 * r;                        // a' = low digit of k
        {
            glimb::L4 v[16];
            #pragma unroll
            for (int b = 0; b < 16; b++) v[b] = glimb::from_u64(xch[(b * 8 + w) * TW + lane]);
            glimb::dft<16, false>(v);
            __syncthreads();                                  // everybody has read the first exchange
            // k = a' + 16 b': times w_L^(k t) (one factor per lane and output: four plain copies, 32 bytes, read coalesced
            // along t from a table that lives in L2), to slot kk = 16 w + b' of the second exchange
            #pragma unroll
            for (int g = 0; g < 8; g++) {                     // two factors (16 registers) in flight
                glimb::W4 wt[2];
                #pragma unroll
                for (int e = 0; e < 2; e++) {
                    const msntt2::Pair* tp = (T <= 16 && V == 1) ? (const msntt2::Pair*)(P.t2 + (((ap + 16 * (2 * g + e)) * T + t) * 4u))
                                                     : (const msntt2::Pair*)(P.t2 + ((size_t)(ap + 16 * (2 * g + e)) * T + t) * 4);      // two 16-byte loads
                    const msntt2::Pair lo = tp[0], hi = tp[1];
                    wt[e] = glimb::w4_from(lo.x, lo.y, hi.x, hi.y);
                }
                #pragma unroll
                for (int e = 0; e < 2; e++) xch[(w * 16 + 2 * g + e) * X2P + lane] = pin(glimb::mul_fold_co(v[2 * g + e], wt[e]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        if constexpr (T <= 16) {
        // the T words of (row, k) pairs: this lane owns pairs pr = threadIdx.x + 512 u, u < T / ... (128 k x RSEL rows = 8192 / T pairs)
        constexpr int NPAIR = (128 * RSEL) / NT;             // pairs per lane and round: 16 / T
        uint64_t out[NPAIR][T];
        #pragma unroll
        for (int u = 0; u < NPAIR; u++) {
            const unsigned pr = threadIdx.x + NT * u;         // pair index: kk fastest (128), then the row
            const unsigned kk = pr & 127, rsel = pr >> 7;
            glimb::L4 v[T];
            #pragma unroll
            for (int tt = 0; tt < T; tt++) v[tt] = glimb::from_u64(xch[kk * X2P + rsel * T + tt]);
            glimb::dft<T, false>(v);
            #pragma unroll
            for (int tt = 0; tt < T; tt++) out[u][tt] = pin(glimb::to_canon(v[tt]));
        }
        __syncthreads();                                      // the second exchange has been read
        if constexpr (NATURAL) {
            // third trip, natural order: slot of (k0 part e = (w' 16 + b') T + s, row) is e (RSEL + 1) + row: the readers run along the rows
            #pragma unroll
            for (int u = 0; u < NPAIR; u++) {
                const unsigned pr = threadIdx.x + NT * u;
                const unsigned kk = pr & 127, rsel = pr >> 7;
                #pragma unroll
                for (int tt = 0; tt < T; tt++) xch[(kk * T + tt) * (RSEL + 1) + rsel] = out[u][tt];
            }
            __syncthreads();
            #pragma unroll
            for (int i = 0; i < 16; i++) {
                const unsigned idx = i * NT + threadIdx.x;    // < 8192 = 128 T x RSEL
                const unsigned rsel = idx % RSEL, e = idx / RSEL;
                const unsigned tt = e & (T - 1), bq = (e >> LOGT) & 15, wq = e >> (LOGT + 4);
                const size_t k0 = (size_t)(wq + 8 * r) + 16 * bq + 256 * (size_t)tt;
                const size_t kout = (size_t)(row0 + rsel) + 256 * k0;
                uint64_t val = xch[e * (RSEL + 1) + rsel];
                if constexpr (OSCALE == 2) val = gld::mmul(val, P.oscale[kout]);       // an inverse transform's n^-1 h^-k
                else if constexpr (OSCALE == 1) val = gld::mmul(val, P.oscale_c);
                NTT2_ST(dst + kout, val, 2);
            }
        } else {
        // third trip: chunk (row, x = bitrev3(w'), c = bitrev4(b')) holds the T outputs in bit-reversed order of t'; the slot
        // of word ad is ad + ad / 16 (16 lanes that write 16 different chunks then hit 16 different banks for every T)
        #pragma unroll
        for (int u = 0; u < NPAIR; u++) {
            const unsigned pr = threadIdx.x + NT * u;
            const unsigned kk = pr & 127, rsel = pr >> 7;
            const unsigned wq = kk >> 4, bq = kk & 15;
            const unsigned chunk = (rsel * 8 + (__brev(wq) >> 29)) * 16 + (__brev(bq) >> 28);
            #pragma unroll
            for (int tt = 0; tt < T; tt++) {
                unsigned rt = 0;
                #pragma unroll
                for (int bit = 0; bit < LOGT; bit++) rt |= ((tt >> bit) & 1u) << (LOGT - 1 - bit);
                const unsigned ad = chunk * T + rt;
                if (V == 3 && (rsel & 3) == 3) continue;      // the idle plane: its slots take the overhang of plane 2's bank offset
                xch[ad + (ad >> 4) + (V == 3 ? 11 * (rsel & 3) : 0)] = out[u][tt];      // (V = 3: the planes of an element on different banks)
            }
        }
        __syncthreads();
        if constexpr (V == 1) {
        // stores: row rev8(k1); inside the row, run (r + 2 x) of 16 T words = chunks c = 0..15 of T words
        #pragma unroll
        for (int i = 0; i < 16; i++) {
            const unsigned idx = i * NT + threadIdx.x;        // < 8192 = RSEL * 8 * 16 * T
            const unsigned tt = idx & (T - 1), c = (idx >> LOGT) & 15, xq = (idx >> (LOGT + 4)) & 7, rsel = idx >> (LOGT + 7);
            const unsigned k1 = row0 + rsel;
            NTT2_ST(dst + ((size_t)(__brev(k1) >> 24) * L + (size_t)(r + 2 * xq) * (16 * T) + c * T + tt), (uint64_t)xch[idx + (idx >> 4)], 2);
        }
        } else {
        // Fq3 stores: the run (row, x) of 16 T elements = 48 T consecutive words, the three planes of an element side by side.  A wave takes
        // four segments of 64 elements = 192 words (three store instructions of 64 consecutive words each): which element of the segment
        // and which plane a lane carries in instruction m is the same for every segment ((64 m + lane) / 3 and the remainder), so an
        // address is a wave-uniform base plus one of three per-lane constants -- no division in the loop.
        {
            unsigned eo[3], lds_off[3];
            #pragma unroll
            for (int m = 0; m < 3; m++) {
                const unsigned wd = 64 * m + lane, el = wd / 3, plane = wd - 3 * el;          // element of the segment (< 64), word of the element
                eo[m] = el;
                const unsigned lo = plane * (128 * T) + (16 * T >= 64 ? el : 0);               // T = 2: the element's share is added per segment
                lds_off[m] = lo + (lo >> 4) + 11 * plane;
            }
            #pragma unroll
            for (int sgi = 0; sgi < 4; sgi++) {
                const unsigned sg = w * 4 + sgi;                                               // elements 64 sg .. 64 sg + 63 of the round: wave-uniform
                #pragma unroll
                for (int m = 0; m < 3; m++) {
                    if constexpr (16 * T >= 64) {
                        const unsigned el0 = 64 * sg, e0 = el0 & (16 * T - 1), xq = (el0 >> (LOGT + 4)) & 7, rw = el0 >> (LOGT + 7);   // scalar unit
                        const unsigned ad0 = ((rw * 4) * 8 + xq) * 16 * T + e0;                                                           // a multiple of 64
                        const size_t g0 = ((size_t)(__brev(row0 + rw) >> 24) * L + (size_t)(r + 2 * xq) * (16 * T) + e0) * 3;
                        NTT2_ST(dst + g0 + 64 * m + lane, (uint64_t)xch[ad0 + (ad0 >> 4) + lds_off[m]], 2);
                    } else {                                                                   // T = 2: a segment spans two runs of 32 elements
                        const unsigned el = 64 * sg + eo[m], e = el & (16 * T - 1), xq = (el >> (LOGT + 4)) & 7, rw = el >> (LOGT + 7);
                        const unsigned ad0 = ((rw * 4) * 8 + xq) * 16 * T + e;
                        const unsigned wd = 64 * m + lane, plane = wd - 3 * eo[m];
                        const size_t g = ((size_t)(__brev(row0 + rw) >> 24) * L + (size_t)(r + 2 * xq) * (16 * T) + e) * 3 + plane;
                        NTT2_ST(dst + g, (uint64_t)xch[ad0 + (ad0 >> 4) + lds_off[m]], 2);
                    }
                }
            }
        }
        }
        }
        } else {
        // ---- T = 16 T1 (T1 = 2, 4): t = t1 + T1 t0, output s = s0 + 16 s1.  Radix 16 over t0 with t1 WAVE-UNIFORM (one (k, row, t1)
        // item per lane: 128 k x RSEL rows x T1 = 512), times w_T^(t1 s0) from scalar registers, third exchange, radix T1 over t1.
        // (the lane index of each block below goes through pin(): otherwise the compiler computes the index arithmetic of BOTH rounds once,
        // ahead of round 0, and keeps ~40 addresses alive through it -- 16 / 19 registers in scratch at T = 64 / 32, and scratch stores reach
        // HBM: a quarter more bytes written than the output itself (round 6, WRITE_SIZE).  Not done for T <= 16: there the shared index
        // arithmetic is 250 instructions per round, 6 % of the kernel, against two spilled registers)
        {
            const unsigned ln = (unsigned)pin((uint64_t)lane);
            const unsigned t1 = msntt2::spin(w & (T1 - 1));   // (and the 16 factors of c3 are loaded again in round 1 rather than parked in vector lanes)
            const unsigned q = (w >> LOGT1) * 64 + ln, kk = q & 127, rsel = q >> 7;
            glimb::L4 v[16];
            #pragma unroll
            for (int t0 = 0; t0 < 16; t0++) v[t0] = glimb::from_u64(xch[kk * X2P + rsel * T + t1 + T1 * t0]);
            glimb::dft<16, false>(v);
            __syncthreads();                                  // the second exchange has been read
            uint64_t* const e3 = xch + (size_t)(rsel * 128 + kk) * ITEMP + t1;
            #pragma unroll
            for (int g = 0; g < 4; g++) {
                glimb::W4 wc[4];
                w4x4_at(P.c3, t1 * 16 + 4 * g, wc);
                __builtin_amdgcn_sched_barrier(0);
                #pragma unroll
                for (int e = 0; e < 4; e++) e3[(4 * g + e) * T1] = pin(glimb::mul_fold_co(v[4 * g + e], wc[e]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        // radix T1 and the stores: wave = a' (as after the first exchange), lane = (position s0r = rev4(s0) of the run, bq); a lane takes the
        // chunks rev4(b') = bq + 4 i of its wave's 16 T-word chunks: T1 consecutive words each (bit-reversed order of s1), 16-byte stores
        // T1 = 4: a lane's four words are 32 consecutive bytes but a store instruction carries 16 per lane -- written straight out, each of
        // the two instructions covers HALF of every 32-byte sector, and the streaming stores are not merged on the way to HBM: 2.04 x the
        // output written (round 5, WRITE_SIZE).  So the lanes of a pair (2 p, 2 p + 1) take the groups p and 8 + p of the 16-group chunk and
        // swap halves through one DPP move per register: every lane then holds words 2 l, 2 l + 1 and 32 + 2 l, 33 + 2 l of the chunk, and
        // each store instruction writes 256 contiguous bytes per 16 lanes.
        {
            const unsigned ln = (unsigned)pin((uint64_t)lane);
            const unsigned s0r = ln & 15, bq = ln >> 4;
            const unsigned grp = T1 == 4 ? ((s0r & 1) << 3) | (s0r >> 1) : s0r, s0 = __brev(grp) >> 28;      // position of this lane's group in the chunk
            #pragma unroll
            for (int i = 0; i < 4 * RSEL; i++) {
                const unsigned brev = bq + 4 * (i & 3), rsel = i >> 2, bp = __brev(brev) >> 28, kk = w * 16 + bp;
                const uint64_t* const e3 = xch + (size_t)(rsel * 128 + kk) * ITEMP + s0 * T1;
                glimb::L4 u[T1];
                #pragma unroll
                for (int tt = 0; tt < T1; tt++) u[tt] = glimb::from_u64(e3[tt]);
                glimb::dft<T1, false>(u);
                const unsigned k1 = row0 + rsel;
                uint64_t* const chunk = dst + (size_t)(__brev(k1) >> 24) * L + (size_t)((r + 2 * (__brev(w) >> 29)) * 16 + brev) * T;
                if constexpr (T1 == 4) {
                    const uint64_t w0 = glimb::to_canon(u[0]), w1 = glimb::to_canon(u[2]), w2 = glimb::to_canon(u[1]), w3 = glimb::to_canon(u[3]);   // memory order
                    const uint64_t n0 = msntt2::lane_xor1(w0), n1 = msntt2::lane_xor1(w1), n2 = msntt2::lane_xor1(w2), n3 = msntt2::lane_xor1(w3);
                    const bool odd = s0r & 1;
                    NTT2_ST((msntt2::Pair*)(chunk + 2 * s0r), (msntt2::Pair{odd ? n2 : w0, odd ? n3 : w1}), 2);
                    NTT2_ST((msntt2::Pair*)(chunk + 32 + 2 * s0r), (msntt2::Pair{odd ? w2 : n0, odd ? w3 : n1}), 2);
                } else {
                    NTT2_ST((msntt2::Pair*)(chunk + s0r * T1), (msntt2::Pair{glimb::to_canon(u[0]), glimb::to_canon(u[1])}), 2);
                }
            }
        }
        }
    }
}

}  // namespace mslde2
