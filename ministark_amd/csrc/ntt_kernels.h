// Goldilocks NTT kernels for gfx950.
//
// Replaces the reference's FftSingle / FftMultiple / BitReverse / MulAssign(scale)
// kernel chain (gpu/src/metal/fft_shaders.h.metal:13-101, gpu/src/plan.rs:378-462:
// log2(n)-10 full read+write passes + a bit-reverse pass + a scale pass) by a
// mixed-radix decomposition with at most ceil(log2(n)/8) passes over HBM and
// NO separate bit-reversal or scaling pass.  Output order is natural, identical
// to ark_poly Radix2EvaluationDomain::{fft,ifft}_in_place.
//
// Decomposition of n = R1*R2*...*Rm (R1 = 256, 16 <= Rp <= 256):
//   input index  j = (j1, j2, ..., jm)   j1 most significant
//   output index k = k1 + R1*k2 + ...    k1 least significant
//   pass 1 (x -> scratch): tile = all j1 x 16 consecutive words of j' = (j2..jm);
//       y[k1][j'] = (h*w_n^k1)^j' * sum_j1 (x[j1][j'] * g^j1) w_R1^(j1 k1),  g = h^(n/R1)
//       stored at (digit-reversed j') * R1 + k1, i.e. layout (jm, ..., j2, k1):
//       every later pass finds its digit at stride R1*...*R(p-1) and the final
//       layout (km, ..., k2, k1) is the natural order -- the "bit reversal" is
//       folded into the 2 KiB-contiguous stores of pass 1.
//   pass p >= 2 (in place on scratch; the last one scratch -> x): tile = all jp x T
//       consecutive low words, T = 4096/Rp; twiddle w_(np)^(j'_p * kp) is a
//       per-tile geometric sequence kept in LDS.
// Every workgroup is 256 threads owning 16 elements each: a radix-16 butterfly
// network in registers, one exchange through LDS, a second radix-(Rp/16) network
// in registers.  Loads and stores are 128 B .. 2 KiB contiguous per wave.
//
// The data words are the reference's Montgomery residues; by linearity they are
// transformed as plain residues with plain twiddles (see gl.h).
// V = u64 words per element: 1 for Fp, 3 for Fq3 (base-field twiddles act
// component-wise, so an Fq3 column is three interleaved Fp transforms).
#pragma once
#include <hip/hip_runtime.h>
#include "gl.h"
#include "gl_dev.h"

namespace msntt {

static constexpr int MAXC = 256;       // columns per launch (grid.y): small columns need many per launch to fill the workgroup slots (round 5: 128 -> 256,
                                       // 4 KiB of pointers in the kernel arguments: 2^15 x 256 columns 0.19 -> 0.245 of HBM, 2^14 x 256 0.11 -> 0.19)
// Every kernel takes its column pointers BY VALUE (2 x MAXC pointers = 4 KiB + the tables): the structs stay below this bound, checked
// where each is defined.  HSA puts no 4 KiB limit on the kernel-argument segment (that is CUDA's); ROCm 7.2 on gfx950 runs 520-column
// launches (tests/test_ntt_parity.py), and a launch a runtime refuses surfaces through hipGetLastError() as MS_ERR_HIP, never silently.
static constexpr size_t MAX_KERNARG_BYTES = 6 * 1024;
static constexpr int TILE = 4096;      // words per workgroup tile
static constexpr int NT = 256;         // threads per workgroup
static constexpr int LDS_PAD_CS = 144; // c-stride (words) of the mid-pass exchange layout (half tile + 16)
#ifndef MS_NTT_WAVES
#define MS_NTT_WAVES 8                 // resident waves per SIMD the pass kernels are register-allocated for
#endif

struct DigitField { unsigned in_shift, out_shift, mask; };

struct PassParams {
    const uint64_t* src[MAXC];
    uint64_t* dst[MAXC];
    const uint64_t* tw_lo;     // w_n^i,        i < 2^lo_bits
    const uint64_t* tw_hi;     // w_n^(i<<lo_bits)
    const uint64_t* wr;        // w_R^e, e < R of this pass
    const uint64_t* aux_lo;    // pass 1: h^i (coset) ; last pass: c*hinv^i (inverse scale)
    const uint64_t* aux_hi;    //         h^(i<<lo_bits)            hinv^(i<<lo_bits)
    const uint64_t* gtab;      // pass 1 coset: g^j1, j1 < 256
    unsigned log_n;
    unsigned V;                // u64 words per element (1 Fp, 3 Fq3)
    unsigned valid_rows;       // pass 1: rows j1 >= valid_rows are implicit zeros (LDE zero padding)
    unsigned lo_bits;          // two-level table split
    unsigned log_s;            // log2 of the element stride of this pass's digit
    unsigned nfields;          // digit-reversal fields
    DigitField fields[3];
    uint64_t scale_const;      // last pass, inverse & offset==1: n^-1 (plain)
};
static_assert(sizeof(PassParams) <= msntt::MAX_KERNARG_BYTES, "kernel-argument block (ntt_kernels.h: MAX_KERNARG_BYTES)");

// w_n^e for e < n via the two-level table
__device__ __forceinline__ uint64_t tw_pow(const PassParams& P, uint64_t e) {
    uint64_t lo = P.tw_lo[e & ((1u << P.lo_bits) - 1)];
    uint64_t hi_i = e >> P.lo_bits;
    return hi_i ? gld::mmul(lo, P.tw_hi[hi_i]) : lo;
}
__device__ __forceinline__ uint64_t aux_pow(const PassParams& P, uint64_t e) {
    uint64_t lo = P.aux_lo[e & ((1u << P.lo_bits) - 1)];
    uint64_t hi_i = e >> P.lo_bits;
    return hi_i ? gld::mmul(lo, P.aux_hi[hi_i]) : lo;
}
__device__ __forceinline__ unsigned digit_rev(const PassParams& P, unsigned x) {
    unsigned r = 0;
    for (unsigned f = 0; f < P.nfields; f++)
        r |= ((x >> P.fields[f].in_shift) & P.fields[f].mask) << P.fields[f].out_shift;
    return r;
}

// The radix-16 / radix-(Rp/16) register networks are gld::dft_lazy (gl_dev.h): inputs
// canonical, outputs weak; every table below is in MONTGOMERY form (w * 2^64 mod p) so that
// gld::mmul(data, table) = data * w, canonical, for weak `data`.

// ---- pass 1 ----------------------------------------------------------------------
// grid = (n*V/256/16, columns).  COSET: input scaled by h^j (offset != 1, forward).
// NA: 16 = dense input; 1 / 2 / 4 = only rows j1 < 16*NA are non-zero (LDE blow-up 16 / 8 / 4): the
// zero rows are neither loaded nor multiplied and the first network collapses (gld::dft16_pruned).
// The LDS exchange runs in two rounds through one 16 KiB buffer (waves 0-1 publish their half of the
// tile, everybody reads; then waves 2-3): half the footprint of a one-shot exchange, so that 8
// workgroups (the VGPR limit) are resident per CU to cover the global-load latency.
template <bool INV, bool COSET, int NA = 16>
__global__ void __launch_bounds__(NT, INV ? MS_NTT_WAVES - 1 : MS_NTT_WAVES) ntt_first_pass(PassParams P) {
    const unsigned V = P.V;
    __shared__ uint64_t lds[TILE / 2];
    uint64_t y[16];
    const uint64_t* __restrict__ src = P.src[blockIdx.y];
    uint64_t* __restrict__ dst = P.dst[blockIdx.y];
    const unsigned tid = threadIdx.x;
    const size_t row_words = ((size_t)1 << (P.log_n - 8)) * V;   // words per j1 row
    const size_t w0 = (size_t)blockIdx.x * 16;

    // phase 1: thread (t, b) owns rows j1 = 16a + b, a = 0..15, word column w0 + t
    {
        const unsigned t = tid & 15, b = tid >> 4;
        uint64_t x[16];
        if constexpr (NA == 16) {
            #pragma unroll
            for (int a = 0; a < 16; a++)
                x[a] = (16u * a + b < P.valid_rows) ? src[(size_t)(16 * a + b) * row_words + w0 + t] : 0;
            if constexpr (COSET) {
                #pragma unroll
                for (int a = 0; a < 16; a++) x[a] = gld::mmul(x[a], P.gtab[16 * a + b]);
            }
            gld::dft_lazy<16, INV>(x);
        } else {
            #pragma unroll
            for (int a = 0; a < NA; a++) x[a] = src[(size_t)(16 * a + b) * row_words + w0 + t];
            if constexpr (COSET) {
                #pragma unroll
                for (int a = 0; a < NA; a++) x[a] = gld::mmul(x[a], P.gtab[16 * a + b]);
            }
            gld::dft16_pruned<NA, INV>(x);
        }
        // internal twiddle w_256^(b c) (also canonicalises: wr[0] = 1), then exchange so that
        // thread (c, t) gets all b
        #pragma unroll
        for (int c = 0; c < 16; c++) x[c] = gld::mmul(x[c], P.wr[(b * c) & 255]);
        // word (t, b & 7, c) of the half tile sits at  (t*128 + (b & 7)*16 + c) ^ (t | (t & 1) << 4):
        // conflict-free for the (t, b)-major 64-bit writes and the (c, t)-major reads.  The write
        // address is A ^ c (re-derived per round, one v_xor each, instead of 16 pinned registers),
        // the read address one of two bases plus an immediate.
        const unsigned c2 = tid & 15, t2 = tid >> 4;
        const unsigned A = (t * 128 + ((b & 7) << 4)) ^ (t | ((t & 1) << 4));
        const unsigned Rb = (t2 * 128 + c2) ^ (t2 | ((t2 & 1) << 4));
        const uint64_t* rd0 = lds + Rb;            // even rows bb
        const uint64_t* rd1 = lds + (Rb ^ 16);     // odd rows
        if (b < 8) {                                         // wave-uniform: waves 0-1
            const unsigned A0 = gld::opaque(A);
            #pragma unroll
            for (int c = 0; c < 16; c++) lds[A0 ^ c] = x[c];
        }
        __syncthreads();
        #pragma unroll
        for (int bb = 0; bb < 8; bb++) y[bb] = (bb & 1) ? rd1[(bb & ~1) << 4] : rd0[bb << 4];
        __syncthreads();
        if (b >= 8) {
            const unsigned A1 = gld::opaque(A);
            #pragma unroll
            for (int c = 0; c < 16; c++) lds[A1 ^ c] = x[c];
        }
        __syncthreads();
        #pragma unroll
        for (int bb = 0; bb < 8; bb++) y[8 + bb] = (bb & 1) ? rd1[(bb & ~1) << 4] : rd0[bb << 4];
    }
    // phase 2: thread (c, t) owns b = 0..15 -> outputs k1 = c + 16 d
    {
        const unsigned tid2 = gld::opaque(tid);              // keep the twiddle loads below the exchange
        const unsigned c = tid2 & 15, t = tid2 >> 4;
        gld::dft_lazy<16, INV>(y);
        const size_t w = w0 + t;
        const unsigned jp = (unsigned)(w / V), v = (unsigned)(w % V);
        const size_t out_base = ((size_t)digit_rev(P, jp) << 8) * V + v;
        // twiddle (h w_n^k1)^j' for k1 = c + 16 d:  A * B^d   (j'*k1 < n: no wrap)
        uint64_t A = tw_pow(P, (uint64_t)jp * c);
        if constexpr (COSET) A = gld::mmul(A, aux_pow(P, jp));
        const uint64_t B = tw_pow(P, (uint64_t)jp * 16);
        uint64_t tw = A;
        #pragma unroll
        for (int d = 0; d < 16; d++) {
            dst[out_base + (size_t)(c + 16 * d) * V] = gld::mmul(y[d], tw);
            if (d < 15) tw = gld::mmul(tw, B);
        }
    }
}

// ---- passes 2..m -------------------------------------------------------------------
// R = 16*RB rows at element stride s = 2^log_s, T = 4096/R = 256/RB consecutive words.
// SCALE (last pass only): 0 none, 1 multiply by scale_const, 2 multiply by c*hinv^k (aux tables)
// BITREV (last pass only): store element k at position bitrev(k) -- Matrix::bit_reverse_rows
// (src/matrix.rs:352-354) fused into the transform: the tile is transposed through LDS so that
// each wave still writes runs of R consecutive elements.
template <int RB, bool INV, bool LAST, int SCALE, bool BITREV = false>
__global__ void __launch_bounds__(NT, BITREV ? 4 : SCALE == 2 ? 5 : MS_NTT_WAVES) ntt_mid_pass(PassParams P) {   // SCALE 2 carries a table walk per output; BITREV holds a whole tile in LDS
    constexpr int R = 16 * RB, T = 256 / RB, G = 16 / RB;
    constexpr int LOGR = (RB == 1) ? 4 : (RB == 2) ? 5 : (RB == 4) ? 6 : (RB == 8) ? 7 : 8;
    static_assert(!BITREV || LAST, "bit-reversed store only exists for the last pass");
    const unsigned V = P.V;
    // exchange in two rounds of half a tile (see ntt_first_pass); the bit-reversed store transposes a whole tile
    constexpr int LDS_WORDS = BITREV ? T * (R + 1) : 16 * LDS_PAD_CS;
    __shared__ uint64_t lds[LDS_WORDS];
    __shared__ uint64_t twl[R];
    const uint64_t* __restrict__ src = P.src[blockIdx.y];
    uint64_t* __restrict__ dst = P.dst[blockIdx.y];
    const unsigned tid = threadIdx.x;
    const size_t sw = ((size_t)1 << P.log_s) * V;            // words per unit of this digit
    const unsigned tiles_per_u = (unsigned)(sw / T);
    const unsigned U = blockIdx.x / tiles_per_u;
    const size_t lo0 = (size_t)(blockIdx.x % tiles_per_u) * T;
    const size_t base = (size_t)U * R * sw + lo0;

    if constexpr (!LAST) {
        // per-tile twiddles w_U^k, w_U = w_(n_p)^(rev(U)) = w_n^(rev(U) * s)
        if (tid < R) {
            const uint64_t nmask = (((uint64_t)1) << P.log_n) - 1;
            uint64_t e = (((uint64_t)digit_rev(P, U) * tid) << P.log_s) & nmask;
            twl[tid] = tw_pow(P, e);
        }
    }
    uint64_t y[16];
    if constexpr (RB == 1) {
        // R = 16: a single network, no exchange
        const unsigned t = tid;
        #pragma unroll
        for (int a = 0; a < 16; a++) y[a] = src[base + (size_t)a * sw + t];
        gld::dft_lazy<16, INV>(y);
        if constexpr (!LAST) __syncthreads();
    } else {
        {
            const unsigned t = tid % T, b = tid / T;
            uint64_t x[16];
            #pragma unroll
            for (int a = 0; a < 16; a++) x[a] = src[base + (size_t)(a * RB + b) * sw + t];
            gld::dft_lazy<16, INV>(x);
            #pragma unroll
            for (int c = 0; c < 16; c++) x[c] = gld::mmul(x[c], P.wr[(b * c) & (R - 1)]);
            // rows b < RB/2 live in threads 0..127 (waves 0-1): b * T + t = tid
            const unsigned t2 = tid % T, cl = tid / T;
            constexpr int HB = RB / 2;
            if (tid < 128) {
                #pragma unroll
                for (int c = 0; c < 16; c++) lds[c * LDS_PAD_CS + tid] = x[c];
            }
            __syncthreads();
            #pragma unroll
            for (int g = 0; g < G; g++) {
                #pragma unroll
                for (int bb = 0; bb < HB; bb++) y[g * RB + bb] = lds[(cl * G + g) * LDS_PAD_CS + bb * T + t2];
            }
            __syncthreads();
            if (tid >= 128) {
                #pragma unroll
                for (int c = 0; c < 16; c++) lds[c * LDS_PAD_CS + tid - 128] = x[c];
            }
            __syncthreads();
            #pragma unroll
            for (int g = 0; g < G; g++) {
                #pragma unroll
                for (int bb = 0; bb < HB; bb++) y[g * RB + HB + bb] = lds[(cl * G + g) * LDS_PAD_CS + bb * T + t2];
            }
        }
        #pragma unroll
        for (int g = 0; g < G; g++) gld::dft_lazy<RB, INV>(y + g * RB);
    }
    // outputs: thread (t, cl) holds k = c + 16 d, c = cl*G + g, d = 0..RB-1 in y[g*RB + d]
    {
        const unsigned tid3 = gld::opaque(tid);              // keep the store addressing below the exchange
        const unsigned t = tid3 % T, cl = tid3 / T;
        if constexpr (BITREV) __syncthreads();               // phase 2 has finished reading lds
        #pragma unroll
        for (int g = 0; g < G; g++) {
            #pragma unroll
            for (int d = 0; d < RB; d++) {
                const unsigned k = (cl * G + g) + 16 * d;
                uint64_t val = y[g * RB + d];
                const size_t pos = base + (size_t)k * sw + t;
                if constexpr (!LAST) {
                    val = gld::mmul(val, twl[k]);
                } else if constexpr (SCALE == 1) {
                    val = gld::mmul(val, P.scale_const);
                } else if constexpr (SCALE == 2) {
                    val = gld::mmul(val, aux_pow(P, pos / V));
                } else {
                    val = gld::canon(val);
                }
                if constexpr (BITREV) lds[t * (R + 1) + (__brev(k) >> (32 - LOGR))] = val;
                else dst[pos] = val;
            }
        }
        if constexpr (BITREV) {
            __syncthreads();
            // lanes run along the bit-reversed k: runs of R consecutive elements per low word
            #pragma unroll 4
            for (unsigned idx = tid; idx < (unsigned)TILE; idx += NT) {
                const unsigned kk = idx % R, tt = idx / R;
                const size_t w = lo0 + tt;                            // word inside the low block (U = 0 in the last pass)
                const size_t e_low = w / V;
                const unsigned v = (unsigned)(w % V);
                const size_t e_rev = P.log_s ? (size_t)(__brevll((unsigned long long)e_low) >> (64 - P.log_s)) : 0;
                dst[((e_rev << LOGR) + kk) * V + v] = lds[tt * (R + 1) + kk];
            }
        }
    }
}

// ---- n = 2^12 / 2^13 / 2^14, Fp columns: the two passes of the (256, 16) / (256, 32) / (256, 64) plan in ONE launch ---------------------
// A column is one tile of each pass (256 rows x 16 words, then 16 rows x 256 words): the workgroup that ran ntt_first_pass on it keeps the
// result in LDS -- in the layout pass 1 stores, (j', k1) at 256 j' + k1 -- and runs ntt_mid_pass<1, .., LAST> on it from there.  The same
// instruction sequence on the same values as the two launches (bit-identical), without the round trip through scratch and, what counts at
// this size, without the second launch: a batch of 2^12-point columns is one wave of workgroups, its time is a workgroup's latency plus the
// launch (2 x 21 us for 512 columns before, profiles/r06_c2_sweep_small.json).  The reference's own bench sizes are 2^11, 2^12, 2^15, 2^18
// (gpu/benches/fft.rs:18).  SCALE as in ntt_mid_pass (1: n^-1, 2: n^-1 h^-k); P.wr / P.fields / P.log_s are pass 1's.
// The column pointers come from a TABLE in memory (cols[2 c] = source, cols[2 c + 1] = destination of column c; the host's pinned staging ring,
// read in place): one launch takes any number of columns -- with the pointers in the kernel arguments a launch ends at MAXC columns, and 256
// workgroups are one per CU.
struct FusedParams {
    const uint64_t* const* cols;
    const uint64_t* tw_lo; const uint64_t* tw_hi; const uint64_t* wr; const uint64_t* wr2; const uint64_t* aux_lo; const uint64_t* aux_hi; const uint64_t* gtab;   // wr: w_256^e (pass 1), wr2: w_R^e (pass 2)
    unsigned log_n, lo_bits, nfields, ncols;      // ncols: columns of this launch (ntt_fused_tiny packs several per workgroup)
    DigitField fields[3];
    uint64_t scale_const;
};
__device__ __forceinline__ uint64_t tw_pow(const FusedParams& P, uint64_t e) {
    uint64_t lo = P.tw_lo[e & ((1u << P.lo_bits) - 1)];
    uint64_t hi_i = e >> P.lo_bits;
    return hi_i ? gld::mmul(lo, P.tw_hi[hi_i]) : lo;
}
__device__ __forceinline__ uint64_t aux_pow(const FusedParams& P, uint64_t e) {
    uint64_t lo = P.aux_lo[e & ((1u << P.lo_bits) - 1)];
    uint64_t hi_i = e >> P.lo_bits;
    return hi_i ? gld::mmul(lo, P.aux_hi[hi_i]) : lo;
}
__device__ __forceinline__ unsigned digit_rev(const FusedParams& P, unsigned x) {
    unsigned r = 0;
    for (unsigned f = 0; f < P.nfields; f++)
        r |= ((x >> P.fields[f].in_shift) & P.fields[f].mask) << P.fields[f].out_shift;
    return r;
}
// LOGN = 12: 256 threads, the (256, 16) plan.  LOGN = 13 (14): 512 (1024) threads, the (256, 32) ((256, 64)) plan -- a column is TWO (FOUR) tiles of
// each pass (pass 1: words 0..15, 16..31, .. of every row; pass 2: low words 0..127 and 128..255 (four runs of 64)) and each 256-thread part
// of the workgroup (`half` below) runs the 256-thread code of ntt_first_pass / ntt_mid_pass<2 (4), .., LAST> on its tile.  Measured, forward /
// inverse, against the two launches: 2^12 x 512 columns 0.12 / 0.11 -> 0.21 / 0.21 of HBM, 2^13 x 512 0.15 / 0.13 -> 0.25 / 0.25, 2^14 x 256
// 0.19 / 0.18 -> 0.235 / 0.225 (profiles/r06_c2_sweep_small.json).  Exchange buffers live inside `col` (pass 1's before the column is written, pass
// 2's after it has been read into registers).
template <int LOGN, bool INV, bool COSET, int SCALE>
__global__ void __launch_bounds__(NT << (LOGN - 12), LOGN == 12 ? 5 : 4) ntt_fused_small(FusedParams P) {     // (2^14: one workgroup of sixteen waves per CU)
    static_assert(LOGN >= 12 && LOGN <= 14, "one workgroup per column: 2^12 .. 2^14 points (128 KiB of LDS at 2^14)");
    constexpr int H = 1 << (LOGN - 12), RB = H, ROW = 16 * H;                  // halves of the workgroup, pass 2's radix / 16, words per pass-1 row
    __shared__ uint64_t col[TILE * H];
    uint64_t y[16];
    const uint64_t* __restrict__ src = P.cols[2 * (size_t)blockIdx.x];
    uint64_t* __restrict__ dst = (uint64_t*)P.cols[2 * (size_t)blockIdx.x + 1];
    const unsigned tid = threadIdx.x & (NT - 1), half = threadIdx.x >> 8;
    uint64_t* const lds = col + (size_t)half * TILE;                           // this half's exchange buffer (2048 words in pass 1, 2304 in pass 2)
    {   // pass 1, phase 1 (ntt_first_pass with row_words = ROW, w0 = 16 half): thread (t, b) owns rows j1 = 16 a + b, word w0 + t
        const unsigned t = tid & 15, b = tid >> 4;
        uint64_t x[16];
        #pragma unroll
        for (int a = 0; a < 16; a++) x[a] = src[(size_t)(16 * a + b) * ROW + 16 * half + t];
        if constexpr (COSET) {
            #pragma unroll
            for (int a = 0; a < 16; a++) x[a] = gld::mmul(x[a], P.gtab[16 * a + b]);
        }
        gld::dft_lazy<16, INV>(x);
        #pragma unroll
        for (int c = 0; c < 16; c++) x[c] = gld::mmul(x[c], P.wr[(b * c) & 255]);
        const unsigned c2 = tid & 15, t2 = tid >> 4;
        const unsigned A = (t * 128 + ((b & 7) << 4)) ^ (t | ((t & 1) << 4));
        const unsigned Rb = (t2 * 128 + c2) ^ (t2 | ((t2 & 1) << 4));
        const uint64_t* rd0 = lds + Rb;
        const uint64_t* rd1 = lds + (Rb ^ 16);
        if (b < 8) {
            const unsigned A0 = gld::opaque(A);
            #pragma unroll
            for (int c = 0; c < 16; c++) lds[A0 ^ c] = x[c];
        }
        __syncthreads();
        #pragma unroll
        for (int bb = 0; bb < 8; bb++) y[bb] = (bb & 1) ? rd1[(bb & ~1) << 4] : rd0[bb << 4];
        __syncthreads();
        if (b >= 8) {
            const unsigned A1 = gld::opaque(A);
            #pragma unroll
            for (int c = 0; c < 16; c++) lds[A1 ^ c] = x[c];
        }
        __syncthreads();
        #pragma unroll
        for (int bb = 0; bb < 8; bb++) y[8 + bb] = (bb & 1) ? rd1[(bb & ~1) << 4] : rd0[bb << 4];
        __syncthreads();                                 // the exchange buffers become the column
    }
    {   // pass 1, phase 2: thread (c, t) owns b = 0..15 -> k1 = c + 16 d of word j' = 16 half + t, times (h w_n^k1)^j'
        const unsigned tid2 = gld::opaque(tid);
        const unsigned c = tid2 & 15, jp = 16 * half + (tid2 >> 4);
        gld::dft_lazy<16, INV>(y);
        const unsigned out_base = digit_rev(P, jp) << 8;
        uint64_t A = tw_pow(P, (uint64_t)jp * c);
        if constexpr (COSET) A = gld::mmul(A, aux_pow(P, jp));
        const uint64_t B = tw_pow(P, (uint64_t)jp * 16);
        uint64_t tw = A;
        #pragma unroll
        for (int d = 0; d < 16; d++) {
            col[out_base + c + 16 * d] = gld::mmul(y[d], tw);
            if (d < 15) tw = gld::mmul(tw, B);
        }
    }
    __syncthreads();
    // pass 2 (ntt_mid_pass<RB, INV, true, SCALE> with sw = 256, tile `half` of H: low words lo0 = half T .. + T)
    constexpr int R = 16 * RB, T = 256 / RB, G = 16 / RB;
    const unsigned lo0 = half * T;
    if constexpr (RB == 1) {
        const unsigned t = gld::opaque(tid);
        #pragma unroll
        for (int a = 0; a < 16; a++) y[a] = col[a * 256 + t];
        gld::dft_lazy<16, INV>(y);
    } else {
        const unsigned t = tid % T, b = tid / T;
        uint64_t x[16];
        #pragma unroll
        for (int a = 0; a < 16; a++) x[a] = col[(size_t)(a * RB + b) * 256 + lo0 + t];
        gld::dft_lazy<16, INV>(x);
        #pragma unroll
        for (int c = 0; c < 16; c++) x[c] = gld::mmul(x[c], P.wr2[(b * c) & (R - 1)]);
        __syncthreads();                                 // every word of the column is in registers: its memory is the exchange buffer now
        const unsigned t2 = tid % T, cl = tid / T;
        constexpr int HB = RB / 2;
        if (tid < 128) {
            #pragma unroll
            for (int c = 0; c < 16; c++) lds[c * LDS_PAD_CS + tid] = x[c];
        }
        __syncthreads();
        #pragma unroll
        for (int g = 0; g < G; g++) {
            #pragma unroll
            for (int bb = 0; bb < HB; bb++) y[g * RB + bb] = lds[(cl * G + g) * LDS_PAD_CS + bb * T + t2];
        }
        __syncthreads();
        if (tid >= 128) {
            #pragma unroll
            for (int c = 0; c < 16; c++) lds[c * LDS_PAD_CS + tid - 128] = x[c];
        }
        __syncthreads();
        #pragma unroll
        for (int g = 0; g < G; g++) {
            #pragma unroll
            for (int bb = 0; bb < HB; bb++) y[g * RB + HB + bb] = lds[(cl * G + g) * LDS_PAD_CS + bb * T + t2];
        }
        #pragma unroll
        for (int g = 0; g < G; g++) gld::dft_lazy<RB, INV>(y + g * RB);
    }
    {   // outputs: thread (t, cl) holds k = c + 16 d, c = cl G + g, d = 0..RB-1 in y[g RB + d]
        const unsigned tid3 = gld::opaque(tid);
        const unsigned t = tid3 % T, cl = tid3 / T;
        #pragma unroll
        for (int g = 0; g < G; g++) {
            #pragma unroll
            for (int d = 0; d < RB; d++) {
                const unsigned k = (cl * G + g) + 16 * d;
                uint64_t val = y[g * RB + d];
                const size_t pos = (size_t)k * 256 + lo0 + t;
                if constexpr (SCALE == 1) val = gld::mmul(val, P.scale_const);
                else if constexpr (SCALE == 2) val = gld::mmul(val, aux_pow(P, pos));
                else val = gld::canon(val);
                dst[pos] = val;
            }
        }
    }
}

// ---- n = 512 / 1024 / 2048, Fp columns: the (256, W) plan, W = n / 256 = 2 / 4 / 8, in one launch; 16 / W columns per workgroup -------------------
// ntt_fused_small's scheme on columns whose pass-1 rows hold only W < 16 words: a workgroup takes PACK = 16 / W columns side by side -- the 16
// "words" of a pass-1 tile row are the W words of column 0, then of column 1, ... (the arithmetic of a lane depends on its word j' = t mod W and
// on nothing else of its column, so every lane works) -- and pass 2 is one radix-W network per thread and column over the W values of its k1:
// no inner twiddle, no exchange.  Replaces ntt_small's log2(n) radix-2 stages with a barrier each (2^11 is one of the reference's own bench
// sizes and its smallest transform, gpu/benches/fft.rs:18, gpu/src/plan.rs:248).  Dispatched for 2^11 only (ms_ntt.cpp: W = 2, 4 measured slower
// than ntt_small, which also keeps Fq3 columns and zero-extended inputs).
template <int LOGN, bool INV, bool COSET, int SCALE>
__global__ void __launch_bounds__(NT, 5) ntt_fused_tiny(FusedParams P) {
    static_assert(LOGN >= 9 && LOGN <= 11, "256 rows of 2, 4 or 8 words");
    constexpr int LOGW = LOGN - 8, W = 1 << LOGW, PACK = 16 / W;
    __shared__ uint64_t col[TILE];                       // PACK columns of 256 W words; its first half doubles as pass 1's exchange buffer
    uint64_t* const lds = col;
    uint64_t y[16];
    const unsigned tid = threadIdx.x;
    const unsigned c0 = blockIdx.x * PACK;                // this workgroup's first column
    {   // pass 1, phase 1: thread (t, b) owns rows j1 = 16 a + b of word t mod W of column c0 + t / W
        const unsigned t = tid & 15, b = tid >> 4;
        const unsigned c = c0 + (t >> LOGW);
        uint64_t x[16];
        if (c < P.ncols) {
            const uint64_t* __restrict__ src = P.cols[2 * (size_t)c];
            #pragma unroll
            for (int a = 0; a < 16; a++) x[a] = src[(size_t)(16 * a + b) * W + (t & (W - 1))];
        } else {
            #pragma unroll
            for (int a = 0; a < 16; a++) x[a] = 0;
        }
        if constexpr (COSET) {
            #pragma unroll
            for (int a = 0; a < 16; a++) x[a] = gld::mmul(x[a], P.gtab[16 * a + b]);
        }
        gld::dft_lazy<16, INV>(x);
        #pragma unroll
        for (int c = 0; c < 16; c++) x[c] = gld::mmul(x[c], P.wr[(b * c) & 255]);
        const unsigned c2 = tid & 15, t2 = tid >> 4;
        const unsigned A = (t * 128 + ((b & 7) << 4)) ^ (t | ((t & 1) << 4));
        const unsigned Rb = (t2 * 128 + c2) ^ (t2 | ((t2 & 1) << 4));
        const uint64_t* rd0 = lds + Rb;
        const uint64_t* rd1 = lds + (Rb ^ 16);
        if (b < 8) {
            const unsigned A0 = gld::opaque(A);
            #pragma unroll
            for (int c = 0; c < 16; c++) lds[A0 ^ c] = x[c];
        }
        __syncthreads();
        #pragma unroll
        for (int bb = 0; bb < 8; bb++) y[bb] = (bb & 1) ? rd1[(bb & ~1) << 4] : rd0[bb << 4];
        __syncthreads();
        if (b >= 8) {
            const unsigned A1 = gld::opaque(A);
            #pragma unroll
            for (int c = 0; c < 16; c++) lds[A1 ^ c] = x[c];
        }
        __syncthreads();
        #pragma unroll
        for (int bb = 0; bb < 8; bb++) y[8 + bb] = (bb & 1) ? rd1[(bb & ~1) << 4] : rd0[bb << 4];
        __syncthreads();                                 // the exchange buffer becomes the columns
    }
    {   // pass 1, phase 2: thread (c, t) owns b = 0..15 -> k1 = c + 16 d of word j' = t mod W of column t / W, times (h w_n^k1)^j'
        const unsigned tid2 = gld::opaque(tid);
        const unsigned c = tid2 & 15, t = tid2 >> 4, jp = t & (W - 1), sub = t >> LOGW;
        gld::dft_lazy<16, INV>(y);
        const unsigned out_base = sub * (256 * W) + (digit_rev(P, jp) << 8);
        uint64_t A = tw_pow(P, (uint64_t)jp * c);
        if constexpr (COSET) A = gld::mmul(A, aux_pow(P, jp));
        const uint64_t B = tw_pow(P, (uint64_t)jp * 16);
        uint64_t tw = A;
        #pragma unroll
        for (int d = 0; d < 16; d++) {
            col[out_base + c + 16 * d] = gld::mmul(y[d], tw);
            if (d < 15) tw = gld::mmul(tw, B);
        }
    }
    __syncthreads();
    {   // pass 2: thread k1 owns the W words j' of every column; X[k1 + 256 k2] = sum_j' col[j'][k1] w_W^(j' k2)
        const unsigned t = gld::opaque(tid);
        #pragma unroll
        for (int sub = 0; sub < PACK; sub++) {
            if (c0 + sub >= P.ncols) break;               // (wave-uniform)
            uint64_t* __restrict__ dst = (uint64_t*)P.cols[2 * (size_t)(c0 + sub) + 1];
            uint64_t z[W];
            #pragma unroll
            for (int a = 0; a < W; a++) z[a] = col[sub * (256 * W) + a * 256 + t];
            gld::dft_lazy<W, INV>(z);
            #pragma unroll
            for (int k = 0; k < W; k++) {
                uint64_t val = z[k];
                const size_t pos = (size_t)k * 256 + t;
                if constexpr (SCALE == 1) val = gld::mmul(val, P.scale_const);
                else if constexpr (SCALE == 2) val = gld::mmul(val, aux_pow(P, pos));
                else val = gld::canon(val);
                dst[pos] = val;
            }
        }
    }
}

// ---- small transforms (n <= 2048): one workgroup per column, everything in LDS ----
// scale_in[j]  (forward coset)  multiplies input j   (nullptr: none)
// scale_out[k] (inverse)        multiplies output k  (nullptr: none)
// tw[i] = w_n^i, i < n/2.
struct SmallParams {
    const uint64_t* const* cols;   // [2 c] = source, [2 c + 1] = destination of column c: a table in memory (the pinned staging ring), so that one
                                   // launch takes any number of columns -- 256 workgroups, the most a by-value pointer array allows, are one per CU
    const uint64_t* tw;
    const uint64_t* scale_in;
    const uint64_t* scale_out;
    unsigned log_n;
    unsigned V;
};
static_assert(sizeof(SmallParams) <= msntt::MAX_KERNARG_BYTES, "kernel-argument block (ntt_kernels.h: MAX_KERNARG_BYTES)");
static __global__ void __launch_bounds__(NT) ntt_small(SmallParams P) {
    __shared__ uint64_t lds[2048];
    const uint64_t* __restrict__ src = P.cols[2 * (size_t)blockIdx.y];
    uint64_t* __restrict__ dst = (uint64_t*)P.cols[2 * (size_t)blockIdx.y + 1];
    const unsigned n = 1u << P.log_n, tid = threadIdx.x, V = P.V;
    for (unsigned v = 0; v < V; v++) {
        for (unsigned j = tid; j < n; j += NT) {
            uint64_t x = src[(size_t)j * V + v];
            if (P.scale_in) x = gld::mmul(x, P.scale_in[j]);
            const unsigned r = P.log_n ? (__brev(j) >> (32 - P.log_n)) : 0;
            lds[r] = x;
        }
        __syncthreads();
        for (unsigned s = 1; s <= P.log_n; s++) {
            const unsigned half = 1u << (s - 1);
            for (unsigned q = tid; q < n / 2; q += NT) {
                const unsigned i = q & (half - 1), lo = ((q >> (s - 1)) << s) + i, hi = lo + half;
                uint64_t u = lds[lo];
                uint64_t t = gld::mmul(lds[hi], P.tw[i << (P.log_n - s)]);
                lds[lo] = gl::add(u, t);
                lds[hi] = gl::sub(u, t);
            }
            __syncthreads();
        }
        for (unsigned k = tid; k < n; k += NT) {
            uint64_t x = lds[k];
            if (P.scale_out) x = gld::mmul(x, P.scale_out[k]);
            dst[(size_t)k * V + v] = x;
        }
        __syncthreads();
    }
}

// ---- bit reversal (gpu/src/metal/fft_shaders.h.metal:32-44) -----------------------
// dst[bitrev(i)] = src[i] for i < 2^log_n, elements of V words; dst may equal src.
// Index i = (hi:5 | mid | lo:5), rev(i) = (rev(lo):5 | rev(mid) | rev(hi):5).  A workgroup
// owns the pair of regions {mid, rev(mid)}: it stages both 32x32-element tiles in LDS
// and writes each to the other's region, so reads and writes are both 32 elements
// contiguous and the permutation is safe in place.
struct BitrevParams {
    const uint64_t* src[MAXC];
    uint64_t* dst[MAXC];
    unsigned log_n;
};
static_assert(sizeof(BitrevParams) <= msntt::MAX_KERNARG_BYTES, "kernel-argument block (ntt_kernels.h: MAX_KERNARG_BYTES)");
template <int V>
__global__ void __launch_bounds__(NT) bit_reverse_tiled(BitrevParams P) {
    __shared__ uint64_t tile[2][32 * 33 * V];
    const uint64_t* __restrict__ src = P.src[blockIdx.y];
    uint64_t* __restrict__ dst = P.dst[blockIdx.y];
    const unsigned log_mid = P.log_n - 10;
    const unsigned mid = blockIdx.x;
    const unsigned rmid = log_mid ? (__brev(mid) >> (32 - log_mid)) : 0;
    if (rmid < mid) return;
    const int ntile = (rmid == mid) ? 1 : 2;
    for (int q = 0; q < ntile; q++) {
        const unsigned m = q ? rmid : mid;
        for (unsigned e = threadIdx.x; e < 32 * 32 * V; e += NT) {
            const unsigned w = e % (32 * V), hi = e / (32 * V);         // w = lo*V + v
            const size_t i = (((size_t)hi << (log_mid + 5)) | ((size_t)m << 5)) * V + w;
            tile[q][hi * 33 * V + w] = src[i];
        }
    }
    __syncthreads();
    for (int q = 0; q < ntile; q++) {
        const unsigned m_out = q ? mid : rmid;                          // data of region m goes to rev(m)
        for (unsigned e = threadIdx.x; e < 32 * 32 * V; e += NT) {
            const unsigned w = e % (32 * V), lo_r = e / (32 * V);       // output row = rev(lo)
            const unsigned hi_r = w / V, v = w % V;                     // output col = rev(hi)
            const unsigned lo = __brev(lo_r) >> 27, hi = __brev(hi_r) >> 27;
            const size_t o = (((size_t)lo_r << (log_mid + 5)) | ((size_t)m_out << 5)) * V + w;
            dst[o] = tile[q][hi * 33 * V + lo * V + v];
        }
    }
}
// small sizes (log_n < 10): one thread per element, OUT OF PLACE only (dst != src)
template <int V>
__global__ void __launch_bounds__(NT) bit_reverse_simple(BitrevParams P) {
    const uint64_t* __restrict__ src = P.src[blockIdx.y];
    uint64_t* __restrict__ dst = P.dst[blockIdx.y];
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= ((size_t)1 << P.log_n)) return;
    const size_t r = P.log_n ? (size_t)(__brevll((unsigned long long)i) >> (64 - P.log_n)) : 0;
    for (int v = 0; v < V; v++) dst[r * V + v] = src[i * V + v];
}

}  // namespace msntt
