// DEEP composition on device: DeepPolyComposer::{get_ood_evals, into_deep_poly}
// (src/composer.rs:43-188), which the reference runs on the host, sequentially per column
// (horner_evaluate src/utils.rs:124-133, divide_out_point(s)_into src/utils.rs:151-175).
//
//  * horner_blocks: out-of-domain evaluations P_c(x) for (column, point) queries.  One workgroup per
//    4096 coefficients and query group: lane t sums c[t + 256 k] (x^256)^k over its sixteen coefficients
//    (coalesced loads), weights the sum with x^t and the workgroup adds the 256 values up in LDS; the block
//    values are the coefficients of a polynomial in x^4096 and go through the same kernel again (ms_deep.cpp).
//  * deep_points: the reference builds  Q(X) = sum_t alpha_t (P_ct(X) - P_ct(z_t)) / (X - z_t)  by synthetic
//    division in coefficient space.  Q has degree <= n-2, so it is determined by its values on any n
//    points: this kernel evaluates the sum at the n points of the coset offset*<w_n> from coset
//    evaluations of the P_c (one lane per point, the few 1/(x - z_k) by one batched Fq3 inversion), and
//    an inverse coset NTT returns exactly the coefficients synthetic division produces (exact field
//    arithmetic, hence bit-identical).
//  * deep_degree_adjust: coefficients of Q(X) * (alpha + beta X)  (src/composer.rs:164-185).
// Points and results live in Fq3 (or in Fp for Fq = Fp AIRs: PW = 1).
#pragma once
#include <hip/hip_runtime.h>
#include "gl.h"
#include "gl_dev.h"
#include "stage_kernels.h"

namespace msdeep {

static constexpr int NT = 256;
static constexpr int MAXCOLS = 96;
static constexpr int MAXPOINTS = 8;

using F1 = msstage::FpT;
using F3 = msstage::Fq3T;

// element of the point field: PW = 1 (Fp) or 3 (Fq3), stored in 3 words
struct Q { uint64_t w[3]; };
template <int PW> __device__ __forceinline__ Q q_zero() { return {{0, 0, 0}}; }
template <int PW> __device__ __forceinline__ Q q_add(const Q& a, const Q& b) {
    if constexpr (PW == 1) return {{gl::add(a.w[0], b.w[0]), 0, 0}};
    else return {{gl::add(a.w[0], b.w[0]), gl::add(a.w[1], b.w[1]), gl::add(a.w[2], b.w[2])}};
}
template <int PW> __device__ __forceinline__ Q q_sub(const Q& a, const Q& b) {
    if constexpr (PW == 1) return {{gl::sub(a.w[0], b.w[0]), 0, 0}};
    else return {{gl::sub(a.w[0], b.w[0]), gl::sub(a.w[1], b.w[1]), gl::sub(a.w[2], b.w[2])}};
}
template <int PW> __device__ __forceinline__ Q q_mul(const Q& a, const Q& b) {
    if constexpr (PW == 1) return {{gld::mmul(a.w[0], b.w[0]), 0, 0}};
    else { const gl::Fq3 r = F3::mul({a.w[0], a.w[1], a.w[2]}, {b.w[0], b.w[1], b.w[2]}); return {{r.c0, r.c1, r.c2}}; }
}
template <int PW> __device__ __forceinline__ Q q_inv(const Q& a) {
    if constexpr (PW == 1) return {{F1::inv(a.w[0]), 0, 0}};
    else { const gl::Fq3 r = F3::inv({a.w[0], a.w[1], a.w[2]}); return {{r.c0, r.c1, r.c2}}; }
}
// load a coefficient / evaluation of a CW-word column as an element of the point field
template <int CW> __device__ __forceinline__ Q q_load(const uint64_t* col, size_t i) {
    if constexpr (CW == 1) return {{col[i], 0, 0}};
    else return {{col[3 * i], col[3 * i + 1], col[3 * i + 2]}};
}

struct HornerParams {
    const uint64_t* cols[MAXCOLS];
    const uint32_t* qcol;      // device: column of each query
    uint64_t* partial;         // device: [nq][nblocks][3]
    size_t n;
    unsigned nblocks, ngroups;
    const uint32_t* group;     // device: [ngroups][2] = first query, number of queries (<= GQ, same column, adjacent)
    // host-built tables per query (the same for every lane of the launch: wave-uniform loads); Montgomery form, 3 words each
    const uint64_t* ypow;      // [nq][16]: (x^256)^k          the factor of a lane's k-th coefficient
    const uint32_t* ylimb;     // [nq][16][3 words][4]: 22 / 22 / 20-bit limbs of ypow's words (Fp coefficient columns)
    const uint64_t* xlo;       // [nq][16]: x^i                 x^t = xlo[t & 15] xhi[t >> 4], the lane's weight in the block
    const uint64_t* xhi;       // [nq][16]: x^(16 i)
    // second level: the block values of the level below are the coefficients (3 words each), query q reads its own row of them
    const uint64_t* self_src;  // [nq][self_nblocks][3] or null
    unsigned self_nblocks;
};
// (ov 2^128 + hi 2^64 + lo) 2^-64 mod p, canonical: the Montgomery reduction of the low 128 bits (felt_u64.h.metal:165-177 as in
// gl::mont_mul) with hi brought below p first, plus ov 2^64 = ov (2^32 - 1)
__device__ __forceinline__ uint64_t reduce_132(uint64_t lo, uint64_t hi, uint32_t top) {
    const uint64_t xl = lo, xh = hi >= gl::P ? hi - gl::P : hi;
    const uint64_t s = xl + (xl << 32);
    const uint64_t ov = s < xl;
    const uint64_t bb = s - (s >> 32) - ov;
    const uint64_t r = xh - bb;
    return gl::add((xh < bb) ? r + gl::P : r, (uint64_t)top * 0xFFFFFFFFull);
}
// sum_k c_k Y_k for sixteen Fp coefficients and wave-uniform factors, without a single carry: c = c0 + 2^32 c1, Y = y0 + 2^22 y1 + 2^44 y2
// (22 / 22 / 20 bits), six columns S[3 i + j] += c_i y_j of 54-bit products -- sixteen of them stay below 2^58 --, one
// v_mad_u64_u32 each; the columns are put together (< 2^134) and reduced once.
__device__ __forceinline__ void limb_mac(uint64_t* S, uint64_t c, const uint32_t* y) {
    const uint32_t c0 = (uint32_t)c, c1 = (uint32_t)(c >> 32);
    S[0] += (uint64_t)c0 * y[0]; S[1] += (uint64_t)c0 * y[1]; S[2] += (uint64_t)c0 * y[2];
    S[3] += (uint64_t)c1 * y[0]; S[4] += (uint64_t)c1 * y[1]; S[5] += (uint64_t)c1 * y[2];
}
__device__ __forceinline__ uint64_t limb_sum_reduce(const uint64_t* S) {
    typedef unsigned __int128 u128;
    const u128 low = (u128)S[0] + ((u128)S[1] << 22) + ((u128)S[2] << 44) + ((u128)S[3] << 32) + ((u128)S[4] << 54);   // < 2^113
    const u128 top = (u128)(S[5] & ((1ull << 52) - 1)) << 76;
    const u128 sum = low + top;
    const uint32_t over = (uint32_t)(S[5] >> 52) + (sum < top ? 1u : 0u);
    return reduce_132((uint64_t)sum, (uint64_t)(sum >> 64), over);
}
// CW: words per coefficient (1 Fp, 3 Fq3); PW: words of the point field (PW >= CW)
// Lane t of block b takes coefficients b*4096 + t + 256*k, k < 16 (a wave reads 64 consecutive coefficients per load):
// A_t = sum_k c[t + 256k] y^k with y = x^256, and the block value is sum_t A_t x^t: every lane weights its A_t with x^t (two table
// factors) and the 256 values are ADDED up through LDS (the last five levels inside one wave).
// Fp coefficients (CW = 1): A_t per component of the point field is limb_mac / limb_sum_reduce above: 6 multiply-adds per coefficient,
// component and query instead of a dependent modular product (nine for an Fq3 point).  Fq3 coefficients (the block values of the
// level below, extension columns) run the Horner chain in y.
// GQ: up to GQ queries on the SAME column (adjacent in the query list) form a group: the workgroup reads its 4096 coefficients once.
template <int CW, int PW, int GQ>
__global__ void __launch_bounds__(NT) horner_blocks(HornerParams P) {
    __shared__ uint64_t sh[NT * 3];
    const unsigned g = blockIdx.x % P.ngroups, b = blockIdx.x / P.ngroups, t = threadIdx.x;
    const unsigned q0 = P.group[2 * g], cnt = P.group[2 * g + 1];          // wave-uniform
    const uint64_t* col = P.self_src ? P.self_src + (size_t)q0 * P.self_nblocks * 3 : P.cols[P.qcol[q0]];
    unsigned qj[GQ];
    #pragma unroll
    for (int j = 0; j < GQ; j++) qj[j] = q0 + (j < (int)cnt ? j : 0);
    Q acc[GQ];
    const size_t start = (size_t)b * 4096 + t;
    if constexpr (CW == 1) {
        uint64_t S[GQ][PW][6];
        #pragma unroll
        for (int j = 0; j < GQ; j++) {
            #pragma unroll
            for (int w = 0; w < PW; w++) {
                #pragma unroll
                for (int i = 0; i < 6; i++) S[j][w][i] = 0;
            }
        }
        // all sixteen loads first (a block lives for one memory latency, not four): clamped addresses, values past the end zeroed
        uint64_t c[16];
        #pragma unroll
        for (int k = 0; k < 16; k++) {
            const size_t i = start + (size_t)k * NT;
            c[k] = col[i < P.n ? i : P.n - 1];
        }
        [[maybe_unused]] constexpr int UNR = PW == 1 ? 16 : 4;   // (an Fq3 point keeps 18 scalar factor words per step: fully unrolled they spill)
        #pragma unroll UNR
        for (int k = 0; k < 16; k++) {
            if (start + (size_t)k * NT >= P.n) c[k] = 0;
            #pragma unroll
            for (int j = 0; j < GQ; j++) {
                #pragma unroll
                for (int w = 0; w < PW; w++) limb_mac(S[j][w], c[k], P.ylimb + (((size_t)qj[j] * 16 + k) * 3 + w) * 4);
            }
        }
        #pragma unroll
        for (int j = 0; j < GQ; j++) {
            acc[j] = q_zero<PW>();
            #pragma unroll
            for (int w = 0; w < PW; w++) acc[j].w[w] = limb_sum_reduce(S[j][w]);
        }
    } else {
        Q y[GQ];
        #pragma unroll
        for (int j = 0; j < GQ; j++) {
            const uint64_t* yp = P.ypow + ((size_t)qj[j] * 16 + 1) * 3;
            y[j] = {{yp[0], yp[1], yp[2]}};
            acc[j] = q_zero<PW>();
        }
        // only the steps the block has coefficients for (a second level of 1 024 block values: 4 of 16)
        const size_t left = P.n - (size_t)b * 4096;
        const int steps = left >= 4096 ? 16 : (int)((left + NT - 1) / NT);
        #pragma unroll 4
        for (int k = steps - 1; k >= 0; k--) {
            const size_t i = start + (size_t)k * NT;
            Q c = q_zero<PW>();
            if (i < P.n) c = q_load<CW>(col, i);
            #pragma unroll
            for (int j = 0; j < GQ; j++) acc[j] = q_add<PW>(q_mul<PW>(acc[j], y[j]), c);
        }
    }
    // per query: sum_t A_t x^t
    #pragma unroll
    for (int j = 0; j < GQ; j++) {
        if (j >= (int)cnt) break;
        if (j > 0) __syncthreads();                 // the previous query's last levels may still be reading
        const uint64_t* lo = P.xlo + ((size_t)qj[j] * 16 + (t & 15)) * 3;
        const uint64_t* hi = P.xhi + ((size_t)qj[j] * 16 + (t >> 4)) * 3;
        Q a = q_mul<PW>(acc[j], q_mul<PW>(Q{{lo[0], lo[1], lo[2]}}, Q{{hi[0], hi[1], hi[2]}}));
        // level s: lanes t < 2 s hold values, lanes t < s add the value of lane t + s.  s = 128, 64 cross waves; from 32 on wave 0 alone
        #pragma unroll
        for (unsigned s_ = NT / 2; s_ >= 1; s_ >>= 1) {
            if (t < 2 * s_) { sh[3 * t] = a.w[0]; sh[3 * t + 1] = a.w[1]; sh[3 * t + 2] = a.w[2]; }
            if (s_ >= 64) __syncthreads(); else gld::wave_lockstep();
            if (t < s_) a = q_add<PW>(a, Q{{sh[3 * (t + s_)], sh[3 * (t + s_) + 1], sh[3 * (t + s_) + 2]}});
        }
        if (t == 0) {
            uint64_t* o = P.partial + ((size_t)qj[j] * P.nblocks + b) * 3;
            o[0] = a.w[0]; o[1] = a.w[1]; o[2] = a.w[2];
        }
    }
}

// col: < nbase base, else ext.  alimb: the 22 / 22 / 20-bit limbs of alpha's words (limb_mac's second operand; filled by the host)
struct Term { uint32_t col, point; uint64_t alpha[3]; uint64_t ood[3]; uint32_t alimb[3][4]; };
struct DeepParams {
    const uint64_t* base[MAXCOLS];     // coset evaluations, natural order, n x Fp
    const uint64_t* ext[MAXCOLS];      //                                  n x Fq3
    const Term* terms;                 // device
    const uint64_t* tw_lo;             // w_n^i two-level table (Montgomery)
    const uint64_t* tw_hi;
    uint64_t points[MAXPOINTS][3];
    uint64_t* out;                     // n x PW words
    uint64_t h_mont;
    size_t n;
    unsigned nbase, nterms, npoints, lo_bits, xshift;
    unsigned term_start[MAXPOINTS + 1];   // terms are sorted by point: those of point k are [term_start[k], term_start[k + 1])
    uint64_t csum[MAXPOINTS][3];          // sum_{t: pt = k} alpha_t ood_t (host): the constant part of a point's numerator
    // ms_deep_rows: the n inputs are ROWS [first, first + n) of LDE columns over a domain of 2^log_dom points in bit-reversed
    // order (log_dom = 0: natural order from 0, the coset of ms_deep_compose); adjust: the result is multiplied by
    // (adj_alpha + adj_beta x), the degree adjustment of src/composer.rs:170-186 applied pointwise
    size_t first;
    unsigned log_dom, adjust;
    uint64_t adj_alpha[3], adj_beta[3];
};
// PTS points per lane (points i, i + NT, ... of the workgroup's NT * PTS: coalesced): their PTS * npoints denominators x - z_k
// share ONE inversion (Montgomery's trick: 72 products for Fp, more for Fq3, against 3 per denominator), and the terms arrive
// sorted by point so that a point's quotient factor multiplies the SUM of its terms:
//     sum_t alpha_t (P_ct(x) - ood_t) / (x - z_pt)  =  sum_k 1/(x - z_k)  sum_{t: pt = k} alpha_t (P_ct(x) - ood_t)
// -- nterms + npoints products per point instead of 2 nterms.  Exact field arithmetic: the same value.
template <int PW, int PTS, int MP, int WAVES = 4>       // MP: the most distinct points this instantiation serves (array sizes); WAVES: waves per workgroup = denominators' pools per inversion
__global__ void __launch_bounds__(64 * WAVES) deep_points(DeepParams P) {
    constexpr int NT = 64 * WAVES;                  // (shadows the namespace's 256: the launch passes 64 WAVES threads)
    const size_t i0 = (size_t)blockIdx.x * (NT * PTS) + threadIdx.x;
    Q d[PTS][MP], pre[PTS][MP];
    uint64_t xv[PTS];
    Q run = {{gl::ONE_MONT, 0, 0}};
    #pragma unroll
    for (int j = 0; j < PTS; j++) {
        const size_t i = i0 + (size_t)j * NT;
        size_t nat = i < P.n ? i : 0;
        if (P.log_dom) nat = (size_t)(__brevll((unsigned long long)(P.first + nat)) >> (64 - P.log_dom));      // position -> point index
        const size_t e = nat << P.xshift;
        uint64_t xs = P.tw_lo[e & ((1u << P.lo_bits) - 1)];
        if (e >> P.lo_bits) xs = gld::mmul(xs, P.tw_hi[e >> P.lo_bits]);
        xs = gld::mmul(xs, P.h_mont);
        xv[j] = xs;
        const Q x = {{xs, 0, 0}};
        #pragma unroll
        for (int k = 0; k < MP; k++) if (k < (int)P.npoints) {
            d[j][k] = q_sub<PW>(x, Q{{P.points[k][0], P.points[k][1], P.points[k][2]}});
            pre[j][k] = run;
            run = q_mul<PW>(run, d[j][k]);
        }
    }
    // The inversion (a Fermat power: ~73 dependent products for Fp, the largest single item of this kernel when it serves only the
    // PTS * npoints denominators of one lane) is pooled over the workgroup: lane l of ONE wave inverts the product of the four waves'
    // lane-l products and hands each its own inverse back (nine more products on that wave, two barriers) -- a quarter of the
    // inversions (an eighth with WAVES = 8: the Fp launch of ms_deep_rows).  The inverting wave rotates with the workgroup so that the serial chains spread over a CU's SIMDs.
    static_assert(WAVES == 4 || WAVES == 8, "the inverting wave is blockIdx.x & (WAVES - 1)");
    __shared__ uint64_t pool[NT * PW];
    {
        const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        #pragma unroll
        for (int w = 0; w < PW; w++) pool[(wv * PW + w) * 64 + lane] = run.w[w];
        __syncthreads();
        if (wv == (blockIdx.x & (WAVES - 1))) {
            Q a[WAVES], pr[WAVES];                               // pr[v] = a0 .. av
            #pragma unroll
            for (int v = 0; v < WAVES; v++) {
                a[v] = q_zero<PW>();
                #pragma unroll
                for (int w = 0; w < PW; w++) a[v].w[w] = pool[(v * PW + w) * 64 + lane];
                pr[v] = v ? q_mul<PW>(pr[v - 1], a[v]) : a[0];
            }
            Q t = q_inv<PW>(pr[WAVES - 1]);
            #pragma unroll
            for (int v = WAVES - 1; v >= 0; v--) {
                const Q r = v ? q_mul<PW>(t, pr[v - 1]) : t;     // 1 / a_v
                if (v) t = q_mul<PW>(t, a[v]);                   // 1 / (a0 .. a(v-1))
                #pragma unroll
                for (int w = 0; w < PW; w++) pool[(v * PW + w) * 64 + lane] = r.w[w];
            }
        }
        __syncthreads();
    }
    Q inv = q_zero<PW>();
    #pragma unroll
    for (int w = 0; w < PW; w++) inv.w[w] = pool[((threadIdx.x >> 6) * PW + w) * 64 + (threadIdx.x & 63)];
    #pragma unroll
    for (int j = PTS - 1; j >= 0; j--) {
        #pragma unroll
        for (int k = MP - 1; k >= 0; k--) if (k < (int)P.npoints) {
            const Q dk = d[j][k];
            d[j][k] = q_mul<PW>(inv, pre[j][k]);          // now 1 / (x_j - z_k)
            inv = q_mul<PW>(inv, dk);
        }
    }
    // terms outermost, the lane's PTS points innermost: PTS independent loads per term.  A point's numerator is
    //     sum_t alpha_t (P_ct(x) - ood_t)  =  sum_t alpha_t P_ct(x)  -  csum_k:
    // for Fp columns the first sum is accumulated as UNREDUCED 54-bit limb products (limb_mac: six multiply-adds per term and component,
    // no carry, no reduction) and reduced once per sixteen terms -- the dependent modular product per term was two thirds of this kernel
    // (round 4; the Horner kernel's trick).  Extension columns keep the Fq3 product.  Exact arithmetic either way: the same value.
    Q acc[PTS];
    #pragma unroll
    for (int j = 0; j < PTS; j++) acc[j] = q_zero<PW>();
    #pragma unroll
    for (int k = 0; k < MP; k++) if (k < (int)P.npoints) {
        Q sum[PTS];
        uint64_t S[PTS][PW][6];
        #pragma unroll
        for (int j = 0; j < PTS; j++) {
            sum[j] = q_zero<PW>();
            #pragma unroll
            for (int w = 0; w < PW; w++) {
                #pragma unroll
                for (int i = 0; i < 6; i++) S[j][w][i] = 0;
            }
        }
        auto flush = [&]() {
            #pragma unroll
            for (int j = 0; j < PTS; j++) {
                Q r = q_zero<PW>();
                #pragma unroll
                for (int w = 0; w < PW; w++) {
                    r.w[w] = limb_sum_reduce(S[j][w]);
                    #pragma unroll
                    for (int i = 0; i < 6; i++) S[j][w][i] = 0;
                }
                sum[j] = q_add<PW>(sum[j], r);
            }
        };
        // windows of at most sixteen terms (the limb columns' headroom); inside a window the loop is unrolled by four so that the loads of
        // four terms are in flight together (one memory latency per four terms instead of one per term)
        for (unsigned t0 = P.term_start[k]; t0 < P.term_start[k + 1]; t0 += 16) {  // wave-uniform bounds and terms
            const unsigned t1 = t0 + 16 < P.term_start[k + 1] ? t0 + 16 : P.term_start[k + 1];
            #pragma unroll 4
            for (unsigned t = t0; t < t1; t++) {
                const Term& T = P.terms[t];
                if (T.col < P.nbase) {
                    uint64_t v[PTS];
                    #pragma unroll
                    for (int j = 0; j < PTS; j++) {
                        const size_t i = i0 + (size_t)j * NT < P.n ? i0 + (size_t)j * NT : 0;
                        v[j] = P.base[T.col][i];
                    }
                    #pragma unroll
                    for (int j = 0; j < PTS; j++) {
                        #pragma unroll
                        for (int w = 0; w < PW; w++) limb_mac(S[j][w], v[j], T.alimb[w]);
                    }
                } else if constexpr (PW == 3) {
                    const Q alpha = {{T.alpha[0], T.alpha[1], T.alpha[2]}};
                    #pragma unroll
                    for (int j = 0; j < PTS; j++) {
                        const size_t i = i0 + (size_t)j * NT < P.n ? i0 + (size_t)j * NT : 0;
                        sum[j] = q_add<PW>(sum[j], q_mul<PW>(q_load<3>(P.ext[T.col - P.nbase], i), alpha));
                    }
                }
            }
            flush();
        }
        const Q ck = {{P.csum[k][0], P.csum[k][1], P.csum[k][2]}};
        #pragma unroll
        for (int j = 0; j < PTS; j++) acc[j] = q_add<PW>(acc[j], q_mul<PW>(q_sub<PW>(sum[j], ck), d[j][k]));
    }
    #pragma unroll
    for (int j = 0; j < PTS; j++) {
        const size_t i = i0 + (size_t)j * NT;
        if (i >= P.n) continue;
        if (P.adjust) {
            const Q a = {{P.adj_alpha[0], P.adj_alpha[1], P.adj_alpha[2]}}, b = {{P.adj_beta[0], P.adj_beta[1], P.adj_beta[2]}};
            Q bx = q_zero<PW>();
            #pragma unroll
            for (int w = 0; w < PW; w++) bx.w[w] = gld::mmul(b.w[w], xv[j]);
            acc[j] = q_mul<PW>(acc[j], q_add<PW>(a, bx));
        }
        #pragma unroll
        for (int w = 0; w < PW; w++) P.out[PW * i + w] = acc[j].w[w];
    }
}

// out_i = alpha * c_i + beta * c_(i-1)   (c_-1 = 0), in place over a separate copy
template <int PW>
__global__ void __launch_bounds__(NT) deep_degree_adjust(uint64_t* dst, const uint64_t* src, size_t n, Q alpha, Q beta) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    Q c = q_zero<PW>(), prev = q_zero<PW>();
    #pragma unroll
    for (int w = 0; w < PW; w++) { c.w[w] = src[PW * i + w]; if (i) prev.w[w] = src[PW * (i - 1) + w]; }
    const Q r = q_add<PW>(q_mul<PW>(c, alpha), q_mul<PW>(prev, beta));
    #pragma unroll
    for (int w = 0; w < PW; w++) dst[PW * i + w] = r.w[w];
}

}  // namespace msdeep

// ---- the same three kernels for the 252-bit field (Fq = Fp = Fp252: one word size, no extension) ----------
namespace msdeep252 {

static constexpr int NT = 256;
using E = f252::E;
__device__ __forceinline__ E ld(const uint64_t* p, size_t i) { return {{p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]}}; }
__device__ __forceinline__ void st(uint64_t* p, size_t i, const E& x) { p[4 * i] = x.l[0]; p[4 * i + 1] = x.l[1]; p[4 * i + 2] = x.l[2]; p[4 * i + 3] = x.l[3]; }

struct HornerParams {
    const uint64_t* cols[msdeep::MAXCOLS];
    const uint32_t* qcol;
    const uint64_t* qpoint;    // 4 words per query
    uint64_t* partial;         // [nq][nblocks][4]
    size_t n;
    unsigned nblocks;
};
static __global__ void __launch_bounds__(NT) horner_blocks(HornerParams P) {
    __shared__ uint64_t sh[NT * 4];
    const unsigned q = blockIdx.y, b = blockIdx.x, t = threadIdx.x;
    const uint64_t* col = P.cols[P.qcol[q]];
    E xp[9];
    xp[0] = ld(P.qpoint, q);
    for (int l = 1; l <= 8; l++) xp[l] = f252::mul(xp[l - 1], xp[l - 1]);
    const size_t start = (size_t)b * 4096 + t;
    E acc = f252::zero();
    for (int k = 15; k >= 0; k--) {
        const size_t i = start + (size_t)k * NT;
        acc = f252::mul(acc, xp[8]);
        if (i < P.n) acc = f252::add(acc, ld(col, i));
    }
    for (unsigned l = 0; l < 8; l++) {
        st(sh, t, acc);
        __syncthreads();
        const unsigned step = 1u << l;
        if ((t & (2 * step - 1)) == 0) acc = f252::add(acc, f252::mul(ld(sh, t + step), xp[l]));
        __syncthreads();
    }
    if (t == 0) st(P.partial, (size_t)q * P.nblocks + b, acc);
}

struct Term { uint32_t col, point; uint64_t alpha[4]; uint64_t ood[4]; };
struct DeepParams {
    const uint64_t* cols[msdeep::MAXCOLS];   // coset evaluations, natural order
    const Term* terms;
    const uint64_t* tw_lo;                   // w_n^i tables of the forward 252 plan
    const uint64_t* tw_hi;
    uint64_t points[msdeep::MAXPOINTS][4];
    uint64_t h[4];                           // coset offset
    uint64_t* out;
    size_t n;
    unsigned nterms, npoints, lo_bits;
};
static __global__ void __launch_bounds__(NT) deep_points(DeepParams P) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= P.n) return;
    E x = ld(P.tw_lo, i & ((1u << P.lo_bits) - 1));
    if (i >> P.lo_bits) x = f252::mul(x, ld(P.tw_hi, i >> P.lo_bits));
    x = f252::mul(x, E{{P.h[0], P.h[1], P.h[2], P.h[3]}});
    // 1 / (x - z_k) for every point with one inversion
    E d[msdeep::MAXPOINTS], pre[msdeep::MAXPOINTS];
    E run = f252::one();
    for (unsigned k = 0; k < P.npoints; k++) {
        d[k] = f252::sub(x, E{{P.points[k][0], P.points[k][1], P.points[k][2], P.points[k][3]}});
        pre[k] = run;
        run = f252::mul(run, d[k]);
    }
    E inv = f252::inv(run);
    for (int k = (int)P.npoints - 1; k >= 0; k--) {
        const E dk = d[k];
        d[k] = f252::mul(inv, pre[k]);
        inv = f252::mul(inv, dk);
    }
    E acc = f252::zero();
    for (unsigned t = 0; t < P.nterms; t++) {
        const Term T = P.terms[t];
        E v = f252::sub(ld(P.cols[T.col], i), E{{T.ood[0], T.ood[1], T.ood[2], T.ood[3]}});
        acc = f252::add(acc, f252::mul(f252::mul(v, d[T.point]), E{{T.alpha[0], T.alpha[1], T.alpha[2], T.alpha[3]}}));
    }
    st(P.out, i, acc);
}
static __global__ void __launch_bounds__(NT) deep_degree_adjust(uint64_t* dst, const uint64_t* src, size_t n, E alpha, E beta) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    const E c = ld(src, i), prev = i ? ld(src, i - 1) : f252::zero();
    st(dst, i, f252::add(f252::mul(c, alpha), f252::mul(prev, beta)));
}

}  // namespace msdeep252
