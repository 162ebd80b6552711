// Throughput-oriented Goldilocks primitives for the NTT kernels (gfx950).
//
// Measured issue costs (profiles/r01_ubench*_instr_rates.txt): plain 32-bit VOP2 ops
// (v_add_u32, v_sub_u32, logic, v_lshrrev_b32, v_mov) 2 cycles per wave64; everything that
// produces or consumes a carry / mask, any VOP3, v_mad_u64_u32 and the 64-bit add/compare: 4.
// So the primitives below minimise *instruction count* of the 4-cycle class, and keep values
// "weak" (any u64 congruent to the residue) wherever a canonical value is not needed:
//
//   mmul(a, bm)      a weak, bm CANONICAL Montgomery-form constant (w*2^64 mod p)
//                    -> CANONICAL a*w mod p          (Montgomery reduction is exact for a < 2^64)
//   add_lazy(u, t)   u weak, t CANONICAL  -> weak u+t        (one overflow fix is enough)
//   sub_lazy(u, t)   u weak, t CANONICAL  -> weak u-t
//   canon(x)         weak -> canonical
//
// Written as portable C++ on 32-bit limbs with clang's carry builtins, so hipcc emits
// v_add_co/v_addc_co chains (and inserts the gfx950 VALU->VCC wait states itself) and g++
// compiles the same source for the host-side simulator tests.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <stdint.h>
#endif
#include "gl.h"

namespace gld {

typedef unsigned __int128 u128;
static constexpr uint32_t EPS32 = 0xFFFFFFFFu;
template <int N> struct int_c { static constexpr int value = N; };     // compile-time index passed by value

MS_HD uint32_t lo32(uint64_t x) { return (uint32_t)x; }
MS_HD uint32_t hi32(uint64_t x) { return (uint32_t)(x >> 32); }
MS_HD uint64_t mk64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

#if defined(__clang__)
MS_HD uint32_t addc(uint32_t a, uint32_t b, uint32_t cin, uint32_t* cout) { return __builtin_addc(a, b, cin, cout); }
MS_HD uint32_t subc(uint32_t a, uint32_t b, uint32_t bin, uint32_t* bout) { return __builtin_subc(a, b, bin, bout); }
#else   // g++ (simulator build): same semantics, no builtin before GCC 14
MS_HD uint32_t addc(uint32_t a, uint32_t b, uint32_t cin, uint32_t* cout) {
    const uint64_t s = (uint64_t)a + b + cin; *cout = (uint32_t)(s >> 32); return (uint32_t)s;
}
MS_HD uint32_t subc(uint32_t a, uint32_t b, uint32_t bin, uint32_t* bout) {
    const uint64_t s = (uint64_t)a - b - bin; *bout = (uint32_t)(s >> 63); return (uint32_t)s;
}
#endif

// Scheduling fence for one 32-bit value: the compiler may not move anything that depends on the result
// above this point (used to keep table loads of a later phase from being hoisted over an LDS exchange,
// where they would only pin registers).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ unsigned opaque(unsigned x) { asm volatile("" : "+v"(x)); return x; }
#else
MS_HD unsigned opaque(unsigned x) { asm volatile("" : "+r"(x)); return x; }
#endif

// Montgomery product (felt_u64.h.metal:165-177 restated on limbs):
//   s = xl + (xl << 32); b = s - (s >> 32) - carry; r = xh - b; if borrow r += p
MS_HD uint64_t mmul(uint64_t a, uint64_t bm) {
    const u128 x = (u128)a * bm;
    const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), x2 = (uint32_t)(x >> 64), x3 = (uint32_t)(x >> 96);
    uint32_t e, f, g, h;
    const uint32_t ah = addc(x1, x0, 0, &e);          // s = (ah : x0), carry e
    const uint32_t bl = subc(x0, ah, e, &f);          // b = s - ah - e
    const uint32_t bh = subc(ah, 0, f, &g);           // (g is always 0)
    uint32_t rl = subc(x2, bl, 0, &f);                // r = xh - b
    uint32_t rh = subc(x3, bh, f, &g);
    const uint32_t m = 0u - g;                        // borrow -> subtract EPS (= add p)
    rl = subc(rl, m, 0, &f);
    rh = subc(rh, 0, f, &h);
    return mk64(rl, rh);
}

// (64-bit form: v_lshl_add_u64 + v_cmp_lt_u64 measured 21 cycles vs 24 for the limb chain)
MS_HD uint64_t add_lazy(uint64_t u, uint64_t t) {
    const uint64_t s = u + t;
    return (s < t) ? s + gl::EPS : s;                 // overflow -> add EPS
}
MS_HD uint64_t sub_lazy(uint64_t u, uint64_t t) {
    uint32_t c, d;
    uint32_t sl = subc(lo32(u), lo32(t), 0, &c);
    uint32_t sh = subc(hi32(u), hi32(t), c, &d);
    const uint32_t m = 0u - d;                        // borrow -> subtract EPS
    sl = subc(sl, m, 0, &c);
    sh = subc(sh, 0, c, &d);
    return mk64(sl, sh);
}
// weak -> canonical.  Deliberately a 64-bit compare, NOT "carry-out of x + EPS": with the
// limb form hipcc (ROCm 7.2) folds addc(hi, 0, c) into the addition that produced `hi` and
// then uses the carry-out of the merged three-input add -- a wrong value whenever `hi` itself
// had wrapped (found by scripts/dbg_bfly.hip, device != host on the same source).
MS_HD uint64_t canon(uint64_t x) { return x >= gl::P ? x + gl::EPS : x; }

// x * 2^S mod p for the in-network twiddles, S in {12,24,...,84}; canonical result.
// (general path: through the Montgomery multiplier with the constant 2^S * 2^64 mod p)
template <int S>
MS_HD uint64_t mul_pow2(uint64_t x) {
    static_assert(S > 0 && S < 96, "shift out of range");
    if constexpr (S >= 64) {
        // x*2^S = (x << r) * 2^64, r = S-64:  y0*2^64 + y1*2^96 + y2*2^128 = y0*EPS - (y2:y1)
        constexpr int r = S - 64;
        const uint32_t y0 = lo32(x) << r;
        const uint32_t y1 = (uint32_t)(x >> (32 - r));
        const uint32_t y2 = hi32(x) >> (32 - r);
        const uint64_t t = (uint64_t)y0 * EPS32;        // < p
        uint32_t c, d;
        uint32_t rl = subc(lo32(t), y1, 0, &c);
        uint32_t rh = subc(hi32(t), y2, c, &d);
        const uint32_t m = 0u - d;
        rl = subc(rl, m, 0, &c);
        rh = subc(rh, 0, c, &d);
        return mk64(rl, rh);                            // in [0, p): see DESIGN.md
    } else {
        if constexpr (S <= 32) {
            // x*2^S = L + H*2^64, H < 2^32:  L + H*EPS, one overflow fix, then canonical
            const uint64_t L = x << S;
            const uint32_t H = (uint32_t)(x >> (64 - S));
            const u128 t = (u128)H * EPS32 + L;
            uint64_t r = (uint64_t)t;
            if ((uint64_t)(t >> 64)) r += gl::EPS;
            return canon(r);          // 47 cycles vs 57 through the multiplier (profiles/r01_ubench3_*)
        } else {
            // 32 < S < 64: the plain shift-and-reduce form measured SLOWER (76 cycles) than the
            // Montgomery multiplier with the constant 2^S * 2^64 mod p (57 cycles)
            constexpr uint64_t CM = (uint64_t)(((u128)1 << (S + 64)) % gl::P);
            return mmul(x, CM);
        }
    }
}

// ---- radix-16 networks ---------------------------------------------------------------
// w_16 = 2^156 = -2^60 is arkworks' 16th root of unity; w_16^e for e = 0..7 as +-2^s:
//   e:      0   1     2     3    4    5    6     7
//   fwd:    1  -2^60 -2^24  2^84 2^48 2^12 -2^72 -2^36
//   inv:    1   2^36  2^72 -2^12 -2^48 -2^84 2^24  2^60       (w_16^-1 = 2^36)
// A negative sign is folded into the butterfly by swapping its add and sub.
template <bool INV, int E> struct W16 {
    static constexpr int shift = INV ? ((36 * E) % 192) : ((156 * E) % 192);
    static constexpr bool negate = shift >= 96;
    static constexpr int s = negate ? shift - 96 : shift;
};

// reference network: canonical arithmetic everywhere (kept as the checked baseline)
__device__ __attribute__((unused)) static const uint64_t W16_FWD_PLAIN[8] = {
    1ull, 17293822564807737345ull, 18446744069397807105ull, 4503599626321920ull,
    281474976710656ull, 4096ull, 18446742969902956801ull, 18446744000695107585ull};
__device__ __attribute__((unused)) static const uint64_t W16_INV_PLAIN[8] = {
    1ull, 68719476736ull, 1099511627520ull, 18446744069414580225ull,
    18446462594437873665ull, 18442240469788262401ull, 16777216ull, 1152921504606846976ull};

template <int N>
MS_HD void bitrev_regs(uint64_t* x) {
    constexpr int LOGN = (N == 2) ? 1 : (N == 4) ? 2 : (N == 8) ? 3 : 4;
    #pragma unroll
    for (int i = 0; i < N; i++) {
        int r = 0;
        #pragma unroll
        for (int b = 0; b < LOGN; b++) r |= ((i >> b) & 1) << (LOGN - 1 - b);
        if (r > i) { uint64_t t = x[i]; x[i] = x[r]; x[r] = t; }
    }
}

template <bool INV>
__device__ __forceinline__ void dft16_ref(uint64_t* x) {
    bitrev_regs<16>(x);
    const uint64_t* W = INV ? W16_INV_PLAIN : W16_FWD_PLAIN;
    #pragma unroll
    for (int s = 1; s <= 4; s++) {
        const int half = 1 << (s - 1);
        #pragma unroll
        for (int blk = 0; blk < 16; blk += 2 * half) {
            #pragma unroll
            for (int i = 0; i < half; i++) {
                const int e = i * (16 >> s);
                uint64_t u = x[blk + i];
                uint64_t t = (e == 0) ? x[blk + i + half] : gl::mul(x[blk + i + half], W[e]);
                x[blk + i] = gl::add(u, t);
                x[blk + i + half] = gl::sub(u, t);
            }
        }
    }
}

// one DIT butterfly with twiddle w_16^E (as +-2^s): inputs u weak, v weak; outputs weak.
// FIRST: v is known canonical (stage 1: values straight from memory / a multiplier).
template <bool INV, int E, bool V_CANON>
MS_HD void bfly(uint64_t& u, uint64_t& v) {
    uint64_t t;
    if constexpr (E == 0) t = V_CANON ? v : canon(v);
    else t = mul_pow2<W16<INV, E>::s>(v);
    const uint64_t a = add_lazy(u, t), b = sub_lazy(u, t);
    if constexpr (E != 0 && W16<INV, E>::negate) { u = b; v = a; } else { u = a; v = b; }
}

// In-register DFT of N <= 16 values, natural order in and out.  Inputs CANONICAL, outputs WEAK.
//   X[c] = sum_a x[a] w_N^(a c),   w_N = w_16^(16/N)   (INV: the inverse root)
// PREBITREV: x[] already holds the inputs in bit-reversed index order (skip the renaming).
template <int N, bool INV, bool PREBITREV = false>
MS_HD void dft_lazy(uint64_t* x) {
    if constexpr (N == 1) return;
    if constexpr (!PREBITREV) bitrev_regs<N>(x);
    // stage 1: all twiddles 1, inputs canonical
    #pragma unroll
    for (int blk = 0; blk < N; blk += 2) bfly<INV, 0, true>(x[blk], x[blk + 1]);
    if constexpr (N >= 4) {
        #pragma unroll
        for (int blk = 0; blk < N; blk += 4) {
            bfly<INV, 0, false>(x[blk], x[blk + 2]);
            bfly<INV, 4, false>(x[blk + 1], x[blk + 3]);
        }
    }
    if constexpr (N >= 8) {
        #pragma unroll
        for (int blk = 0; blk < N; blk += 8) {
            bfly<INV, 0, false>(x[blk], x[blk + 4]);
            bfly<INV, 2, false>(x[blk + 1], x[blk + 5]);
            bfly<INV, 4, false>(x[blk + 2], x[blk + 6]);
            bfly<INV, 6, false>(x[blk + 3], x[blk + 7]);
        }
    }
    if constexpr (N >= 16) {
        bfly<INV, 0, false>(x[0], x[8]);
        bfly<INV, 1, false>(x[1], x[9]);
        bfly<INV, 2, false>(x[2], x[10]);
        bfly<INV, 3, false>(x[3], x[11]);
        bfly<INV, 4, false>(x[4], x[12]);
        bfly<INV, 5, false>(x[5], x[13]);
        bfly<INV, 6, false>(x[6], x[14]);
        bfly<INV, 7, false>(x[7], x[15]);
    }
}
template <bool INV>
MS_HD void dft16(uint64_t* x) { dft_lazy<16, INV>(x); }

// Radix-16 network whose inputs x[a] are zero for a >= NA (zero-padded LDE input, NA in {1,2,4}):
//   X[M c2 + c1] = sum_{a<NA} (x_a w^(a c1)) w_NA^(a c2),  M = 16/NA
// i.e. M twisted size-NA transforms; each is a handful of the same butterflies.  x[0..NA) canonical
// in, x[0..16) weak out (natural order).
template <int NA, bool INV>
MS_HD void dft16_pruned(uint64_t* x) {
    static_assert(NA == 1 || NA == 2 || NA == 4, "prune factor");
    if constexpr (NA == 1) {
        #pragma unroll
        for (int c = 1; c < 16; c++) x[c] = x[0];
    } else if constexpr (NA == 2) {
        const uint64_t x0 = x[0], x1 = x[1];
        #pragma unroll
        for (int c1 = 0; c1 < 8; c1++) {
            uint64_t u = x0, v = x1;
            if (c1 == 0) bfly<INV, 0, true>(u, v);
            if (c1 == 1) bfly<INV, 1, true>(u, v);
            if (c1 == 2) bfly<INV, 2, true>(u, v);
            if (c1 == 3) bfly<INV, 3, true>(u, v);
            if (c1 == 4) bfly<INV, 4, true>(u, v);
            if (c1 == 5) bfly<INV, 5, true>(u, v);
            if (c1 == 6) bfly<INV, 6, true>(u, v);
            if (c1 == 7) bfly<INV, 7, true>(u, v);
            x[c1] = u; x[c1 + 8] = v;
        }
    } else {
        const uint64_t x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
        auto quad = [&](auto C1) {
            constexpr int c1 = decltype(C1)::value;
            uint64_t s0 = x0, s1 = x2, p = x1, q = x3;
            bfly<INV, 2 * c1, true>(s0, s1);          // x0 +- w^(2 c1) x2
            bfly<INV, 2 * c1, true>(p, q);            // x1 +- w^(2 c1) x3
            bfly<INV, c1, false>(s0, p);              // X[c1], X[c1 + 8]
            bfly<INV, c1 + 4, false>(s1, q);          // X[c1 + 4], X[c1 + 12]
            x[c1] = s0; x[c1 + 8] = p; x[c1 + 4] = s1; x[c1 + 12] = q;
        };
        quad(int_c<0>{});
        quad(int_c<1>{});
        quad(int_c<2>{});
        quad(int_c<3>{});
    }
}

// "Every lane of this wave has executed the accesses above before any lane executes those below."  On the hardware that holds by
// construction (a wave issues each instruction for all its lanes, in order, and its LDS / memory instructions execute in order),
// so this is only a fence for the compiler's scheduler; the simulator of tests/emu runs the lanes one after the other between
// barriers and needs a real one.
#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
__device__ __forceinline__ void wave_lockstep() { __builtin_amdgcn_wave_barrier(); }
#else
inline void wave_lockstep() { __syncthreads(); }
#endif

// The 64-bit word the neighbouring lane (lane ^ 1) holds: one DPP move per half (quad_perm [1, 0, 3, 2]), no LDS.  Both lanes of the
// pair must be active.  The simulator has no lock-step lanes: there the exchange goes through a block-wide buffer between two barriers
// (every live thread of the block calls it the same number of times).
#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
__device__ __forceinline__ uint64_t lane_xor1(uint64_t v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(uint32_t)v, 0xB1, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(uint32_t)(v >> 32), 0xB1, 0xF, 0xF, true);
    return ((uint64_t)hi << 32) | lo;
}
#else
static inline uint64_t lane_xor1(uint64_t v) {
    static uint64_t buf[1024];
    buf[threadIdx.x] = v;
    __syncthreads();
    const uint64_t r = buf[threadIdx.x ^ 1];
    __syncthreads();
    return r;
}
#endif

// Which of its wave's 64 consecutive elements a lane takes when the elements are wider than one store instruction carries (32-byte
// elements, 16 bytes per lane and instruction): even lanes the first 32 (lane / 2), odd lanes the second 32.  After the lanes of a pair
// swap halves (lane_xor1) lane l holds bytes 16 l .. 16 l + 15 of the wave's first KiB and of its second: two fully contiguous stores.
__device__ __forceinline__ unsigned wave_elem(unsigned tid) { return (tid & ~63u) | ((tid & 1u) << 5) | ((tid & 63u) >> 1); }

}  // namespace gld
