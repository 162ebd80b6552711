// Goldilocks arithmetic on four redundant 24-bit limbs, for the radix-16 networks of the NTT (gfx950).
//
// Why: on gfx950 a plain 32-bit VOP2 add/sub/logic op issues in ~2.4 cycles per wave64, every carry
// producing/consuming op, every VOP3 and v_mad_u64_u32 in ~4.2 (profiles/r01_ubench*_instr_rates.txt).
// A modular add on a 64-bit residue costs 5 carry-class instructions (~21 cycles); a butterfly network
// is almost only adds and multiplications by powers of two.  So inside a network a value is kept as
//
//      x  =  l0 + l1*2^24 + l2*2^48 + l3*2^72      (mod 2^96 + 1),   l_i signed 32-bit
//
// p = 2^64 - 2^32 + 1 divides 2^96 + 1, so arithmetic mod 2^96 + 1 is exact mod p.  Then
//   * add / sub        = 4 x v_add_u32 / v_sub_u32, no carries (8 bits of headroom per limb),
//   * times 2^(24 m)   = a rotation of the limbs with a sign flip on the wrapped ones (2^96 = -1):
//                        free, it only renames registers and swaps an add for a sub,
//   * times 2^12       = one pass of (and, ashr, shift-add) per limb, which also renormalises,
// and arkworks' 16th root of unity is -2^60 (gl_dev.h), so a radix-16 network needs nothing else.
//
// The general twiddle w that follows a network is applied WITHOUT leaving the representation first:
//      x*w = sum_i l_i * (w * 2^(24 i) mod p)
// i.e. eight v_mad_u64_u32 against four pre-multiplied copies W_i of the twiddle, accumulated carry-free
// into two 64-bit columns (lo words, hi words); `fold` brings the 95-bit result to a weak 64-bit residue.
// 2^24 = w_8^5 (w_8 = arkworks' 8th root), so the W_i are entries of the SAME twiddle table at
// exponent offsets 5 i n/8 -- no extra tables.  Limbs must be non-negative for the unsigned
// multiply-adds: a bias congruent to 0 (32 * (2^96 + 1), spread as ~2^29 per limb) is added to input 0
// of every network, which reaches every output with coefficient 1.
//
// Restates nothing of the reference (its butterflies are one Montgomery product + add + sub per pair,
// fft_shaders.h.metal:13-31); results are bit-identical because every step is exact mod p.
#pragma once
#include "gl.h"

namespace glimb {

typedef unsigned __int128 u128;
static constexpr uint32_t M24 = 0xFFFFFFu;

template <int N> struct ic { static constexpr int value = N; };
struct L4 { uint32_t l[4]; };          // two's-complement limbs; value = sum l[i] * 2^(24 i)  (mod 2^96 + 1)

// bias ≡ 0 (mod 2^96 + 1):  32 * (2^96 + 1) = (2^29 + 32) + (2^29 - 32) 2^24 + (2^29 - 32) 2^48 + (2^29 - 32) 2^72
static constexpr uint32_t BIAS0 = (1u << 29) + 32u, BIASK = (1u << 29) - 32u;

#if defined(__HIP_DEVICE_COMPILE__)
// v_perm_b32: byte i of the result = byte sel[i] of {hi:lo} (0..3 = lo, 4..7 = hi, 0x0c = zero)
__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
#else
MS_HD uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) {
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t s = (sel >> (8 * i)) & 0xFF;
        const uint32_t b = (s < 8) ? (uint32_t)((v >> (8 * s)) & 0xFF) : 0u;
        r |= b << (8 * i);
    }
    return r;
}
#endif
static constexpr uint32_t SEL_345 = 0x0c050403u;     // bits 24..47 of {hi:lo}
static constexpr uint32_t SEL_234 = 0x0c040302u;     // bits 16..39 of {hi:lo}

// any 64-bit residue -> limbs (l0, l1 < 2^24, l2 < 2^16, l3 = 0)
MS_HD L4 from_u64(uint64_t x) {
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    L4 r;
    r.l[0] = lo & M24;
    r.l[1] = perm(hi, lo, SEL_345);
    r.l[2] = hi >> 16;
    r.l[3] = 0;
    return r;
}

// a * t (both any 64-bit residues) -> limbs, |l_i| < 2^24.  The 128-bit product is cut at 24-bit
// boundaries; its top word (weight 2^96 = -1) is subtracted from the two low limbs.
MS_HD L4 mul_to_limbs(uint64_t a, uint64_t t) {
    const u128 pr = (u128)a * t;
    const uint32_t w0 = (uint32_t)pr, w1 = (uint32_t)(pr >> 32), w2 = (uint32_t)(pr >> 64), w3 = (uint32_t)(pr >> 96);
    L4 r;
    r.l[0] = (w0 & M24) - (w3 & M24);
    r.l[1] = perm(w1, w0, SEL_345) - (w3 >> 24);
    r.l[2] = perm(w2, w1, SEL_234);
    r.l[3] = w2 >> 8;
    return r;
}

MS_HD void add_bias(L4& x) { x.l[0] += BIAS0; x.l[1] += BIASK; x.l[2] += BIASK; x.l[3] += BIASK; }

// x * 2^12, renormalised: |out_i| < 2^24 + |x|_max / 2^12.  Limbs are signed: floor split l = hi*2^12 + lo.
MS_HD L4 half_shift(const L4& x) {
    uint32_t lo[4], hi[4];
    #pragma unroll
    for (int i = 0; i < 4; i++) { lo[i] = x.l[i] & 0xFFFu; hi[i] = (uint32_t)((int32_t)x.l[i] >> 12); }
    L4 r;
    r.l[0] = (lo[0] << 12) - hi[3];                 // hi[3] has weight 2^96 = -1
    r.l[1] = (lo[1] << 12) + hi[0];
    r.l[2] = (lo[2] << 12) + hi[1];
    r.l[3] = (lo[3] << 12) + hi[2];
    return r;
}

// One DIT butterfly  (u, v) -> (u + t, u - t),  t = v * 2^S  (S a multiple of 12, 0 <= S < 192).
// 2^96 = -1 folds into the add/sub choice; 2^(24 m) into which limb pairs with which.
template <int S>
MS_HD void bfly(L4& u, L4& v) {
    static_assert(S >= 0 && S < 192 && S % 12 == 0, "twiddle must be a power of 2^12");
    constexpr bool NEG = (S >= 96);
    constexpr int s = NEG ? S - 96 : S;
    constexpr int m = s / 24;
    L4 w = v;
    if constexpr ((s % 24) != 0) w = half_shift(v);
    L4 a, b;
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        const int j = (i - m) & 3;
        const bool wrap = (i < m);                   // limb came around 2^96
        const bool minus = (wrap != NEG);            // t_i = -w_j
        a.l[i] = minus ? u.l[i] - w.l[j] : u.l[i] + w.l[j];
        b.l[i] = minus ? u.l[i] + w.l[j] : u.l[i] - w.l[j];
    }
    u = a; v = b;
}

// exponent of 2 for w_16^E:  forward root w_16 = 2^156, inverse root 2^36  (gl_dev.h W16)
template <bool INV, int E> struct Sh16 { static constexpr int value = INV ? ((36 * E) % 192) : ((156 * E) % 192); };

template <int N>
MS_HD void bitrev_regs(L4* x) {
    constexpr int LOGN = (N == 2) ? 1 : (N == 4) ? 2 : (N == 8) ? 3 : 4;
    #pragma unroll
    for (int i = 0; i < N; i++) {
        int r = 0;
        #pragma unroll
        for (int b = 0; b < LOGN; b++) r |= ((i >> b) & 1) << (LOGN - 1 - b);
        if (r > i) { L4 t = x[i]; x[i] = x[r]; x[r] = t; }
    }
}

// In-register DFT of N <= 16 values, natural order in and out:  X[c] = sum_a x[a] w_N^(a c), w_N = w_16^(16/N).
// Inputs |l_i| < 2^24 (+ the bias on x[0] when BIAS); outputs: bias + a signed swing < 2^28.2 per limb.
template <int N, bool INV, bool BIAS = true>
MS_HD void dft(L4* x) {
    if constexpr (N == 1) { if constexpr (BIAS) add_bias(x[0]); return; }
    if constexpr (BIAS) add_bias(x[0]);
    bitrev_regs<N>(x);
    #pragma unroll
    for (int blk = 0; blk < N; blk += 2) bfly<0>(x[blk], x[blk + 1]);
    if constexpr (N >= 4) {
        #pragma unroll
        for (int blk = 0; blk < N; blk += 4) {
            bfly<0>(x[blk], x[blk + 2]);
            bfly<Sh16<INV, 4>::value>(x[blk + 1], x[blk + 3]);
        }
    }
    if constexpr (N >= 8) {
        #pragma unroll
        for (int blk = 0; blk < N; blk += 8) {
            bfly<0>(x[blk], x[blk + 4]);
            bfly<Sh16<INV, 2>::value>(x[blk + 1], x[blk + 5]);
            bfly<Sh16<INV, 4>::value>(x[blk + 2], x[blk + 6]);
            bfly<Sh16<INV, 6>::value>(x[blk + 3], x[blk + 7]);
        }
    }
    if constexpr (N >= 16) {
        bfly<0>(x[0], x[8]);
        bfly<Sh16<INV, 1>::value>(x[1], x[9]);
        bfly<Sh16<INV, 2>::value>(x[2], x[10]);
        bfly<Sh16<INV, 3>::value>(x[3], x[11]);
        bfly<Sh16<INV, 4>::value>(x[4], x[12]);
        bfly<Sh16<INV, 5>::value>(x[5], x[13]);
        bfly<Sh16<INV, 6>::value>(x[6], x[14]);
        bfly<Sh16<INV, 7>::value>(x[7], x[15]);
    }
}

// Radix-16 network whose inputs x[a] are zero for a >= NA (zero-padded LDE input, NA in {1, 2, 4}):
//   X[M c2 + c1] = sum_{a<NA} (x_a w^(a c1)) w_NA^(a c2),  M = 16/NA.   x[0..NA) in, x[0..16) out.
template <int NA, bool INV>
MS_HD void dft16_pruned(L4* x) {
    static_assert(NA == 1 || NA == 2 || NA == 4, "prune factor");
    add_bias(x[0]);
    if constexpr (NA == 1) {
        #pragma unroll
        for (int c = 1; c < 16; c++) x[c] = x[0];
    } else if constexpr (NA == 2) {
        const L4 x0 = x[0], x1 = x[1];
        auto one = [&](auto C1) {
            constexpr int c1 = decltype(C1)::value;
            L4 u = x0, v = x1;
            bfly<Sh16<INV, c1>::value>(u, v);
            x[c1] = u; x[c1 + 8] = v;
        };
        one(ic<0>{}); one(ic<1>{}); one(ic<2>{}); one(ic<3>{});
        one(ic<4>{}); one(ic<5>{}); one(ic<6>{}); one(ic<7>{});
    } else {
        const L4 x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
        auto quad = [&](auto C1) {
            constexpr int c1 = decltype(C1)::value;
            L4 s0 = x0, s1 = x2, p = x1, q = x3;
            bfly<Sh16<INV, 2 * c1>::value>(s0, s1);          // x0 +- w^(2 c1) x2
            bfly<Sh16<INV, 2 * c1>::value>(p, q);            // x1 +- w^(2 c1) x3
            bfly<Sh16<INV, c1>::value>(s0, p);               // X[c1], X[c1 + 8]
            bfly<Sh16<INV, c1 + 4>::value>(s1, q);           // X[c1 + 4], X[c1 + 12]
            x[c1] = s0; x[c1 + 8] = p; x[c1 + 4] = s1; x[c1 + 12] = q;
        };
        quad(ic<0>{}); quad(ic<1>{}); quad(ic<2>{}); quad(ic<3>{});
    }
}

// ---- leaving the representation ---------------------------------------------------------------------
// Four pre-multiplied copies of a twiddle: W[i] = w * 2^(24 i) mod p (any representative < 2^64), as 32-bit halves.
struct W4 { uint32_t lo[4], hi[4]; };

MS_HD W4 w4_from(uint64_t w0, uint64_t w1, uint64_t w2, uint64_t w3) {
    W4 r;
    r.lo[0] = (uint32_t)w0; r.hi[0] = (uint32_t)(w0 >> 32);
    r.lo[1] = (uint32_t)w1; r.hi[1] = (uint32_t)(w1 >> 32);
    r.lo[2] = (uint32_t)w2; r.hi[2] = (uint32_t)(w2 >> 32);
    r.lo[3] = (uint32_t)w3; r.hi[3] = (uint32_t)(w3 >> 32);
    return r;
}

// (acc_lo, acc_hi) with value acc_lo + acc_hi * 2^32 < 2^96  ->  64-bit residue.
//   y = a0 + (a1 + b0) 2^32 + (b1 + carry) 2^64;   2^64 = 2^32 - 1 (mod p):   z = (m : a0) + t * (2^32 - 1), t < 2^32.
//   z is computed wrapping (one v_mad_u64_u32); it overflowed iff z < (m : a0), and then z < 2^63.1 + 2^32, so the
//   +EPS that accounts for the lost 2^64 cannot overflow again -- and lands below p.  CANON: the same +EPS also
//   maps a z in [p, 2^64) to z - p, so the canonical form costs one more compare.
template <bool CANON>
MS_HD uint64_t fold_t(uint64_t acc_lo, uint64_t acc_hi) {
    const uint32_t a0 = (uint32_t)acc_lo, a1 = (uint32_t)(acc_lo >> 32), b0 = (uint32_t)acc_hi, b1 = (uint32_t)(acc_hi >> 32);
    const uint32_t m = a1 + b0;
    const uint32_t t = b1 + (uint32_t)(m < a1);
    const uint64_t base = ((uint64_t)m << 32) | a0;
    const uint64_t z = (uint64_t)t * 0xFFFFFFFFull + base;
    bool fix = z < base;
    if (CANON) fix = fix || (z >= gl::P);
    return z + (fix ? gl::EPS : 0ull);
}
MS_HD uint64_t fold(uint64_t acc_lo, uint64_t acc_hi) { return fold_t<false>(acc_lo, acc_hi); }

// x * w as a weak residue.  Limbs must be in [0, 2^30) (network outputs carry the bias).
template <bool CANON = false>
MS_HD uint64_t mul_fold(const L4& x, const W4& w) {
    uint64_t alo = (uint64_t)x.l[0] * w.lo[0], ahi = (uint64_t)x.l[0] * w.hi[0];
    #pragma unroll
    for (int i = 1; i < 4; i++) { alo += (uint64_t)x.l[i] * w.lo[i]; ahi += (uint64_t)x.l[i] * w.hi[i]; }
    return fold_t<CANON>(alo, ahi);
}

// ---- round 3: the same product with the middle carry inside the multiply-adds --------------------------------------
// The second accumulator starts from the high word of the first, so the product is  a0 + H 2^32  with H < 2^64 exact
// (limbs < 2^30, halves < 2^32: both sums stay below 2^64) and only H's high word h1 (weight 2^64 = EPS) is left to fold:
//      z = (h0 : a0) + h1 EPS          one v_mad_u64_u32 whose CARRY-OUT (VOP3B sdst) says whether 2^64 was lost,
//      z += carry ? EPS : 0            a second one; it cannot wrap again: z < h1 EPS <= 2^64 - 2^33 + 1 after a wrap.
// Against fold_t: no 32-bit add / compare / add-with-carry for the middle word, no 64-bit compare, no 64-bit add.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint64_t fold_h(uint32_t a0, uint64_t H) {
    const uint64_t base = ((uint64_t)(uint32_t)H << 32) | a0;
    const uint32_t h1 = (uint32_t)(H >> 32);
    uint64_t z, cm;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(z), "=s"(cm) : "v"(h1), "v"(base));      // -1 = 0xFFFFFFFF = EPS
    uint32_t c01;
    asm("v_cndmask_b32 %0, 0, 1, %1" : "=v"(c01) : "s"(cm));
    return (uint64_t)c01 * 0xFFFFFFFFull + z;
}
#else
MS_HD uint64_t fold_h(uint32_t a0, uint64_t H) {
    const uint64_t base = ((uint64_t)(uint32_t)H << 32) | a0;
    const u128 t = (u128)(uint32_t)(H >> 32) * 0xFFFFFFFFull + base;
    return (uint64_t)t + ((uint64_t)(t >> 64) ? gl::EPS : 0ull);
}
#endif
template <bool CANON = false>
MS_HD uint64_t mul_fold_co(const L4& x, const W4& w) {
    uint64_t alo = (uint64_t)x.l[0] * w.lo[0];
    #pragma unroll
    for (int i = 1; i < 4; i++) alo += (uint64_t)x.l[i] * w.lo[i];
    uint64_t H = (uint64_t)x.l[0] * w.hi[0] + (alo >> 32);
    #pragma unroll
    for (int i = 1; i < 4; i++) H += (uint64_t)x.l[i] * w.hi[i];
    const uint64_t z = fold_h((uint32_t)alo, H);
    return (CANON && z >= gl::P) ? z + gl::EPS : z;
}

// A factor on the way INTO a network: x (any 64-bit residue, cut into limbs of 24, 24, 16 bits) times q given as three
// pre-shifted copies Q_i = q 2^(24 i) mod p.  The 90-bit result  a0 + H 2^32  is cut at the limb boundaries directly (no
// reduction at all: limbs are a representation mod 2^96 + 1), 14 instructions against the 21 of mul_to_limbs.
struct Q3 { uint32_t lo[3], hi[3]; };
MS_HD Q3 q3_from(uint64_t q0, uint64_t q1, uint64_t q2) {
    Q3 r;
    r.lo[0] = (uint32_t)q0; r.hi[0] = (uint32_t)(q0 >> 32);
    r.lo[1] = (uint32_t)q1; r.hi[1] = (uint32_t)(q1 >> 32);
    r.lo[2] = (uint32_t)q2; r.hi[2] = (uint32_t)(q2 >> 32);
    return r;
}
MS_HD L4 mul3_to_limbs(uint64_t x, const Q3& q) {
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    const uint32_t l0 = lo & M24, l1 = perm(hi, lo, SEL_345), l2 = hi >> 16;
    const uint64_t alo = (uint64_t)l0 * q.lo[0] + (uint64_t)l1 * q.lo[1] + (uint64_t)l2 * q.lo[2];      // < 2^58
    uint64_t H = (uint64_t)l0 * q.hi[0] + (alo >> 32);
    H += (uint64_t)l1 * q.hi[1];
    H += (uint64_t)l2 * q.hi[2];                                                                        // < 2^58
    const uint32_t a0 = (uint32_t)alo, h0 = (uint32_t)H, h1 = (uint32_t)(H >> 32);
    L4 r;
    r.l[0] = a0 & M24;
    r.l[1] = perm(h0, a0, SEL_345);
    r.l[2] = perm(h1, h0, SEL_234);
    r.l[3] = h1 >> 8;
    return r;
}

// x as a weak residue (no twiddle): the W_i are the constants 2^(24 i) mod p
//   1, 2^24, 2^48, 2^72 = 2^40 - 2^8.
template <bool CANON = false>
MS_HD uint64_t to_weak(const L4& x) {
    const uint64_t alo = (uint64_t)x.l[0] + (uint64_t)x.l[1] * 0x1000000ull + (uint64_t)x.l[3] * 0xFFFFFF00ull;
    const uint64_t ahi = (uint64_t)x.l[2] * 0x10000ull + (uint64_t)x.l[3] * 0xFFull;
    return fold_t<CANON>(alo, ahi);
}
MS_HD uint64_t to_canon(const L4& x) { return to_weak<true>(x); }

}  // namespace glimb
