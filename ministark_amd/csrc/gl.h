// Goldilocks field  p = 2^64 - 2^32 + 1  for gfx950 (device) and the host-side
// plan code.  Replaces the reference's Metal field class
// gpu/src/metal/felt_u64.h.metal:9-178 (Fp) and :183-279 (Fq3).
//
// Memory format is the reference's (arkworks): Montgomery residues a*2^64 mod p,
// canonical in [0,p).  Two multiplications are provided:
//   mont_mul(aR, bR) = abR      -- element (x) element, what the stage kernels need
//   mul(a, w)        = a*w      -- plain product mod p.  The NTT is linear, so
//                                  NTT(xR) = R*NTT(x): kernels treat the stored
//                                  words as plain residues and multiply them by
//                                  PLAIN twiddles; no conversion at the boundary.
//
// gfx950 cost model measured with scripts/ubench.hip (profiles/r01_ubench_*):
// v_add_u32-class VOP2 = 2 cycles per wave64, everything else we need
// (v_mad_u64_u32, v_lshl_add_u64, v_cmp_*_u64, carry ops, VOP3) = 4 cycles.
// So the code below prefers the single-instruction 64-bit add/compare forms
// the compiler emits for plain uint64_t arithmetic over carry chains.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <stdint.h>
#else                            // hiprtc (eval_jit.h) has no system headers
typedef unsigned char uint8_t;
typedef unsigned int uint32_t;
typedef int int32_t;
typedef unsigned long long uint64_t;
typedef long long int64_t;
#endif

#if defined(__HIPCC__)
#define MS_HD __host__ __device__ __forceinline__
#else
#define MS_HD inline
#endif

namespace gl {

typedef unsigned __int128 u128;
static constexpr uint64_t P = 0xFFFFFFFF00000001ull;
static constexpr uint64_t EPS = 0xFFFFFFFFull;          // 2^64 mod p = 2^32 - 1 = Montgomery ONE
static constexpr uint64_t ONE_MONT = EPS;
static constexpr uint64_t R2 = 18446744065119617025ull; // 2^128 mod p
static constexpr uint64_t GENERATOR = 7;
static constexpr uint64_t TWO_ADIC_ROOT = 1753635133440165772ull; // 7^((p-1)/2^32), canonical

// canonical + canonical -> canonical
MS_HD uint64_t add(uint64_t a, uint64_t b) {
    uint64_t t = P - b;            // a + b = a - (p - b)
    uint64_t x = a - t;
    return (a < t) ? x + P : x;
}
MS_HD uint64_t sub(uint64_t a, uint64_t b) {
    uint64_t x = a - b;
    return (a < b) ? x + P : x;
}
MS_HD uint64_t neg(uint64_t a) { return a ? P - a : 0; }

// any u64 + canonical -> u64 congruent mod p (may be >= p)
MS_HD uint64_t add_lazy(uint64_t a, uint64_t b_canon) {
    uint64_t s = a + b_canon;
    return (s < b_canon) ? s + EPS : s;
}
// any u64 - canonical -> u64 congruent mod p
MS_HD uint64_t sub_lazy(uint64_t a, uint64_t b_canon) {
    uint64_t d = a - b_canon;
    return (a < b_canon) ? d - EPS : d;
}
MS_HD uint64_t canon(uint64_t a) { return a >= P ? a - P : a; }

// 128-bit value (hi:lo) -> canonical residue.   2^64 = EPS, 2^96 = -1 (mod p)
MS_HD uint64_t reduce128(uint64_t lo, uint64_t hi) {
    uint32_t hh = (uint32_t)(hi >> 32), hl = (uint32_t)hi;
    uint64_t t0 = lo - hh;
    if (lo < hh) t0 -= EPS;                       // +p
    uint64_t t1 = ((uint64_t)hl << 32) - hl;      // hl * EPS
    uint64_t r = t0 + t1;
    if (r < t1) r += EPS;                         // -p (cannot wrap twice: t1 <= 2^64 - 2^33 + 1)
    return canon(r);
}
// plain product a*b mod p, any u64 inputs, canonical output
MS_HD uint64_t mul(uint64_t a, uint64_t b) {
    u128 x = (u128)a * b;
    return reduce128((uint64_t)x, (uint64_t)(x >> 64));
}
// Montgomery product a*b*2^-64 mod p (felt_u64.h.metal:165-177), canonical in/out
MS_HD uint64_t mont_mul(uint64_t a, uint64_t b) {
    u128 x = (u128)a * b;
    uint64_t xl = (uint64_t)x, xh = (uint64_t)(x >> 64);
    uint64_t s = xl + (xl << 32);
    uint64_t ov = s < xl;
    uint64_t bb = s - (s >> 32) - ov;
    uint64_t r = xh - bb;
    return (xh < bb) ? r + P : r;
}
MS_HD uint64_t to_mont(uint64_t canon_v) { return mont_mul(canon_v, R2); }
MS_HD uint64_t from_mont(uint64_t m) { return mont_mul(m, 1); }

// plain-domain helpers (host plan code and a few device paths)
MS_HD uint64_t pow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) { if (e & 1) r = mul(r, a); a = mul(a, a); e >>= 1; }
    return r;
}
MS_HD uint64_t inv(uint64_t a) { return pow(a, P - 2); }
MS_HD uint64_t root_of_unity(unsigned log_n) {       // canonical, arkworks get_root_of_unity(2^log_n)
    uint64_t r = TWO_ADIC_ROOT;
    for (unsigned i = log_n; i < 32; i++) r = mul(r, r);
    return r;
}

// Montgomery-domain helpers for the element-wise stage kernels
MS_HD uint64_t mont_pow(uint64_t a, uint64_t e) {
    uint64_t r = ONE_MONT;
    while (e) { if (e & 1) r = mont_mul(r, a); a = mont_mul(a, a); e >>= 1; }
    return r;
}
MS_HD uint64_t mont_sqn(uint64_t a, int n) { for (int i = 0; i < n; i++) a = mont_mul(a, a); return a; }
// x^(p-2), addition chain with 72 multiplications (felt_u64.h.metal:97-109); inv(0) = 0
MS_HD uint64_t mont_inv(uint64_t x) {
    uint64_t t2 = mont_mul(mont_sqn(x, 1), x);
    uint64_t t3 = mont_mul(mont_sqn(t2, 1), x);
    uint64_t t6 = mont_mul(mont_sqn(t3, 3), t3);
    uint64_t t12 = mont_mul(mont_sqn(t6, 6), t6);
    uint64_t t24 = mont_mul(mont_sqn(t12, 12), t12);
    uint64_t t30 = mont_mul(mont_sqn(t24, 6), t6);
    uint64_t t31 = mont_mul(mont_sqn(t30, 1), x);
    uint64_t t63 = mont_mul(mont_sqn(t31, 32), t31);
    return mont_mul(mont_sqn(t63, 1), x);
}

// ---- Fq3 = Fp[x]/(x^3 - 2), Montgomery components ------------------------
struct Fq3 { uint64_t c0, c1, c2; };
MS_HD Fq3 add(Fq3 a, Fq3 b) { return {add(a.c0, b.c0), add(a.c1, b.c1), add(a.c2, b.c2)}; }
MS_HD Fq3 sub(Fq3 a, Fq3 b) { return {sub(a.c0, b.c0), sub(a.c1, b.c1), sub(a.c2, b.c2)}; }
MS_HD Fq3 neg(Fq3 a) { return {neg(a.c0), neg(a.c1), neg(a.c2)}; }
MS_HD uint64_t dbl(uint64_t a) { return add(a, a); }
// Karatsuba-style 6-multiplication product (same count as felt_u64.h.metal:205-231);
// multiplication by the non-residue 2 is a doubling, not a field multiply.
MS_HD Fq3 mont_mul(Fq3 a, Fq3 b) {
    uint64_t ad = mont_mul(a.c0, b.c0), be = mont_mul(a.c1, b.c1), cf = mont_mul(a.c2, b.c2);
    uint64_t x = sub(sub(mont_mul(add(a.c1, a.c2), add(b.c1, b.c2)), be), cf);   // b f + c e
    uint64_t y = sub(sub(mont_mul(add(a.c0, a.c1), add(b.c0, b.c1)), ad), be);   // a e + b d
    uint64_t z = add(sub(sub(mont_mul(add(a.c0, a.c2), add(b.c0, b.c2)), ad), cf), be); // a f + c d + b e
    return {add(ad, dbl(x)), add(y, dbl(cf)), z};
}
MS_HD Fq3 mont_mul_fp(Fq3 a, uint64_t s) { return {mont_mul(a.c0, s), mont_mul(a.c1, s), mont_mul(a.c2, s)}; }
MS_HD Fq3 mont_pow(Fq3 a, uint64_t e) {
    Fq3 r = {ONE_MONT, 0, 0};
    while (e) { if (e & 1) r = mont_mul(r, a); a = mont_mul(a, a); e >>= 1; }
    return r;
}
// The reference leaves the Fq3 inverse out (evaluation_shaders.h.metal:393,401 commented,
// src/eval_gpu.rs:338 todo!()); this is the norm-based formula arkworks'
// CubicExtField::inverse uses.  inv(0) = 0.
MS_HD Fq3 mont_inv(Fq3 a) {
    uint64_t s0 = sub(mont_mul(a.c0, a.c0), dbl(mont_mul(a.c1, a.c2)));
    uint64_t s1 = sub(dbl(mont_mul(a.c2, a.c2)), mont_mul(a.c0, a.c1));
    uint64_t s2 = sub(mont_mul(a.c1, a.c1), mont_mul(a.c0, a.c2));
    uint64_t n = add(mont_mul(a.c0, s0), dbl(add(mont_mul(a.c2, s1), mont_mul(a.c1, s2))));
    uint64_t ni = mont_inv(n);
    return {mont_mul(s0, ni), mont_mul(s1, ni), mont_mul(s2, ni)};
}

}  // namespace gl
