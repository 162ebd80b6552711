// The 252-bit StarkWare prime field  p = 2^251 + 17*2^192 + 1  (host + device).
// Replaces gpu/src/metal/felt_u256.h.metal:9-204 (+ u256.h.metal, u128.h.metal); the Rust type is
// ark_ff::Fp256<MontBackend<_, 4>> (gpu/src/fields.rs:239-264): 4 little-endian u64 limbs,
// Montgomery radix R = 2^256, generator 3, two-adicity 192.
//
// p = 1 (mod 2^64), so -p^-1 = 2^64 - 1 (felt_u256.h.metal:104 N_PRIME's low limb) and a
// Montgomery reduction step  t += m*p  with  m = -t0  costs ONE wide multiply: m*p = m + m*p3*2^192.
// This field is compute-bound on any GPU (>= 20 64x64 products per multiplication); no attempt is
// made here to reach an HBM roofline.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <stdint.h>
#endif
#include "gl.h"   // MS_HD

namespace f252 {

typedef unsigned __int128 u128;
struct E { uint64_t l[4]; };

static constexpr uint64_t P0 = 1ull, P3 = 0x0800000000000011ull;                 // p = P3*2^192 + 1
// R mod p  ("ONE", felt_u256.h.metal:101) and R^2 mod p (felt_u256.h.metal:103)
static constexpr uint64_t ONE_L[4] = {18446744073709551585ull, 18446744073709551615ull, 18446744073709551615ull, 576460752303422960ull};
static constexpr uint64_t R2_L[4] = {18446741271209837569ull, 5151653887ull, 18446744073700081664ull, 576413109808302096ull};

MS_HD E zero() { return {{0, 0, 0, 0}}; }
MS_HD E one() { return {{ONE_L[0], ONE_L[1], ONE_L[2], ONE_L[3]}}; }
MS_HD bool is_zero(const E& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
MS_HD bool geq_p(const E& a) {           // a >= p ?
    if (a.l[3] != P3) return a.l[3] > P3;
    return (a.l[2] | a.l[1]) != 0 || a.l[0] >= P0;
}
MS_HD E sub_p(const E& a) {              // a - p (caller guarantees a >= p)
    E r;
    u128 d = (u128)a.l[0] - P0;
    r.l[0] = (uint64_t)d; uint64_t b = (uint64_t)(d >> 64) & 1;
    d = (u128)a.l[1] - b; r.l[1] = (uint64_t)d; b = (uint64_t)(d >> 64) & 1;
    d = (u128)a.l[2] - b; r.l[2] = (uint64_t)d; b = (uint64_t)(d >> 64) & 1;
    r.l[3] = a.l[3] - P3 - b;
    return r;
}
MS_HD E add(const E& a, const E& b) {    // canonical in, canonical out (a + b < 2p < 2^253: no carry out)
    E r;
    u128 s = (u128)a.l[0] + b.l[0]; r.l[0] = (uint64_t)s;
    s = (u128)a.l[1] + b.l[1] + (uint64_t)(s >> 64); r.l[1] = (uint64_t)s;
    s = (u128)a.l[2] + b.l[2] + (uint64_t)(s >> 64); r.l[2] = (uint64_t)s;
    r.l[3] = a.l[3] + b.l[3] + (uint64_t)(s >> 64);
    return geq_p(r) ? sub_p(r) : r;
}
MS_HD E neg(const E& a) {
    if (is_zero(a)) return a;
    E r;
    u128 d = (u128)P0 - a.l[0]; r.l[0] = (uint64_t)d; uint64_t b = (uint64_t)(d >> 64) & 1;
    d = (u128)0 - a.l[1] - b; r.l[1] = (uint64_t)d; b = (uint64_t)(d >> 64) & 1;
    d = (u128)0 - a.l[2] - b; r.l[2] = (uint64_t)d; b = (uint64_t)(d >> 64) & 1;
    r.l[3] = P3 - a.l[3] - b;
    return r;
}
MS_HD E sub(const E& a, const E& b) { return add(a, neg(b)); }      // felt_u256.h.metal:134-140

// Montgomery product a*b*2^-256 mod p (CIOS, 4 limbs), canonical in/out
MS_HD E mul(const E& a, const E& b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        #pragma unroll
        for (int j = 0; j < 4; j++) {
            c += (u128)a.l[j] * b.l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        // m = -t0 ; t += m*p = m + m*P3*2^192 ; then shift one limb
        const uint64_t m = 0 - t[0];
        u128 d = (u128)t[0] + m;                  // low limb becomes 0, carry = (t0 != 0)
        uint64_t carry = (uint64_t)(d >> 64);
        d = (u128)t[1] + carry; t[0] = (uint64_t)d; carry = (uint64_t)(d >> 64);
        d = (u128)t[2] + carry; t[1] = (uint64_t)d; carry = (uint64_t)(d >> 64);
        d = (u128)m * P3 + t[3] + carry; t[2] = (uint64_t)d;
        d = (u128)t[4] + (uint64_t)(d >> 64); t[3] = (uint64_t)d;
        t[4] = t[5] + (uint64_t)(d >> 64);
    }
    E r = {{t[0], t[1], t[2], t[3]}};
    return (t[4] || geq_p(r)) ? sub_p(r) : r;
}
MS_HD E to_mont(const E& canon) { return mul(canon, E{{R2_L[0], R2_L[1], R2_L[2], R2_L[3]}}); }
MS_HD E from_mont(const E& m) { return mul(m, E{{1, 0, 0, 0}}); }
MS_HD E pow(E a, const uint64_t* e, int nlimbs) {
    E r = one();
    for (int i = 0; i < nlimbs; i++) {
        uint64_t w = e[i];
        for (int b = 0; b < 64; b++) {
            if (w & 1) r = mul(r, a);
            a = mul(a, a);
            w >>= 1;
        }
    }
    return r;
}
MS_HD E pow_u64(E a, uint64_t e) {
    E r = one();
    while (e) { if (e & 1) r = mul(r, a); e >>= 1; if (e) a = mul(a, a); }
    return r;
}
MS_HD E inv(const E& a) {                 // a^(p-2); inv(0) = 0
    const uint64_t e[4] = {P0 - 2, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull, P3 - 1};   // p - 2 (borrow through the zero limbs)
    return pow(a, e, 4);
}
MS_HD bool eq(const E& a, const E& b) { return a.l[0] == b.l[0] && a.l[1] == b.l[1] && a.l[2] == b.l[2] && a.l[3] == b.l[3]; }

// generator 3 (gpu/src/fields.rs:241), two-adicity 192: root of unity of order 2^log_n, Montgomery form
inline E root_of_unity(unsigned log_n) {
    // 3^((p-1)/2^192) = 3^P3
    E g = to_mont(E{{3, 0, 0, 0}});
    E r = pow_u64(g, P3);
    for (unsigned i = log_n; i < 192; i++) r = mul(r, r);
    return r;
}

}  // namespace f252
