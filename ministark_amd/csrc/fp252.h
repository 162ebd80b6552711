// The 252-bit StarkWare prime field  p = 2^251 + 17*2^192 + 1  (host + device).
// Replaces gpu/src/metal/felt_u256.h.metal:9-204 (+ u256.h.metal, u128.h.metal); the Rust type is
// ark_ff::Fp256<MontBackend<_, 4>> (gpu/src/fields.rs:239-264): 4 little-endian u64 limbs,
// Montgomery radix R = 2^256, generator 3, two-adicity 192.
//
// p = 1 (mod 2^64), so -p^-1 = 2^64 - 1 (felt_u256.h.metal:104 N_PRIME's low limb) and a
// Montgomery reduction step  t += m*p  with  m = -t0  costs ONE wide multiply: m*p = m + m*p3*2^192.
// This field is compute-bound on any GPU (81 32x32 products per multiplication); no attempt is
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

// Montgomery product a*b*2^-256 mod p, canonical in/out.
// On gfx950 every carry-consuming instruction issues at half rate and the compiler's expansion of a 4-limb
// CIOS with 128-bit temporaries costs ~440 VALU instructions (214 of them add-with-carry).  Here the operands
// are cut into nine 28-bit digits: the 81 digit products accumulate into 64-bit columns that cannot overflow
// (9 * 2^56 < 2^60), so the whole product is 81 v_mad_u64_u32 and no carry handling; the reduction stays in
// the same lazy columns (p = 1 mod 2^28: m = -c_k mod 2^28, and m*p = m + 17m*2^192 + m*2^251 is two more
// multiply-adds into columns k+6 and k+8); eight rounds of 28 bits and one of 32 make R = 2^256 exactly, so
// the value is the arkworks Montgomery product bit for bit.  ~250 instructions.
//
// mul_t<false> leaves out the final conditional subtraction (result < 2p) and accepts a LAZY first operand: any a < 2^256
// with a canonical b keeps every column below 2^62 (a's top digit has 32 bits, so one term per column is < 2^60), and
// (a b + m p) / R < a p / R + p < 2p because p / R < 1/31.9.  The NTT tiles (fp252_ntt_kernels.h) run on such values.
// The product in pieces (round 5: the constraint evaluator accumulates SEVERAL products in the same columns before ONE reduction,
// eval_kernels.h): nine 28-bit digits of an operand (the top one has 32 bits) ...
MS_HD void digits9(const E& a, uint32_t* x) {
    constexpr uint32_t M = (1u << 28) - 1;
    const uint64_t w0 = a.l[0], w1 = a.l[1], w2 = a.l[2], w3 = a.l[3];
    x[0] = (uint32_t)w0 & M; x[1] = (uint32_t)(w0 >> 28) & M; x[2] = (uint32_t)((w0 >> 56) | (w1 << 8)) & M;
    x[3] = (uint32_t)(w1 >> 20) & M; x[4] = (uint32_t)((w1 >> 48) | (w2 << 16)) & M; x[5] = (uint32_t)(w2 >> 12) & M;
    x[6] = (uint32_t)((w2 >> 40) | (w3 << 24)) & M; x[7] = (uint32_t)(w3 >> 4) & M; x[8] = (uint32_t)(w3 >> 32);
}
// ... columns c[k] += sum_{i+j=k} x_i y_j: one product adds less than 9 * 2^60 / 16 to a column (canonical operands: every digit
// below 2^28 but the top ones, < 2^27.1), so SIXTEEN products fit a 64-bit column together with the reduction's own additions ...
MS_HD void mac81(uint64_t* c, const uint32_t* x, const uint32_t* y) {
    #pragma unroll
    for (int i = 0; i < 9; i++) {
        #pragma unroll
        for (int j = 0; j < 9; j++) c[i + j] += (uint64_t)x[i] * y[j];
    }
}
// ... and the Montgomery reduction of the columns: (sum of the products) * 2^-256 mod p.  With K canonical products in the columns
// the value is below (K / 31.9 + 1) p: one conditional subtraction makes it canonical for K <= 30.
template <bool CANON>
MS_HD E reduce_columns(uint64_t* c) {
    constexpr uint32_t M = (1u << 28) - 1;
    // eight reduction rounds of 28 bits: m = -c_k mod 2^28, c += m * p * 2^(28k), p = 1 + 17*2^192 + 2^251
    uint64_t carry = 0;
    #pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint64_t t = c[k] + carry;
        const uint32_t m = (0u - (uint32_t)t) & M;
        carry = (t + m) >> 28;
        c[k + 6] += (uint64_t)m * (17u << 24);
        c[k + 8] += (uint64_t)m << 27;
    }
    // one round of 32 bits completes R = 2^256 (8*28 + 32)
    {
        c[8] += carry;
        const uint32_t low = (uint32_t)c[8] + ((uint32_t)c[9] << 28);
        const uint32_t m = 0u - low;
        c[8] += m;
        c[14] += ((uint64_t)m * 17u) << 24;
        c[16] += (uint64_t)m << 27;
    }
    // digits d_8 .. d_18 (d_8 = 0 and the low 4 bits of d_9 are 0), result = sum_j d_j 2^(28(j-9) - 4)
    uint32_t d[19];
    uint64_t cr = 0;
    #pragma unroll
    for (int k = 8; k < 19; k++) { const uint64_t t = c[k] + cr; d[k] = (uint32_t)t & M; cr = t >> 28; }
    E r;
    r.l[0] = (uint64_t)(d[9] >> 4) | ((uint64_t)d[10] << 24) | ((uint64_t)d[11] << 52);
    r.l[1] = (uint64_t)(d[11] >> 12) | ((uint64_t)d[12] << 16) | ((uint64_t)d[13] << 44);
    r.l[2] = (uint64_t)(d[13] >> 20) | ((uint64_t)d[14] << 8) | ((uint64_t)d[15] << 36);
    r.l[3] = (uint64_t)d[16] | ((uint64_t)d[17] << 28) | ((uint64_t)d[18] << 56);
    if constexpr (!CANON) return r;
    return geq_p(r) ? sub_p(r) : r;
}
template <bool CANON>
MS_HD E mul_t(const E& a, const E& b) {
    uint32_t x[9], y[9];
    digits9(a, x);
    digits9(b, y);
    uint64_t c[19];
    #pragma unroll
    for (int k = 0; k < 19; k++) c[k] = 0;
    mac81(c, x, y);
    return reduce_columns<CANON>(c);
}
MS_HD E mul(const E& a, const E& b) { return mul_t<true>(a, b); }
// a * a * 2^-256 mod p for a CANONICAL a: the 81 digit products are 9 squares and 36 products taken twice -- 45 multiply-adds, the second
// factor of the pairs doubled once (digits < 2^28, the top one < 2^27.1: the doubles fit 32 bits and a column holds what mul's holds).
MS_HD E sqr(const E& a) {
    uint32_t x[9], x2[9];
    digits9(a, x);
    #pragma unroll
    for (int i = 0; i < 9; i++) x2[i] = x[i] << 1;
    uint64_t c[19];
    #pragma unroll
    for (int k = 0; k < 19; k++) c[k] = 0;
    #pragma unroll
    for (int i = 0; i < 9; i++) {
        c[2 * i] += (uint64_t)x[i] * x[i];
        #pragma unroll
        for (int j = i + 1; j < 9; j++) c[i + j] += (uint64_t)x[i] * x2[j];
    }
    return reduce_columns<true>(c);
}

// ---- lazy forms for long butterfly chains: residues kept below 2^256 ~ 31.9 p, no conditional subtractions -------------
MS_HD E add_lazy(const E& a, const E& b) {          // a + b, the caller guarantees a + b < 2^256
    E r;
    u128 s = (u128)a.l[0] + b.l[0]; r.l[0] = (uint64_t)s;
    s = (u128)a.l[1] + b.l[1] + (uint64_t)(s >> 64); r.l[1] = (uint64_t)s;
    s = (u128)a.l[2] + b.l[2] + (uint64_t)(s >> 64); r.l[2] = (uint64_t)s;
    r.l[3] = a.l[3] + b.l[3] + (uint64_t)(s >> 64);
    return r;
}
template <int K>
MS_HD E kp_minus(const E& t) {                       // K p - t for t < K p  (K = 2, 4: K p = K + K P3 2^192)
    E r;
    u128 d = (u128)(uint64_t)K - t.l[0]; r.l[0] = (uint64_t)d; uint64_t b = (uint64_t)(d >> 64) & 1;
    d = (u128)0 - t.l[1] - b; r.l[1] = (uint64_t)d; b = (uint64_t)(d >> 64) & 1;
    d = (u128)0 - t.l[2] - b; r.l[2] = (uint64_t)d; b = (uint64_t)(d >> 64) & 1;
    r.l[3] = (uint64_t)K * P3 - t.l[3] - b;
    return r;
}
// canonical residue of any x < 2^256: q = floor(x / 2^251) is floor(x / p) or one more, so x - q p lies in [-p, p)
MS_HD E reduce_lazy(const E& x) {
    const uint64_t q = x.l[3] >> 59;
    E r;
    u128 d = (u128)x.l[0] - q; r.l[0] = (uint64_t)d; uint64_t b = (uint64_t)(d >> 64) & 1;
    d = (u128)x.l[1] - b; r.l[1] = (uint64_t)d; b = (uint64_t)(d >> 64) & 1;
    d = (u128)x.l[2] - b; r.l[2] = (uint64_t)d; b = (uint64_t)(d >> 64) & 1;
    d = (u128)x.l[3] - q * P3 - b; r.l[3] = (uint64_t)d;
    const bool negative = ((uint64_t)(d >> 64) & 1) != 0;
    if (negative) {
        E s;
        u128 a = (u128)r.l[0] + P0; s.l[0] = (uint64_t)a;
        a = (u128)r.l[1] + (uint64_t)(a >> 64); s.l[1] = (uint64_t)a;
        a = (u128)r.l[2] + (uint64_t)(a >> 64); s.l[2] = (uint64_t)a;
        s.l[3] = r.l[3] + P3 + (uint64_t)(a >> 64);
        return s;
    }
    return r;
}

MS_HD E to_mont(const E& canon) { return mul(canon, E{{R2_L[0], R2_L[1], R2_L[2], R2_L[3]}}); }
MS_HD E from_mont(const E& m) { return mul(m, E{{1, 0, 0, 0}}); }
MS_HD E pow(E a, const uint64_t* e, int nlimbs) {
    E r = one();
    for (int i = 0; i < nlimbs; i++) {
        uint64_t w = e[i];
        for (int b = 0; b < 64; b++) {
            if (w & 1) r = mul(r, a);
            a = sqr(a);
            w >>= 1;
        }
    }
    return r;
}
MS_HD E pow_u64(E a, uint64_t e) {
    E r = one();
    while (e) { if (e & 1) r = mul(r, a); e >>= 1; if (e) a = sqr(a); }
    return r;
}
MS_HD E sqr_n(E a, int k) {               // a^(2^k)
    #pragma unroll 1
    for (int i = 0; i < k; i++) a = sqr(a);
    return a;
}
// a^(p-2); inv(0) = 0.  p - 2 = 2^192 c - 1 with c = 2^59 + 17, and 2^192 c - 1 = c (2^192 - 1) + (c - 1): with b = a^c the inverse is
// b^(2^192 - 1) a^(c-1) -- 250 squarings and 11 products (the run of 192 ones through 2^k - 1 for k = 2, 3, 6, 12, ..., 192) where
// square-and-multiply over the bits of p - 2 takes 256 and 194.  The inverse is unique: the same words either way.
MS_HD E inv(const E& a) {
    const E a16 = sqr_n(a, 4);
    const E am = mul(sqr_n(a16, 55), a16);                   // a^(2^59 + 16) = a^(c-1)
    const E b = mul(am, a);                                  // a^c
    E x = mul(sqr(b), b);                                    // b^(2^2 - 1)
    x = mul(sqr(x), b);                                      // b^(2^3 - 1)
    #pragma unroll 1
    for (int k = 3; k < 192; k *= 2) x = mul(sqr_n(x, k), x);    // b^(2^2k - 1) = (b^(2^k - 1))^(2^k) b^(2^k - 1)
    return mul(x, am);
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
