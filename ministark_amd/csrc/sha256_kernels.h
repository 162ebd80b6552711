// SHA-256 Merkle commitments for gfx950.
//
// Replaces the reference's CPU path (the reference has no GPU SHA-256):
//   hash_rows           src/merkle.rs:412-436 + Sha256HashFn::hash_elements src/hash.rs:92-99
//                       leaf[r] = SHA-256( ||_c serialize_uncompressed(M[c][r]) ), i.e. the
//                       canonical (non-Montgomery) integer of every limb, little-endian, 8 bytes
//                       per Goldilocks word (Fq3 = c0||c1||c2)
//   build_merkle_nodes  src/merkle.rs:438-508: nodes[k] = SHA-256(nodes[2k] || nodes[2k+1]),
//                       leaf pairs hash into nodes[n/2 .. n), nodes[1] is the root, nodes[0] unused.
// One row (or node) per lane: the matrix is column-major so a wave reads 64 consecutive rows of a
// column in one coalesced 512 B access, converts out of Montgomery form in registers and feeds
// the words straight into the message schedule -- the rows are never materialised.
// This phase is integer-ALU bound (64 rounds per 64-byte block), not HBM bound.
#pragma once
#include <hip/hip_runtime.h>
#include "gl.h"
#include "gl_dev.h"
#include "fp252.h"

namespace mssha {

static constexpr int MAXCOLS = 128;
static constexpr int NT = 256;

__device__ static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

// K256[i] + W[i] for the second block of a 64-byte message (0x80, zeros, bit length 512): that
// block is a constant, so its whole message schedule is folded into the round constants.
__device__ static const uint32_t KW_PAD64[64] = {
    0xc28a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf374,
    0x649b69c1, 0xf0fe4786, 0x0fe1edc6, 0x240cf254, 0x4fe9346f, 0x6cc984be, 0x61b9411e, 0x16f988fa,
    0xf2c65152, 0xa88e5a6d, 0xb019fc65, 0xb9d99ec7, 0x9a1231c3, 0xe70eeaa0, 0xfdb1232b, 0xc7353eb0,
    0x3069bad5, 0xcb976d5f, 0x5a0f118f, 0xdc1eeefd, 0x0a35b689, 0xde0b7a04, 0x58f4ca9d, 0xe15d5b16,
    0x007f3e86, 0x37088980, 0xa507ea32, 0x6fab9537, 0x17406110, 0x0d8cd6f1, 0xcdaa3b6d, 0xc0bbbe37,
    0x83613bda, 0xdb48a363, 0x0b02e931, 0x6fd15ca7, 0x521afaca, 0x31338431, 0x6ed41a95, 0x6d437890,
    0xc39c91f2, 0x9eccabbd, 0xb5c9a0e6, 0x532fb63c, 0xd2c741c6, 0x07237ea3, 0xa4954b68, 0x4c191d76,};

__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }   // -> v_alignbit_b32
// a ^ b ^ c in ONE instruction: gfx950's v_bitop3_b32 (any boolean function of three operands; 0x96 = parity).  The compiler
// finds Ch and Maj by itself but keeps the sigma functions as two v_xor_b32 each -- 352 of the 2 864 VALU instructions of
// a Merkle node (every instruction costs an issue slot here: profiles/r03_sha256_isa.txt).
// (truth tables: src0 = 0xF0, src1 = 0xCC, src2 = 0xAA; Ch = e ? f : g = 0xCA, Maj = 0xE8)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ uint32_t ch3(uint32_t e, uint32_t f, uint32_t g) { return __builtin_amdgcn_bitop3_b32(e, f, g, 0xCA); }
__device__ __forceinline__ uint32_t maj3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }
#else
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return a ^ b ^ c; }
__device__ __forceinline__ uint32_t ch3(uint32_t e, uint32_t f, uint32_t g) { return (e & f) ^ (~e & g); }
__device__ __forceinline__ uint32_t maj3(uint32_t a, uint32_t b, uint32_t c) { return (a & b) ^ (a & c) ^ (b & c); }
#endif
__device__ __forceinline__ uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }                  // -> v_perm_b32

struct Sha {
    uint32_t h[8];
    uint32_t w[16];
    __device__ __forceinline__ void init() {
        h[0] = 0x6a09e667; h[1] = 0xbb67ae85; h[2] = 0x3c6ef372; h[3] = 0xa54ff53a;
        h[4] = 0x510e527f; h[5] = 0x9b05688c; h[6] = 0x1f83d9ab; h[7] = 0x5be0cd19;
    }
    // one compression of the 16 big-endian words in w[]
    __device__ __forceinline__ void compress() {
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        #pragma unroll
        for (int i = 0; i < 64; i++) {
            uint32_t wi;
            if (i < 16) wi = w[i];
            else {
                const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
                const uint32_t s0 = xor3(rotr(w15, 7), rotr(w15, 18), w15 >> 3);
                const uint32_t s1 = xor3(rotr(w2, 17), rotr(w2, 19), w2 >> 10);
                wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
                w[i & 15] = wi;
            }
            const uint32_t S1 = xor3(rotr(e, 6), rotr(e, 11), rotr(e, 25));
            const uint32_t ch = ch3(e, f, g);
            const uint32_t t1 = hh + S1 + ch + K256[i] + wi;
            const uint32_t S0 = xor3(rotr(a, 2), rotr(a, 13), rotr(a, 22));
            const uint32_t mj = maj3(a, b, c);
            const uint32_t t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    // compression of a block whose 16 words are constants: kw[i] = K256[i] + W[i] precomputed (wave-uniform table)
    __device__ __forceinline__ void compress_kw(const uint32_t* __restrict__ kw) {
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        #pragma unroll
        for (int i = 0; i < 64; i++) {
            const uint32_t S1 = xor3(rotr(e, 6), rotr(e, 11), rotr(e, 25));
            const uint32_t ch = ch3(e, f, g);
            const uint32_t t1 = hh + S1 + ch + kw[i];
            const uint32_t S0 = xor3(rotr(a, 2), rotr(a, 13), rotr(a, 22));
            const uint32_t mj = maj3(a, b, c);
            const uint32_t t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    // compression of the constant padding block that follows a 64-byte message
    __device__ __forceinline__ void compress_pad64() {
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        #pragma unroll
        for (int i = 0; i < 64; i++) {
            const uint32_t S1 = xor3(rotr(e, 6), rotr(e, 11), rotr(e, 25));
            const uint32_t ch = ch3(e, f, g);
            const uint32_t t1 = hh + S1 + ch + KW_PAD64[i];
            const uint32_t S0 = xor3(rotr(a, 2), rotr(a, 13), rotr(a, 22));
            const uint32_t mj = maj3(a, b, c);
            const uint32_t t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
};

struct RowsParams {
    const uint64_t* cols[MAXCOLS];
    uint8_t* leaves;          // nrows x 32 bytes
    size_t nrows;
    unsigned ncols;
    unsigned V;               // u64 words per element
    unsigned row_stride;      // words between consecutive rows of one column (V when columns are dense)
    unsigned fold_last;       // the last block holds only padding + length (row length a multiple of 64 bytes):
    uint32_t kw_last[64];     //   its message schedule is constant and comes folded into the round constants
};
// host: K256[i] + W[i] of the block {0x80000000, 0, ..., 0, bits_hi, bits_lo}
static inline void sha256_fold_pad_block(uint64_t bits, uint32_t kw[64]) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
        0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
        0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
        0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
        0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
        0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
        0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t w[64] = {0};
    w[0] = 0x80000000u; w[14] = (uint32_t)(bits >> 32); w[15] = (uint32_t)bits;
    auto rr = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
    for (int i = 16; i < 64; i++) {
        const uint32_t s0 = rr(w[i - 15], 7) ^ rr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rr(w[i - 2], 17) ^ rr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    for (int i = 0; i < 64; i++) kw[i] = K[i] + w[i];
}

// One row per lane.  The message is a stream of 8-byte slots: slot i < nslots is limb (i % V) of the
// element of column i / V, as its canonical value x, contributing the big-endian words
// bswap32(lo32(x)), bswap32(hi32(x)) (little-endian bytes); slot nslots holds the 0x80 pad, the last
// slot of the last block the bit length.  Blocks of 8 slots are filled with compile-time register
// indices (no dynamic indexing of the schedule).
static __global__ void __launch_bounds__(NT) sha256_rows(RowsParams P) {
    const size_t r = (size_t)blockIdx.x * NT + threadIdx.x;
    if (r >= P.nrows) return;
    Sha s;
    s.init();
    const unsigned V = P.V;
    const unsigned nslots = P.ncols * V;
    const unsigned nblocks = (nslots + 2 + 7) / 8;
    const uint64_t bits = (uint64_t)nslots * 64;
    f252::E big = f252::zero();
    const unsigned data_blocks = P.fold_last ? nblocks - 1 : nblocks;
    for (unsigned blk = 0; blk < data_blocks; blk++) {
        #pragma unroll
        for (int j = 0; j < 8; j++) {
            const unsigned i = blk * 8 + j;
            uint32_t w0 = 0, w1 = 0;
            if (i < nslots) {
                const unsigned c = i / V, v = i - c * V;
                const uint64_t* __restrict__ col = P.cols[c];
                uint64_t x;
                if (V == 4) {
                    const uint64_t* e = col + r * P.row_stride;
                    if (v == 0) big = f252::from_mont(f252::E{{e[0], e[1], e[2], e[3]}});
                    x = v == 0 ? big.l[0] : v == 1 ? big.l[1] : v == 2 ? big.l[2] : big.l[3];
                } else {
                    x = gld::mmul(col[r * P.row_stride + v], 1);       // out of Montgomery form, canonical
                }
                w0 = bswap32((uint32_t)x); w1 = bswap32((uint32_t)(x >> 32));
            } else if (i == nslots) {
                w0 = 0x80000000u;
            }
            if (j == 7 && blk == nblocks - 1) { w0 = (uint32_t)(bits >> 32); w1 = (uint32_t)bits; }
            s.w[2 * j] = w0; s.w[2 * j + 1] = w1;
        }
        s.compress();
    }
    if (P.fold_last) s.compress_kw(P.kw_last);
    uint4* out = (uint4*)(P.leaves + r * 32);
    out[0] = make_uint4(bswap32(s.h[0]), bswap32(s.h[1]), bswap32(s.h[2]), bswap32(s.h[3]));
    out[1] = make_uint4(bswap32(s.h[4]), bswap32(s.h[5]), bswap32(s.h[6]), bswap32(s.h[7]));
}

// nodes[out0 + i] = SHA-256(src[2i] || src[2i+1]) for i < count; digests are 32 raw bytes
static __global__ void __launch_bounds__(NT) sha256_merge_level(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t count) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= count) return;
    Sha s;
    s.init();
    const uint4* in = (const uint4*)(src + i * 64);
    #pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint4 v = in[q];
        s.w[4 * q] = bswap32(v.x); s.w[4 * q + 1] = bswap32(v.y); s.w[4 * q + 2] = bswap32(v.z); s.w[4 * q + 3] = bswap32(v.w);
    }
    s.compress();
    s.compress_pad64();
    uint4* out = (uint4*)(dst + i * 32);
    out[0] = make_uint4(bswap32(s.h[0]), bswap32(s.h[1]), bswap32(s.h[2]), bswap32(s.h[3]));
    out[1] = make_uint4(bswap32(s.h[4]), bswap32(s.h[5]), bswap32(s.h[6]), bswap32(s.h[7]));
}

// The upper part of the tree in few launches: a level of <= 2^17 parents is latency-bound as a launch of its own (two dependent
// compressions on one lane, 5-6 us, whatever its size), and a 2^24-leaf tree has seventeen of them.  Workgroup b takes the NT parents
// [b NT, b NT + NT) of a level of `count` parents (count = NT: the top of the tree, one workgroup; count = k NT: k subtrees at once),
// keeps the current level in LDS and climbs log2(NT) more levels to ONE node; every level is also written to its slot of nodes[]
// (level of c nodes at nodes[c ..]).  `src` holds 2 * count digests.  count < NT: a single workgroup, the tree's last levels.
// PER = 2 (the widest of these levels, 2^17 parents: 256 workgroups instead of 512): a lane computes TWO adjacent parents and their parent
// without leaving its registers.  With 512 workgroups every SIMD hosts two waves for all nine levels -- a wave with one live lane still
// takes its issue slots -- and a level costs 8.4 us instead of the 5.2 us of a wave that has its SIMD to itself (scripts/merkle_top_probe.py).
template <int PER>
static __global__ void __launch_bounds__(NT) sha256_merkle_top(const uint8_t* __restrict__ src, uint8_t* __restrict__ nodes, unsigned count) {
    __shared__ uint32_t lvl[2][NT * 8];
    const unsigned t = threadIdx.x, b = blockIdx.x;
    // the launch that ends in the root also clears the unused slot 0 of nodes[] (src/merkle.rs:145-147: n slots, the root in nodes[1]) -- a
    // memset command of its own was one more launch latency per tree, eight trees per proof
    if (count <= (unsigned)NT && b == 0 && t < 8) ((uint32_t*)nodes)[t] = 0;
    unsigned mine = count < (unsigned)NT ? count : (unsigned)NT;          // nodes of the current level this workgroup computes
    size_t level = count;                                                 // nodes of the current level in the whole tree
    Sha s;
    auto first = [&](size_t node) {                                       // parent `node` of the level of `count` parents, from src, written to its slot
        s.init();
        const uint4* in = (const uint4*)(src + node * 64);
        #pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 v = in[q];
            s.w[4 * q] = bswap32(v.x); s.w[4 * q + 1] = bswap32(v.y); s.w[4 * q + 2] = bswap32(v.z); s.w[4 * q + 3] = bswap32(v.w);
        }
        s.compress();
        s.compress_pad64();
    };
    auto put = [&](size_t slot) {
        uint4* out = (uint4*)(nodes + slot * 32);
        out[0] = make_uint4(bswap32(s.h[0]), bswap32(s.h[1]), bswap32(s.h[2]), bswap32(s.h[3]));
        out[1] = make_uint4(bswap32(s.h[4]), bswap32(s.h[5]), bswap32(s.h[6]), bswap32(s.h[7]));
    };
    if constexpr (PER == 2) {                                            // count is a multiple of 2 NT here
        const size_t n0 = (size_t)b * 2 * NT + 2 * t;
        uint4 second[4];                                                  // both messages are requested before the first compression:
        const uint4* in2 = (const uint4*)(src + (n0 + 1) * 64);          // the level below was written by another launch, its lines come from memory
        #pragma unroll
        for (int q = 0; q < 4; q++) second[q] = in2[q];
        first(n0); put(level + n0);
        uint32_t left[8];
        #pragma unroll
        for (int q = 0; q < 8; q++) left[q] = s.h[q];
        s.init();
        #pragma unroll
        for (int q = 0; q < 4; q++) {
            s.w[4 * q] = bswap32(second[q].x); s.w[4 * q + 1] = bswap32(second[q].y); s.w[4 * q + 2] = bswap32(second[q].z); s.w[4 * q + 3] = bswap32(second[q].w);
        }
        s.compress();
        s.compress_pad64();
        put(level + n0 + 1);
        #pragma unroll
        for (int q = 0; q < 8; q++) { s.w[8 + q] = s.h[q]; s.w[q] = left[q]; }
        level >>= 1;
        s.init();
        s.compress();
        s.compress_pad64();
    } else if (t < mine) first((size_t)b * NT + t);
    int cur = 0;
    for (;;) {
        if (t < mine) {
            put(level + (size_t)b * mine + t);
            #pragma unroll
            for (int q = 0; q < 8; q++) lvl[cur][t * 8 + q] = s.h[q];         // big-endian words, ready for the next schedule
        }
        if (mine == 1) break;
        __syncthreads();
        mine >>= 1; level >>= 1;
        if (t < mine) {
            #pragma unroll
            for (int q = 0; q < 16; q++) s.w[q] = lvl[cur][t * 16 + q];
            s.init();
            s.compress();
            s.compress_pad64();
        }
        cur ^= 1;
    }
}

// Proof-of-work grinding: PublicCoin::grind_proof_of_work (src/random.rs:48-55)
// = the smallest nonce >= 1 with leading_zeros(SHA-256(seed || nonce.to_be_bytes())) >= bits
// (verify_proof_of_work src/random.rs:129-132, merge_with_int src/hash.rs:84-89, leading_zeros :181-192).
// One nonce per lane over a window [base, base + count); the minimum hit is kept with atomicMin.
// The 40-byte message is one block: seed words are loaded once (wave-uniform).
struct PowParams { uint32_t seed[8]; unsigned long long base; unsigned long long count; unsigned bits; unsigned long long* found; };
static __global__ void __launch_bounds__(NT) sha256_pow_grind(PowParams P) {
    const unsigned long long i = (unsigned long long)blockIdx.x * NT + threadIdx.x;
    if (i >= P.count) return;
    const unsigned long long nonce = P.base + i;
    Sha s;
    s.init();
    #pragma unroll
    for (int q = 0; q < 8; q++) s.w[q] = P.seed[q];
    s.w[8] = (uint32_t)(nonce >> 32); s.w[9] = (uint32_t)nonce;       // big-endian u64
    s.w[10] = 0x80000000u; s.w[11] = 0; s.w[12] = 0; s.w[13] = 0; s.w[14] = 0; s.w[15] = 320;
    s.compress();
    // leading zero bits of the big-endian digest
    unsigned lz = 0;
    bool done = false;
    #pragma unroll
    for (int q = 0; q < 8; q++) {
        if (!done) {
            const unsigned z = s.h[q] ? (unsigned)__clz(s.h[q]) : 32u;
            lz += z;
            if (z != 32) done = true;
        }
    }
    if (lz >= P.bits) atomicMin(P.found, nonce);
}

}  // namespace mssha
