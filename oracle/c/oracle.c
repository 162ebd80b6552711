/* ORACLE -- test infrastructure, NOT product code.
 *
 * Plain-C / OpenMP restatement ("port") of the reference's CPU path for the
 * gpu-poly hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load the shared object built from this file; the
 * product library (libministark_hip.so) never links or calls it.
 *
 * The reference's own CPU path is arkworks (ark-poly 0.4.2 / ark-ff 0.4.2 /
 * ark-ff-optimized 0.4.1, un-vendored crates pinned in Cargo.lock) + sha2 +
 * rayon; none of it can be built here (no Rust toolchain), so this file
 * restates the algorithms at the reference's call sites:
 *
 *   Goldilocks Montgomery arithmetic ... gpu/src/metal/felt_u64.h.metal:147-177
 *   NTT / iNTT, coset pre/post scaling . src/matrix.rs:119-139,166-190;
 *                                        gpu/src/plan.rs:254-263,300-309,386-424
 *   bit reversal ...................... gpu/src/utils.rs:4-41
 *   LDE ............................... src/prover.rs:50-51; src/matrix.rs:225-251
 *   row hashing + Merkle nodes ........ src/merkle.rs:412-508; src/hash.rs:77-99
 *   FRI fold (apply_drp) .............. src/fri.rs:526-567
 *   element-wise stages ............... gpu/src/metal/evaluation_shaders.h.metal:11-168
 *
 * All field data is in the reference's memory format: Montgomery residues
 * (R = 2^64) as u64; an Fq3 element is 3 consecutive u64 (c0,c1,c2).
 * "V" below is the number of u64 words per element (1 = Fp, 3 = Fq3); base
 * field twiddles act component-wise (ark-poly DomainCoeff), so an Fq3
 * transform is 3 interleaved Fp transforms.
 *
 * Pinned against: oracle/pyref (independent big-int Python) on the
 * reference's test shapes, and the reference's in-tree constants
 * (tests/test_oracle_kat.py).  Parity for FFT/LDE/FRI outputs is otherwise
 * unpinned by stored vectors (the reference stores none).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
#define GL_P 0xFFFFFFFF00000001ULL
#define GL_ONE 0xFFFFFFFFULL            /* R mod p   (felt_u64.h.metal:118) */
#define GL_R2 18446744065119617025ULL   /* R^2 mod p (felt_u64.h.metal:127) */

/* ---- Goldilocks, Montgomery form ------------------------------------- */
static inline uint64_t gl_add(uint64_t a, uint64_t b) {
    uint64_t tmp = GL_P - b;               /* a + b = a - (p - b) */
    uint64_t x = a - tmp;
    return (a < tmp) ? x + GL_P : x;
}
static inline uint64_t gl_sub(uint64_t a, uint64_t b) {
    uint64_t x = a - b;
    return (a < b) ? x + GL_P : x;
}
static inline uint64_t gl_neg(uint64_t a) { return a ? GL_P - a : 0; }
/* Montgomery product a*b*R^-1 mod p */
static inline uint64_t gl_mul(uint64_t a, uint64_t b) {
    u128 x = (u128)a * b;
    uint64_t xl = (uint64_t)x, xh = (uint64_t)(x >> 64);
    uint64_t tmp = xl << 32;
    uint64_t s = xl + tmp;
    uint64_t ov = s < xl;
    uint64_t bb = s - (s >> 32) - ov;
    uint64_t r = xh - bb;
    return (xh < bb) ? r + GL_P : r;
}
static inline uint64_t gl_to_mont(uint64_t canon) { return gl_mul(canon, GL_R2); }
static inline uint64_t gl_from_mont(uint64_t m) { return gl_mul(m, 1); }
static uint64_t gl_pow(uint64_t a, uint64_t e) {
    uint64_t r = GL_ONE;
    while (e) { if (e & 1) r = gl_mul(r, a); a = gl_mul(a, a); e >>= 1; }
    return r;
}
static inline uint64_t gl_inv(uint64_t a) { return gl_pow(a, GL_P - 2); }

uint64_t oracle_gl_to_mont(uint64_t c) { return gl_to_mont(c % GL_P); }
uint64_t oracle_gl_from_mont(uint64_t m) { return gl_from_mont(m); }
uint64_t oracle_gl_mul(uint64_t a, uint64_t b) { return gl_mul(a, b); }
uint64_t oracle_gl_add(uint64_t a, uint64_t b) { return gl_add(a, b); }
uint64_t oracle_gl_sub(uint64_t a, uint64_t b) { return gl_sub(a, b); }
uint64_t oracle_gl_inv(uint64_t a) { return gl_inv(a); }
uint64_t oracle_gl_pow(uint64_t a, uint64_t e) { return gl_pow(a, e); }

/* 2^32-th root of unity 7^((p-1)/2^32), canonical */
#define GL_TWO_ADIC_ROOT 1753635133440165772ULL
uint64_t oracle_gl_root_of_unity(unsigned log_n) { /* Montgomery form */
    uint64_t r = gl_to_mont(GL_TWO_ADIC_ROOT);
    for (unsigned i = log_n; i < 32; i++) r = gl_mul(r, r);
    return r;
}

/* ---- Fq3 = Fp[x]/(x^3-2), Montgomery components ----------------------- */
typedef struct { uint64_t c0, c1, c2; } fq3;
static inline fq3 fq3_add(fq3 a, fq3 b) { fq3 r = {gl_add(a.c0,b.c0), gl_add(a.c1,b.c1), gl_add(a.c2,b.c2)}; return r; }
static inline fq3 fq3_sub(fq3 a, fq3 b) { fq3 r = {gl_sub(a.c0,b.c0), gl_sub(a.c1,b.c1), gl_sub(a.c2,b.c2)}; return r; }
static inline fq3 fq3_neg(fq3 a) { fq3 r = {gl_neg(a.c0), gl_neg(a.c1), gl_neg(a.c2)}; return r; }
static inline uint64_t gl_dbl(uint64_t a) { return gl_add(a, a); }
static inline fq3 fq3_mul(fq3 a, fq3 b) {       /* schoolbook, x^3 = 2 */
    uint64_t t00 = gl_mul(a.c0,b.c0), t01 = gl_mul(a.c0,b.c1), t02 = gl_mul(a.c0,b.c2);
    uint64_t t10 = gl_mul(a.c1,b.c0), t11 = gl_mul(a.c1,b.c1), t12 = gl_mul(a.c1,b.c2);
    uint64_t t20 = gl_mul(a.c2,b.c0), t21 = gl_mul(a.c2,b.c1), t22 = gl_mul(a.c2,b.c2);
    fq3 r;
    r.c0 = gl_add(t00, gl_dbl(gl_add(t12, t21)));
    r.c1 = gl_add(gl_add(t01, t10), gl_dbl(t22));
    r.c2 = gl_add(gl_add(t02, t11), t20);
    return r;
}
static inline fq3 fq3_mul_fp(fq3 a, uint64_t s) { fq3 r = {gl_mul(a.c0,s), gl_mul(a.c1,s), gl_mul(a.c2,s)}; return r; }
static fq3 fq3_pow(fq3 a, uint64_t e) {
    fq3 r = {GL_ONE, 0, 0};
    while (e) { if (e & 1) r = fq3_mul(r, a); a = fq3_mul(a, a); e >>= 1; }
    return r;
}
/* inverse via norm: for t = x^3 - 2.  N(a) = a * a^p * a^(p^2) in Fp.  We use
 * the adjugate formula of the multiplication matrix (exact, any correct
 * formula agrees with arkworks' CubicExtField::inverse). */
static fq3 fq3_inv(fq3 a) {
    /* a = c0 + c1 x + c2 x^2, x^3 = 2:
       s0 = c0^2 - 2 c1 c2 ; s1 = 2 c2^2 - c0 c1 ; s2 = c1^2 - c0 c2
       n  = c0 s0 + 2 (c2 s1 + c1 s2) ;  a^-1 = (s0, s1, s2) / n            */
    uint64_t s0 = gl_sub(gl_mul(a.c0,a.c0), gl_dbl(gl_mul(a.c1,a.c2)));
    uint64_t s1 = gl_sub(gl_dbl(gl_mul(a.c2,a.c2)), gl_mul(a.c0,a.c1));
    uint64_t s2 = gl_sub(gl_mul(a.c1,a.c1), gl_mul(a.c0,a.c2));
    uint64_t n = gl_add(gl_mul(a.c0,s0), gl_dbl(gl_add(gl_mul(a.c2,s1), gl_mul(a.c1,s2))));
    uint64_t ni = gl_inv(n);
    fq3 r = {gl_mul(s0,ni), gl_mul(s1,ni), gl_mul(s2,ni)};
    return r;
}
void oracle_fq3_mul(const uint64_t* a, const uint64_t* b, uint64_t* out) {
    fq3 x = {a[0],a[1],a[2]}, y = {b[0],b[1],b[2]}; fq3 r = fq3_mul(x,y);
    out[0]=r.c0; out[1]=r.c1; out[2]=r.c2;
}
void oracle_fq3_inv(const uint64_t* a, uint64_t* out) {
    fq3 x = {a[0],a[1],a[2]}; fq3 r = fq3_inv(x); out[0]=r.c0; out[1]=r.c1; out[2]=r.c2;
}

/* ---- bit reversal ------------------------------------------------------ */
static inline size_t bitrev(size_t i, unsigned log_n) {
    size_t r = 0;
    for (unsigned b = 0; b < log_n; b++) { r = (r << 1) | (i & 1); i >>= 1; }
    return r;
}
/* in-place, elements of V words; only the first `prefix` positions take part
 * (prefix = n for a full reversal; prefix < n is prover.rs:185-194's
 * bit_reverse_ce_trace, a reversal of the size-`prefix` prefix). */
void oracle_bit_reverse(uint64_t* data, unsigned log_n, unsigned V) {
    size_t n = (size_t)1 << log_n;
    enum { T = 7 };                                           /* tiles of 128 x 128 elements */
    if (log_n < 2 * T + 2) {
        #pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; i++) {
            size_t j = bitrev(i, log_n);
            if (j > i) for (unsigned v = 0; v < V; v++) {
                uint64_t t = data[i*V+v]; data[i*V+v] = data[j*V+v]; data[j*V+v] = t;
            }
        }
        return;
    }
    /* The same permutation tile by tile (round 5: the element-by-element swap above misses the cache twice per element and was 40 %
     * of the 2^24-point transform's time): i = (hi, mid, lo) with 7-bit hi and lo; new[(hi, mid, lo)] = old[(rev lo, rev mid, rev hi)],
     * so the tile `mid` is the transposed, row- and column-reversed tile `rev mid`: both are read with contiguous rows into buffers
     * and written back with contiguous rows. */
    const unsigned mb = log_n - 2 * T;
    const size_t R = (size_t)1 << T, tile = R * R;
    size_t rv[1 << T];
    for (size_t i = 0; i < R; i++) rv[i] = bitrev(i, T);
    #pragma omp parallel
    {
        uint64_t* A = (uint64_t*)malloc(tile * V * sizeof(uint64_t));
        uint64_t* B = (uint64_t*)malloc(tile * V * sizeof(uint64_t));
        #pragma omp for schedule(dynamic, 4)
        for (size_t mid = 0; mid < ((size_t)1 << mb); mid++) {
            size_t rm = bitrev(mid, mb);
            if (rm < mid) continue;
            for (size_t hi = 0; hi < R; hi++) {
                memcpy(A + hi * R * V, data + (((hi << mb) | mid) << T) * V, R * V * sizeof(uint64_t));
                if (rm != mid) memcpy(B + hi * R * V, data + (((hi << mb) | rm) << T) * V, R * V * sizeof(uint64_t));
            }
            const uint64_t* from_rm = rm != mid ? B : A;
            for (size_t hi = 0; hi < R; hi++) {
                uint64_t* o1 = data + (((hi << mb) | mid) << T) * V;       /* new tile mid  <- old tile rm  */
                uint64_t* o2 = data + (((hi << mb) | rm) << T) * V;        /* new tile rm   <- old tile mid */
                for (size_t lo = 0; lo < R; lo++) {
                    const size_t src = (rv[lo] * R + rv[hi]) * V;
                    for (unsigned v = 0; v < V; v++) o1[lo * V + v] = from_rm[src + v];
                    if (rm != mid) for (unsigned v = 0; v < V; v++) o2[lo * V + v] = A[src + v];
                }
            }
        }
        free(A); free(B);
    }
}

/* ---- NTT --------------------------------------------------------------- */
/* In-place radix-2 transform of one column of n elements x V words.
 * root, offset, scale in Montgomery form.
 *   forward: x_i *= offset^i ; DFT(root)
 *   inverse: DFT(root) ; x_i *= scale * offset^i   (root = gen^-1,
 *            offset = offset^-1, scale = n^-1 supplied by the caller)      */
static void ntt_core_plain(uint64_t* a, unsigned log_n, unsigned V, const uint64_t* tw /* n/2 powers of root */) {
    size_t n = (size_t)1 << log_n;
    oracle_bit_reverse(a, log_n, V);
    for (unsigned s = 1; s <= log_n; s++) {
        size_t m = (size_t)1 << s, half = m >> 1, tstride = n >> s;
        #pragma omp parallel for schedule(static)
        for (size_t k = 0; k < n / 2; k++) {
            size_t blk = k / half, i = k % half;
            size_t lo = blk * m + i, hi = lo + half;
            uint64_t w = tw[i * tstride];
            for (unsigned v = 0; v < V; v++) {
                uint64_t u = a[lo*V+v], t = gl_mul(a[hi*V+v], w);
                a[lo*V+v] = gl_add(u, t);
                a[hi*V+v] = gl_sub(u, t);
            }
        }
    }
}
/* The same butterflies -- the same pairs, the same twiddle tw[(lo mod half) * (n >> s)], exact field arithmetic, hence the same words --
 * taken in a cache-friendly order (round 5: the plain loop above sweeps the whole column once per stage, 24 sweeps of 128 MiB at
 * 2^24 points, and made the CPU baseline of bench.py look slower than a parallel CPU FFT is): stages 1..12 block by block on
 * contiguous 4096-point blocks, the later stages ten at a time on tiles of 1024 rows x 8 adjacent points gathered into a buffer
 * that stays in the core's cache (the shape of ark-poly's parallel FFT: independent sub-FFTs per chunk, src/domain/radix2/fft.rs). */
static void ntt_core(uint64_t* a, unsigned log_n, unsigned V, const uint64_t* tw) {
    enum { B0 = 12, G = 10, W = 8 };
    if (log_n <= B0) { ntt_core_plain(a, log_n, V, tw); return; }
    size_t n = (size_t)1 << log_n;
    oracle_bit_reverse(a, log_n, V);
    #pragma omp parallel for schedule(static)
    for (size_t b = 0; b < (n >> B0); b++) {
        uint64_t* x = a + (b << B0) * V;
        for (unsigned s = 1; s <= B0; s++) {
            size_t m = (size_t)1 << s, half = m >> 1, tstride = n >> s;
            for (size_t k = 0; k < ((size_t)1 << (B0 - 1)); k++) {
                size_t blk = k / half, i = k % half, lo = blk * m + i, hi = lo + half;
                uint64_t w = tw[i * tstride];
                for (unsigned v = 0; v < V; v++) {
                    uint64_t u = x[lo*V+v], t = gl_mul(x[hi*V+v], w);
                    x[lo*V+v] = gl_add(u, t);
                    x[hi*V+v] = gl_sub(u, t);
                }
            }
        }
    }
    for (unsigned s0 = B0; s0 < log_n; ) {
        unsigned s1 = s0 + G < log_n ? s0 + G : log_n;
        size_t rows = (size_t)1 << (s1 - s0), lows = (size_t)1 << s0, tiles = (n >> s1) * (lows / W);
        #pragma omp parallel
        {
            uint64_t* buf = (uint64_t*)malloc(rows * W * V * sizeof(uint64_t));
            #pragma omp for schedule(static)
            for (size_t t = 0; t < tiles; t++) {
                size_t h = t / (lows / W), i0 = (t % (lows / W)) * W;
                uint64_t* base = a + ((h << s1) + i0) * V;
                for (size_t q = 0; q < rows; q++) memcpy(buf + q * W * V, base + (q << s0) * V, W * V * sizeof(uint64_t));
                for (unsigned s = s0 + 1; s <= s1; s++) {
                    size_t hq = (size_t)1 << (s - 1 - s0), tstride = n >> s;
                    for (size_t q = 0; q < rows; q++) {
                        if (q & hq) continue;
                        size_t lowpart = ((q & (hq - 1)) << s0) + i0;          /* lo mod half, for j = 0 */
                        for (size_t j = 0; j < W; j++) {
                            uint64_t w = tw[(lowpart + j) * tstride];
                            uint64_t* plo = buf + (q * W + j) * V;
                            uint64_t* phi = buf + ((q + hq) * W + j) * V;
                            for (unsigned v = 0; v < V; v++) {
                                uint64_t u = plo[v], tt = gl_mul(phi[v], w);
                                plo[v] = gl_add(u, tt);
                                phi[v] = gl_sub(u, tt);
                            }
                        }
                    }
                }
                for (size_t q = 0; q < rows; q++) memcpy(base + (q << s0) * V, buf + q * W * V, W * V * sizeof(uint64_t));
            }
            free(buf);
        }
        s0 = s1;
    }
}
/* test hook: the plain stage-by-stage loop on its own (tests compare the blocked order with it at sizes the big-integer oracle cannot reach) */
void oracle_ntt_stages_plain(uint64_t* a, unsigned log_n, unsigned V, const uint64_t* tw_unused, uint64_t root_mont);
static uint64_t* make_twiddles(unsigned log_n, uint64_t root) {
    size_t half = ((size_t)1 << log_n) / 2;
    if (half == 0) half = 1;
    uint64_t* tw = (uint64_t*)malloc(half * sizeof(uint64_t));
    /* chunked powers like gpu/src/utils.rs:13-30 (fill_twiddles) */
    size_t chunk = 1024;
    #pragma omp parallel for schedule(static)
    for (size_t c = 0; c < (half + chunk - 1) / chunk; c++) {
        size_t s = c * chunk, e = s + chunk < half ? s + chunk : half;
        uint64_t x = gl_pow(root, s);
        for (size_t i = s; i < e; i++) { tw[i] = x; x = gl_mul(x, root); }
    }
    return tw;
}
static void distribute_powers(uint64_t* a, size_t n, unsigned V, uint64_t g, uint64_t c) {
    /* a_i *= c * g^i   (gpu/src/utils.rs:139-156) */
    size_t chunk = 4096;
    #pragma omp parallel for schedule(static)
    for (size_t b = 0; b < (n + chunk - 1) / chunk; b++) {
        size_t s = b * chunk, e = s + chunk < n ? s + chunk : n;
        uint64_t x = gl_mul(c, gl_pow(g, s));
        for (size_t i = s; i < e; i++) {
            for (unsigned v = 0; v < V; v++) a[i*V+v] = gl_mul(a[i*V+v], x);
            x = gl_mul(x, g);
        }
    }
}
void oracle_ntt_stages_plain(uint64_t* a, unsigned log_n, unsigned V, const uint64_t* tw_unused, uint64_t root_mont) {
    (void)tw_unused;
    uint64_t* tw = make_twiddles(log_n, root_mont);
    ntt_core_plain(a, log_n, V, tw);
    free(tw);
}
void oracle_ntt_stages_blocked(uint64_t* a, unsigned log_n, unsigned V, uint64_t root_mont) {
    uint64_t* tw = make_twiddles(log_n, root_mont);
    ntt_core(a, log_n, V, tw);
    free(tw);
}
/* offset_canon: canonical (non-Montgomery) integer of the coset offset. */
void oracle_ntt(uint64_t* a, unsigned log_n, unsigned V, int inverse, uint64_t offset_canon) {
    size_t n = (size_t)1 << log_n;
    uint64_t gen = oracle_gl_root_of_unity(log_n);
    uint64_t off = gl_to_mont(offset_canon % GL_P);
    if (!inverse) {
        if (off != GL_ONE) distribute_powers(a, n, V, off, GL_ONE);
        uint64_t* tw = make_twiddles(log_n, gen);
        ntt_core(a, log_n, V, tw);
        free(tw);
    } else {
        uint64_t* tw = make_twiddles(log_n, gl_inv(gen));
        ntt_core(a, log_n, V, tw);
        free(tw);
        uint64_t size_inv = gl_inv(gl_to_mont((uint64_t)n % GL_P));
        distribute_powers(a, n, V, gl_inv(off), size_inv);
    }
}
void oracle_ntt_columns(uint64_t** cols, unsigned ncols, unsigned log_n, unsigned V, int inverse, uint64_t offset_canon) {
    for (unsigned c = 0; c < ncols; c++) oracle_ntt(cols[c], log_n, V, inverse, offset_canon);
}

/* LDE of one column: in (n elems) -> out (n*2^log_blowup elems), bit-reversed
 * when bit_reversed != 0.  src/prover.rs:50-51. `in` is left untouched. */
void oracle_lde(const uint64_t* in, uint64_t* out, unsigned log_n, unsigned log_blowup, unsigned V,
                uint64_t offset_canon, int bit_reversed) {
    size_t n = (size_t)1 << log_n, N = n << log_blowup;
    memcpy(out, in, n * V * sizeof(uint64_t));
    oracle_ntt(out, log_n, V, 1, 1);
    memset(out + n * V, 0, (N - n) * V * sizeof(uint64_t));
    oracle_ntt(out, log_n + log_blowup, V, 0, offset_canon);
    if (bit_reversed) oracle_bit_reverse(out, log_n + log_blowup, V);
}

/* ---- SHA-256 ----------------------------------------------------------- */
static const uint32_t K256[64] = {
0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,
0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,
0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,
0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,
0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,
0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,
0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,
0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};
#define ROR(x,n) (((x) >> (n)) | ((x) << (32-(n))))
static void sha256_block(uint32_t st[8], const uint8_t blk[64]) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
        w[i] = ((uint32_t)blk[4*i] << 24) | ((uint32_t)blk[4*i+1] << 16) | ((uint32_t)blk[4*i+2] << 8) | blk[4*i+3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ROR(w[i-15],7) ^ ROR(w[i-15],18) ^ (w[i-15] >> 3);
        uint32_t s1 = ROR(w[i-2],17) ^ ROR(w[i-2],19) ^ (w[i-2] >> 10);
        w[i] = w[i-16] + s0 + w[i-7] + s1;
    }
    uint32_t a=st[0],b=st[1],c=st[2],d=st[3],e=st[4],f=st[5],g=st[6],h=st[7];
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = ROR(e,6) ^ ROR(e,11) ^ ROR(e,25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = h + S1 + ch + K256[i] + w[i];
        uint32_t S0 = ROR(a,2) ^ ROR(a,13) ^ ROR(a,22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        h=g; g=f; f=e; e=d+t1; d=c; c=b; b=a; a=t1+t2;
    }
    st[0]+=a; st[1]+=b; st[2]+=c; st[3]+=d; st[4]+=e; st[5]+=f; st[6]+=g; st[7]+=h;
}
static void sha256(const uint8_t* msg, size_t len, uint8_t out[32]) {
    uint32_t st[8] = {0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19};
    size_t i = 0;
    for (; i + 64 <= len; i += 64) sha256_block(st, msg + i);
    uint8_t tail[128]; size_t rem = len - i;
    memset(tail, 0, sizeof tail);
    memcpy(tail, msg + i, rem);
    tail[rem] = 0x80;
    size_t tl = (rem + 9 <= 64) ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int b = 0; b < 8; b++) tail[tl - 1 - b] = (uint8_t)(bits >> (8*b));
    sha256_block(st, tail);
    if (tl == 128) sha256_block(st, tail + 64);
    for (int k = 0; k < 8; k++) { out[4*k]=st[k]>>24; out[4*k+1]=st[k]>>16; out[4*k+2]=st[k]>>8; out[4*k+3]=st[k]; }
}
void oracle_sha256(const uint8_t* msg, size_t len, uint8_t* out) { sha256(msg, len, out); }

/* leaves[r] = SHA256( || _c  LE8(canonical(cols[c][r*V + v])) , v < V )
 * src/merkle.rs:412-436 + src/hash.rs:92-99                                */
void oracle_sha256_rows(const uint64_t* const* cols, unsigned ncols, unsigned V, size_t nrows, uint8_t* leaves) {
    size_t rowbytes = (size_t)ncols * V * 8;
    #pragma omp parallel
    {
        uint8_t* buf = (uint8_t*)malloc(rowbytes);
        #pragma omp for schedule(static)
        for (size_t r = 0; r < nrows; r++) {
            uint8_t* p = buf;
            for (unsigned c = 0; c < ncols; c++)
                for (unsigned v = 0; v < V; v++) {
                    uint64_t x = gl_from_mont(cols[c][r*V+v]);
                    for (int b = 0; b < 8; b++) *p++ = (uint8_t)(x >> (8*b));
                }
            sha256(buf, rowbytes, leaves + 32*r);
        }
        free(buf);
    }
}
/* nodes: n x 32 bytes; nodes[1] = root; nodes[0] zeroed.  src/merkle.rs:485-508 */
void oracle_sha256_merkle(const uint8_t* leaves, size_t n, uint8_t* nodes) {
    memset(nodes, 0, 32);
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n/2; i++) sha256(leaves + 64*i, 64, nodes + 32*(n/2 + i));
    for (size_t size = n/4; size >= 1; size >>= 1) {
        #pragma omp parallel for schedule(static)
        for (size_t i = size; i < 2*size; i++) sha256(nodes + 64*i, 64, nodes + 32*i);
    }
}

/* ---- FRI fold (src/fri.rs:526-567) ------------------------------------ */
/* evals: n elems x V words, bit-reversed order; out: n/ff elems, bit-reversed.
 * alpha: V words (Montgomery). domain_offset canonical. */
void oracle_fri_fold(const uint64_t* evals, uint64_t* out, unsigned log_n, unsigned V, unsigned ff,
                     const uint64_t* alpha, uint64_t offset_canon) {
    size_t n = (size_t)1 << log_n, m = n / ff;
    unsigned log_ff = 0; while ((1u << log_ff) < ff) log_ff++;
    uint64_t* co = (uint64_t*)malloc(n * V * 8);
    memcpy(co, evals, n * V * 8);
    oracle_bit_reverse(co, log_n, V);
    oracle_ntt(co, log_n, V, 1, offset_canon);
    uint64_t fold = gl_to_mont(ff);
    for (size_t i = 0; i < n * V; i++) co[i] = gl_mul(co[i], fold);
    if (V == 1) {
        uint64_t ap[16]; ap[0] = GL_ONE;
        for (unsigned k = 1; k < ff; k++) ap[k] = gl_mul(ap[k-1], alpha[0]);
        #pragma omp parallel for schedule(static)
        for (size_t i = 0; i < m; i++) {
            uint64_t acc = 0;
            for (unsigned k = 0; k < ff; k++) acc = gl_add(acc, gl_mul(co[i*ff+k], ap[k]));
            out[i] = acc;
        }
    } else {
        fq3 ap[16]; fq3 one = {GL_ONE,0,0}; ap[0] = one; fq3 al = {alpha[0],alpha[1],alpha[2]};
        for (unsigned k = 1; k < ff; k++) ap[k] = fq3_mul(ap[k-1], al);
        #pragma omp parallel for schedule(static)
        for (size_t i = 0; i < m; i++) {
            fq3 acc = {0,0,0};
            for (unsigned k = 0; k < ff; k++) {
                fq3 c = {co[(i*ff+k)*3], co[(i*ff+k)*3+1], co[(i*ff+k)*3+2]};
                acc = fq3_add(acc, fq3_mul(c, ap[k]));
            }
            out[i*3] = acc.c0; out[i*3+1] = acc.c1; out[i*3+2] = acc.c2;
        }
    }
    free(co);
    uint64_t off_ff = gl_from_mont(gl_pow(gl_to_mont(offset_canon % GL_P), ff));
    oracle_ntt(out, log_n - log_ff, V, 0, off_ff);
    oracle_bit_reverse(out, log_n - log_ff, V);
}

/* ---- element-wise stages (evaluation_shaders.h.metal:11-168) ---------- */
/* op codes shared with include/ministark_hip.h */
enum { OP_ADD = 0, OP_MUL = 1 };
enum { UN_NEG = 0, UN_INV = 1, UN_EXP = 2 };
static inline fq3 ld3(const uint64_t* p, size_t i, unsigned V) {
    fq3 r; if (V == 3) { r.c0 = p[3*i]; r.c1 = p[3*i+1]; r.c2 = p[3*i+2]; } else { r.c0 = p[i]; r.c1 = 0; r.c2 = 0; } return r;
}
/* dst[i] = lhs[i] op rhs[(i+shift) % n]; lhs/dst have VL words, rhs VR (VR<=VL) */
void oracle_binary(int op, unsigned VL, unsigned VR, size_t n, uint64_t* dst, const uint64_t* lhs,
                   const uint64_t* rhs, size_t shift) {
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        size_t j = (i + shift) % n;
        if (VL == 1) {
            dst[i] = op == OP_ADD ? gl_add(lhs[i], rhs[j]) : gl_mul(lhs[i], rhs[j]);
        } else {
            fq3 a = ld3(lhs, i, 3), b = ld3(rhs, j, VR), r;
            if (op == OP_ADD) r = fq3_add(a, b);
            else r = (VR == 1) ? fq3_mul_fp(a, b.c0) : fq3_mul(a, b);
            dst[3*i] = r.c0; dst[3*i+1] = r.c1; dst[3*i+2] = r.c2;
        }
    }
}
void oracle_binary_const(int op, unsigned VL, unsigned VR, size_t n, uint64_t* dst, const uint64_t* lhs,
                         const uint64_t* c) {
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        if (VL == 1) {
            dst[i] = op == OP_ADD ? gl_add(lhs[i], c[0]) : gl_mul(lhs[i], c[0]);
        } else {
            fq3 a = ld3(lhs, i, 3), b = ld3(c, 0, VR), r;
            if (op == OP_ADD) r = fq3_add(a, b);
            else r = (VR == 1) ? fq3_mul_fp(a, b.c0) : fq3_mul(a, b);
            dst[3*i] = r.c0; dst[3*i+1] = r.c1; dst[3*i+2] = r.c2;
        }
    }
}
/* dst[i] = lhs[i] * rhs[(i+shift)%n]^e */
void oracle_mul_pow(unsigned VL, unsigned VR, size_t n, uint64_t* dst, const uint64_t* lhs,
                    const uint64_t* rhs, unsigned e, size_t shift) {
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        size_t j = (i + shift) % n;
        if (VL == 1) dst[i] = gl_mul(lhs[i], gl_pow(rhs[j], e));
        else {
            fq3 a = ld3(lhs, i, 3), r;
            if (VR == 1) r = fq3_mul_fp(a, gl_pow(rhs[j], e));
            else r = fq3_mul(a, fq3_pow(ld3(rhs, j, 3), e));
            dst[3*i] = r.c0; dst[3*i+1] = r.c1; dst[3*i+2] = r.c2;
        }
    }
}
void oracle_unary(int op, unsigned V, size_t n, uint64_t* dst, const uint64_t* src, unsigned e) {
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        if (V == 1) {
            uint64_t x = src[i];
            dst[i] = op == UN_NEG ? gl_neg(x) : op == UN_INV ? (x ? gl_inv(x) : 0) : gl_pow(x, e);
        } else {
            fq3 x = ld3(src, i, 3), r;
            if (op == UN_NEG) r = fq3_neg(x);
            else if (op == UN_INV) { if (x.c0 | x.c1 | x.c2) r = fq3_inv(x); else r = x; }
            else r = fq3_pow(x, e);
            dst[3*i] = r.c0; dst[3*i+1] = r.c1; dst[3*i+2] = r.c2;
        }
    }
}
/* dst = sum of columns (src/matrix.rs:357-394) */
void oracle_sum_columns(const uint64_t* const* cols, unsigned ncols, unsigned V, size_t n, uint64_t* dst) {
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n * V; i++) {
        uint64_t acc = 0;
        for (unsigned c = 0; c < ncols; c++) acc = gl_add(acc, cols[c][i]);
        dst[i] = acc;
    }
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ======================================================================================================
 * Constraint evaluation: restatement of eval_cpu::eval (src/eval_cpu.rs:33-150).
 *
 * The reference walks the expression DAG once per CHUNK of 512 consecutive LDE points (eval_cpu.rs:44-52,
 * `chunks_mut(CHUNK_SIZE)` under rayon), every node producing an array of 512 values; a division is
 * x * batch_inverse(y) over the chunk (eval_cpu.rs:280-294, ark_ff::batch_inversion: zeros stay zero); a
 * node is Fp iff all its operands are Fp, else Fq (eval_cpu.rs:306-428); Trace(col, off) reads row
 * (i + lde_step * off) mod n (eval_cpu.rs:115-134); the result is returned as Fq (into_fq_array, :262-275).
 *
 * The DAG arrives flattened in topological order (children first): node k = (kind, a, b)
 *   0 X            value offset * w^i
 *   1 CONST        a = word offset into consts[] (Montgomery words), b = 1 if Fq
 *   2 CHALLENGE    a = index   (Fq when fq_words = 3, Fp when fq_words = 1 / the 252-bit field)
 *   3 HINT         a = index
 *   4 TRACE        a = column (base columns first, then extension columns), b = offset (int32)
 *   5 PERIODIC     a = table index: periodic[a] holds periodic_len[a] Fp values, value = table[i mod len]
 *   6 NEG a        7 ADD a b      8 MUL a b      9 DIV a b      10 POW a, exponent b
 * mode 0: Goldilocks, Fq = Fq3 (fq_words 3) or Fq = Fp (fq_words 1); mode 1: the 252-bit field (4 words, Fq = Fp).
 * ====================================================================================================== */
#define EV_CHUNK 512

/* ---- the 252-bit StarkWare prime, Montgomery R = 2^256 (felt_u256.h.metal:101-203), 4 LE limbs ---- */
static const uint64_t F252_P[4] = {1ULL, 0ULL, 0ULL, 0x0800000000000011ULL};
static const uint64_t F252_NPRIME = 0xFFFFFFFFFFFFFFFFULL;          /* -p^-1 mod 2^64 (p = 1 mod 2^64) */
typedef struct { uint64_t l[4]; } f252;
static inline int f252_geq_p(const uint64_t* a) {
    for (int i = 3; i >= 0; i--) { if (a[i] > F252_P[i]) return 1; if (a[i] < F252_P[i]) return 0; }
    return 1;
}
static inline f252 f252_add(f252 a, f252 b) {
    f252 r; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (c || f252_geq_p(r.l)) { u128 bw = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)r.l[i] - F252_P[i] - bw; r.l[i] = (uint64_t)d; bw = (d >> 64) & 1; } }
    return r;
}
static inline f252 f252_sub(f252 a, f252 b) {
    f252 r; u128 bw = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)a.l[i] - b.l[i] - bw; r.l[i] = (uint64_t)d; bw = (d >> 64) & 1; }
    if (bw) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)r.l[i] + F252_P[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
    return r;
}
/* CIOS Montgomery product (the textbook form arkworks' MontBackend computes) */
static f252 f252_mul(f252 a, f252 b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * F252_NPRIME;
        c = (u128)m * F252_P[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * F252_P[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    f252 r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || f252_geq_p(r.l)) { u128 bw = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)r.l[i] - F252_P[i] - bw; r.l[i] = (uint64_t)d; bw = (d >> 64) & 1; } }
    return r;
}
static const f252 F252_ONE = {{0xFFFFFFFFFFFFFFE1ULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0x07FFFFFFFFFFFDF0ULL}};   /* R mod p (felt_u256.h.metal:101) */
static f252 f252_pow(f252 a, const uint64_t* e, int elimbs) {
    f252 r = F252_ONE;
    for (int i = elimbs - 1; i >= 0; i--)
        for (int b = 63; b >= 0; b--) { r = f252_mul(r, r); if ((e[i] >> b) & 1) r = f252_mul(r, a); }
    return r;
}
static f252 f252_inv(f252 a) {
    const uint64_t ee[4] = {0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL, 0x0800000000000010ULL};   /* p - 2 */
    int zero = !(a.l[0] | a.l[1] | a.l[2] | a.l[3]);
    if (zero) return a;
    return f252_pow(a, ee, 4);
}
void oracle_f252_mul(const uint64_t* a, const uint64_t* b, uint64_t* out) { f252 x, y; memcpy(x.l, a, 32); memcpy(y.l, b, 32); x = f252_mul(x, y); memcpy(out, x.l, 32); }
void oracle_f252_inv(const uint64_t* a, uint64_t* out) { f252 x; memcpy(x.l, a, 32); x = f252_inv(x); memcpy(out, x.l, 32); }

typedef struct { int kind; int64_t a, b; } ev_node;

/* one value slot of a chunk: up to 4 words per point; `w` = words per point of this node (1, 3 or 4) */
static void ev_batch_inverse(uint64_t* v, unsigned w, size_t cnt, int mode) {
    /* ark_ff::batch_inversion: prefix products of the non-zero entries, one inversion, walk back; zeros stay zero */
    uint64_t* pre = (uint64_t*)malloc(cnt * w * 8);
    if (mode == 1) {
        f252 acc = F252_ONE;
        for (size_t i = 0; i < cnt; i++) { f252 x; memcpy(x.l, v + 4 * i, 32); memcpy(pre + 4 * i, acc.l, 32); if (x.l[0] | x.l[1] | x.l[2] | x.l[3]) acc = f252_mul(acc, x); }
        acc = f252_inv(acc);
        for (size_t i = cnt; i-- > 0;) {
            f252 x, p; memcpy(x.l, v + 4 * i, 32); memcpy(p.l, pre + 4 * i, 32);
            if (!(x.l[0] | x.l[1] | x.l[2] | x.l[3])) continue;
            f252 inv = f252_mul(acc, p); acc = f252_mul(acc, x); memcpy(v + 4 * i, inv.l, 32);
        }
    } else if (w == 1) {
        uint64_t acc = GL_ONE;
        for (size_t i = 0; i < cnt; i++) { pre[i] = acc; if (v[i]) acc = gl_mul(acc, v[i]); }
        acc = gl_inv(acc);
        for (size_t i = cnt; i-- > 0;) { if (!v[i]) continue; uint64_t inv = gl_mul(acc, pre[i]); acc = gl_mul(acc, v[i]); v[i] = inv; }
    } else {
        fq3 acc = {GL_ONE, 0, 0};
        for (size_t i = 0; i < cnt; i++) { fq3 x = {v[3*i], v[3*i+1], v[3*i+2]}; pre[3*i] = acc.c0; pre[3*i+1] = acc.c1; pre[3*i+2] = acc.c2; if (x.c0 | x.c1 | x.c2) acc = fq3_mul(acc, x); }
        acc = fq3_inv(acc);
        for (size_t i = cnt; i-- > 0;) {
            fq3 x = {v[3*i], v[3*i+1], v[3*i+2]}, p = {pre[3*i], pre[3*i+1], pre[3*i+2]};
            if (!(x.c0 | x.c1 | x.c2)) continue;
            fq3 inv = fq3_mul(acc, p); acc = fq3_mul(acc, x); v[3*i] = inv.c0; v[3*i+1] = inv.c1; v[3*i+2] = inv.c2;
        }
    }
    free(pre);
}

/* returns 0 on success.  out: n * out_words words (out_words = fq_words in mode 0, 4 in mode 1). */
int oracle_eval_expr(const int32_t* nodes3, unsigned nnodes, const uint64_t* consts, int mode, unsigned fq_words,
                     unsigned log_n, unsigned lde_step, uint64_t offset_canon, const uint64_t* offset252,
                     const uint64_t* const* base_cols, unsigned nbase, const uint64_t* const* ext_cols,
                     const uint64_t* challenges, const uint64_t* hints,
                     const uint64_t* const* periodic, const uint32_t* periodic_len, const uint64_t* w252, uint64_t* out) {
    const size_t n = (size_t)1 << log_n;
    const unsigned pw = mode == 1 ? 4 : 1, qw = mode == 1 ? 4 : fq_words;
    /* static typing pass: words per point of every node */
    unsigned* ww = (unsigned*)malloc(nnodes * sizeof(unsigned));
    for (unsigned k = 0; k < nnodes; k++) {
        const int kind = nodes3[3 * k]; const int64_t a = nodes3[3 * k + 1], b = nodes3[3 * k + 2];
        switch (kind) {
        case 0: ww[k] = pw; break;
        case 1: ww[k] = b ? qw : pw; break;
        case 2: case 3: ww[k] = qw; break;
        case 4: ww[k] = ((unsigned)a < nbase) ? pw : qw; break;
        case 5: ww[k] = pw; break;
        case 6: case 10: ww[k] = ww[a]; break;
        case 7: case 8: case 9: ww[k] = ww[a] > ww[b] ? ww[a] : ww[b]; break;
        default: free(ww); return -1;
        }
    }
    const uint64_t wroot = mode == 0 ? oracle_gl_root_of_unity(log_n) : 0;
    const uint64_t hm = mode == 0 ? gl_to_mont(offset_canon % GL_P) : 0;
    int err = 0;
    const size_t nchunks = (n + EV_CHUNK - 1) / EV_CHUNK;
    #pragma omp parallel for schedule(dynamic, 4)
    for (size_t ch = 0; ch < nchunks; ch++) {
        const size_t i0 = ch * EV_CHUNK, cnt = (n - i0 < EV_CHUNK) ? n - i0 : EV_CHUNK;
        uint64_t** val = (uint64_t**)calloc(nnodes, sizeof(uint64_t*));
        for (unsigned k = 0; k < nnodes; k++) {
            const int kind = nodes3[3 * k]; const int64_t a = nodes3[3 * k + 1], b = nodes3[3 * k + 2];
            const unsigned w = ww[k];
            uint64_t* v = (uint64_t*)malloc(cnt * w * 8);
            val[k] = v;
            if (kind == 0) {
                if (mode == 0) { uint64_t x = gl_mul(hm, gl_pow(wroot, i0)); for (size_t i = 0; i < cnt; i++) { v[i] = x; x = gl_mul(x, wroot); } }
                else {
                    f252 wr, x, e; memcpy(wr.l, w252, 32); memcpy(x.l, offset252, 32);
                    uint64_t ee[1] = {i0}; e = f252_pow(wr, ee, 1); x = f252_mul(x, e);
                    for (size_t i = 0; i < cnt; i++) { memcpy(v + 4 * i, x.l, 32); x = f252_mul(x, wr); }
                }
            } else if (kind == 1) { for (size_t i = 0; i < cnt; i++) memcpy(v + w * i, consts + a, w * 8); }
            else if (kind == 2) { for (size_t i = 0; i < cnt; i++) memcpy(v + w * i, challenges + (size_t)a * qw, w * 8); }
            else if (kind == 3) { for (size_t i = 0; i < cnt; i++) memcpy(v + w * i, hints + (size_t)a * qw, w * 8); }
            else if (kind == 4) {
                const uint64_t* col = ((unsigned)a < nbase) ? base_cols[a] : ext_cols[a - nbase];
                const int64_t shift = (int64_t)lde_step * b;
                for (size_t i = 0; i < cnt; i++) {
                    const size_t j = (size_t)(((int64_t)(i0 + i) + shift) % (int64_t)n + (int64_t)n) % n;
                    memcpy(v + w * i, col + w * j, w * 8);
                }
            } else if (kind == 5) {
                const uint64_t* t = periodic[a]; const size_t len = periodic_len[a];
                for (size_t i = 0; i < cnt; i++) memcpy(v + w * i, t + w * ((i0 + i) % len), w * 8);
            } else if (kind == 6) {
                const uint64_t* x = val[a];
                if (mode == 1) { for (size_t i = 0; i < cnt; i++) { f252 z = {{0,0,0,0}}, y; memcpy(y.l, x + 4 * i, 32); y = f252_sub(z, y); memcpy(v + 4 * i, y.l, 32); } }
                else for (size_t i = 0; i < cnt * w; i++) v[i] = gl_neg(x[i]);
            } else if (kind == 7 || kind == 8 || kind == 9) {
                const unsigned wa = ww[a], wb = ww[b];
                uint64_t* y = val[b];
                uint64_t* yinv = NULL;
                if (kind == 9) { yinv = (uint64_t*)malloc(cnt * wb * 8); memcpy(yinv, y, cnt * wb * 8); ev_batch_inverse(yinv, wb, cnt, mode); y = yinv; }
                const uint64_t* x = val[a];
                for (size_t i = 0; i < cnt; i++) {
                    if (mode == 1) {
                        f252 p, q; memcpy(p.l, x + 4 * i, 32); memcpy(q.l, y + 4 * i, 32);
                        p = (kind == 7) ? f252_add(p, q) : f252_mul(p, q); memcpy(v + 4 * i, p.l, 32);
                    } else if (w == 1) {
                        v[i] = (kind == 7) ? gl_add(x[i], y[i]) : gl_mul(x[i], y[i]);
                    } else {
                        fq3 p = wa == 3 ? (fq3){x[3*i], x[3*i+1], x[3*i+2]} : (fq3){x[i], 0, 0};
                        fq3 q = wb == 3 ? (fq3){y[3*i], y[3*i+1], y[3*i+2]} : (fq3){y[i], 0, 0};
                        fq3 r;
                        if (kind == 7) r = fq3_add(p, q);
                        else if (wa == 3 && wb == 3) r = fq3_mul(p, q);
                        else if (wa == 3) r = fq3_mul_fp(p, y[i]);
                        else r = fq3_mul_fp(q, x[i]);
                        v[3*i] = r.c0; v[3*i+1] = r.c1; v[3*i+2] = r.c2;
                    }
                }
                if (yinv) free(yinv);
            } else if (kind == 10) {
                const uint64_t* x = val[a];
                for (size_t i = 0; i < cnt; i++) {
                    if (mode == 1) { f252 p; memcpy(p.l, x + 4 * i, 32); uint64_t ee[1] = {(uint64_t)b}; p = f252_pow(p, ee, 1); memcpy(v + 4 * i, p.l, 32); }
                    else if (w == 1) v[i] = gl_pow(x[i], (uint64_t)b);
                    else { fq3 p = {x[3*i], x[3*i+1], x[3*i+2]}; p = fq3_pow(p, (uint64_t)b); v[3*i] = p.c0; v[3*i+1] = p.c1; v[3*i+2] = p.c2; }
                }
            }
        }
        /* into_fq_array */
        const unsigned wl = ww[nnodes - 1];
        const uint64_t* r = val[nnodes - 1];
        for (size_t i = 0; i < cnt; i++) {
            uint64_t* o = out + (i0 + i) * qw;
            if (wl == qw) memcpy(o, r + (size_t)qw * i, qw * 8);
            else { o[0] = r[i]; for (unsigned t = 1; t < qw; t++) o[t] = 0; }
        }
        for (unsigned k = 0; k < nnodes; k++) free(val[k]);
        free(val);
    }
    free(ww);
    return err;
}

/* ======================================================================================================
 * DEEP composition: horner_evaluate (src/utils.rs:124-133), divide_out_point(s)_into (src/utils.rs:151-175),
 * DeepPolyComposer::into_deep_poly's sum and degree adjustment (src/composer.rs:100-188).  Coefficients are
 * V-word elements (1 = Fp, 3 = Fq3), points / alphas are PW-word Fq elements (PW = 3, or 1 when Fq = Fp).
 * ====================================================================================================== */
static inline fq3 ld_q(const uint64_t* p, unsigned w) { fq3 r = {p[0], w == 3 ? p[1] : 0, w == 3 ? p[2] : 0}; return r; }
void oracle_horner_eval(const uint64_t* coeffs, size_t n, unsigned V, const uint64_t* point, unsigned PW, uint64_t* out) {
    if (PW == 1 && V == 1) { uint64_t acc = 0; for (size_t i = n; i-- > 0;) acc = gl_add(gl_mul(acc, point[0]), coeffs[i]); out[0] = acc; return; }
    fq3 z = ld_q(point, PW), acc = {0, 0, 0};
    for (size_t i = n; i-- > 0;) acc = fq3_add(fq3_mul(acc, z), ld_q(coeffs + (size_t)V * i, V));
    out[0] = acc.c0; if (PW == 3) { out[1] = acc.c1; out[2] = acc.c2; }
}
/* acc[i] += sum_k cs[k] * (coefficient i of coeffs(X) / (X - zs[k]))   -- divide_out_points_into, accumulated */
void oracle_divide_out_points_acc(const uint64_t* coeffs, size_t n, unsigned V, const uint64_t* zs, const uint64_t* cs, unsigned k,
                                  unsigned PW, uint64_t* acc) {
    fq3 rem[64], z[64], c[64];
    if (k > 64) return;
    for (unsigned t = 0; t < k; t++) { rem[t] = (fq3){0, 0, 0}; z[t] = ld_q(zs + (size_t)PW * t, PW); c[t] = ld_q(cs + (size_t)PW * t, PW); }
    for (size_t i = n; i-- > 0;) {
        const fq3 tmp = ld_q(coeffs + (size_t)V * i, V);
        fq3 s = {0, 0, 0};
        for (unsigned t = 0; t < k; t++) { s = fq3_add(s, fq3_mul(c[t], rem[t])); rem[t] = fq3_add(fq3_mul(z[t], rem[t]), tmp); }
        uint64_t* o = acc + (size_t)PW * i;
        o[0] = gl_add(o[0], s.c0); if (PW == 3) { o[1] = gl_add(o[1], s.c1); o[2] = gl_add(o[2], s.c2); }
    }
}
/* out[i] = acc[i] * da + acc[i-1] * db  (composer.rs:172-187; db == 0 skips the shifted term) */
void oracle_degree_adjust(const uint64_t* acc, size_t n, unsigned PW, const uint64_t* da, const uint64_t* db, uint64_t* out) {
    const fq3 a = ld_q(da, PW), b = ld_q(db, PW);
    fq3 last = {0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        const fq3 cur = ld_q(acc + (size_t)PW * i, PW);
        fq3 r = fq3_add(fq3_mul(cur, a), fq3_mul(last, b));
        last = cur;
        out[(size_t)PW * i] = r.c0; if (PW == 3) { out[(size_t)PW * i + 1] = r.c1; out[(size_t)PW * i + 2] = r.c2; }
    }
}

/* ---- Fp252 NTT / LDE -------------------------------------------------------------------------------------------------
   What the reference computes for Fp252 columns: ark-poly 0.4's Radix2EvaluationDomain::{fft_in_place, ifft_in_place} on a
   coset (third-party dependency, ark-poly = "0.4" in Cargo.toml; called from GpuFft / GpuIfft's contract gpu/src/plan.rs:236-325
   and from Matrix::into_evaluations / interpolate, src/matrix.rs:142-251): forward  y_k = sum_j (h^j x_j) w^(jk), inverse
   x_j = h^-j n^-1 sum_k y_k w^(-jk), w = 3^((p-1)/n) (gpu/src/fields.rs:241).  Restated as bit reversal + iterative radix-2
   butterflies over a twiddle table; checked against the big-integer oracle/pyref/ntt.py in tests/test_fp252_parity.py. */
static const f252 F252_R2 = {{18446741271209837569ULL, 5151653887ULL, 18446744073700081664ULL, 576413109808302096ULL}};   /* R^2 mod p (felt_u256.h.metal:103) */
static f252 f252_from_u64(uint64_t v) { f252 x = {{v, 0, 0, 0}}; return f252_mul(x, F252_R2); }
static f252 f252_root_of_unity(unsigned log_n) {
    const uint64_t e[1] = {0x0800000000000011ULL};                     /* (p - 1) / 2^192 */
    f252 r = f252_pow(f252_from_u64(3), e, 1);
    for (unsigned i = log_n; i < 192; i++) r = f252_mul(r, r);
    return r;
}
void oracle_f252_ntt(uint64_t* data, unsigned log_n, int inverse, const uint64_t* offset_mont) {
    const size_t n = (size_t)1 << log_n;
    f252* x = (f252*)data;
    f252 h; memcpy(h.l, offset_mont, 32);
    f252 w = f252_root_of_unity(log_n);
    if (inverse) w = f252_inv(w);
    if (!inverse) { f252 s = F252_ONE; for (size_t i = 0; i < n; i++) { x[i] = f252_mul(x[i], s); s = f252_mul(s, h); } }
    for (size_t i = 0; i < n; i++) {
        size_t r = 0;
        for (unsigned b = 0; b < log_n; b++) r |= ((i >> b) & 1) << (log_n - 1 - b);
        if (r > i) { f252 t = x[i]; x[i] = x[r]; x[r] = t; }
    }
    const size_t half_n = n > 1 ? n / 2 : 1;
    f252* tw = (f252*)malloc(half_n * sizeof(f252));
    tw[0] = F252_ONE;
    for (size_t i = 1; i < half_n; i++) tw[i] = f252_mul(tw[i - 1], w);
    for (unsigned s = 1; s <= log_n; s++) {
        const size_t half = (size_t)1 << (s - 1);
        #pragma omp parallel for schedule(static)
        for (size_t b = 0; b < n / 2; b++) {
            const size_t j = b & (half - 1), lo = ((b >> (s - 1)) << s) + j, hi = lo + half;
            const f252 t = f252_mul(x[hi], tw[j << (log_n - s)]), u = x[lo];
            x[lo] = f252_add(u, t);
            x[hi] = f252_sub(u, t);
        }
    }
    free(tw);
    if (inverse) {
        const f252 hinv = f252_inv(h);
        f252 s = f252_inv(f252_from_u64((uint64_t)n));
        for (size_t i = 0; i < n; i++) { x[i] = f252_mul(x[i], s); s = f252_mul(s, hinv); }
    }
}
/* interpolate on the subgroup, zero-extend, evaluate on the coset h<w_N>, optionally bit-reversed (src/prover.rs:50-51) */
void oracle_f252_lde(const uint64_t* in, uint64_t* out, unsigned log_n, unsigned log_blowup, const uint64_t* offset_mont, int bit_reversed) {
    const size_t n = (size_t)1 << log_n, N = n << log_blowup;
    memcpy(out, in, n * 32);
    oracle_f252_ntt(out, log_n, 1, F252_ONE.l);
    memset(out + 4 * n, 0, (N - n) * 32);
    oracle_f252_ntt(out, log_n + log_blowup, 0, offset_mont);
    if (bit_reversed) oracle_bit_reverse(out, log_n + log_blowup, 4);
}
