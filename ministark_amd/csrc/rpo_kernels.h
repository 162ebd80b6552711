// Rescue-Prime-Optimized (RPO-256, eprint 2022/1577) hashing over Goldilocks for gfx950:
// the reference's Rpo256AbsorbColumnsAndPermute / Rpo256AbsorbRowsAndPermute /
// Rpo256GenMerkleNodes{FirstRow,Row} kernels (gpu/src/metal/hash_shaders.h.metal:215-380) and their
// front-ends GpuRpo256ColumnMajor / GpuRpo256RowMajor / gen_rpo_merkle_tree (gpu/src/plan.rs:32-174).
// (Built but not wired into the reference prover: README.md:90 "coming soon".)
//   state: 12 elements, capacity [0,4), rate [4,12); 7 rounds of
//          MDS ; + RC0 ; x^7 ; MDS ; + RC1 ; x^(1/7)          (hash_shaders.h.metal:235-261)
//   absorb: the 8 rate elements are OVERWRITTEN with the input (:283-290), digest = state[4..8)
//   padding (plan.rs:82-97, stage.rs:1246-1252): if the number of columns is not a multiple of 8,
//          capacity[0] starts at 1 and the last block is completed by a 1 followed by zeros
//   Merkle: node = permute(0,0,0,0, left, right)[4..8)         (:336-380), nodes[1] = root
// MDS is the circulant with first row (7,23,8,26,13,10,9,7,6,22,21,8) (:41-54, Montgomery form there);
// its entries are < 2^5, so a row of the product is accumulated as a 69-bit integer and reduced
// once (no field multiplications) -- same values as the reference's frequency-domain variant.
// One lane per row: every absorb of a row is fused into one kernel, the state stays in registers
// and the per-absorb state/digest round trips of the reference disappear.  ALU-bound: the
// inverse S-box costs 72 field multiplications per element per round.
// The round constants are data taken from the reference (Montgomery form, :57-76).
#pragma once
#include <hip/hip_runtime.h>
#include "gl.h"
#include "gl_dev.h"

namespace msrpo {

static constexpr int NT = 256;
static constexpr int MAXCOLS = 128;

__device__ static const uint64_t RC0[84] = {
    6936159699454947676ull, 6871277616928621393ull, 4226339945476756083ull, 2261225084505152444ull,
    16808067423291017741ull, 12862191241011323277ull, 345720808813194915ull, 10126368034161173654ull,
    840649715788759894ull, 18155600607269645987ull, 16577339120870559289ull, 13749826054300849029ull,
    16047969944113931191ull, 10474334246235299199ull, 15773847146013662260ull, 14401231158322525155ull,
    6009395255763488383ull, 2108579439821148946ull, 13820200715803196660ull, 15968614366574245570ull,
    7529997729792773654ull, 9429194013557833999ull, 11639903126146281421ull, 15759666882357935738ull,
    14807658266593669785ull, 17258259860767641342ull, 9534132615398591413ull, 358719342502509866ull,
    7123090532818864651ull, 734193187930710962ull, 14873184913735487023ull, 17965359964069906568ull,
    12664837478844326631ull, 15575491070113731145ull, 7221479899469196675ull, 7328957460733188967ull,
    15088355010936495340ull, 16762963605345901631ull, 15278161326153175940ull, 6257793333052173411ull,
    8418953127708045776ull, 6523475766574412380ull, 15192936988185261803ull, 1578086224854546096ull,
    10840553425559156784ull, 7453417405109536362ull, 5173069484734008228ull, 3284492202065476384ull,
    1724586709636399686ull, 17997633752581871175ull, 1284825320737914582ull, 960534381847281815ull,
    6708901808183456837ull, 8975591106768797316ull, 52515315389099119ull, 10009391031874081397ull,
    3091228317422201238ull, 1063858230459024983ull, 3396548655473917480ull, 15046057790353688034ull,
    4867464583127666756ull, 13816959924674544309ull, 13931201815459591565ull, 11494116713280125381ull,
    16823081743980874023ull, 6760771226809185048ull, 5346741505458044699ull, 15124596060558844029ull,
    5332565678905773189ull, 17640389307200936126ull, 14049814539797608740ull, 8882709539093378074ull,
    10507930462458090835ull, 10669463960502417047ull, 16753662827442720769ull, 12967456627495301601ull,
    2989815121821278695ull, 5894674479204135685ull, 14187454698288462352ull, 14795723369628125345ull,
    17260571099239679821ull, 16009836214833755168ull, 2009092225887788829ull, 10838446069154019765ull,};
__device__ static const uint64_t RC1[84] = {
    8939123259393952351ull, 14708045228210488368ull, 18125168669810517809ull, 9309821433754818185ull,
    4714467145607136006ull, 1302482025306688824ull, 34829973686821040ull, 5637233680011148778ull,
    227119480134509573ull, 2530972937109017559ull, 7210163798538732239ull, 955913576003606833ull,
    4449617297638325218ull, 10843671682695268638ull, 13198957499160452915ull, 11541825028620451829ull,
    10963484480734735121ull, 4752902142121643229ull, 3015289210993491059ull, 16344286514680205966ull,
    1811079964700766606ull, 12735664961476037524ull, 5775391330037813314ull, 18223625362487900986ull,
    7222477607687412281ull, 4215615082079701144ull, 6177508277476483691ull, 3491362079220677263ull,
    10961785333913978630ull, 1935408839283360916ull, 13974192629927279950ull, 18013556876298568088ull,
    7565676920589638093ull, 9265825103386412558ull, 8061587790235022972ull, 6806849270604947860ull,
    8066442548506952806ull, 12791828131640457742ull, 9268748809821748950ull, 17496234860625277598ull,
    13583894547367420658ull, 13920282495726802458ull, 3933141341199584259ull, 6658057712176150702ull,
    16812362035931029194ull, 15160401867587809089ull, 16411108749946146942ull, 3390826434320009844ull,
    18405475140095477472ull, 13864039573264702148ull, 496144052468360460ull, 9791523668470936672ull,
    528582340156917005ull, 15864481364569144493ull, 682830611952089590ull, 347158833826327515ull,
    13752775429919623417ull, 10254722988306758482ull, 8794150602427420596ull, 2480344122229837853ull,
    15462337562022968595ull, 6729968753311049611ull, 9250220857258211097ull, 12031447985684644003ull,
    14538803180331344696ull, 4055445230671851890ull, 14764039661528567501ull, 2047787218814287270ull,
    8977863094202715520ull, 6560450968915612407ull, 9976241128570886075ull, 17877509887772213755ull,
    3549624494907837709ull, 4253629935471652443ull, 2859199883984623807ull, 1087607721547343649ull,
    7907517619951970198ull, 11306402795121903516ull, 10168009948206732524ull, 9177440083248248246ull,
    13169036816957726187ull, 12924186209140199217ull, 9673006056831483321ull, 747828276541750689ull,};
__device__ static const uint32_t MDS_ROW[12] = {7, 23, 8, 26, 13, 10, 9, 7, 6, 22, 21, 8};

struct State { uint64_t s[12]; };

// out[m] = sum_n MDS[m][n] * s[n],  MDS[m][n] = MDS_ROW[(n - m) mod 12]; canonical in, canonical out
__device__ __forceinline__ void apply_mds(State& st) {
    uint64_t out[12];
    #pragma unroll
    for (int m = 0; m < 12; m++) {
        unsigned __int128 acc = 0;
        #pragma unroll
        for (int n = 0; n < 12; n++) acc += (unsigned __int128)st.s[n] * MDS_ROW[(n - m + 12) % 12];
        out[m] = gl::reduce128((uint64_t)acc, (uint64_t)(acc >> 64));
    }
    #pragma unroll
    for (int m = 0; m < 12; m++) st.s[m] = out[m];
}
__device__ __forceinline__ uint64_t sqn(uint64_t x, int n) { for (int i = 0; i < n; i++) x = gld::mmul(x, x); return x; }
__device__ __forceinline__ uint64_t pow7(uint64_t x) {
    const uint64_t x2 = gld::mmul(x, x), x4 = gld::mmul(x2, x2);
    return gld::mmul(gld::mmul(x4, x2), x);
}
// x^10540996611094048183 = x^(1/7): 72 multiplications (felt_u64.h.metal:59-73)
__device__ __forceinline__ uint64_t pow_inv7(uint64_t x) {
    const uint64_t t1 = gld::mmul(x, x);
    const uint64_t t2 = gld::mmul(t1, t1);
    const uint64_t t3 = gld::mmul(sqn(t2, 3), t2);
    const uint64_t t4 = gld::mmul(sqn(t3, 6), t3);
    const uint64_t t5 = gld::mmul(sqn(t4, 12), t4);
    const uint64_t t6 = gld::mmul(sqn(t5, 6), t3);
    const uint64_t t7 = gld::mmul(sqn(t6, 31), t6);
    const uint64_t t8 = sqn(gld::mmul(gld::mmul(t7, t7), t6), 2);
    return gld::mmul(gld::mmul(gld::mmul(t1, t2), x), t8);
}
// The same chain on the 12 state elements AT ONCE, step by step: twelve independent products are in flight at every step of the
// addition chain instead of one element's 72 dependent ones after the other (the S-box is 90 % of a permutation and a lane runs at two
// waves per SIMD: the dependent chains were issue-latency-bound at half the vector ALU rate).
__device__ __forceinline__ void sq_all(uint64_t* x, int n) {
    for (int i = 0; i < n; i++) {
        #pragma unroll
        for (int j = 0; j < 12; j++) x[j] = gld::mmul(x[j], x[j]);
    }
}
__device__ __forceinline__ void mul_all(uint64_t* x, const uint64_t* y) {
    #pragma unroll
    for (int j = 0; j < 12; j++) x[j] = gld::mmul(x[j], y[j]);
}
__device__ __forceinline__ void pow_inv7_all(uint64_t* s) {
    uint64_t head[12], t3[12], cur[12];
    #pragma unroll
    for (int j = 0; j < 12; j++) {
        const uint64_t t1 = gld::mmul(s[j], s[j]);
        const uint64_t t2 = gld::mmul(t1, t1);
        head[j] = gld::mmul(gld::mmul(t1, t2), s[j]);      // t1 t2 x, the factor of the last step
        cur[j] = t2; t3[j] = t2;
    }
    sq_all(cur, 3); mul_all(cur, t3);                        // t3 = t2^(2^3) t2
    #pragma unroll
    for (int j = 0; j < 12; j++) t3[j] = cur[j];
    sq_all(cur, 6); mul_all(cur, t3);                        // t4 = t3^(2^6) t3
    uint64_t t4[12];
    #pragma unroll
    for (int j = 0; j < 12; j++) t4[j] = cur[j];
    sq_all(cur, 12); mul_all(cur, t4);                       // t5 = t4^(2^12) t4
    sq_all(cur, 6); mul_all(cur, t3);                        // t6 = t5^(2^6) t3
    #pragma unroll
    for (int j = 0; j < 12; j++) t3[j] = cur[j];             // t3 <- t6
    sq_all(cur, 31); mul_all(cur, t3);                       // t7 = t6^(2^31) t6
    sq_all(cur, 1); mul_all(cur, t3);                        // t7^2 t6
    sq_all(cur, 2);                                          // t8
    #pragma unroll
    for (int j = 0; j < 12; j++) s[j] = gld::mmul(head[j], cur[j]);
}
__device__ __forceinline__ void permute(State& st) {
    for (int r = 0; r < 7; r++) {
        apply_mds(st);
        #pragma unroll
        for (int j = 0; j < 12; j++) st.s[j] = pow7(gl::add(st.s[j], RC0[r * 12 + j]));
        apply_mds(st);
        #pragma unroll
        for (int j = 0; j < 12; j++) st.s[j] = gl::add(st.s[j], RC1[r * 12 + j]);
        pow_inv7_all(st.s);
    }
}

struct RowsParams {
    const uint64_t* cols[MAXCOLS];
    uint64_t* digests;       // nrows x 4 elements
    size_t nrows;
    unsigned ncols;
    unsigned row_stride;     // words between rows of one column: 1 (column-major) or ncols (row-major matrix)
};
static __global__ void __launch_bounds__(NT) rpo256_rows(RowsParams P) {
    const size_t r = (size_t)blockIdx.x * NT + threadIdx.x;
    if (r >= P.nrows) return;
    State st;
    const bool pad = (P.ncols % 8) != 0;
    st.s[0] = pad ? gl::ONE_MONT : 0; st.s[1] = 0; st.s[2] = 0; st.s[3] = 0;
    #pragma unroll
    for (int j = 4; j < 12; j++) st.s[j] = 0;
    const unsigned nabsorb = (P.ncols + 7) / 8;
    for (unsigned a = 0; a < nabsorb; a++) {
        #pragma unroll
        for (int j = 0; j < 8; j++) {
            const unsigned c = a * 8 + j;
            uint64_t v = 0;
            if (c < P.ncols) v = P.cols[c][r * P.row_stride];
            else if (c == P.ncols) v = gl::ONE_MONT;          // "a single 1 element followed by zeros"
            st.s[4 + j] = v;
        }
        permute(st);
    }
    uint64_t* o = P.digests + 4 * r;
    o[0] = st.s[4]; o[1] = st.s[5]; o[2] = st.s[6]; o[3] = st.s[7];
}
// dst[i] = merge(src[2i], src[2i+1]) for i < count; digests are 4 elements
static __global__ void __launch_bounds__(NT) rpo256_merge_level(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, size_t count) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= count) return;
    State st;
    st.s[0] = 0; st.s[1] = 0; st.s[2] = 0; st.s[3] = 0;
    #pragma unroll
    for (int j = 0; j < 8; j++) st.s[4 + j] = src[8 * i + j];
    permute(st);
    uint64_t* o = dst + 4 * i;
    o[0] = st.s[4]; o[1] = st.s[5]; o[2] = st.s[6]; o[3] = st.s[7];
}

// The upper levels of a tree: a level of `count` nodes on one lane each takes one permutation's latency (~230 us: 7 rounds of
// 76 dependent-chain steps on 12 elements) however few nodes it has -- the 16 levels of <= 2^15 nodes were 5 of the 12.5 ms of a
// 2^20-leaf commitment.  Here SIXTEEN lanes share a node: lane j < 12 owns state element j, runs its S-box chains alone (76 products
// per round instead of 12 x 76) and the two MDS products of a round gather the twelve elements through LDS (one 8-byte write, twelve
// reads, the lane's own rotation of the circulant row in registers).  Same integer arithmetic per element as `permute`: same digests.
static constexpr int NTW = 64;           // one wave: four nodes
__device__ __forceinline__ uint64_t mds_wide(uint64_t s, uint64_t* sh, unsigned j, bool owner, const uint32_t* row) {
    if (owner) sh[j] = s;
    __syncthreads();
    unsigned __int128 acc = 0;
    #pragma unroll
    for (int n = 0; n < 12; n++) acc += (unsigned __int128)sh[n] * row[n];
    __syncthreads();
    return gl::reduce128((uint64_t)acc, (uint64_t)(acc >> 64));
}
static __global__ void __launch_bounds__(NTW) rpo256_merge_level_wide(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, size_t count) {
    __shared__ uint64_t sh_all[NTW / 16][12];
    const unsigned g = threadIdx.x / 16, j16 = threadIdx.x % 16;
    const bool owner = j16 < 12;
    const unsigned j = owner ? j16 : j16 - 12;                       // the four spare lanes shadow elements 0..3 (results unused)
    const size_t i = (size_t)blockIdx.x * (NTW / 16) + g;
    const bool live = i < count;
    uint32_t row[12];
    #pragma unroll
    for (int n = 0; n < 12; n++) row[n] = MDS_ROW[(n + 12 - j) % 12];
    uint64_t s = (live && j16 >= 4 && owner) ? src[8 * i + (j16 - 4)] : 0;
    uint64_t* sh = sh_all[g];
    for (int r = 0; r < 7; r++) {
        s = mds_wide(s, sh, j, owner, row);
        s = pow7(gl::add(s, RC0[r * 12 + j]));
        s = mds_wide(s, sh, j, owner, row);
        s = pow_inv7(gl::add(s, RC1[r * 12 + j]));
    }
    if (live && j16 >= 4 && j16 < 8) dst[4 * i + (j16 - 4)] = s;
}

}  // namespace msrpo
