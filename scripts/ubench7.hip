// Round-2 micro-benchmark, part 4: one radix-16 level of the NTT (network + the general factors that follow it) on the
// MATRIX cores, against the limb-form network + multiply-fold of gl_limb.h.
//
//   y_k = t_k * sum_j w_16^(j k) x_j        (t_k = w_256^(k b): the factor between the two networks of a radix-256 pass)
//
// is a 16 x 16 matrix of field elements M[k][j] that depends only on the row digit b.  With the data word x_j cut into
// its 8 bytes and every entry pre-multiplied by 2^(8 beta) and cut into 8 balanced signed bytes,
//
//   S[k, a] = sum_{j, beta} byte_a(M[k][j] 2^(8 beta) mod p) * (byte_beta(x_j) - 128) + 2^22       (a, beta < 8)
//
// is a 128 x 128 by 128 x N signed-byte product = 16 v_mfma_i32_16x16x64_i8 per 16 columns (256 elements), and
//   y'_k = sum_a 2^(8 a) S[k, a]  =  y_k - C sum_j M[k][j] + Bc        (C = 0x8080..80, Bc = 2^22 * 0x0101..01)
// The -128 (bytes as signed) and the +2^22 (partial sums non-negative: the accumulator's initial value) are UNIFORM
// offsets of all inputs / outputs of a DFT, so they only ever reach output k = 0 of the next level (sum_j w^(j k) = 0
// otherwise): a real pass would fix one element in 16; this benchmark checks y' against the formula above.
//
// Operand layout (CK's WarpGemmAttributeMfmaImpl_i32_16x16x64_i8: A lane l = row l % 16, 16 consecutive K of block
// l / 16; B likewise per column; D lane l = column l % 16, rows 4 (l / 16) + r): lane (g, n) holds the elements
// j in {2g, 2g+1, 8+2g, 9+2g} of column n as plain 64-bit words -- the B operands are the data registers themselves --
// and receives k in the same set: row 16 rt + 4 g' + r of the byte matrix is (k = 2 g' + (tp & 1) + 8 (tp >> 1),
// a = r + 4 (rt & 1)), tp = rt >> 1.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ministark_amd/csrc scripts/ubench7.hip -o scripts/ubench7
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "gl_limb.h"
#include "gl_dev.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
static constexpr uint64_t XOR80 = 0x8080808080808080ull;

// 8 partial sums (each in [0, 2^23)) -> a weak 64-bit residue of sum_a 2^(8a) S_a
__device__ __forceinline__ uint64_t reduce8(const int* S) {
    uint64_t lo = (uint32_t)S[0], hi = (uint32_t)S[4];
    lo += (uint64_t)(uint32_t)S[1] << 8;  hi += (uint64_t)(uint32_t)S[5] << 8;
    lo += (uint64_t)(uint32_t)S[2] << 16; hi += (uint64_t)(uint32_t)S[6] << 16;
    lo += (uint64_t)(uint32_t)S[3] << 24; hi += (uint64_t)(uint32_t)S[7] << 24;
    // value = lo + hi 2^32, lo, hi < 2^48.  hi = h1 2^32 + h0:  hi 2^32 = h0 2^32 + h1 (2^32 - 1)  (mod p)
    const uint32_t h0 = (uint32_t)hi, h1 = (uint32_t)(hi >> 32);
    const uint64_t z = (uint64_t)h1 * 0xFFFFFFFFull + lo;                 // < 2^49
    const uint64_t s = z + ((uint64_t)h0 << 32);
    return s < z ? s + gl::EPS : s;                                         // wrapped: 2^64 = 2^32 - 1
}

// one level on the matrix cores; A fragments [rt][kh] resident in registers (64 VGPRs), 4 column groups per lane.
// PIPE: the four MFMAs of the NEXT output element are issued before the reduction of the current one, so that the
// reduction (VALU) runs while the matrix pipe works.
template <bool PIPE>
__global__ void __launch_bounds__(256, 4) k_mfma(uint64_t* data, const v4i* __restrict__ amat, int iters) {
    const unsigned lane = threadIdx.x & 63;
    v4i A[8][2];
    #pragma unroll
    for (int rt = 0; rt < 8; rt++)
        #pragma unroll
        for (int kh = 0; kh < 2; kh++) A[rt][kh] = amat[(rt * 2 + kh) * 64 + lane];
    uint64_t x[4][4];
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    #pragma unroll
    for (int q = 0; q < 4; q++)
        #pragma unroll
        for (int e = 0; e < 4; e++) x[q][e] = data[base + q * 4 + e];
    const v4i bias = {1 << 22, 1 << 22, 1 << 22, 1 << 22};
    auto issue = [&](const v4i& B0, const v4i& B1, int tp, v4i& lo, v4i& hi) {
        lo = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2 * tp][0], B0, bias, 0, 0, 0);
        hi = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2 * tp + 1][0], B0, bias, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2 * tp][1], B1, lo, 0, 0, 0);
        hi = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2 * tp + 1][1], B1, hi, 0, 0, 0);
    };
    auto red = [&](const v4i& lo, const v4i& hi) {
        int S[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return reduce8(S);
    };
    for (int it = 0; it < iters; it++) {
        v4i B0[4], B1[4];
        #pragma unroll
        for (int q = 0; q < 4; q++) {
            uint64_t xs[4];
            #pragma unroll
            for (int e = 0; e < 4; e++) xs[e] = x[q][e] ^ XOR80;
            B0[q][0] = (int)(uint32_t)xs[0]; B0[q][1] = (int)(uint32_t)(xs[0] >> 32); B0[q][2] = (int)(uint32_t)xs[1]; B0[q][3] = (int)(uint32_t)(xs[1] >> 32);
            B1[q][0] = (int)(uint32_t)xs[2]; B1[q][1] = (int)(uint32_t)(xs[2] >> 32); B1[q][2] = (int)(uint32_t)xs[3]; B1[q][3] = (int)(uint32_t)(xs[3] >> 32);
        }
        if constexpr (PIPE) {
            v4i lo[2], hi[2];
            issue(B0[0], B1[0], 0, lo[0], hi[0]);
            #pragma unroll
            for (int u = 0; u < 16; u++) {                    // u = 4 q + tp
                if (u < 15) issue(B0[(u + 1) >> 2], B1[(u + 1) >> 2], (u + 1) & 3, lo[(u + 1) & 1], hi[(u + 1) & 1]);
                x[u >> 2][u & 3] = red(lo[u & 1], hi[u & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            #pragma unroll
            for (int u = 0; u < 16; u++) {
                v4i lo, hi;
                issue(B0[u >> 2], B1[u >> 2], u & 3, lo, hi);
                x[u >> 2][u & 3] = red(lo, hi);
            }
        }
    }
    #pragma unroll
    for (int q = 0; q < 4; q++)
        #pragma unroll
        for (int e = 0; e < 4; e++) data[base + q * 4 + e] = x[q][e];
}

// the same level in limb form (ubench5's k_full4): convert, network, multiply-fold by wave-uniform factors
__global__ void __launch_bounds__(256, 4) k_limb(uint64_t* data, const uint64_t* __restrict__ wt, int iters) {
    using namespace glimb;
    uint64_t x[16];
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    for (int a = 0; a < 16; a++) x[a] = data[base + a];
    for (int it = 0; it < iters; it++) {
        L4 v[16];
        #pragma unroll
        for (int a = 0; a < 16; a++) v[a] = from_u64(x[a]);
        dft<16, false>(v);
        #pragma unroll
        for (int c = 0; c < 16; c++) {
            const uint64_t* wp = wt + c * 4;
            x[c] = mul_fold(v[c], w4_from(wp[0], wp[1], wp[2], wp[3]));
        }
    }
    for (int a = 0; a < 16; a++) data[base + a] = x[a];
}

// MFMA issue only (no reduction): the matrix pipe's own time for the 16 instructions per group
__global__ void __launch_bounds__(256, 4) k_mfma_only(uint64_t* data, const v4i* __restrict__ amat, int iters) {
    const unsigned lane = threadIdx.x & 63;
    v4i A[8][2];
    #pragma unroll
    for (int rt = 0; rt < 8; rt++)
        #pragma unroll
        for (int kh = 0; kh < 2; kh++) A[rt][kh] = amat[(rt * 2 + kh) * 64 + lane];
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    v4i B0, B1;
    for (int e = 0; e < 4; e++) { B0[e] = (int)data[base + e]; B1[e] = (int)data[base + 4 + e]; }
    v4i acc[8];
    #pragma unroll
    for (int rt = 0; rt < 8; rt++) acc[rt] = v4i{0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int q = 0; q < 4; q++)
            #pragma unroll
            for (int rt = 0; rt < 8; rt++) {
                acc[rt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[rt][0], B0, acc[rt], 0, 0, 0);
                acc[rt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[rt][1], B1, acc[rt], 0, 0, 0);
            }
    }
    int s = 0;
    #pragma unroll
    for (int rt = 0; rt < 8; rt++) s ^= acc[rt][0] ^ acc[rt][1] ^ acc[rt][2] ^ acc[rt][3];
    data[base] = (uint64_t)(uint32_t)s;
}

// Do the matrix pipe and the vector ALU overlap at all?  MODE 0: 16 MFMAs per step; 1: NV independent v_mad_u64_u32
// per step; 2: both in the same wave, no dependence between the two streams.
template <int MODE, int NV, bool VOP2 = false>
__global__ void __launch_bounds__(256, 4) k_mix(uint64_t* data, const v4i* __restrict__ amat, int iters) {
    const unsigned lane = threadIdx.x & 63;
    v4i A[4][2];
    #pragma unroll
    for (int rt = 0; rt < 4; rt++)
        #pragma unroll
        for (int kh = 0; kh < 2; kh++) A[rt][kh] = amat[(rt * 2 + kh) * 64 + lane];
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    v4i B0, B1;
    for (int e = 0; e < 4; e++) { B0[e] = (int)data[base + e]; B1[e] = (int)data[base + 4 + e]; }
    v4i acc[4];
    #pragma unroll
    for (int rt = 0; rt < 4; rt++) acc[rt] = v4i{0, 0, 0, 0};
    uint64_t c[8];
    #pragma unroll
    for (int i = 0; i < 8; i++) c[i] = data[base + 8 + i];
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int q = 0; q < 4; q++) {
            if (MODE != 1) {
                #pragma unroll
                for (int rt = 0; rt < 4; rt++) {
                    acc[rt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[rt][0], B0, acc[rt], 0, 0, 0);
                    acc[rt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[rt][1], B1, acc[rt], 0, 0, 0);
                }
            }
            if (MODE != 0) {
                #pragma unroll
                for (int v = 0; v < NV; v++) {
                    if (VOP2) { uint32_t lo = (uint32_t)c[v & 7], o = (uint32_t)c[(v + 3) & 7]; lo = (lo + o) ^ (o >> 3); c[v & 7] = (c[v & 7] & 0xFFFFFFFF00000000ull) | lo; }   // add, shift, xor: plain 32-bit VOP2
                    else c[v & 7] = (uint64_t)(uint32_t)c[v & 7] * 0x9E3779B1u + c[(v + 3) & 7];
                }
            }
        }
    }
    int s = 0;
    #pragma unroll
    for (int rt = 0; rt < 4; rt++) s ^= acc[rt][0] ^ acc[rt][1] ^ acc[rt][2] ^ acc[rt][3];
    uint64_t t = 0;
    #pragma unroll
    for (int i = 0; i < 8; i++) t ^= c[i];
    data[base] = (uint64_t)(uint32_t)s ^ t;
}

// ---------------------------------------------------------------------------------------------------------------- host
static void balanced_bytes(uint64_t v, int8_t* d) {       // canonical v < p -> 8 signed bytes of v or v - p
    // representative r with sum d_a 256^a = r, d_a in [-128, 127]:  r in [-(128/255)(2^64 - 1), (127/255)(2^64 - 1)]
    const unsigned __int128 maxpos = ((unsigned __int128)127 * 0xFFFFFFFFFFFFFFFFull) / 255;
    __int128 r = (__int128)v;
    if ((unsigned __int128)v > maxpos) r -= (__int128)gl::P;
    for (int a = 0; a < 8; a++) {
        int dgt = (int)(((r % 256) + 256) % 256);
        if (dgt >= 128) dgt -= 256;
        d[a] = (int8_t)dgt;
        r = (r - dgt) / 256;
    }
    if (r != 0) { printf("balanced digits overflow\n"); exit(1); }
}

template <typename K, typename... Args>
static double time_kernel(K kern, int blocks, Args... args) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, args...);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, args...);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    return best;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d\n", prop.name, prop.multiProcessorCount);
    const unsigned b = 5;
    const uint64_t w256 = gl::root_of_unity(8), w16 = gl::pow(w256, 16);
    uint64_t M[16][16];
    for (int k = 0; k < 16; k++) for (int j = 0; j < 16; j++) M[k][j] = gl::mul(gl::pow(w16, (uint64_t)(j * k) % 16), gl::pow(w256, (uint64_t)k * b));
    // A fragments: [rt][kh][lane] 16 bytes
    std::vector<int8_t> frag((size_t)16 * 64 * 16);
    for (int rt = 0; rt < 8; rt++) for (int kh = 0; kh < 2; kh++) for (int lane = 0; lane < 64; lane++) {
        const int i = lane & 15, gk = lane >> 4, gp = i >> 2, r = i & 3, tp = rt >> 1;
        const int k = 2 * gp + (tp & 1) + 8 * (tp >> 1), a = r + 4 * (rt & 1);
        for (int q = 0; q < 16; q++) {
            const int j = 8 * kh + 2 * gk + (q >> 3), beta = q & 7;
            int8_t d[8];
            balanced_bytes(gl::mul(M[k][j], gl::pow(2, 8 * beta)), d);
            frag[(((size_t)(rt * 2 + kh) * 64) + lane) * 16 + q] = d[a];
        }
    }
    v4i* d_amat; CK(hipMalloc(&d_amat, frag.size())); CK(hipMemcpy(d_amat, frag.data(), frag.size(), hipMemcpyHostToDevice));
    const int blocks = 256 * 8;
    const size_t words = (size_t)blocks * 256 * 16;
    std::vector<uint64_t> h(words), out(words);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (auto& v : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = s; }      // any 64-bit word (weak residues)
    uint64_t *d_data, *d_wt; CK(hipMalloc(&d_data, words * 8)); CK(hipMalloc(&d_wt, 64 * 8));
    std::vector<uint64_t> wt(64);
    for (auto& v : wt) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = s % gl::P; }
    CK(hipMemcpy(d_wt, wt.data(), 64 * 8, hipMemcpyHostToDevice));

    // ---- correctness of one level
    CK(hipMemcpy(d_data, h.data(), words * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mfma<true>, dim3(blocks), dim3(256), 0, 0, d_data, d_amat, 1);
    CK(hipMemcpy(out.data(), d_data, words * 8, hipMemcpyDeviceToHost));
    size_t bad = 0, checked = 0;
    const uint64_t Cc = XOR80 % gl::P, Bc = gl::mul((uint64_t)1 << 22, 0x0101010101010101ull % gl::P);
    for (int blk : {0, 7, blocks - 1}) for (int wave = 0; wave < 4; wave++) for (int q = 0; q < 4; q++) for (int n = 0; n < 16; n++) {
        // column (blk, wave, q, n): element j held by lane (g, n), slot e:  j = 2g + (e & 1) + 8 (e >> 1)
        uint64_t xin[16], yout[16];
        for (int g = 0; g < 4; g++) for (int e = 0; e < 4; e++) {
            const size_t idx = ((size_t)blk * 256 + wave * 64 + g * 16 + n) * 16 + q * 4 + e;
            const int j = 2 * g + (e & 1) + 8 * (e >> 1);
            xin[j] = h[idx]; yout[j] = out[idx];
        }
        for (int k = 0; k < 16; k++) {
            uint64_t acc = Bc;
            for (int j = 0; j < 16; j++) acc = gl::add(acc, gl::mul(M[k][j], gl::sub(xin[j] % gl::P, Cc)));
            checked++;
            if (yout[k] % gl::P != acc) { if (bad < 5) printf("MISMATCH blk %d wave %d q %d n %d k %d: %016llx vs %016llx\n", blk, wave, q, n, k, (unsigned long long)(yout[k] % gl::P), (unsigned long long)acc); bad++; }
        }
    }
    printf("CHECK one MFMA level against the field formula: %zu outputs, %zu mismatches\n", checked, bad);

    // ---- timing
    const int iters = 64;
    for (int bl : {256 * 8, 256 * 2}) {
        const double wave_elems = (double)bl * 4 * 16 * iters;
        auto cyc = [&](double ms) { return ms * 1e-3 * 2.4e9 * 1024.0 / wave_elems; };
        double t1 = time_kernel(k_limb, bl, d_data, (const uint64_t*)d_wt, iters);
        double t2 = time_kernel(k_mfma<false>, bl, d_data, (const v4i*)d_amat, iters);
        double t4 = time_kernel(k_mfma<true>, bl, d_data, (const v4i*)d_amat, iters);
        double t3 = time_kernel(k_mfma_only, bl, d_data, (const v4i*)d_amat, iters);
        printf("LEVEL blocks=%5d  limb network + multiply-fold   %8.3f ms => %6.1f cycles per element per SIMD @2.4GHz\n", bl, t1, cyc(t1));
        printf("LEVEL blocks=%5d  matrix cores + reduction       %8.3f ms => %6.1f cycles per element per SIMD @2.4GHz\n", bl, t2, cyc(t2));
        printf("LEVEL blocks=%5d  matrix cores + reduction, MFMAs one element ahead %8.3f ms => %6.1f cycles per element per SIMD @2.4GHz\n", bl, t4, cyc(t4));
        printf("LEVEL blocks=%5d  matrix cores, MFMA issue only  %8.3f ms => %6.1f cycles per element per SIMD @2.4GHz  (%.0f TOPS)\n", bl, t3, cyc(t3),
               (double)bl * 4 * iters * 4 * 16 * 2.0 * 16 * 16 * 64 / (t3 * 1e-3) / 1e12);
    }
    {   // overlap probe: per step 8 MFMAs and 40 multiply-adds
        const int bl = 256 * 8;
        const double steps = (double)bl * 4 * iters * 4;                     // wave-steps
        auto cyc = [&](double ms) { return ms * 1e-3 * 2.4e9 * 1024.0 / steps; };
        double a = time_kernel(k_mix<0, 40>, bl, d_data, (const v4i*)d_amat, iters);
        double b2 = time_kernel(k_mix<1, 40>, bl, d_data, (const v4i*)d_amat, iters);
        double c2 = time_kernel(k_mix<2, 40>, bl, d_data, (const v4i*)d_amat, iters);
        printf("OVERLAP per step and SIMD: 8 MFMAs alone %6.1f cycles, 40 v_mad_u64_u32 alone %6.1f cycles, both in one wave %6.1f cycles\n", cyc(a), cyc(b2), cyc(c2));
        double b3 = time_kernel(k_mix<1, 40, true>, bl, d_data, (const v4i*)d_amat, iters);
        double c3 = time_kernel(k_mix<2, 40, true>, bl, d_data, (const v4i*)d_amat, iters);
        printf("OVERLAP per step and SIMD: 8 MFMAs alone %6.1f cycles, 40 x (add, shift, xor) alone %6.1f cycles, both in one wave %6.1f cycles\n", cyc(a), cyc(b3), cyc(c3));
    }
    return bad ? 1 : 0;
}
