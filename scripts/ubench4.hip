// Round-2 micro-benchmarks that size the two-pass (4096 x 4096) NTT on MI355X:
//   (A) cycles per element of one radix-16 network + general twiddle, 24-bit-limb form (gl_limb.h)
//       against the 64-bit form of round 1 (gl_dev.h)
//   (B) HBM behaviour of the two passes' access patterns on 8 cold 128 MiB columns, with a dummy
//       VALU load between the loads and the stores (how well do memory and arithmetic overlap at
//       1 x 1024-thread or 2 x 512-thread workgroups per CU?)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ministark_amd/csrc scripts/ubench4.hip -o scripts/ubench4
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "gl_limb.h"
#include "gl_dev.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

using namespace glimb;

// ------------------------------------------------------------------ (A)
__global__ void __launch_bounds__(256) k_net_uniform(uint64_t* data, const uint64_t* __restrict__ wt, int iters) {
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
            const uint64_t* wp = wt + (size_t)(it & 7) * 64 + c * 4;   // uniform address -> scalar loads
            x[c] = mul_fold(v[c], w4_from(wp[0], wp[1], wp[2], wp[3]));
        }
    }
    for (int a = 0; a < 16; a++) data[base + a] = x[a];
}
// twiddles from a 4096-entry table in LDS, four lookups per element at lane-dependent indices
__global__ void __launch_bounds__(256, 4) k_net_lds(uint64_t* data, const uint64_t* __restrict__ wt, int iters) {
    __shared__ uint64_t tab[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) tab[i] = wt[i & 511] + i;
    __syncthreads();
    uint64_t x[16];
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    for (int a = 0; a < 16; a++) x[a] = data[base + a];
    const unsigned cidx = (threadIdx.x >> 3) & 15, m0 = (threadIdx.x >> 7);
    for (int it = 0; it < iters; it++) {
        L4 v[16];
        #pragma unroll
        for (int a = 0; a < 16; a++) v[a] = from_u64(x[a]);
        dft<16, false>(v);
        #pragma unroll
        for (int c = 0; c < 16; c++) {
            const unsigned e = (cidx * (m0 + 16 * c + it)) & 4095;
            const uint64_t* wp = tab + e;
            x[c] = mul_fold(v[c], w4_from(wp[0], wp[2560], wp[1024], wp[3584]));
        }
    }
    for (int a = 0; a < 16; a++) data[base + a] = x[a];
}
__global__ void __launch_bounds__(256) k_net_old(uint64_t* data, const uint64_t* __restrict__ wt, int iters) {
    uint64_t x[16];
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    for (int a = 0; a < 16; a++) x[a] = gld::canon(data[base + a]);
    for (int it = 0; it < iters; it++) {
        gld::dft_lazy<16, false>(x);
        #pragma unroll
        for (int c = 0; c < 16; c++) x[c] = gld::mmul(x[c], wt[(it & 7) * 64 + c * 4]);
    }
    for (int a = 0; a < 16; a++) data[base + a] = x[a];
}
// pass-boundary pieces: to_weak + canon + Montgomery twiddle (store side), mul_to_limbs (load side)
__global__ void __launch_bounds__(256) k_boundary(uint64_t* data, const uint64_t* __restrict__ wt, int iters) {
    uint64_t x[16];
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    for (int a = 0; a < 16; a++) x[a] = data[base + a];
    uint64_t tw = wt[threadIdx.x & 63], B = wt[64 + (threadIdx.x & 31)];
    for (int it = 0; it < iters; it++) {
        L4 v[16];
        #pragma unroll
        for (int a = 0; a < 16; a++) { v[a] = mul_to_limbs(x[a], tw); add_bias(v[a]); }
        #pragma unroll
        for (int c = 0; c < 16; c++) { x[c] = gld::mmul(to_weak(v[c]), tw); tw = gld::mmul(tw, B); }
    }
    for (int a = 0; a < 16; a++) data[base + a] = x[a];
}

template <typename K>
static void run_net(const char* name, K kern, uint64_t* d_data, const uint64_t* d_wt, int blocks) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 64;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_data, d_wt, iters);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_data, d_wt, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    const double wave_elems = (double)blocks * 4 * 16 * iters;     // element-steps per lane, summed over waves
    const double cyc = best * 1e-3 * 2.4e9 * 1024.0 / wave_elems;
    printf("NET %-14s blocks=%5d  %8.3f ms  => %6.1f cycles per element (network + twiddle) per SIMD @2.4GHz\n", name, blocks, best, cyc);
    fflush(stdout);
}

// ------------------------------------------------------------------ (B)
__device__ __forceinline__ uint64_t spin_work(uint64_t x, int spin) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    for (int i = 0; i < spin; i++) {
        asm volatile("v_mad_u32_u24 %0, %0, %1, %1\n v_mad_u32_u24 %1, %1, %0, %0" : "+v"(lo), "+v"(hi));
    }
    return ((uint64_t)hi << 32) | lo;
}
template <int QN, int NT> struct Map {
    static constexpr int NW = NT / 64;
    // step-1 (load) mapping: returns row for register (h, a)
    __device__ static void lane(unsigned tid, unsigned& q, unsigned& c0, unsigned& b0) {
        const unsigned l = tid & 63, w = tid >> 6;
        if (QN == 8) { q = l & 7; c0 = l >> 3; b0 = w; } else { q = l & 3; c0 = l >> 2; b0 = w; }
    }
    __device__ static unsigned row(unsigned c0, unsigned b0, int h, int a) {
        if (QN == 8) return 256u * a + 16u * b0 + c0 + 8u * h;
        return 256u * a + 16u * (b0 + 8u * h) + c0;
    }
    // output index k (12 bits) for register (h, c') with the store mapping
    __device__ static unsigned kout(unsigned tid, int h, int cp) {
        const unsigned l = tid & 63, w = tid >> 6;
        if (QN == 8) return (l >> 3) + 8u * h + 16u * w + 256u * cp;
        return (l >> 2) + 16u * (w + 8u * h) + 256u * cp;
    }
};
__device__ __forceinline__ unsigned swz_tile(unsigned bid, unsigned ntiles) { return (bid % 8) * (ntiles / 8) + bid / 8; }

// pass 1: x[row * 4096 + QN*J + q]  ->  S[(k1 / QN)][j2][k1 % QN]
template <int QN, int NT>
__global__ void __launch_bounds__(NT) k_p1(const uint64_t* __restrict__ x, uint64_t* __restrict__ S, int spin) {
    const unsigned J = swz_tile(blockIdx.x, 4096 / QN), tid = threadIdx.x;
    unsigned q, c0, b0; Map<QN, NT>::lane(tid, q, c0, b0);
    uint64_t v[2][16];
    #pragma unroll
    for (int h = 0; h < 2; h++)
        #pragma unroll
        for (int a = 0; a < 16; a++) v[h][a] = x[(size_t)Map<QN, NT>::row(c0, b0, h, a) * 4096 + QN * J + q];
    #pragma unroll
    for (int h = 0; h < 2; h++)
        #pragma unroll
        for (int a = 0; a < 16; a++) v[h][a] = spin_work(v[h][a], spin);
    #pragma unroll
    for (int h = 0; h < 2; h++)
        #pragma unroll
        for (int cp = 0; cp < 16; cp++) {
            const unsigned k1 = Map<QN, NT>::kout(tid, h, cp);
            S[((size_t)(k1 / QN) * 4096 + QN * J + q) * QN + (k1 % QN)] = v[h][cp];
        }
}
// pass 2: S[K][j2][kappa] -> y[(QN*K + kappa) + 4096 * k2]
template <int QN, int NT>
__global__ void __launch_bounds__(NT) k_p2(const uint64_t* __restrict__ S, uint64_t* __restrict__ y, int spin) {
    const unsigned K = swz_tile(blockIdx.x, 4096 / QN), tid = threadIdx.x;
    unsigned kap, c0, b0; Map<QN, NT>::lane(tid, kap, c0, b0);
    uint64_t v[2][16];
    #pragma unroll
    for (int h = 0; h < 2; h++)
        #pragma unroll
        for (int a = 0; a < 16; a++) v[h][a] = S[((size_t)K * 4096 + Map<QN, NT>::row(c0, b0, h, a)) * QN + kap];
    #pragma unroll
    for (int h = 0; h < 2; h++)
        #pragma unroll
        for (int a = 0; a < 16; a++) v[h][a] = spin_work(v[h][a], spin);
    #pragma unroll
    for (int h = 0; h < 2; h++)
        #pragma unroll
        for (int cp = 0; cp < 16; cp++) {
            const unsigned k2 = Map<QN, NT>::kout(tid, h, cp);
            y[(size_t)k2 * 4096 + QN * K + kap] = v[h][cp];
        }
}

template <int QN, int NT>
static void run_pattern(uint64_t* const* cols, int ncols, uint64_t* const* scr, int nscr, int spin, int mode) {
    // mode 0: pass 1 only; 1: pass 2 only; 2: column by column p1,p2 (one scratch); 3: batched (p1 all columns, p2 all columns)
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid(4096 / QN), blk(NT);
    std::vector<float> t;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0));
        if (mode == 0) for (int c = 0; c < ncols; c++) hipLaunchKernelGGL((k_p1<QN, NT>), grid, blk, 0, 0, cols[c], scr[c % nscr], spin);
        if (mode == 1) for (int c = 0; c < ncols; c++) hipLaunchKernelGGL((k_p2<QN, NT>), grid, blk, 0, 0, scr[c % nscr], cols[c], spin);
        if (mode == 2) for (int c = 0; c < ncols; c++) {
            hipLaunchKernelGGL((k_p1<QN, NT>), grid, blk, 0, 0, cols[c], scr[0], spin);
            hipLaunchKernelGGL((k_p2<QN, NT>), grid, blk, 0, 0, scr[0], cols[c], spin);
        }
        if (mode == 3) {
            for (int c = 0; c < ncols; c++) hipLaunchKernelGGL((k_p1<QN, NT>), grid, blk, 0, 0, cols[c], scr[c % nscr], spin);
            for (int c = 0; c < ncols; c++) hipLaunchKernelGGL((k_p2<QN, NT>), grid, blk, 0, 0, scr[c % nscr], cols[c], spin);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    const float ms = t[t.size() / 2];
    const double per_col_us = ms * 1e3 / ncols;
    const double passes = (mode >= 2) ? 2.0 : 1.0;
    static const char* const names[4] = {"pass1 only", "pass2 only", "p1,p2 per column (1 scratch)", "p1 x8 then p2 x8"};
    printf("PATTERN cols=%d threads=%4d spin=%3d %-30s %8.1f us/column  %7.1f GB/s (r+w per pass)\n", QN, NT, spin, names[mode],
           per_col_us, passes * 2.0 * 134217728.0 / per_col_us / 1e3);
    fflush(stdout);
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d\n", prop.name, prop.multiProcessorCount);
    const bool only_a = argc > 1 && argv[1][0] == 'a', only_b = argc > 1 && argv[1][0] == 'b';

    if (!only_b) {
        const int blocks = 256 * 8;
        uint64_t *d_data, *d_wt;
        CK(hipMalloc(&d_data, (size_t)blocks * 256 * 16 * 8)); CK(hipMalloc(&d_wt, 8192 * 8));
        std::vector<uint64_t> h((size_t)blocks * 256 * 16), w(8192);
        uint64_t s = 0x9E3779B97F4A7C15ull;
        for (auto& v : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = s % gl::P; }
        for (auto& v : w) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = s % gl::P; }
        CK(hipMemcpy(d_data, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_wt, w.data(), w.size() * 8, hipMemcpyHostToDevice));
        for (int b : {256 * 8, 256 * 4}) {
            run_net("old 64-bit", k_net_old, d_data, d_wt, b);
            run_net("limb uniform", k_net_uniform, d_data, d_wt, b);
            run_net("limb lds-table", k_net_lds, d_data, d_wt, b);
            run_net("boundary", k_boundary, d_data, d_wt, b);
        }
        CK(hipFree(d_data)); CK(hipFree(d_wt));
    }
    if (!only_a) {
        const int NC = 8;
        uint64_t* cols[NC]; uint64_t* scr[NC];
        for (int c = 0; c < NC; c++) { CK(hipMalloc(&cols[c], 134217728)); CK(hipMemset(cols[c], c + 1, 134217728)); }
        for (int c = 0; c < NC; c++) { CK(hipMalloc(&scr[c], 134217728)); CK(hipMemset(scr[c], 0, 134217728)); }
        for (int spin : {0, 30, 60}) {
            for (int mode = 0; mode < 4; mode++) {
                run_pattern<8, 1024>(cols, NC, scr, NC, spin, mode);
                run_pattern<4, 512>(cols, NC, scr, NC, spin, mode);
            }
        }
    }
    return 0;
}
