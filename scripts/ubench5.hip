// Round-2 micro-benchmarks, part 2:
//   (A) isolates the pieces of the limb-form network step (convert, network, multiply-accumulate, fold)
//   (B) strided-tile reads / writes on cold 128 MiB columns: ROWS x QN-word tiles at row stride n/ROWS,
//       separately for the read side and the write side, against a contiguous sweep.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ministark_amd/csrc scripts/ubench5.hip -o scripts/ubench5
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
#define NETK(name, WAVES, ...) \
__global__ void __launch_bounds__(256, WAVES) name(uint64_t* data, const uint64_t* __restrict__ wt, int iters) { \
    uint64_t x[16]; \
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; \
    for (int a = 0; a < 16; a++) x[a] = data[base + a]; \
    for (int it = 0; it < iters; it++) { __VA_ARGS__ } \
    for (int a = 0; a < 16; a++) data[base + a] = x[a]; \
}
// network only: convert, network, recombine cheaply (xor of limbs) so nothing is optimised away
NETK(k_conv_net, 4,
    L4 v[16];
    _Pragma("unroll") for (int a = 0; a < 16; a++) v[a] = from_u64(x[a]);
    dft<16, false>(v);
    _Pragma("unroll") for (int c = 0; c < 16; c++) x[c] = ((uint64_t)(v[c].l[0] ^ v[c].l[2]) << 32) | (v[c].l[1] ^ v[c].l[3]);
)
// convert only
NETK(k_conv, 4,
    L4 v[16];
    _Pragma("unroll") for (int a = 0; a < 16; a++) v[a] = from_u64(x[a]);
    _Pragma("unroll") for (int c = 0; c < 16; c++) x[c] = ((uint64_t)(v[c].l[0] ^ v[c].l[2]) << 32) | (v[c].l[1] + v[(c + 1) & 15].l[0]);
)
// multiply-accumulate + fold only (limbs taken from the words of x)
NETK(k_mulfold, 4,
    _Pragma("unroll") for (int c = 0; c < 16; c++) {
        L4 v; v.l[0] = (uint32_t)x[c] >> 3; v.l[1] = (uint32_t)(x[c] >> 32) >> 3; v.l[2] = v.l[0] ^ 0x155555; v.l[3] = v.l[1] ^ 0x0aaaaa;
        const uint64_t* wp = wt + (size_t)(it & 7) * 64 + c * 4;
        x[c] = mul_fold(v, w4_from(wp[0], wp[1], wp[2], wp[3]));
    }
)
// multiply-accumulate without the fold
NETK(k_mulonly, 4,
    _Pragma("unroll") for (int c = 0; c < 16; c++) {
        L4 v; v.l[0] = (uint32_t)x[c] >> 3; v.l[1] = (uint32_t)(x[c] >> 32) >> 3; v.l[2] = v.l[0] ^ 0x155555; v.l[3] = v.l[1] ^ 0x0aaaaa;
        const uint64_t* wp = wt + (size_t)(it & 7) * 64 + c * 4;
        W4 w = w4_from(wp[0], wp[1], wp[2], wp[3]);
        uint64_t alo = (uint64_t)v.l[0] * w.lo[0], ahi = (uint64_t)v.l[0] * w.hi[0];
        _Pragma("unroll") for (int i = 1; i < 4; i++) { alo += (uint64_t)v.l[i] * w.lo[i]; ahi += (uint64_t)v.l[i] * w.hi[i]; }
        x[c] = alo ^ ahi;
    }
)
NETK(k_full4, 4,
    L4 v[16];
    _Pragma("unroll") for (int a = 0; a < 16; a++) v[a] = from_u64(x[a]);
    dft<16, false>(v);
    _Pragma("unroll") for (int c = 0; c < 16; c++) {
        const uint64_t* wp = wt + (size_t)(it & 7) * 64 + c * 4;
        x[c] = mul_fold(v[c], w4_from(wp[0], wp[1], wp[2], wp[3]));
    }
)
NETK(k_full2, 2,
    L4 v[16];
    _Pragma("unroll") for (int a = 0; a < 16; a++) v[a] = from_u64(x[a]);
    dft<16, false>(v);
    _Pragma("unroll") for (int c = 0; c < 16; c++) {
        const uint64_t* wp = wt + (size_t)(it & 7) * 64 + c * 4;
        x[c] = mul_fold(v[c], w4_from(wp[0], wp[1], wp[2], wp[3]));
    }
)
NETK(k_old, 4,
    gld::dft_lazy<16, false>(x);
    _Pragma("unroll") for (int c = 0; c < 16; c++) x[c] = gld::mmul(x[c], wt[(it & 7) * 64 + c * 4]);
)
NETK(k_old_net, 4,
    gld::dft_lazy<16, false>(x);
    _Pragma("unroll") for (int c = 0; c < 16; c++) x[c] = gld::canon(x[c]);
)
NETK(k_old_mul, 4,
    _Pragma("unroll") for (int c = 0; c < 16; c++) x[c] = gld::mmul(x[c], wt[(it & 7) * 64 + c * 4]);
)
// plain VOP2 mix: 4 adds + 4 subs on 8 accumulators per "element" x 16, to see the rate of compiler-scheduled fast ops
NETK(k_addsub, 4,
    uint32_t l[32];
    _Pragma("unroll") for (int a = 0; a < 16; a++) { l[2 * a] = (uint32_t)x[a]; l[2 * a + 1] = (uint32_t)(x[a] >> 32); }
    _Pragma("unroll") for (int r = 0; r < 8; r++) {
        _Pragma("unroll") for (int a = 0; a < 16; a++) { const uint32_t u = l[a], v = l[a + 16]; l[a] = u + v; l[a + 16] = u - v; }
        _Pragma("unroll") for (int a = 0; a < 32; a += 2) { const uint32_t u = l[a], v = l[a + 1]; l[a] = u + v; l[a + 1] = u - v; }
    }
    _Pragma("unroll") for (int a = 0; a < 16; a++) x[a] = ((uint64_t)l[2 * a + 1] << 32) | l[2 * a];
)

template <typename K>
static void run_net(const char* name, K kern, uint64_t* d_data, const uint64_t* d_wt, int blocks, double scale = 1.0) {
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
    const double wave_elems = (double)blocks * 4 * 16 * iters;
    const double cyc = best * 1e-3 * 2.4e9 * 1024.0 / wave_elems;
    printf("PIECE %-22s blocks=%5d  %8.3f ms  => %6.1f cycles per element per SIMD @2.4GHz%s\n", name, blocks, best, cyc * scale, scale != 1.0 ? " (per 4 add/sub pairs)" : "");
    fflush(stdout);
}

// ------------------------------------------------------------------ (B)
// tile = ROWS rows x QN words at row stride 2^24/ROWS words; 32 words per lane; per instruction a wave
// touches 64/QN consecutive rows x QN words.
template <int QN, int ROWS, int NT, bool RD, bool WR, bool RSTRIDED, bool WSTRIDED>
__global__ void __launch_bounds__(NT) k_tile(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst) {
    constexpr unsigned NTILES = (1u << 24) / (ROWS * QN), RPI = 64 / QN, NW = NT / 64, STRIDE = (1u << 24) / ROWS;
    const unsigned T = (blockIdx.x % 8) * (NTILES / 8) + blockIdx.x / 8, tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const unsigned q = l % QN, r0 = l / QN;
    uint64_t v[32];
    #pragma unroll
    for (int i = 0; i < 32; i++) {
        const unsigned row = r0 + RPI * (w + NW * i);
        const size_t strided = (size_t)row * STRIDE + (size_t)T * QN + q;
        const size_t contig = (size_t)T * (ROWS * QN) + (size_t)(w + NW * i) * 64 + l;
        v[i] = RD ? src[RSTRIDED ? strided : contig] : (uint64_t)(tid + i);
    }
    if (WR) {
        #pragma unroll
        for (int i = 0; i < 32; i++) {
            const unsigned row = r0 + RPI * (w + NW * i);
            const size_t strided = (size_t)row * STRIDE + (size_t)T * QN + q;
            const size_t contig = (size_t)T * (ROWS * QN) + (size_t)(w + NW * i) * 64 + l;
            dst[WSTRIDED ? strided : contig] = v[i] + 1;
        }
    } else {
        uint64_t s = 0;
        #pragma unroll
        for (int i = 0; i < 32; i++) s ^= v[i];
        if (s == 0x123456789abcdefull) dst[tid] = s;
    }
}
template <int QN, int ROWS, int NT, bool RD, bool WR, bool RS, bool WS>
static void run_tile(const char* what, uint64_t* const* cols, uint64_t* const* scr, int ncols, bool hot) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid((1u << 24) / (ROWS * QN)), blk(NT);
    std::vector<float> t;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0));
        for (int c = 0; c < ncols; c++) hipLaunchKernelGGL((k_tile<QN, ROWS, NT, RD, WR, RS, WS>), grid, blk, 0, 0, cols[hot ? 0 : c], scr[hot ? 0 : c]);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    const double us = t[2] * 1e3 / ncols;
    const double bytes = 134217728.0 * ((RD ? 1 : 0) + (WR ? 1 : 0));
    printf("TILE %4d rows x %2d words (%3d B) threads=%4d %-28s %s %7.1f us/column  %7.1f GB/s\n", ROWS, QN, QN * 8, NT, what, hot ? "hot " : "cold", us, bytes / us / 1e3);
    fflush(stdout);
}
template <int QN, int ROWS, int NT>
static void run_shape(uint64_t* const* cols, uint64_t* const* scr, int ncols) {
    run_tile<QN, ROWS, NT, true, false, true, false>("read strided", cols, scr, ncols, false);
    run_tile<QN, ROWS, NT, false, true, false, true>("write strided", cols, scr, ncols, false);
    run_tile<QN, ROWS, NT, true, true, true, false>("read strided, write contig", cols, scr, ncols, false);
    run_tile<QN, ROWS, NT, true, true, false, true>("read contig, write strided", cols, scr, ncols, false);
    run_tile<QN, ROWS, NT, true, true, true, true>("read + write strided", cols, scr, ncols, false);
    run_tile<QN, ROWS, NT, true, true, true, true>("read + write strided", cols, scr, ncols, true);
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
        for (int b : {256 * 8, 256 * 2}) {
            run_net("old net+mmul", k_old, d_data, d_wt, b);
            run_net("old net (+canon)", k_old_net, d_data, d_wt, b);
            run_net("old mmul", k_old_mul, d_data, d_wt, b);
            run_net("limb full (4 w/SIMD)", k_full4, d_data, d_wt, b);
            run_net("limb full (2 w/SIMD)", k_full2, d_data, d_wt, b);
            run_net("limb convert+net", k_conv_net, d_data, d_wt, b);
            run_net("limb convert", k_conv, d_data, d_wt, b);
            run_net("limb mul+fold", k_mulfold, d_data, d_wt, b);
            run_net("limb mul only", k_mulonly, d_data, d_wt, b);
            run_net("add/sub mix", k_addsub, d_data, d_wt, b, 1.0 / 8.0);
        }
        CK(hipFree(d_data)); CK(hipFree(d_wt));
    }
    if (!only_a) {
        const int NC = 8;
        uint64_t* cols[NC]; uint64_t* scr[NC];
        for (int c = 0; c < NC; c++) { CK(hipMalloc(&cols[c], 134217728)); CK(hipMemset(cols[c], c + 1, 134217728)); }
        for (int c = 0; c < NC; c++) { CK(hipMalloc(&scr[c], 134217728)); CK(hipMemset(scr[c], 0, 134217728)); }
        run_tile<64, 512, 1024, true, true, false, false>("contiguous copy", cols, scr, NC, false);
        run_tile<64, 512, 1024, true, true, false, false>("contiguous copy", cols, scr, NC, true);
        run_shape<4, 8192, 1024>(cols, scr, NC);
        run_shape<8, 4096, 1024>(cols, scr, NC);
        run_shape<16, 2048, 1024>(cols, scr, NC);
        run_shape<32, 1024, 1024>(cols, scr, NC);
        run_shape<64, 512, 1024>(cols, scr, NC);
        run_shape<8, 2048, 512>(cols, scr, NC);
        run_shape<16, 1024, 512>(cols, scr, NC);
        run_shape<16, 512, 256>(cols, scr, NC);
        run_shape<64, 256, 512>(cols, scr, NC);
    }
    return 0;
}
