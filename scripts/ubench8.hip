// Round-3 micro-benchmark:
//   (A) the fold of a limb-form product through the carry-out of v_mad_u64_u32 (glimb::fold_co) and the three-copy product
//       on loads (glimb::mul3_to_limbs): checked against the host field formula, timed against the round-2 forms;
//   (B) the three radix-256 passes as bare access patterns IN PLACE on one 128 MiB column (footprint < the 256 MiB
//       Infinity Cache): per-column chains over 8 columns, and the same column again and again (the shape of the
//       reference's criterion harness, gpu/benches/fft.rs:36-43).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ministark_amd/csrc scripts/ubench8.hip -o scripts/ubench8
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

// ------------------------------------------------------------------ (A) correctness
__global__ void k_check_fold(const uint32_t* __restrict__ limbs, const uint64_t* __restrict__ w4, uint64_t* out_old, uint64_t* out_new, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    L4 v; for (int k = 0; k < 4; k++) v.l[k] = limbs[4 * i + k];
    const W4 w = w4_from(w4[4 * i], w4[4 * i + 1], w4[4 * i + 2], w4[4 * i + 3]);
    out_old[i] = mul_fold<false>(v, w);
    out_new[i] = mul_fold_co<false>(v, w);
}
__global__ void k_check_canon(const uint32_t* __restrict__ limbs, const uint64_t* __restrict__ w4, uint64_t* out_new, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    L4 v; for (int k = 0; k < 4; k++) v.l[k] = limbs[4 * i + k];
    const W4 w = w4_from(w4[4 * i], w4[4 * i + 1], w4[4 * i + 2], w4[4 * i + 3]);
    out_new[i] = mul_fold_co<true>(v, w);
}
__global__ void k_check_mul3(const uint64_t* __restrict__ x, const uint64_t* __restrict__ q3, uint64_t* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Q3 q = q3_from(q3[3 * i], q3[3 * i + 1], q3[3 * i + 2]);
    const L4 v = mul3_to_limbs(x[i], q);
    L4 b = v; add_bias(b);
    out[i] = to_canon(b);
}

static uint64_t sm_state = 0x6d696e69;
static uint64_t splitmix() { uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

static int check() {
    const int n = 1 << 20;
    std::vector<uint32_t> limbs(4 * n);
    std::vector<uint64_t> w(n), w4(4 * n), x(n), q(n), q3(3 * n);
    const uint64_t sh[4] = {1, (uint64_t)1 << 24, (uint64_t)1 << 48, gl::pow(2, 72)};
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 4; k++) {
            uint32_t l = (uint32_t)splitmix() & 0x3FFFFFFFu;            // < 2^30
            if ((i & 15) == 0) l = 0x3FFFFFFFu;                          // the extreme
            if ((i & 15) == 1) l = 0;
            limbs[4 * i + k] = l;
        }
        uint64_t ww = splitmix(); if (ww >= gl::P) ww -= gl::P;
        if ((i & 31) == 2) ww = gl::P - 1;
        if ((i & 31) == 3) ww = 0xFFFFFFFFull;
        w[i] = ww;
        for (int k = 0; k < 4; k++) w4[4 * i + k] = gl::mul(ww, sh[k]);
        if ((i & 7) == 5) for (int k = 0; k < 4; k++) w4[4 * i + k] += ((w4[4 * i + k] < 0xFFFFFFFEull) ? gl::P : 0);   // non-canonical copies (any representative < 2^64)
        x[i] = splitmix(); if ((i & 63) == 7) x[i] = ~0ull; if ((i & 63) == 8) x[i] = gl::P;
        uint64_t qq = splitmix(); if (qq >= gl::P) qq -= gl::P;
        q[i] = qq;
        for (int k = 0; k < 3; k++) q3[3 * i + k] = gl::mul(qq, sh[k]);
    }
    uint32_t* d_l; uint64_t *d_w4, *d_o, *d_n, *d_x, *d_q3;
    CK(hipMalloc(&d_l, 16 * n)); CK(hipMalloc(&d_w4, 32 * n)); CK(hipMalloc(&d_o, 8 * n)); CK(hipMalloc(&d_n, 8 * n)); CK(hipMalloc(&d_x, 8 * n)); CK(hipMalloc(&d_q3, 24 * n));
    CK(hipMemcpy(d_l, limbs.data(), 16 * n, hipMemcpyHostToDevice)); CK(hipMemcpy(d_w4, w4.data(), 32 * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_x, x.data(), 8 * n, hipMemcpyHostToDevice)); CK(hipMemcpy(d_q3, q3.data(), 24 * n, hipMemcpyHostToDevice));
    std::vector<uint64_t> o(n), nn(n), cn(n), m3(n);
    hipLaunchKernelGGL(k_check_fold, dim3(n / 256), dim3(256), 0, 0, d_l, d_w4, d_o, d_n, n);
    CK(hipMemcpy(o.data(), d_o, 8 * n, hipMemcpyDeviceToHost)); CK(hipMemcpy(nn.data(), d_n, 8 * n, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k_check_canon, dim3(n / 256), dim3(256), 0, 0, d_l, d_w4, d_n, n);
    CK(hipMemcpy(cn.data(), d_n, 8 * n, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k_check_mul3, dim3(n / 256), dim3(256), 0, 0, d_x, d_q3, d_o, n);
    CK(hipMemcpy(m3.data(), d_o, 8 * n, hipMemcpyDeviceToHost));
    long bad_old = 0, bad_new = 0, bad_canon = 0, bad_m3 = 0;
    for (int i = 0; i < n; i++) {
        uint64_t val = 0;
        for (int k = 0; k < 4; k++) val = gl::add(val, gl::mul(limbs[4 * i + k] % gl::P, sh[k]));
        const uint64_t want = gl::mul(val, w[i]);
        if (gl::canon(o[i]) != want) bad_old++;
        if (gl::canon(nn[i]) != want) bad_new++;
        if (cn[i] != want) bad_canon++;
        if (m3[i] != gl::mul(gl::canon(x[i] >= gl::P ? x[i] - gl::P : x[i]), q[i])) bad_m3++;
    }
    printf("CHECK mul_fold (round 2)  mismatches %ld / %d\n", bad_old, n);
    printf("CHECK mul_fold_co         mismatches %ld / %d\n", bad_new, n);
    printf("CHECK mul_fold_co<CANON>  mismatches %ld / %d\n", bad_canon, n);
    printf("CHECK mul3_to_limbs       mismatches %ld / %d\n", bad_m3, n);
    fflush(stdout);
    return (bad_new || bad_canon || bad_m3) ? 1 : 0;
}

// ------------------------------------------------------------------ (A) timing
#define NETK(name, WAVES, ...) \
__global__ void __launch_bounds__(256, WAVES) name(uint64_t* data, const uint64_t* __restrict__ wt, int iters) { \
    uint64_t x[16]; \
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; \
    for (int a = 0; a < 16; a++) x[a] = data[base + a]; \
    for (int it = 0; it < iters; it++) { __VA_ARGS__ } \
    for (int a = 0; a < 16; a++) data[base + a] = x[a]; \
}
NETK(k_full_old, 4,
    L4 v[16];
    _Pragma("unroll") for (int a = 0; a < 16; a++) v[a] = from_u64(x[a]);
    dft<16, false>(v);
    _Pragma("unroll") for (int c = 0; c < 16; c++) {
        const uint64_t* wp = wt + (size_t)(it & 7) * 64 + c * 4;
        x[c] = mul_fold(v[c], w4_from(wp[0], wp[1], wp[2], wp[3]));
    }
)
NETK(k_full_new, 4,
    L4 v[16];
    _Pragma("unroll") for (int a = 0; a < 16; a++) v[a] = from_u64(x[a]);
    dft<16, false>(v);
    _Pragma("unroll") for (int c = 0; c < 16; c++) {
        const uint64_t* wp = wt + (size_t)(it & 7) * 64 + c * 4;
        x[c] = mul_fold_co(v[c], w4_from(wp[0], wp[1], wp[2], wp[3]));
    }
)
NETK(k_mulfold_old, 4,
    _Pragma("unroll") for (int c = 0; c < 16; c++) {
        L4 v; v.l[0] = (uint32_t)x[c] >> 3; v.l[1] = (uint32_t)(x[c] >> 32) >> 3; v.l[2] = v.l[0] ^ 0x155555; v.l[3] = v.l[1] ^ 0x0aaaaa;
        const uint64_t* wp = wt + (size_t)(it & 7) * 64 + c * 4;
        x[c] = mul_fold(v, w4_from(wp[0], wp[1], wp[2], wp[3]));
    }
)
NETK(k_mulfold_new, 4,
    _Pragma("unroll") for (int c = 0; c < 16; c++) {
        L4 v; v.l[0] = (uint32_t)x[c] >> 3; v.l[1] = (uint32_t)(x[c] >> 32) >> 3; v.l[2] = v.l[0] ^ 0x155555; v.l[3] = v.l[1] ^ 0x0aaaaa;
        const uint64_t* wp = wt + (size_t)(it & 7) * 64 + c * 4;
        x[c] = mul_fold_co(v, w4_from(wp[0], wp[1], wp[2], wp[3]));
    }
)
NETK(k_in_old, 4,           // per-lane factor on the loads, round 2: 128-bit product cut into limbs
    L4 v[16];
    const uint64_t q = wt[threadIdx.x + (it & 7) * 256];
    _Pragma("unroll") for (int a = 0; a < 16; a++) v[a] = mul_to_limbs(x[a], q);
    dft<16, false>(v);
    _Pragma("unroll") for (int c = 0; c < 16; c++) x[c] = ((uint64_t)(v[c].l[0] ^ v[c].l[2]) << 32) | (v[c].l[1] ^ v[c].l[3]);
)
NETK(k_in_new, 4,           // ... three pre-shifted copies of the factor
    L4 v[16];
    const uint64_t* qp = wt + 3 * (threadIdx.x + (it & 7) * 256);
    const Q3 q = q3_from(qp[0], qp[1], qp[2]);
    _Pragma("unroll") for (int a = 0; a < 16; a++) v[a] = mul3_to_limbs(x[a], q);
    dft<16, false>(v);
    _Pragma("unroll") for (int c = 0; c < 16; c++) x[c] = ((uint64_t)(v[c].l[0] ^ v[c].l[2]) << 32) | (v[c].l[1] ^ v[c].l[3]);
)
NETK(k_in_plain, 4,
    L4 v[16];
    _Pragma("unroll") for (int a = 0; a < 16; a++) v[a] = from_u64(x[a]);
    dft<16, false>(v);
    _Pragma("unroll") for (int c = 0; c < 16; c++) x[c] = ((uint64_t)(v[c].l[0] ^ v[c].l[2]) << 32) | (v[c].l[1] ^ v[c].l[3]);
)

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
    const double wave_elems = (double)blocks * 4 * 16 * iters;
    printf("PIECE %-34s blocks=%5d  %8.3f ms  => %6.1f cycles per element per SIMD @2.4GHz\n", name, blocks, best, best * 1e-3 * 2.4e9 * 1024.0 / wave_elems);
    fflush(stdout);
}

// ------------------------------------------------------------------ (B) in-place pass patterns
static constexpr int NCOL = 8;
// PASS 1 / 3 pattern: 256 rows at stride 2^16 words, 64 words at 64 T (in place);  PASS 2: 256 rows at stride 256 words inside block U.
template <int PASS>
__global__ void __launch_bounds__(512, 4) k_pass(uint64_t* __restrict__ col) {
    const unsigned T = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    size_t rbase, rstride;
    if (PASS == 2) { const unsigned U = T >> 2, q = T & 3; rbase = (size_t)U * 65536 + 64 * q; rstride = 256; }
    else { rbase = (size_t)T * 64; rstride = 65536; }
    uint64_t v[32];
    #pragma unroll
    for (int i = 0; i < 32; i++) v[i] = col[rbase + (size_t)(w + 8 * i) * rstride + lane];
    #pragma unroll
    for (int i = 0; i < 32; i++) v[i] += 1;
    #pragma unroll
    for (int i = 0; i < 32; i++) col[rbase + (size_t)(w + 8 * i) * rstride + lane] = v[i];
}
template <typename F>
static void timeit(const char* name, int per, F body) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t;
    body(); CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0, 0));
        body();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    const double us = t[2] * 1e3 / per;
    printf("INPLACE %-72s %7.1f us/column  (%6.1f GB/s algorithmic)\n", name, us, 268435456.0 / us / 1e3);
    fflush(stdout);
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d\n", prop.name, prop.multiProcessorCount);
    int rc = check();
    {
        const int blocks = 2048;
        uint64_t *d_data, *d_wt;
        CK(hipMalloc(&d_data, (size_t)blocks * 256 * 16 * 8)); CK(hipMalloc(&d_wt, 3 * 8 * 256 * 8 + 4096));
        std::vector<uint64_t> h((size_t)blocks * 256 * 16), hw(3 * 8 * 256 + 512);
        for (auto& v : h) v = splitmix();
        for (auto& v : hw) { v = splitmix(); if (v >= gl::P) v -= gl::P; }
        CK(hipMemcpy(d_data, h.data(), h.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_wt, hw.data(), hw.size() * 8, hipMemcpyHostToDevice));
        run_net("level, round-2 fold", k_full_old, d_data, d_wt, blocks);
        run_net("level, carry-out fold", k_full_new, d_data, d_wt, blocks);
        run_net("mul + fold, round 2", k_mulfold_old, d_data, d_wt, blocks);
        run_net("mul + fold, carry-out", k_mulfold_new, d_data, d_wt, blocks);
        run_net("convert + network", k_in_plain, d_data, d_wt, blocks);
        run_net("128-bit product on loads + network", k_in_old, d_data, d_wt, blocks);
        run_net("three-copy product on loads + network", k_in_new, d_data, d_wt, blocks);
        CK(hipFree(d_data)); CK(hipFree(d_wt));
    }
    if (argc > 1 && atoi(argv[1]) == 0) return rc;
    uint64_t* col[NCOL];
    const size_t bytes = (size_t)8 << 24;
    for (int c = 0; c < NCOL; c++) { CK(hipMalloc(&col[c], bytes)); CK(hipMemset(col[c], c + 1, bytes)); }
    auto P = [&](int pass, uint64_t* c) {
        if (pass == 2) hipLaunchKernelGGL(k_pass<2>, dim3(1024), dim3(512), 0, 0, c);
        else hipLaunchKernelGGL(k_pass<1>, dim3(1024), dim3(512), 0, 0, c);
    };
    timeit("3 in-place passes, chain per column over 8 columns (cold first pass)", NCOL, [&] { for (int c = 0; c < NCOL; c++) { P(1, col[c]); P(2, col[c]); P(3, col[c]); } });
    timeit("3 in-place passes, batch order over 8 columns", NCOL, [&] { for (int p = 1; p <= 3; p++) for (int c = 0; c < NCOL; c++) P(p, col[c]); });
    timeit("3 in-place passes, the same column 8 times (reference harness shape)", NCOL, [&] { for (int c = 0; c < NCOL; c++) { P(1, col[0]); P(2, col[0]); P(3, col[0]); } });
    timeit("3 in-place passes, chain per column over 2 columns", NCOL, [&] { for (int c = 0; c < NCOL; c++) { P(1, col[c & 1]); P(2, col[c & 1]); P(3, col[c & 1]); } });
    timeit("strided pass alone, 8 columns (cold)", NCOL, [&] { for (int c = 0; c < NCOL; c++) P(1, col[c]); });
    timeit("strided pass alone, same column (hot)", NCOL, [&] { for (int c = 0; c < NCOL; c++) P(1, col[0]); });
    timeit("local pass alone, 8 columns (cold)", NCOL, [&] { for (int c = 0; c < NCOL; c++) P(2, col[c]); });
    timeit("local pass alone, same column (hot)", NCOL, [&] { for (int c = 0; c < NCOL; c++) P(2, col[0]); });
    return rc;
}
