// Round 4: the three kernels of the 2^24 forward coset transform (and the transform in batch order) under compile-time variants of
// ntt2_kernels.h: -DVAR_NTL (non-temporal tile loads), -DVAR_NTS (non-temporal tile stores), -DVAR_HOT (every workgroup on tiles 0..3:
// arithmetic + exchange + issue alone), -DVAR_... added as they are tried.  Tables hold arbitrary residues (timing only).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ministark_amd/csrc [-DVAR_x] scripts/ntt_pass_bench2.hip -o scripts/ntt_pass_bench2_x
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#if defined(__HIP_DEVICE_COMPILE__)
template <class T> __device__ __forceinline__ T var_nt_ld(const T* p) { return __builtin_nontemporal_load(p); }
template <class T> __device__ __forceinline__ void var_nt_st(T* p, T v) {
    if constexpr (sizeof(T) == 16) {
        typedef unsigned long long v2 __attribute__((ext_vector_type(2)));
        v2 t; __builtin_memcpy(&t, &v, 16); __builtin_nontemporal_store(t, (v2*)p);
    } else __builtin_nontemporal_store(v, p);
}
#else
template <class T> __host__ __device__ T var_nt_ld(const T* p) { return *p; }
template <class T> __host__ __device__ void var_nt_st(T* p, T v) { *p = v; }
#endif
// VAR_NTL / VAR_NTS: bit masks of the passes (1, 2, 4 = pass 1, 2, 3) whose tile loads / stores are non-temporal
#ifdef VAR_NTL
#define NTT2_LD(p, pass) (((VAR_NTL >> ((pass) - 1)) & 1) ? var_nt_ld(p) : *(p))
#endif
#ifdef VAR_NTS
#define NTT2_ST(p, v, pass) do { if ((VAR_NTS >> ((pass) - 1)) & 1) var_nt_st(p, v); else *(p) = (v); } while (0)
#endif
#ifdef VAR_HOT
#define NTT2_BX (blockIdx.x & 3)
#endif
#include "ntt_kernels.h"
#include "ntt2_kernels.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

static uint64_t* dev_table(size_t words, uint64_t seed) {
    std::vector<uint64_t> h(words);
    uint64_t s = seed;
    for (auto& v : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = s % gl::P; if (!v) v = 1; }
    uint64_t* d; CK(hipMalloc(&d, words * 8)); CK(hipMemcpy(d, h.data(), words * 8, hipMemcpyHostToDevice));
    return d;
}
template <class F>
static double time_us(F launch, int reps = 9) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t;
    for (int i = 0; i < 3; i++) launch();
    CK(hipDeviceSynchronize());
    for (int i = 0; i < reps; i++) {
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2] * 1e3;
}
int main(int argc, char** argv) {
    const char* tag = argc > 1 ? argv[1] : "base";
    const unsigned log_n = 24, NC = 8;
    const size_t n = (size_t)1 << log_n;
    uint64_t *cols[NC], *scr[NC];
    for (unsigned c = 0; c < NC; c++) { cols[c] = dev_table(n, 17 + c); CK(hipMalloc(&scr[c], n * 8)); CK(hipMemset(scr[c], 1, n * 8)); }
    uint64_t* wr4 = dev_table(256 * 4, 1); uint64_t* twu4 = dev_table((size_t)256 * 256 * 4, 2); uint64_t* sc4 = dev_table(4, 3);
    uint64_t* gp = dev_table(256 * 4, 4); uint64_t* tw_lo = dev_table(4096, 5); uint64_t* tw_hi = dev_table(4096, 6);
    uint64_t* tin4 = dev_table((size_t)256 * 256 * 4, 11); uint64_t* tout4 = dev_table((size_t)256 * 16 * 4, 12);
    uint64_t* aux_lo = dev_table(4096, 7); uint64_t* aux_hi = dev_table(4096, 8);
    msntt2::Params Q; memset(&Q, 0, sizeof Q);
    Q.wr4 = wr4; Q.twu4 = twu4; Q.sc4 = sc4; Q.g4 = gp; Q.tw_lo = tw_lo; Q.tw_hi = tw_hi; Q.aux_lo = aux_lo; Q.aux_hi = aux_hi;
    Q.log_n = log_n; Q.V = 1; Q.valid_rows = 256; Q.lo_bits = 12; Q.tin4 = tin4; Q.tout4 = tout4; Q.r3 = 8;
#ifdef VAR_XCD
    Q.xcd_map = 1;
#endif
    const msntt::DigitField f1[2] = {{0, 8, 255}, {8, 0, 255}};
    const dim3 b2(msntt2::NT);
    unsigned scr_of = 0xFFFFFFFFu;      // chain modes: every column through scratch column scr_of
    auto pass = [&](int q, hipStream_t st, unsigned c0, unsigned nc) {
        msntt2::Params A = Q;
        const dim3 g((unsigned)(n / msntt2::TILE), nc);
        if (q == 0) {
            A.log_s = 0; A.nfields = 2; A.fields[0] = f1[0]; A.fields[1] = f1[1];
            for (unsigned c = 0; c < nc; c++) { A.src[c] = cols[c0 + c]; A.dst[c] = scr[scr_of == 0xFFFFFFFFu ? c0 + c : scr_of + c]; }
            hipLaunchKernelGGL((msntt2::ntt2_first_pass<true, false, true, 16, true, true>), g, b2, 0, st, A);
        } else if (q == 1) {
            A.log_s = 8; A.nfields = 1; A.fields[0] = {0, 0, 255};
            for (unsigned c = 0; c < nc; c++) { A.src[c] = A.dst[c] = scr[scr_of == 0xFFFFFFFFu ? c0 + c : scr_of + c]; }
            hipLaunchKernelGGL((msntt2::ntt2_mid_pass<true, false, false, 0, true, true>), g, b2, 0, st, A);
        } else {
            A.log_s = 16; A.nfields = 0;
            for (unsigned c = 0; c < nc; c++) { A.src[c] = scr[scr_of == 0xFFFFFFFFu ? c0 + c : scr_of + c]; A.dst[c] = cols[c0 + c]; }
            hipLaunchKernelGGL((msntt2::ntt2_mid_pass<true, false, true, 0>), g, b2, 0, st, A);
        }
    };
    for (int i = 0; i < 400; i++) pass(1, 0, 0, NC);      // settle the clocks
    CK(hipDeviceSynchronize());
    for (int round = 0; round < 3; round++) {
        const double t1 = time_us([&] { pass(0, 0, 0, NC); }) / NC, t2 = time_us([&] { pass(1, 0, 0, NC); }) / NC, t3 = time_us([&] { pass(2, 0, 0, NC); }) / NC;
        const double tt = time_us([&] { pass(0, 0, 0, NC); pass(1, 0, 0, NC); pass(2, 0, 0, NC); }) / NC;
        printf("%-10s pass 1 %6.1f  pass 2 %6.1f  pass 3 %6.1f  sum %6.1f   transform (batch order) %6.1f us/column = %.3f of 8 TB/s\n",
               tag, t1, t2, t3, t1 + t2 + t3, tt, 268435456.0 / (tt * 1e-6) / 8e12);
        // chain per column through ONE scratch column (128 MiB: stays in the Infinity Cache), one stream; and two chains on two streams
        scr_of = 0;
        const double tc = time_us([&] { for (unsigned c = 0; c < NC; c++) { pass(0, 0, c, 1); pass(1, 0, c, 1); pass(2, 0, c, 1); } }) / NC;
        static hipStream_t s2[2] = {nullptr, nullptr};
        static hipEvent_t ev[3];
        if (!s2[0]) { for (int i = 0; i < 2; i++) CK(hipStreamCreateWithFlags(&s2[i], hipStreamNonBlocking)); for (int i = 0; i < 3; i++) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
        const double tc2 = time_us([&] {
            CK(hipEventRecord(ev[2], 0));
            for (int i = 0; i < 2; i++) CK(hipStreamWaitEvent(s2[i], ev[2], 0));
            for (unsigned c = 0; c < NC; c++) { scr_of = c & 1; pass(0, s2[c & 1], c, 1); pass(1, s2[c & 1], c, 1); pass(2, s2[c & 1], c, 1); }
            for (int i = 0; i < 2; i++) { CK(hipEventRecord(ev[i], s2[i])); CK(hipStreamWaitEvent(0, ev[i], 0)); }
        }) / NC;
        scr_of = 0xFFFFFFFFu;
        printf("%-10s chain per column, one scratch column: %6.1f us/column = %.3f;  two chains on two streams (two scratch columns): %6.1f = %.3f\n",
               tag, tc, 268435456.0 / (tc * 1e-6) / 8e12, tc2, 268435456.0 / (tc2 * 1e-6) / 8e12);
        fflush(stdout);
    }
    return 0;
}
