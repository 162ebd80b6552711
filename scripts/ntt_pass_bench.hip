// Times single NTT pass kernels (round-1 and limb-form) on 8 x 2^24 synthetic columns: for exploring kernel
// variants without rebuilding the library.  Tables hold arbitrary non-zero residues (timing only).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ministark_amd/csrc [-DVARIANT...] scripts/ntt_pass_bench.hip -o scripts/ntt_pass_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
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
static double time_us(F launch, int reps = 7) {
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
    const unsigned log_n = 24, NC = 8;
    const size_t n = (size_t)1 << log_n;
    uint64_t *cols[NC], *scr[NC];
    for (unsigned c = 0; c < NC; c++) { cols[c] = dev_table(n, 17 + c); CK(hipMalloc(&scr[c], n * 8)); CK(hipMemset(scr[c], 1, n * 8)); }
    uint64_t* wr4 = dev_table(256 * 4, 1); uint64_t* twu4 = dev_table((size_t)256 * 256 * 4, 2); uint64_t* sc4 = dev_table(4, 3);
    uint64_t* gp = dev_table(256, 4); uint64_t* tw_lo = dev_table(4096, 5); uint64_t* tw_hi = dev_table(4096, 6);
    uint64_t* tin4 = dev_table((size_t)256 * 256 * 4, 11); uint64_t* tout4 = dev_table((size_t)256 * 16 * 4, 12);
    uint64_t* aux_lo = dev_table(4096, 7); uint64_t* aux_hi = dev_table(4096, 8); uint64_t* wr = dev_table(256, 9); uint64_t* gtab = dev_table(256, 10);

    msntt2::Params Q; memset(&Q, 0, sizeof Q);
    msntt::PassParams P; memset(&P, 0, sizeof P);
    Q.wr4 = wr4; Q.twu4 = twu4; Q.sc4 = sc4; Q.g4 = gp; Q.tw_lo = tw_lo; Q.tw_hi = tw_hi; Q.aux_lo = aux_lo; Q.aux_hi = aux_hi;
    Q.log_n = log_n; Q.V = 1; Q.valid_rows = 256; Q.lo_bits = 12; Q.tin4 = tin4; Q.tout4 = tout4; Q.r3 = 8;
    P.tw_lo = tw_lo; P.tw_hi = tw_hi; P.wr = wr; P.aux_lo = aux_lo; P.aux_hi = aux_hi; P.gtab = gtab; P.log_n = log_n; P.V = 1; P.valid_rows = 256; P.lo_bits = 12;
    // pass 1 of an (8, 8, 8) plan: j' = (j2, j3) -> layout (j3, j2)
    msntt::DigitField f1[2] = {{0, 8, 255}, {8, 0, 255}};
    auto set_pass = [&](int q) {
        Q.log_s = P.log_s = 8 * q;
        Q.nfields = P.nfields = (q == 0) ? 2 : (q == 1 ? 1 : 0);
        if (q == 0) { Q.fields[0] = P.fields[0] = f1[0]; Q.fields[1] = P.fields[1] = f1[1]; }
        if (q == 1) { Q.fields[0] = P.fields[0] = {0, 0, 255}; }
        for (unsigned c = 0; c < NC; c++) {
            Q.src[c] = P.src[c] = (q == 0) ? cols[c] : scr[c];
            Q.dst[c] = P.dst[c] = (q == 2) ? cols[c] : scr[c];
        }
    };
    const dim3 g2((unsigned)(n / msntt2::TILE), NC), b2(msntt2::NT), g1((unsigned)(n / msntt::TILE), NC), b1(msntt::NT);
    double t;
    set_pass(0);
    // settle the clocks (about one second of work)
    for (int i = 0; i < 1200; i++) hipLaunchKernelGGL((msntt::ntt_first_pass<false, true>), g1, b1, 0, 0, P);
    CK(hipDeviceSynchronize());
  for (int round = 0; round < 2; round++) {
    set_pass(0);
    t = time_us([&] { hipLaunchKernelGGL((msntt::ntt_first_pass<false, true>), g1, b1, 0, 0, P); });   printf("round-1 pass 1 (coset)   %7.1f us/column\n", t / NC);
    t = time_us([&] { hipLaunchKernelGGL((msntt2::ntt2_first_pass<true, false, true, 16>), g2, b2, 0, 0, Q); }); printf("limb    pass 1 (coset)   %7.1f us/column\n", t / NC);
    t = time_us([&] { hipLaunchKernelGGL((msntt2::ntt2_first_pass<true, false, false, 16>), g2, b2, 0, 0, Q); }); printf("limb    pass 1 (subgroup) %6.1f us/column\n", t / NC);
    t = time_us([&] { hipLaunchKernelGGL((msntt2::ntt2_first_pass<true, false, true, 16, true>), g2, b2, 0, 0, Q); }); printf("limb    pass 1 (coset, uniform factor)    %7.1f us/column\n", t / NC);
    t = time_us([&] { hipLaunchKernelGGL((msntt2::ntt2_first_pass<true, false, false, 16, true>), g2, b2, 0, 0, Q); }); printf("limb    pass 1 (subgroup, uniform factor) %7.1f us/column\n", t / NC);
    t = time_us([&] { hipLaunchKernelGGL((msntt2::ntt2_first_pass<true, false, true, 16, true, true>), g2, b2, 0, 0, Q); }); printf("limb    pass 1 (coset, uniform, permuted rows)  %5.1f us/column\n", t / NC);
    set_pass(1);
    for (unsigned c = 0; c < NC; c++) Q.dst[c] = cols[c];
    t = time_us([&] { hipLaunchKernelGGL((msntt2::ntt2_mid_pass<true, false, false, 0, true, true>), g2, b2, 0, 0, Q); });   printf("limb    pass 2 (load factor, permuted rows in)  %5.1f us/column\n", t / NC);
    set_pass(1);
    t = time_us([&] { hipLaunchKernelGGL((msntt2::ntt2_mid_pass<true, false, false, 0, true>), g2, b2, 0, 0, Q); });   printf("limb    pass 2 (per-lane load factor)     %7.1f us/column\n", t / NC);
    t = time_us([&] { hipLaunchKernelGGL((msntt::ntt_mid_pass<16, false, false, 0>), g1, b1, 0, 0, P); }); printf("round-1 pass 2           %7.1f us/column\n", t / NC);
    t = time_us([&] { hipLaunchKernelGGL((msntt2::ntt2_mid_pass<true, false, false, 0>), g2, b2, 0, 0, Q); });   printf("limb    pass 2           %7.1f us/column\n", t / NC);
    // the middle passes of the (256, R, 256) plans (2^17 .. 2^23 points), run here over the same 2^24 words per "column" (blocks of
    // R rows of 256 words; in place on permuted rows, as in the plans): time per 2^24 words
    set_pass(1);
    t = time_us([&] { hipLaunchKernelGGL((msntt2::ntt2_small_mid_pass<true, false, 1, true>), g2, b2, 0, 0, Q); });   printf("limb    middle pass R = 2   (registers only)       %5.1f us per 2^24 words\n", t / NC);
    t = time_us([&] { hipLaunchKernelGGL((msntt2::ntt2_small_mid_pass<true, false, 2, true>), g2, b2, 0, 0, Q); });   printf("limb    middle pass R = 4   (registers only)       %5.1f us per 2^24 words\n", t / NC);
    t = time_us([&] { hipLaunchKernelGGL((msntt2::ntt2_small_mid_pass<true, false, 3, true>), g2, b2, 0, 0, Q); });   printf("limb    middle pass R = 8   (registers only)       %5.1f us per 2^24 words\n", t / NC);
    t = time_us([&] { hipLaunchKernelGGL((msntt2::ntt2_small_mid_pass<true, false, 4, true>), g2, b2, 0, 0, Q); });   printf("limb    middle pass R = 16  (registers only)       %5.1f us per 2^24 words\n", t / NC);
    t = time_us([&] { hipLaunchKernelGGL((msntt2::ntt2_mid_pass_r<true, false, 1, true>), g2, b2, 0, 0, Q); });       printf("limb    middle pass R = 32  (16 x 2, one exchange)  %5.1f us per 2^24 words\n", t / NC);
    t = time_us([&] { hipLaunchKernelGGL((msntt2::ntt2_mid_pass_r<true, false, 2, true>), g2, b2, 0, 0, Q); });       printf("limb    middle pass R = 64  (16 x 4, one exchange)  %5.1f us per 2^24 words\n", t / NC);
    t = time_us([&] { hipLaunchKernelGGL((msntt2::ntt2_mid_pass_r<true, false, 3, true>), g2, b2, 0, 0, Q); });       printf("limb    middle pass R = 128 (16 x 8, one exchange)  %5.1f us per 2^24 words\n", t / NC);
    set_pass(2);
    t = time_us([&] { hipLaunchKernelGGL((msntt::ntt_mid_pass<16, false, true, 0>), g1, b1, 0, 0, P); });  printf("round-1 pass 3           %7.1f us/column\n", t / NC);
    t = time_us([&] { hipLaunchKernelGGL((msntt2::ntt2_mid_pass<true, false, true, 0>), g2, b2, 0, 0, Q); });    printf("limb    pass 3           %7.1f us/column\n", t / NC);
    t = time_us([&] { hipLaunchKernelGGL((msntt::ntt_mid_pass<16, false, true, 0, true>), g1, b1, 0, 0, P); });  printf("round-1 pass 3, bit-reversed store  %7.1f us/column\n", t / NC);
    t = time_us([&] { hipLaunchKernelGGL(msntt2::ntt2_last_pass_bitrev<true>, g2, b2, 0, 0, Q); });    printf("limb    pass 3, bit-reversed store  %7.1f us/column\n", t / NC);
  }
    // whole transforms (uniform factor + permuted rows): launch orders
    {
        hipStream_t sts[4]; hipEvent_t evs[5];
        for (int i = 0; i < 4; i++) CK(hipStreamCreateWithFlags(&sts[i], hipStreamNonBlocking));
        for (int i = 0; i < 5; i++) CK(hipEventCreateWithFlags(&evs[i], hipEventDisableTiming));
        auto passes = [&](hipStream_t st, unsigned c0, unsigned nc, unsigned scr0) {
            msntt2::Params A = Q;
            const dim3 g((unsigned)(n / msntt2::TILE), nc);
            A.log_s = 0; A.nfields = 2; A.fields[0] = f1[0]; A.fields[1] = f1[1];
            for (unsigned c = 0; c < nc; c++) { A.src[c] = cols[c0 + c]; A.dst[c] = scr[scr0 + c]; }
            hipLaunchKernelGGL((msntt2::ntt2_first_pass<true, false, true, 16, true, true>), g, b2, 0, st, A);
            A.log_s = 8; A.nfields = 1; A.fields[0] = {0, 0, 255};
            for (unsigned c = 0; c < nc; c++) { A.src[c] = scr[scr0 + c]; A.dst[c] = cols[c0 + c]; }
            hipLaunchKernelGGL((msntt2::ntt2_mid_pass<true, false, false, 0, true, true>), g, b2, 0, st, A);
            A.log_s = 16; A.nfields = 0;
            for (unsigned c = 0; c < nc; c++) { A.src[c] = cols[c0 + c]; A.dst[c] = cols[c0 + c]; }
            hipLaunchKernelGGL((msntt2::ntt2_mid_pass<true, false, true, 0>), g, b2, 0, st, A);
        };
        for (int round = 0; round < 2; round++) {
            t = time_us([&] { passes(0, 0, NC, 0); });
            printf("transform, batch order (3 launches x 8 columns)        %7.1f us/column\n", t / NC);
            t = time_us([&] { for (unsigned c = 0; c < NC; c += 2) passes(0, c, 2, 0); });
            printf("transform, 2 columns per launch                        %7.1f us/column\n", t / NC);
            t = time_us([&] { for (unsigned c = 0; c < NC; c++) passes(0, c, 1, 0); });
            printf("transform, chain per column, one stream                %7.1f us/column\n", t / NC);
            for (int k : {2, 4}) {
                t = time_us([&] {
                    CK(hipEventRecord(evs[4], 0));
                    for (int i = 0; i < k; i++) CK(hipStreamWaitEvent(sts[i], evs[4], 0));
                    for (unsigned c = 0; c < NC; c++) passes(sts[c % k], c, 1, c % k);
                    for (int i = 0; i < k; i++) { CK(hipEventRecord(evs[i], sts[i])); CK(hipStreamWaitEvent(0, evs[i], 0)); }
                });
                printf("transform, chain per column on %d streams               %7.1f us/column\n", k, t / NC);
            }
            for (int k : {2, 4}) {
                t = time_us([&] {
                    CK(hipEventRecord(evs[4], 0));
                    for (int i = 0; i < k; i++) CK(hipStreamWaitEvent(sts[i], evs[4], 0));
                    for (int i = 0; i < k; i++) passes(sts[i], i * (NC / k), NC / k, i * (NC / k));
                    for (int i = 0; i < k; i++) { CK(hipEventRecord(evs[i], sts[i])); CK(hipStreamWaitEvent(0, evs[i], 0)); }
                });
                printf("transform, %d streams x %d columns per launch            %7.1f us/column\n", k, NC / k, t / NC);
            }
        }
    }
    CK(hipDeviceSynchronize());
    return 0;
}
