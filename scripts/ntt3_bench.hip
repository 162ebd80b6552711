// Round 3: the 2^24 transform fully IN PLACE (no scratch column): ntt2_first_pass<.., TEAM> + ntt2_mid_pass<.., PERM> in place
// + the last pass in place, timed under different launch orders on 8 columns and on one column again and again.
// Tables hold arbitrary residues (timing only; parity is the library's job).
// The rendezvous variant of pass 1 (template flag TEAM) is NOT in the product: apply scripts/ntt3_inplace_team.patch first
// (patch -p1 < scripts/ntt3_inplace_team.patch), build, and revert.  Result kept in profiles/r03_ntt3_inplace.txt.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ministark_amd/csrc scripts/ntt3_bench.hip -o scripts/ntt3_bench
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
    CK(hipDeviceSynchronize()); CK(hipGetLastError());
    return t[t.size() / 2] * 1e3;
}
int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const unsigned log_n = 24, NC = 8;
    const size_t n = (size_t)1 << log_n;
    uint64_t *cols[NC], *scr[NC];
    for (unsigned c = 0; c < NC; c++) { cols[c] = dev_table(n, 17 + c); CK(hipMalloc(&scr[c], n * 8)); CK(hipMemset(scr[c], 1, n * 8)); }
    uint64_t* wr4 = dev_table(256 * 4, 1); uint64_t* twu4 = dev_table((size_t)256 * 256 * 4, 2); uint64_t* sc4 = dev_table(4, 3);
    uint64_t* gp = dev_table(256 * 4, 4); uint64_t* tw_lo = dev_table(4096, 5); uint64_t* tw_hi = dev_table(4096, 6);
    uint64_t* tin4 = dev_table((size_t)256 * 256 * 4, 11); uint64_t* tout4 = dev_table((size_t)256 * 16 * 4, 12);
    uint64_t* aux_lo = dev_table(4096, 7); uint64_t* aux_hi = dev_table(4096, 8);
    const size_t SYNCW = 1 + 2 * 256 * NC;
    unsigned* sync; CK(hipMalloc(&sync, 4 * SYNCW * 4)); CK(hipMemset(sync, 0, 4 * SYNCW * 4));     // one area per stream

    msntt2::Params Q; memset(&Q, 0, sizeof Q);
    Q.wr4 = wr4; Q.twu4 = twu4; Q.sc4 = sc4; Q.g4 = gp; Q.tw_lo = tw_lo; Q.tw_hi = tw_hi; Q.aux_lo = aux_lo; Q.aux_hi = aux_hi;
    Q.log_n = log_n; Q.V = 1; Q.valid_rows = 256; Q.lo_bits = 12; Q.tin4 = tin4; Q.tout4 = tout4; Q.r3 = 8; Q.sync = sync;
    const msntt::DigitField f1[2] = {{0, 8, 255}, {8, 0, 255}};
    const dim3 b2(msntt2::NT);
    const unsigned ntiles = (unsigned)(n / msntt2::TILE);
    // settle the clocks
    {
        msntt2::Params A = Q; A.log_s = 16; for (unsigned c = 0; c < NC; c++) { A.src[c] = cols[c]; A.dst[c] = cols[c]; }
        for (int i = 0; i < 300; i++) hipLaunchKernelGGL((msntt2::ntt2_mid_pass<true, false, true, 0>), dim3(ntiles, NC), b2, 0, 0, A);
        CK(hipDeviceSynchronize());
    }
    // in place: pass 1 (team), pass 2 (permuted rows, in place), pass 3 (in place)
    auto inplace = [&](hipStream_t st, unsigned c0, unsigned nc, int which = 7, unsigned area = 0) {
        msntt2::Params A = Q;
        A.sync = sync + area * SYNCW;
        const dim3 g(ntiles, nc);
        for (unsigned c = 0; c < nc; c++) { A.src[c] = cols[c0 + c]; A.dst[c] = cols[c0 + c]; }
        A.ntiles = ntiles; A.ncols = nc; A.nslabs = nc * (ntiles / 4);
        A.log_s = 0; A.nfields = 2; A.fields[0] = f1[0]; A.fields[1] = f1[1];
        if (which & 1) hipLaunchKernelGGL((msntt2::ntt2_first_pass<true, false, true, 16, true, true, true>), g, b2, 0, st, A);
        A.log_s = 8; A.nfields = 1; A.fields[0] = {0, 0, 255};
        if (which & 2) hipLaunchKernelGGL((msntt2::ntt2_mid_pass<true, false, false, 0, true, true>), g, b2, 0, st, A);
        A.log_s = 16; A.nfields = 0;
        if (which & 4) hipLaunchKernelGGL((msntt2::ntt2_mid_pass<true, false, true, 0>), g, b2, 0, st, A);
    };
    // round 2: through a scratch column (col -> scr, scr -> col, col)
    auto scratch = [&](hipStream_t st, unsigned c0, unsigned nc, unsigned scr0) {
        msntt2::Params A = Q;
        const dim3 g(ntiles, nc);
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
    hipStream_t sts[4]; hipEvent_t evs[5];
    for (int i = 0; i < 4; i++) CK(hipStreamCreateWithFlags(&sts[i], hipStreamNonBlocking));
    for (int i = 0; i < 5; i++) CK(hipEventCreateWithFlags(&evs[i], hipEventDisableTiming));
    double t;
    printf("settled\n"); fflush(stdout);
    inplace(0, 0, 1, 1); CK(hipDeviceSynchronize()); printf("pass 1 team ok\n"); fflush(stdout);
    inplace(0, 0, 1, 2); CK(hipDeviceSynchronize()); printf("pass 2 ok\n"); fflush(stdout);
    inplace(0, 0, 1, 4); CK(hipDeviceSynchronize()); printf("pass 3 ok\n"); fflush(stdout);
    scratch(0, 0, 1, 0); CK(hipDeviceSynchronize()); printf("scratch 1 col ok\n"); fflush(stdout);
    scratch(0, 0, NC, 0); CK(hipDeviceSynchronize()); printf("scratch 8 col ok\n"); fflush(stdout);
    inplace(0, 0, NC, 1); CK(hipDeviceSynchronize()); printf("inplace 8 col p1 ok\n"); fflush(stdout);
    inplace(0, 0, NC, 6); CK(hipDeviceSynchronize()); printf("inplace 8 col p23 ok\n"); fflush(stdout);
    for (int round = 0; round < 2; round++) {
        t = time_us([&] { scratch(0, 0, NC, 0); });                                   printf("scratch,  batch order (3 launches x 8 columns)           %7.1f us/column\n", t / NC);
        t = time_us([&] { inplace(0, 0, NC); });                                      printf("in place, batch order (3 launches x 8 columns)           %7.1f us/column\n", t / NC);
        t = time_us([&] { for (unsigned c = 0; c < NC; c++) inplace(0, c, 1); });     printf("in place, chain per column                               %7.1f us/column\n", t / NC);
        t = time_us([&] { for (unsigned c = 0; c < NC; c += 2) inplace(0, c, 2); });  printf("in place, chain of 2 columns per launch                  %7.1f us/column\n", t / NC);
        t = time_us([&] { for (unsigned c = 0; c < NC; c++) inplace(0, 0, 1); });     printf("in place, the same column 8 times                        %7.1f us/column\n", t / NC);
        t = time_us([&] { for (unsigned c = 0; c < NC; c++) scratch(0, 0, 1, 0); });  printf("scratch,  the same column 8 times                        %7.1f us/column\n", t / NC);
        for (int k : {2, 4}) {
            t = time_us([&] {
                CK(hipEventRecord(evs[4], 0));
                for (int i = 0; i < k; i++) CK(hipStreamWaitEvent(sts[i], evs[4], 0));
                for (unsigned c = 0; c < NC; c++) inplace(sts[c % k], c, 1, 7, c % k);
                for (int i = 0; i < k; i++) { CK(hipEventRecord(evs[i], sts[i])); CK(hipStreamWaitEvent(0, evs[i], 0)); }
            });
            printf("in place, chain per column on %d streams                  %7.1f us/column\n", k, t / NC);
        }
        t = time_us([&] { inplace(0, 0, NC, 1); });                                   printf("  pass 1 alone (team, in place), 8 columns               %7.1f us/column\n", t / NC);
        t = time_us([&] { inplace(0, 0, NC, 2); });                                   printf("  pass 2 alone (permuted rows, in place), 8 columns      %7.1f us/column\n", t / NC);
        t = time_us([&] { inplace(0, 0, NC, 4); });                                   printf("  pass 3 alone (in place), 8 columns                     %7.1f us/column\n", t / NC);
        t = time_us([&] { for (unsigned c = 0; c < NC; c++) inplace(0, 0, 1, 1); });  printf("  pass 1 alone, the same column 8 times                  %7.1f us/column\n", t / NC);
        t = time_us([&] { for (unsigned c = 0; c < NC; c++) inplace(0, 0, 1, 2); });  printf("  pass 2 alone, the same column 8 times                  %7.1f us/column\n", t / NC);
        t = time_us([&] { for (unsigned c = 0; c < NC; c++) inplace(0, 0, 1, 4); });  printf("  pass 3 alone, the same column 8 times                  %7.1f us/column\n", t / NC);
    }
    // the counters must be back at zero
    std::vector<unsigned> h(SYNCW * 4);
    CK(hipMemcpy(h.data(), sync, h.size() * 4, hipMemcpyDeviceToHost));
    unsigned nz = 0; for (unsigned v : h) nz += v != 0;
    printf("sync words left non-zero: %u\n", nz);
    return 0;
}
