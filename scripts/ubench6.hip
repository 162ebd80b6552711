// Round-2 micro-benchmark, part 3: does the 256 MiB Infinity Cache (MALL) pay for a three-pass 2^24 transform?
// Pure access patterns of the three radix-256 passes (same tiles, runs and strides as ntt2_kernels.h), optional
// arithmetic stand-in (SPIN dependent v_mad_u64_u32 per element), in different LAUNCH ORDERS:
//   batch      : pass p over all 8 columns, then pass p+1 (what the library did in round 2)
//   chain      : all passes of column c, then column c+1 (col -> scratch -> scratch -> col, or in -> out in place)
//   grouped    : pass 1 and pass 2 interleaved in groups of 32 MiB (pass 2 is local to 512 KiB blocks and a
//                quarter of pass 1's tiles produces a quarter of those blocks), then pass 3
//   streams    : the chains of different columns on 2 / 4 streams
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench6.hip -o scripts/ubench6
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

static constexpr int NCOL = 8;
struct Cols { const uint64_t* src[NCOL]; uint64_t* dst[NCOL]; };

// PASS 1: tile T = (j2, g): reads rows j1 (stride 2^16 words) x 64 words at j' = 256 j2 + 64 g;
//         writes 64 runs of 256 words (k1) at position (j3 256 + j2) 256, j3 = 64 g + t.
// PASS 2: tile T = (U, q): 256 rows at stride 256 words inside block U (2^16 words), 64 words at 64 q; in place.
// PASS 3: tile T: 256 rows at stride 2^16 words, 64 words at 64 T.
// tile index = blockIdx.x * tmul + tadd (selects a group).
template <int PASS, int SPIN, bool NTL, bool NTS>
__global__ void __launch_bounds__(512, 4) k_pass(Cols C, unsigned tmul, unsigned tadd) {
    const uint64_t* __restrict__ src = C.src[blockIdx.y];
    uint64_t* __restrict__ dst = C.dst[blockIdx.y];
    const unsigned T = blockIdx.x * tmul + tadd, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    size_t rbase, rstride;
    if (PASS == 1 || PASS == 4 || PASS == 5) { const unsigned j2 = T >> 2, g = T & 3; rbase = (size_t)j2 * 256 + 64 * g; rstride = 65536; }
    else if (PASS == 2 || PASS == 6) { const unsigned U = T >> 2, q = T & 3; rbase = (size_t)U * 65536 + 64 * q; rstride = 256; }
    else { rbase = (size_t)T * 64; rstride = 65536; }
    uint64_t v[32];
    if (PASS == 6) {        // pass 2 reading pass 5's permuted rows: natural k1 = 64 q + lane sits at (a'>>3) 128 + 32 q + (dl>>1) 16 + (a'&7) 2 + (dl&1)
        const unsigned U = T >> 2, q = T & 3, ap = lane & 15, dl = lane >> 4;
        const unsigned pos = (ap >> 3) * 128 + 32 * q + (dl >> 1) * 16 + (ap & 7) * 2 + (dl & 1);
        #pragma unroll
        for (int i = 0; i < 32; i++) v[i] = src[(size_t)U * 65536 + (size_t)(w + 8 * i) * 256 + pos];
    } else {
        #pragma unroll
        for (int i = 0; i < 32; i++) {
            const uint64_t* p = src + rbase + (size_t)(w + 8 * i) * rstride + lane;
            v[i] = NTL ? __builtin_nontemporal_load(p) : *p;
        }
    }
    if (SPIN) {
        #pragma unroll
        for (int i = 0; i < 32; i++) {
            uint64_t a = v[i];
            #pragma unroll
            for (int s = 0; s < SPIN; s++) a = (uint64_t)(uint32_t)a * 0x9E3779B9u + (a >> 7);
            v[i] = a;
        }
    } else {
        #pragma unroll
        for (int i = 0; i < 32; i++) v[i] += 1;
    }
    if (PASS == 1) {
        const unsigned j2 = T >> 2, g = T & 3;
        #pragma unroll
        for (int i = 0; i < 32; i++) {
            const unsigned e = i * 512 + tid, t = e >> 8, k1 = e & 255;
            uint64_t* p = dst + ((size_t)(64 * g + t) * 256 + j2) * 256 + k1;
            if (NTS) __builtin_nontemporal_store(v[i], p); else *p = v[i];
        }
    } else if (PASS == 5) {                // pass 1, permuted rows: a lane's outputs d, d + 1 adjacent -> 16-byte stores, 8 lanes = one 128-byte line
        const unsigned j2 = T >> 2, g = T & 3, c3 = lane & 7, tl = lane >> 3;
        #pragma unroll
        for (int i = 0; i < 32; i += 2) {
            const unsigned r = i >> 4, d = i & 15, t = 8 * w + tl;
            const unsigned pos = r * 128 + (d >> 1) * 16 + c3 * 2;
            ulonglong2* p = (ulonglong2*)(dst + ((size_t)(64 * g + t) * 256 + j2) * 256 + pos);
            *p = make_ulonglong2(v[i], v[i + 1]);
        }
    } else if (PASS == 4) {                // pass 1 with the store mapping of ntt2_first_pass: 8 lanes x 8 B = 64-byte pieces
        const unsigned j2 = T >> 2, g = T & 3, c3 = lane & 7, tl = lane >> 3;
        #pragma unroll
        for (int i = 0; i < 32; i++) {
            const unsigned r = i >> 4, d = i & 15, k1 = c3 + 8 * r + 16 * d, t = 8 * w + tl;
            uint64_t* p = dst + ((size_t)(64 * g + t) * 256 + j2) * 256 + k1;
            if (NTS) __builtin_nontemporal_store(v[i], p); else *p = v[i];
        }
    } else {
        #pragma unroll
        for (int i = 0; i < 32; i++) {
            uint64_t* p = dst + rbase + (size_t)(w + 8 * i) * rstride + lane;
            if (NTS) __builtin_nontemporal_store(v[i], p); else *p = v[i];
        }
    }
}

struct Bufs { uint64_t* in[NCOL]; uint64_t* out[NCOL]; uint64_t* scr[NCOL]; };

template <int PASS, int SPIN, bool NTL = false, bool NTS = false>
static void launch(hipStream_t st, const uint64_t* const* src, uint64_t* const* dst, int ncols, int group /* -1: all */) {
    Cols C;
    for (int c = 0; c < ncols; c++) { C.src[c] = src[c]; C.dst[c] = dst[c]; }
    const unsigned ntiles = 1024;
    if (group < 0) hipLaunchKernelGGL((k_pass<PASS, SPIN, NTL, NTS>), dim3(ntiles, ncols), dim3(512), 0, st, C, 1u, 0u);
    else if (PASS == 1) hipLaunchKernelGGL((k_pass<PASS, SPIN, NTL, NTS>), dim3(256, ncols), dim3(512), 0, st, C, 4u, (unsigned)group);
    else hipLaunchKernelGGL((k_pass<PASS, SPIN, NTL, NTS>), dim3(256, ncols), dim3(512), 0, st, C, 1u, 256u * group);
}

template <typename F>
static void timeit(const char* name, int spin, F body) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t;
    body(); CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0, 0));
        body();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    const double us = t[2] * 1e3 / NCOL;
    printf("ORDER spin=%3d %-64s %7.1f us/column  (%6.1f GB/s algorithmic)\n", spin, name, us, 268435456.0 / us / 1e3);
    fflush(stdout);
}

static hipStream_t ST[4];
static hipEvent_t EV[4];
// fork: make streams 1..k-1 wait for what is on stream 0 now; join: stream 0 waits for all
static void fork(int k) { CK(hipEventRecord(EV[0], 0)); for (int s = 1; s < k; s++) CK(hipStreamWaitEvent(ST[s], EV[0], 0)); }
static void join(int k) { for (int s = 1; s < k; s++) { CK(hipEventRecord(EV[s], ST[s])); CK(hipStreamWaitEvent(0, EV[s], 0)); } }

template <int SPIN>
static void run_all(Bufs& B) {
    auto one = [](uint64_t* p) { return p; };
    (void)one;
    // batch: in place through 8 scratch columns
    timeit("batch: 3 launches x 8 columns (col->scr, scr, scr->col)", SPIN, [&] {
        launch<1, SPIN>(0, B.in, B.scr, NCOL, -1); launch<2, SPIN>(0, B.scr, B.scr, NCOL, -1); launch<3, SPIN>(0, B.scr, B.in, NCOL, -1);
    });
    timeit("batch: 3 launches x 8 columns out of place (in->out, out, out)", SPIN, [&] {
        launch<1, SPIN>(0, B.in, B.out, NCOL, -1); launch<2, SPIN>(0, B.out, B.out, NCOL, -1); launch<3, SPIN>(0, B.out, B.out, NCOL, -1);
    });
    timeit("chain per column, one scratch (col->scr, scr, scr->col)", SPIN, [&] {
        for (int c = 0; c < NCOL; c++) {
            launch<1, SPIN>(0, &B.in[c], &B.scr[0], 1, -1); launch<2, SPIN>(0, &B.scr[0], &B.scr[0], 1, -1); launch<3, SPIN>(0, &B.scr[0], &B.in[c], 1, -1);
        }
    });
    timeit("chain per column out of place (in->out, out, out)", SPIN, [&] {
        for (int c = 0; c < NCOL; c++) {
            launch<1, SPIN>(0, &B.in[c], &B.out[c], 1, -1); launch<2, SPIN>(0, &B.out[c], &B.out[c], 1, -1); launch<3, SPIN>(0, &B.out[c], &B.out[c], 1, -1);
        }
    });
    timeit("chain per column out of place, nt loads in pass 1", SPIN, [&] {
        for (int c = 0; c < NCOL; c++) {
            launch<1, SPIN, true>(0, &B.in[c], &B.out[c], 1, -1); launch<2, SPIN>(0, &B.out[c], &B.out[c], 1, -1); launch<3, SPIN>(0, &B.out[c], &B.out[c], 1, -1);
        }
    });
    timeit("chain per column one scratch, nt loads p1 + nt stores p3", SPIN, [&] {
        for (int c = 0; c < NCOL; c++) {
            launch<1, SPIN, true>(0, &B.in[c], &B.scr[0], 1, -1); launch<2, SPIN>(0, &B.scr[0], &B.scr[0], 1, -1); launch<3, SPIN, false, true>(0, &B.scr[0], &B.in[c], 1, -1);
        }
    });
    timeit("chain 2 columns per launch, two scratch", SPIN, [&] {
        for (int c = 0; c < NCOL; c += 2) {
            launch<1, SPIN>(0, &B.in[c], &B.scr[0], 2, -1); launch<2, SPIN>(0, &B.scr[0], &B.scr[0], 2, -1); launch<3, SPIN>(0, &B.scr[0], &B.in[c], 2, -1);
        }
    });
    timeit("grouped per column: 4 x [p1(g) p2(g)] then p3, one scratch", SPIN, [&] {
        for (int c = 0; c < NCOL; c++) {
            for (int g = 0; g < 4; g++) { launch<1, SPIN>(0, &B.in[c], &B.scr[0], 1, g); launch<2, SPIN>(0, &B.scr[0], &B.scr[0], 1, g); }
            launch<3, SPIN>(0, &B.scr[0], &B.in[c], 1, -1);
        }
    });
    timeit("grouped, 2 columns per launch, two scratch", SPIN, [&] {
        for (int c = 0; c < NCOL; c += 2) {
            for (int g = 0; g < 4; g++) { launch<1, SPIN>(0, &B.in[c], &B.scr[0], 2, g); launch<2, SPIN>(0, &B.scr[0], &B.scr[0], 2, g); }
            launch<3, SPIN>(0, &B.scr[0], &B.in[c], 2, -1);
        }
    });
    timeit("grouped, 4 columns per launch, four scratch", SPIN, [&] {
        for (int c = 0; c < NCOL; c += 4) {
            for (int g = 0; g < 4; g++) { launch<1, SPIN>(0, &B.in[c], &B.scr[0], 4, g); launch<2, SPIN>(0, &B.scr[0], &B.scr[0], 4, g); }
            launch<3, SPIN>(0, &B.scr[0], &B.in[c], 4, -1);
        }
    });
    for (int k : {2, 4}) {
        char nm[96];
        snprintf(nm, sizeof nm, "chain per column on %d streams (scratch per stream)", k);
        timeit(nm, SPIN, [&] {
            fork(k);
            for (int c = 0; c < NCOL; c++) {
                hipStream_t st = ST[c % k]; uint64_t* s = B.scr[c % k];
                launch<1, SPIN>(st, &B.in[c], &s, 1, -1); launch<2, SPIN>(st, &s, &s, 1, -1); launch<3, SPIN>(st, &s, &B.in[c], 1, -1);
            }
            join(k);
        });
        snprintf(nm, sizeof nm, "grouped per column on %d streams (scratch per stream)", k);
        timeit(nm, SPIN, [&] {
            fork(k);
            for (int c = 0; c < NCOL; c++) {
                hipStream_t st = ST[c % k]; uint64_t* s = B.scr[c % k];
                for (int g = 0; g < 4; g++) { launch<1, SPIN>(st, &B.in[c], &s, 1, g); launch<2, SPIN>(st, &s, &s, 1, g); }
                launch<3, SPIN>(st, &s, &B.in[c], 1, -1);
            }
            join(k);
        });
    }
    // single passes, for reference
    timeit("pass 1 alone x 8 columns (col->scr)  [x1/3 of a transform]", SPIN, [&] { launch<1, SPIN>(0, B.in, B.scr, NCOL, -1); });
    timeit("pass 1 alone, stores in 64-byte pieces as ntt2_first_pass", SPIN, [&] { launch<4, SPIN>(0, B.in, B.scr, NCOL, -1); });
    timeit("pass 1 alone, permuted rows: 16-byte stores, 8 lanes per 128-byte line", SPIN, [&] { launch<5, SPIN>(0, B.in, B.scr, NCOL, -1); });
    timeit("pass 2 alone reading the permuted rows (2 x 256 B per row), out of place (scr->out)", SPIN, [&] { launch<6, SPIN>(0, B.scr, B.out, NCOL, -1); });
    timeit("pass 2 alone x 8 columns (scr)", SPIN, [&] { launch<2, SPIN>(0, B.scr, B.scr, NCOL, -1); });
    timeit("pass 3 alone x 8 columns (scr->col)", SPIN, [&] { launch<3, SPIN>(0, B.scr, B.in, NCOL, -1); });
    timeit("pass 2 alone, same column 8 times (hot)", SPIN, [&] { for (int c = 0; c < NCOL; c++) launch<2, SPIN>(0, &B.scr[0], &B.scr[0], 1, -1); });
    timeit("pass 3 alone in place, same column 8 times (hot)", SPIN, [&] { for (int c = 0; c < NCOL; c++) launch<3, SPIN>(0, &B.scr[0], &B.scr[0], 1, -1); });
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d\n", prop.name, prop.multiProcessorCount);
    Bufs B;
    const size_t bytes = (size_t)8 << 24;
    for (int c = 0; c < NCOL; c++) {
        CK(hipMalloc(&B.in[c], bytes)); CK(hipMalloc(&B.out[c], bytes)); CK(hipMalloc(&B.scr[c], bytes));
        CK(hipMemset(B.in[c], c + 1, bytes)); CK(hipMemset(B.out[c], 0, bytes)); CK(hipMemset(B.scr[c], 0, bytes));
    }
    ST[0] = 0;
    for (int s = 1; s < 4; s++) CK(hipStreamCreateWithFlags(&ST[s], hipStreamNonBlocking));
    for (int s = 0; s < 4; s++) CK(hipEventCreateWithFlags(&EV[s], hipEventDisableTiming));
    const int which = argc > 1 ? atoi(argv[1]) : -1;
    if (which < 0 || which == 0) run_all<0>(B);
    if (which < 0 || which == 40) run_all<40>(B);
    return 0;
}
