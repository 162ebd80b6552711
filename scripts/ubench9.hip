// Round-4 micro-benchmark: do the strided 256 x 64-word tiles of the radix-256 passes run faster with 16 bytes per lane?
// The product's passes move 8 bytes per lane (global_load/store_dwordx2, one 512-byte row run per wave instruction) and sit at the time
// of their bare access pattern (profiles/r02_ubench6_*, r03_ntt_pass_bench.txt).  MI355X_MICROARCH.md prices 8-byte accesses at
// 0.54-0.70x the 16-byte rate and T21 (cdna_hip_programming.md) shows dwordx2 stores to be issue-bound per instruction.
// Variants (same tiles, same bytes):
//   L8/S8    as the product
//   L16/S16  a wave instruction covers TWO row runs (lanes 0-31 one row, lanes 32-63 another), each lane 2 adjacent words; one
//            v_permlane32_swap per dword puts the 16 rows of one word column into one lane (what a radix-16 network needs)
//   W128     tile of 256 rows x 128 words, 1024 threads: one 1 KiB row run per wave instruction, no swap
//   PERSIST  512 resident workgroups loop over the tiles and request the next tile's rows before working on the current one
// SPIN = units of stand-in arithmetic per element (4 plain 32-bit ops + 1 v_mad_u64_u32 each; 17 units = the passes' ~86 instructions).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench9.hip -o scripts/ubench9
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

static constexpr int NCOL = 8;
struct Cols { const uint64_t* src[NCOL]; uint64_t* dst[NCOL]; };
struct alignas(16) Pair { uint64_t x, y; };

template <int SPIN>
__device__ __forceinline__ uint64_t work(uint64_t v) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    #pragma unroll
    for (int s = 0; s < SPIN; s++) {
        lo += hi; hi ^= lo; lo += 0x9E3779B9u; hi += lo;
        const uint64_t m = (uint64_t)lo * 0x85EBCA6Bu + hi;
        lo = (uint32_t)m; hi = (uint32_t)(m >> 32);
    }
    if (SPIN == 0) lo += 1;
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ void swap32(uint64_t& a, uint64_t& b) {     // a.upper half-wave <-> b.lower half-wave
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    auto r0 = __builtin_amdgcn_permlane32_swap(al, bl, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(ah, bh, false, false);
    a = ((uint64_t)r1[0] << 32) | r0[0];
    b = ((uint64_t)r1[1] << 32) | r0[1];
}

// tile geometry: MODE 2 = pass 2 (block U of 2^16 words, rows at stride 256 words, run q), MODE 3 = pass 3 (rows at stride 2^16 words)
template <int MODE>
__device__ __forceinline__ void geom(unsigned T, size_t& rbase, size_t& rstride) {
    if (MODE == 2) { const unsigned U = T >> 2, q = T & 3; rbase = (size_t)U * 65536 + 64 * q; rstride = 256; }
    else { rbase = (size_t)T * 64; rstride = 65536; }
}

template <int MODE, int LW, int SW, int SPIN>
__global__ void __launch_bounds__(512, 4) k_tile(Cols C) {
    const uint64_t* __restrict__ src = C.src[blockIdx.y];
    uint64_t* __restrict__ dst = C.dst[blockIdx.y];
    const unsigned tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l32 = lane & 31;
    size_t rbase, rstride; geom<MODE>(blockIdx.x, rbase, rstride);
    __shared__ uint64_t occ[8192];                       // 64 KiB: two workgroups per CU, as the product's passes
    occ[tid] = tid;
    uint64_t v[2][16];
    #pragma unroll
    for (int h = 0; h < 2; h++) {
        if (LW == 8) {
            #pragma unroll
            for (int a = 0; a < 16; a++) v[h][a] = src[rbase + (size_t)(w + 8 * h + 16 * a) * rstride + lane];
        } else {
            #pragma unroll
            for (int a = 0; a < 8; a++) {
                const Pair t = *(const Pair*)(src + rbase + (size_t)(w + 8 * h + 16 * (a + 8 * half)) * rstride + 2 * l32);
                v[h][a] = t.x; v[h][a + 8] = t.y;
            }
        }
    }
    if (LW == 16) {
        #pragma unroll
        for (int h = 0; h < 2; h++)
            #pragma unroll
            for (int a = 0; a < 8; a++) swap32(v[h][a], v[h][a + 8]);
    }
    #pragma unroll
    for (int h = 0; h < 2; h++)
        #pragma unroll
        for (int a = 0; a < 16; a++) v[h][a] = work<SPIN>(v[h][a]);
    __syncthreads();
    v[0][0] += occ[tid ^ 64] >> 20;
    #pragma unroll
    for (int h = 0; h < 2; h++) {
        if (SW == 8) {
            #pragma unroll
            for (int a = 0; a < 16; a++) dst[rbase + (size_t)(w + 8 * h + 16 * a) * rstride + lane] = v[h][a];
        } else {
            #pragma unroll
            for (int a = 0; a < 16; a += 2) {
                uint64_t p = v[h][a], q = v[h][a + 1];
                if (LW == 16) swap32(p, q);                      // (with L8 the data is in the wrong lanes for this; the bytes moved are the same)
                *(Pair*)(dst + rbase + (size_t)(w + 8 * h + 16 * (a + half)) * rstride + 2 * l32) = Pair{p, q};
            }
        }
    }
}

// L8 / S8 with non-temporal loads and stores; MODE 1 = pass 1 (rows at stride 2^16 words in, 64 runs of 256 words out with the product's
// permuted 16-byte stores -- ntt2_first_pass<.., PERM>), MODE 2 / 3 as above
// MAP: which tile a workgroup takes (workgroup b runs on XCD b % 8): 0 = b; 1 = every XCD walks its own contiguous eighth of the tiles;
// 2 = the 10-bit reversal of b (concurrent workgroups far apart); 3 = b with its low three bits moved to the top
template <int MAP> __device__ __forceinline__ unsigned tile_of(unsigned b) {
    if (MAP == 1) return (b & 7) * 128 + (b >> 3);
    if (MAP == 2) return __brev(b) >> 22;
    if (MAP == 3) return ((b & 7) << 7) | (b >> 3);
    if (MAP == 4) { const unsigned x = b & 7, i = b >> 3, j2 = x * 32 + (i & 31), g = i >> 5; return j2 * 4 + g; }     // pass 1: an XCD walks consecutive j2 of one g
    if (MAP == 5) { const unsigned x = b & 7, i = b >> 3, g = x & 3, j2 = (x >> 2) * 128 + i; return j2 * 4 + g; }      // pass 1: g fixed per XCD pair
    if (MAP == 7) { const unsigned x = b & 7, i = b >> 3, g = x & 3, j2 = 2 * i + (x >> 2); return j2 * 4 + g; }                   // pass 1: two XCDs per g interleave consecutive j2
    if (MAP == 8) { const unsigned x = b & 7, i = b >> 3, g = x >> 1, j2 = 2 * i + (x & 1); return j2 * 4 + g; }                    // the same with the XCD pairs adjacent
    if (MAP == 9) { const unsigned x = b & 7, i = b >> 3, g = x & 3, j2 = (x >> 2) * 128 + ((i & 1) ? 127 - (i >> 1) : (i >> 1)); return j2 * 4 + g; }   // map 5 walking from both ends
    if (MAP == 6) { const unsigned x = b & 7, i = b >> 3; return (i >> 2) * 32 + x * 4 + (i & 3); }                      // groups of 4 consecutive tiles per XCD
    return b;
}
template <int MODE, int SPIN, bool NT, int MAP = 0>
__global__ void __launch_bounds__(512, 4) k_tile_nt(Cols C) {
    const uint64_t* __restrict__ src = C.src[blockIdx.y];
    uint64_t* __restrict__ dst = C.dst[blockIdx.y];
    const unsigned tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    __shared__ uint64_t occ[8192];
    occ[tid] = tid;
    size_t rbase, rstride;
    const unsigned bx = tile_of<MAP>(blockIdx.x);
    if (MODE == 1) { const unsigned j2 = bx >> 2, g = bx & 3; rbase = (size_t)j2 * 256 + 64 * g; rstride = 65536; }
    else geom<MODE>(bx, rbase, rstride);
    uint64_t v[32];
    #pragma unroll
    for (int i = 0; i < 32; i++) { const uint64_t* p = src + rbase + (size_t)(w + 8 * i) * rstride + lane; v[i] = NT ? __builtin_nontemporal_load(p) : *p; }
    #pragma unroll
    for (int i = 0; i < 32; i++) v[i] = work<SPIN>(v[i]);
    __syncthreads();
    v[0] += occ[tid ^ 64] >> 20;
    if (MODE == 1) {
        typedef unsigned long long v2 __attribute__((ext_vector_type(2)));
        const unsigned j2 = bx >> 2, g = bx & 3, c3 = lane & 7, tl = lane >> 3;
        #pragma unroll
        for (int i = 0; i < 32; i += 2) {
            const unsigned r = i >> 4, d = i & 15, t = 8 * w + tl;
            const unsigned pos = r * 128 + (d >> 1) * 16 + c3 * 2;
            v2* p = (v2*)(dst + ((size_t)(64 * g + t) * 256 + j2) * 256 + pos);
            v2 val; val.x = v[i]; val.y = v[i + 1];
            if (NT) __builtin_nontemporal_store(val, p); else *p = val;
        }
    } else {
        #pragma unroll
        for (int i = 0; i < 32; i++) { uint64_t* p = dst + rbase + (size_t)(w + 8 * i) * rstride + lane; if (NT) __builtin_nontemporal_store(v[i], p); else *p = v[i]; }
    }
}

// 256 rows x 128 words, 1024 threads, one 1 KiB row run per wave instruction; 16 words per lane
template <int MODE, int SPIN>
__global__ void __launch_bounds__(1024) k_tile_w128(Cols C) {
    const uint64_t* __restrict__ src = C.src[blockIdx.y];
    uint64_t* __restrict__ dst = C.dst[blockIdx.y];
    const unsigned tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    size_t rbase, rstride;
    if (MODE == 2) { const unsigned U = blockIdx.x >> 1, q = blockIdx.x & 1; rbase = (size_t)U * 65536 + 128 * q; rstride = 256; }
    else { rbase = (size_t)blockIdx.x * 128; rstride = 65536; }
    __shared__ uint64_t occ[16384];                      // 128 KiB: one workgroup per CU
    occ[tid] = tid;
    Pair v[16];
    #pragma unroll
    for (int a = 0; a < 16; a++) v[a] = *(const Pair*)(src + rbase + (size_t)(w + 16 * a) * rstride + 2 * lane);
    __syncthreads();
    v[0].x += occ[tid ^ 64] >> 20;
    #pragma unroll
    for (int a = 0; a < 16; a++) { v[a].x = work<SPIN>(v[a].x); v[a].y = work<SPIN>(v[a].y); }
    #pragma unroll
    for (int a = 0; a < 16; a++) *(Pair*)(dst + rbase + (size_t)(w + 16 * a) * rstride + 2 * lane) = v[a];
}

// persistent: gridDim.x resident workgroups walk the HALF tiles (16 words per lane) of tiles blockIdx.x, + gridDim.x, ...; the next
// half is requested before the work on the current one (32 data registers in flight + 32 being worked on)
template <int MODE, int LW, int SPIN>
__global__ void __launch_bounds__(512, 4) k_tile_persist(Cols C, unsigned ntiles) {
    const uint64_t* __restrict__ src = C.src[blockIdx.y];
    uint64_t* __restrict__ dst = C.dst[blockIdx.y];
    const unsigned tid = threadIdx.x, lane = tid & 63, w = tid >> 6, half = lane >> 5, l32 = lane & 31;
    __shared__ uint64_t occ[8192];
    occ[tid] = tid;
    __syncthreads();
    uint64_t v[16], nx[16];
    auto load = [&](unsigned H, uint64_t* o) {             // half tile H = 2 T + h
        size_t rbase, rstride; geom<MODE>(H >> 1, rbase, rstride);
        const unsigned h = H & 1;
        if (LW == 8) {
            #pragma unroll
            for (int a = 0; a < 16; a++) o[a] = src[rbase + (size_t)(w + 8 * h + 16 * a) * rstride + lane];
        } else {
            #pragma unroll
            for (int a = 0; a < 8; a++) {
                const Pair t = *(const Pair*)(src + rbase + (size_t)(w + 8 * h + 16 * (a + 8 * half)) * rstride + 2 * l32);
                o[a] = t.x; o[a + 8] = t.y;
            }
        }
    };
    const unsigned nh = 2 * ntiles, hstep = 2 * gridDim.x;
    auto next_of = [&](unsigned H) { return (H & 1) ? (H - 1 + hstep) : H + 1; };
    unsigned H = 2 * blockIdx.x;
    load(H, v);
    v[0] += occ[tid ^ 64] >> 20;
    while (H < nh) {
        const unsigned Hn = next_of(H);
        if (Hn < nh) load(Hn, nx);
        size_t rbase, rstride; geom<MODE>(H >> 1, rbase, rstride);
        const unsigned h = H & 1;
        if (LW == 16) {
            #pragma unroll
            for (int a = 0; a < 8; a++) swap32(v[a], v[a + 8]);
        }
        #pragma unroll
        for (int a = 0; a < 16; a++) v[a] = work<SPIN>(v[a]);
        if (LW == 8) {
            #pragma unroll
            for (int a = 0; a < 16; a++) dst[rbase + (size_t)(w + 8 * h + 16 * a) * rstride + lane] = v[a];
        } else {
            #pragma unroll
            for (int a = 0; a < 16; a += 2) {
                uint64_t p = v[a], q = v[a + 1];
                swap32(p, q);
                *(Pair*)(dst + rbase + (size_t)(w + 8 * h + 16 * (a + half)) * rstride + 2 * l32) = Pair{p, q};
            }
        }
        #pragma unroll
        for (int a = 0; a < 16; a++) v[a] = nx[a];
        H = Hn;
    }
}

template <int W>
__global__ void __launch_bounds__(256) k_copy(const uint64_t* __restrict__ s, uint64_t* __restrict__ d, size_t n) {
    const size_t step = (size_t)gridDim.x * blockDim.x;
    if (W == 8) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) d[i] = s[i] + 1; }
    else { const Pair* sp = (const Pair*)s; Pair* dp = (Pair*)d; for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / 2; i += step) { Pair t = sp[i]; t.x += 1; dp[i] = t; } }
}

template <typename F>
static void timeit(const char* name, F body) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t;
    body(); body(); CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 7; rep++) {
        CK(hipEventRecord(e0, 0)); body(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    const double us = t[3] * 1e3 / NCOL;
    printf("%-86s %7.1f us/column  (%6.1f GB/s read+write)\n", name, us, 268435456.0 / us / 1e3);
    fflush(stdout);
}

static uint64_t *IN[NCOL], *SCR[NCOL];
template <int MODE> static Cols cols() {
    Cols C;
    for (int c = 0; c < NCOL; c++) { C.src[c] = SCR[c]; C.dst[c] = MODE == 2 ? SCR[c] : IN[c]; }
    return C;
}

template <int MODE, int SPIN>
static void run_mode() {
    const Cols C = cols<MODE>();
    char nm[128];
    auto name = [&](const char* what) { snprintf(nm, sizeof nm, "pass %d pattern, %2d units of arithmetic: %s", MODE, SPIN, what); return nm; };
    timeit(name("L8 / S8 (the product)"), [&] { hipLaunchKernelGGL((k_tile<MODE, 8, 8, SPIN>), dim3(1024, NCOL), dim3(512), 0, 0, C); });
    timeit(name("L16 / S8"), [&] { hipLaunchKernelGGL((k_tile<MODE, 16, 8, SPIN>), dim3(1024, NCOL), dim3(512), 0, 0, C); });
    timeit(name("L8 / S16"), [&] { hipLaunchKernelGGL((k_tile<MODE, 8, 16, SPIN>), dim3(1024, NCOL), dim3(512), 0, 0, C); });
    timeit(name("L16 / S16 (two row runs per instruction + half-wave swaps)"), [&] { hipLaunchKernelGGL((k_tile<MODE, 16, 16, SPIN>), dim3(1024, NCOL), dim3(512), 0, 0, C); });
    timeit(name("256 x 128 words, 1024 threads, 1 KiB per instruction"), [&] { hipLaunchKernelGGL((k_tile_w128<MODE, SPIN>), dim3(512, NCOL), dim3(1024), 0, 0, C); });
    timeit(name("persistent 512 workgroups, next tile requested first, L8 / S8"), [&] { hipLaunchKernelGGL((k_tile_persist<MODE, 8, SPIN>), dim3(512 / NCOL, NCOL), dim3(512), 0, 0, C, 1024u); });
    timeit(name("persistent 512 workgroups, next tile requested first, L16 / S16"), [&] { hipLaunchKernelGGL((k_tile_persist<MODE, 16, SPIN>), dim3(512 / NCOL, NCOL), dim3(512), 0, 0, C, 1024u); });
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d\n", prop.name, prop.multiProcessorCount);
    const size_t bytes = (size_t)8 << 24;
    for (int c = 0; c < NCOL; c++) {
        CK(hipMalloc(&IN[c], bytes)); CK(hipMalloc(&SCR[c], bytes));
        CK(hipMemset(IN[c], c + 1, bytes)); CK(hipMemset(SCR[c], 3, bytes));
    }
    // settle the clocks
    for (int i = 0; i < 300; i++) hipLaunchKernelGGL((k_tile<3, 8, 8, 17>), dim3(1024, NCOL), dim3(512), 0, 0, cols<3>());
    CK(hipDeviceSynchronize());
    {
        uint64_t *a, *b; const size_t n = (size_t)1 << 27;      // 1 GiB each
        CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMemset(a, 1, n * 8));
        for (int blocks : {2048, 8192, 65536}) {
            char nm[96];
            snprintf(nm, sizeof nm, "copy 1 GiB, 8 bytes per lane, %d blocks (x8 columns' worth of bytes)", blocks);
            timeit(nm, [&] { hipLaunchKernelGGL(k_copy<8>, dim3(blocks), dim3(256), 0, 0, a, b, n); });
            snprintf(nm, sizeof nm, "copy 1 GiB, 16 bytes per lane, %d blocks", blocks);
            timeit(nm, [&] { hipLaunchKernelGGL(k_copy<16>, dim3(blocks), dim3(256), 0, 0, a, b, n); });
        }
        CK(hipFree(a)); CK(hipFree(b));
    }
    for (int round = 0; round < 2; round++) {
        run_mode<3, 0>(); run_mode<2, 0>();
        run_mode<3, 17>(); run_mode<2, 17>();
    }
    run_mode<3, 8>();
    run_mode<3, 26>();
    for (int round = 0; round < 2; round++) {
        Cols C1; for (int c = 0; c < NCOL; c++) { C1.src[c] = IN[c]; C1.dst[c] = SCR[c]; }
        const Cols C2 = cols<2>(), C3 = cols<3>();
        timeit("pass 1 pattern (permuted 16-byte stores), default policy, 17 units", [&] { hipLaunchKernelGGL((k_tile_nt<1, 17, false>), dim3(1024, NCOL), dim3(512), 0, 0, C1); });
        timeit("pass 1 pattern (permuted 16-byte stores), non-temporal,   17 units", [&] { hipLaunchKernelGGL((k_tile_nt<1, 17, true>), dim3(1024, NCOL), dim3(512), 0, 0, C1); });
        timeit("pass 2 pattern, default policy, 17 units", [&] { hipLaunchKernelGGL((k_tile_nt<2, 17, false>), dim3(1024, NCOL), dim3(512), 0, 0, C2); });
        timeit("pass 2 pattern, non-temporal,   17 units", [&] { hipLaunchKernelGGL((k_tile_nt<2, 17, true>), dim3(1024, NCOL), dim3(512), 0, 0, C2); });
        timeit("pass 3 pattern, default policy, 17 units", [&] { hipLaunchKernelGGL((k_tile_nt<3, 17, false>), dim3(1024, NCOL), dim3(512), 0, 0, C3); });
        timeit("pass 3 pattern, non-temporal,   17 units", [&] { hipLaunchKernelGGL((k_tile_nt<3, 17, true>), dim3(1024, NCOL), dim3(512), 0, 0, C3); });
        timeit("pass 3 pattern, non-temporal,    0 units", [&] { hipLaunchKernelGGL((k_tile_nt<3, 0, true>), dim3(1024, NCOL), dim3(512), 0, 0, C3); });
        timeit("pass 2 pattern, non-temporal,    0 units", [&] { hipLaunchKernelGGL((k_tile_nt<2, 0, true>), dim3(1024, NCOL), dim3(512), 0, 0, C2); });
        timeit("pass 3 pattern, nt, 17 units, tile map 1 (an eighth of the tiles per XCD)", [&] { hipLaunchKernelGGL((k_tile_nt<3, 17, true, 1>), dim3(1024, NCOL), dim3(512), 0, 0, C3); });
        timeit("pass 3 pattern, nt, 17 units, tile map 2 (bit-reversed)", [&] { hipLaunchKernelGGL((k_tile_nt<3, 17, true, 2>), dim3(1024, NCOL), dim3(512), 0, 0, C3); });
        timeit("pass 1 pattern, nt, 17 units, tile map 1", [&] { hipLaunchKernelGGL((k_tile_nt<1, 17, true, 1>), dim3(1024, NCOL), dim3(512), 0, 0, C1); });
        timeit("pass 1 pattern, nt, 17 units, tile map 2", [&] { hipLaunchKernelGGL((k_tile_nt<1, 17, true, 2>), dim3(1024, NCOL), dim3(512), 0, 0, C1); });
        timeit("pass 1 pattern, nt, 17 units, tile map 4 (XCD: consecutive j2 of one g)", [&] { hipLaunchKernelGGL((k_tile_nt<1, 17, true, 4>), dim3(1024, NCOL), dim3(512), 0, 0, C1); });
        timeit("pass 1 pattern, nt, 17 units, tile map 5 (g fixed per XCD)", [&] { hipLaunchKernelGGL((k_tile_nt<1, 17, true, 5>), dim3(1024, NCOL), dim3(512), 0, 0, C1); });
        timeit("pass 1 pattern, nt, 17 units, tile map 7 (two XCDs per g, interleaved j2)", [&] { hipLaunchKernelGGL((k_tile_nt<1, 17, true, 7>), dim3(1024, NCOL), dim3(512), 0, 0, C1); });
        timeit("pass 1 pattern, nt, 17 units, tile map 8 (adjacent XCD pairs per g)", [&] { hipLaunchKernelGGL((k_tile_nt<1, 17, true, 8>), dim3(1024, NCOL), dim3(512), 0, 0, C1); });
        timeit("pass 1 pattern, nt, 17 units, tile map 6 (4 consecutive tiles per XCD)", [&] { hipLaunchKernelGGL((k_tile_nt<1, 17, true, 6>), dim3(1024, NCOL), dim3(512), 0, 0, C1); });
        timeit("pass 3 pattern, nt, 17 units, tile map 6 (4 consecutive tiles per XCD)", [&] { hipLaunchKernelGGL((k_tile_nt<3, 17, true, 6>), dim3(1024, NCOL), dim3(512), 0, 0, C3); });
        timeit("pass 3 pattern, nt, 17 units, tile map 3", [&] { hipLaunchKernelGGL((k_tile_nt<3, 17, true, 3>), dim3(1024, NCOL), dim3(512), 0, 0, C3); });
        timeit("pass 2 pattern, nt, 17 units, tile map 1", [&] { hipLaunchKernelGGL((k_tile_nt<2, 17, true, 1>), dim3(1024, NCOL), dim3(512), 0, 0, C2); });
        timeit("pass 2 pattern, nt, 17 units, tile map 2", [&] { hipLaunchKernelGGL((k_tile_nt<2, 17, true, 2>), dim3(1024, NCOL), dim3(512), 0, 0, C2); });
    }
    return 0;
}
