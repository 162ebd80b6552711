// Round-5 micro-benchmark: can a 2^24-point natural-order transform be TWO fabric passes when the exchange that the granule rule
// forbids in LDS (docs/DESIGN_HISTORY.md 8.1-5; result: DESIGN.md 9.1) goes through the XCD's 4 MiB L2 instead?  n = 4096 x 4096 words, i = i0 + 4096 i1:
//   pass A  super-tile = 4096 rows (i1, stride 32 KiB) x W words (W = 64: 2 MiB, 512-byte runs; W = 32: 1 MiB, 256-byte runs).
//           A TEAM of 16 workgroups that sit on ONE XCD (found from HW_REG_XCC_ID, never assumed) does radix 4096 = 256 x 16:
//           A1  member g loads rows g + 16 m (nt), radix-256 stand-in (two halves through 64 KiB of LDS, like ntt2_first_pass),
//               stores rows m + 256 g of the destination with the default policy (the lines stay dirty in L2);
//           team barrier (one monotonic counter per team, relaxed agent atomics, s_waitcnt vmcnt(0) before the arrival);
//           A2  member j loads rows 16 j + q + 256 g (L2 hits, nt = past L1), radix-16 stand-in in registers, stores IN PLACE (nt).
//   pass B  row pass: a workgroup owns 4 consecutive rows ka (128 KiB contiguous), whole-row radix-4096 stand-in (two exchanges
//           through LDS), stores X[ka + 4096 kb]: 32-byte pieces.  The 16 workgroups that hold the 64 rows of one 512-byte run are
//           placed next to each other on one XCD so that its write-back L2 can complete the lines before it evicts them
//           (k_rowsB), or they are a team that stages the pieces in a 2 MiB L2-resident buffer and copies whole runs out (k_teamB).
// Reported per variant: microseconds per column (8 columns per launch) and, under rocprofv3 --pmc, FETCH_SIZE / WRITE_SIZE.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench11.hip -o scripts/ubench11
// Run:   scripts/ubench11            (all variants)        scripts/ubench11 <variant> <launches>   (one variant, for the PMC passes)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

static constexpr int NCOL = 8;
static constexpr unsigned LOGN = 24;
static constexpr unsigned ROWW = 4096;                    // words per row = rows per column
struct Cols { const uint64_t* src[NCOL]; uint64_t* dst[NCOL]; };
struct Ctl {
    unsigned xcd_count[8];
    unsigned registered, bad, pad[6];
    unsigned bar[64 * 32];                                // one counter per team, each on its own 128-byte line
};

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

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7u;
}
__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned add_relaxed(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// uniform base + 32-bit lane offset: one address register per access (the saddr form of global_load / global_store)
__device__ __forceinline__ uint64_t ld_nt(const uint64_t* base, uint32_t word) { return __builtin_nontemporal_load((const uint64_t*)((const char*)base + (word << 3))); }
__device__ __forceinline__ void st_nt(uint64_t* base, uint32_t word, uint64_t v) { __builtin_nontemporal_store(v, (uint64_t*)((char*)base + (word << 3))); }
__device__ __forceinline__ void st_def(uint64_t* base, uint32_t word, uint64_t v) { *(uint64_t*)((char*)base + (word << 3)) = v; }

// Team formation: rank inside the XCD this workgroup really runs on.  -> false when the placement left this team incomplete.
template <int TEAMS>
__device__ __forceinline__ bool join_team(Ctl* ctl, unsigned& xcc, unsigned& team, unsigned& member) {
    __shared__ unsigned s_x, s_r, s_ok;
    if (threadIdx.x == 0) {
        const unsigned x = xcc_id();
        const unsigned r = add_relaxed(&ctl->xcd_count[x], 1);
        add_relaxed(&ctl->registered, 1);
        unsigned ok = 1;
        const unsigned need = (r / 16 + 1) * 16;
        unsigned spins = 0;
        while (ld_relaxed(&ctl->xcd_count[x]) < need) {
            if (ld_relaxed(&ctl->registered) >= gridDim.x || ++spins > (1u << 20)) { if (ld_relaxed(&ctl->xcd_count[x]) < need) ok = 0; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (r / 16 >= (unsigned)TEAMS) ok = 0;
        if (!ok) add_relaxed(&ctl->bad, 1);
        s_x = x; s_r = r; s_ok = ok;
    }
    __syncthreads();
    xcc = s_x; team = s_r / 16; member = s_r % 16;
    return s_ok != 0;
}

// every wave's stores are acknowledged by L2, then one arrival per workgroup; the team meets at generation gen (1, 2, ...)
__device__ __forceinline__ void team_barrier(unsigned* ctr, unsigned gen) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        add_relaxed(ctr, 1);
        unsigned spins = 0;                               // bounded: a lost team mate must not hang the box (the run is then void)
        while (ld_relaxed(ctr) < 16u * gen && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
}

// the radix-256 stand-in of a 256 x W tile held as PER words per lane: SPIN units, both halves through 64 KiB of LDS, SPIN units
template <int PER, int SPIN>
__device__ __forceinline__ void tile_work(uint64_t (&v)[PER], uint64_t* lds) {
    const unsigned t = threadIdx.x;
    constexpr int H = PER / 2;
    constexpr unsigned HALF = 512u * H;
    #pragma unroll
    for (int h = 0; h < 2; h++) {
        #pragma unroll
        for (int i = 0; i < H; i++) lds[t + 512 * i] = work<SPIN>(v[h * H + i]);
        __syncthreads();
        #pragma unroll
        for (int i = 0; i < H; i++) v[h * H + i] = work<SPIN>(lds[((t + 512 * i) * 17u) % HALF]);
        __syncthreads();
    }
}

// ---- pass A as a team ------------------------------------------------------------------------------------------------
// INPLACE: A1 stores into dst and A2 re-reads / overwrites dst.  !INPLACE: A1 stores into a per-team staging slab (stays in L2 for
// the whole launch), A2 reads it and stores to dst; a second barrier protects the slab.
template <int W, int SPIN1, int SPIN2, int TEAMS, bool INPLACE>
__global__ void __launch_bounds__(512, 4) k_teamA(Cols C, Ctl* ctl, uint64_t* staging, int ncol) {
    __shared__ uint64_t lds[8192];
    unsigned xcc, team, g;
    if (!join_team<TEAMS>(ctl, xcc, team, g)) return;
    const unsigned T = xcc * TEAMS + team, NT_ = 8 * TEAMS;
    unsigned* ctr = &ctl->bar[T * 32];
    constexpr int PER = W / 2;                           // 256 rows x W words over 512 lanes
    constexpr unsigned RPI = 512 / W;                    // rows covered by one load instruction of the workgroup
    const unsigned t = threadIdx.x, c_ = t % W, r0_ = t / W;
    const unsigned tiles_per_col = ROWW / W, ntiles = (unsigned)ncol * tiles_per_col;
    uint64_t* slab = staging + (size_t)T * (4096u * W);
    unsigned gen = 0;
    for (unsigned tile = T; tile < ntiles; tile += NT_) {
        const unsigned col = tile / tiles_per_col, tc = tile % tiles_per_col;
        const uint64_t* __restrict__ src = C.src[col] + (size_t)tc * W;
        uint64_t* __restrict__ dst = C.dst[col] + (size_t)tc * W;
        unsigned c = c_, r0 = r0_;
        asm volatile("" : "+v"(c), "+v"(r0));               // the offsets are recomputed per tile instead of living in 96 hoisted registers
        uint64_t v[PER];
        #pragma unroll
        for (int i = 0; i < PER; i++) { const unsigned m = r0 + RPI * i; v[i] = ld_nt(src, (g + 16 * m) * ROWW + c); }
        tile_work<PER, SPIN1>(v, lds);
        #pragma unroll
        for (int i = 0; i < PER; i++) {
            const unsigned m = r0 + RPI * i, row = m + 256 * g;
            if (INPLACE) st_def(dst, row * ROWW + c, v[i]); else st_def(slab, row * W + c, v[i]);
        }
        team_barrier(ctr, ++gen);
        asm volatile("" : "+v"(c), "+v"(r0));
        #pragma unroll
        for (int i = 0; i < PER; i++) {
            const unsigned ri = r0 + RPI * i, row = 16 * g + (ri & 15) + 256 * (ri >> 4);
            v[i] = INPLACE ? ld_nt(dst, row * ROWW + c) : ld_nt(slab, row * W + c);
        }
        #pragma unroll
        for (int i = 0; i < PER; i++) v[i] = work<SPIN2>(v[i]);
        #pragma unroll
        for (int i = 0; i < PER; i++) {
            const unsigned ri = r0 + RPI * i, row = 16 * g + (ri & 15) + 256 * (ri >> 4);
            st_nt(dst, row * ROWW + c, v[i]);
        }
        if (!INPLACE) team_barrier(ctr, ++gen);
    }
}

// ---- pass B, no synchronisation: 32-byte pieces completed by the XCD's L2 ---------------------------------------------------
// MAP 1: block b -> XCD b % 8, the 16 workgroups of a 64-row group consecutive on that XCD.  MAP 0: launch order.
template <int SPIN, bool NTST, int MAP, int VEC>
__global__ void __launch_bounds__(512, 4) k_rowsB(Cols C) {
    __shared__ uint64_t lds[8192];
    unsigned b = blockIdx.x;
    unsigned G, m;
    if (MAP == 1) { const unsigned x = b & 7, i = b >> 3; G = x + 8 * (i >> 4); m = i & 15; } else { G = b >> 4; m = b & 15; }
    const uint64_t* __restrict__ src = C.src[blockIdx.y] + (size_t)(64 * G + 4 * m) * ROWW;
    uint64_t* __restrict__ dst = C.dst[blockIdx.y] + 64 * G + 4 * m;
    const unsigned t = threadIdx.x;
    uint64_t v[32];
    #pragma unroll
    for (int i = 0; i < 32; i++) v[i] = ld_nt(src, t + 512 * i);
    tile_work<32, SPIN>(v, lds);
    if (VEC == 1) {
        #pragma unroll
        for (int i = 0; i < 32; i++) {
            const unsigned e = t + 512 * i, r = e & 3, kb = e >> 2;
            if (NTST) st_nt(dst, kb * ROWW + r, v[i]); else st_def(dst, kb * ROWW + r, v[i]);
        }
    } else {
        #pragma unroll
        for (int i = 0; i < 16; i++) {
            const unsigned e = t + 512 * i, rp = e & 1, kb = e >> 1;
            typedef uint64_t u2 __attribute__((ext_vector_type(2)));
            u2 w; w.x = v[2 * i]; w.y = v[2 * i + 1];
            u2* p = (u2*)((char*)dst + ((kb * ROWW + 2 * rp) << 3));
            if (NTST) __builtin_nontemporal_store(w, p); else *p = w;
        }
    }
}

// ---- pass B as a team: pieces into an L2-resident slab, whole 512-byte runs out ------------------------------------------------
template <int SPIN, int TEAMS>
__global__ void __launch_bounds__(512, 4) k_teamB(Cols C, Ctl* ctl, uint64_t* staging, int ncol) {
    __shared__ uint64_t lds[8192];
    unsigned xcc, team, m;
    if (!join_team<TEAMS>(ctl, xcc, team, m)) return;
    const unsigned T = xcc * TEAMS + team, NT_ = 8 * TEAMS;
    unsigned* ctr = &ctl->bar[T * 32];
    const unsigned t_ = threadIdx.x;
    uint64_t* slab = staging + (size_t)T * (4096u * 64);
    const unsigned groups_per_col = ROWW / 64, ngroups = (unsigned)ncol * groups_per_col;
    unsigned gen = 0;
    for (unsigned grp = T; grp < ngroups; grp += NT_) {
        const unsigned col = grp / groups_per_col, G = grp % groups_per_col;
        const uint64_t* __restrict__ src = C.src[col] + (size_t)(64 * G + 4 * m) * ROWW;
        uint64_t* __restrict__ dst = C.dst[col] + 64 * G;
        unsigned t = t_;
        asm volatile("" : "+v"(t));
        uint64_t v[32];
        #pragma unroll
        for (int i = 0; i < 32; i++) v[i] = ld_nt(src, t + 512 * i);
        tile_work<32, SPIN>(v, lds);
        #pragma unroll
        for (int i = 0; i < 32; i++) { const unsigned e = t + 512 * i, r = e & 3, kb = e >> 2; st_def(slab, kb * 64 + 4 * m + r, v[i]); }
        team_barrier(ctr, ++gen);
        asm volatile("" : "+v"(t));
        #pragma unroll
        for (int i = 0; i < 32; i++) { const unsigned e = t + 512 * i; v[i] = ld_nt(slab, m * 16384 + e); }
        #pragma unroll
        for (int i = 0; i < 32; i++) { const unsigned e = t + 512 * i, kb = 256 * m + (e >> 6), cc = e & 63; st_nt(dst, kb * ROWW + cc, v[i]); }
        team_barrier(ctr, ++gen);
    }
}

// ---- reference patterns of the shipped three-pass plan (for the same box, same binary) -------------------------------------------
// a 256 x 64-word tile at row stride `stride` words, runs of 64 words: pass 1 (stride 2^16) / pass 3 of ntt2_kernels.h, out of place
template <int SPIN>
__global__ void __launch_bounds__(512, 4) k_tile256(Cols C, unsigned log_stride) {
    __shared__ uint64_t lds[8192];
    const unsigned t = threadIdx.x, c = t & 63, r0 = t >> 6, b = blockIdx.x;
    const size_t stride = (size_t)1 << log_stride;
    const unsigned runs_per_row = (unsigned)(stride >> 6);
    const size_t base = (size_t)(b / runs_per_row) * (stride << 8) + (size_t)(b % runs_per_row) * 64;
    const uint64_t* __restrict__ src = C.src[blockIdx.y] + base;
    uint64_t* __restrict__ dst = C.dst[blockIdx.y] + base;
    uint64_t v[32];
    #pragma unroll
    for (int i = 0; i < 32; i++) v[i] = ld_nt(src, ((r0 + 8 * i) << log_stride) + c);
    tile_work<32, SPIN>(v, lds);
    #pragma unroll
    for (int i = 0; i < 32; i++) st_nt(dst, ((r0 + 8 * i) << log_stride) + c, v[i]);
}

// ---- is the L2 write-back at all?  every workgroup stores the SAME 64 KiB of its own TIMES times (waiting for the acknowledgements in between) --
template <int TIMES, bool NTST>
__global__ void __launch_bounds__(512, 4) k_rewrite(uint64_t* buf) {
    uint64_t* mine = buf + (size_t)blockIdx.x * 8192;
    const unsigned t = threadIdx.x;
    for (int r = 0; r < TIMES; r++) {
        #pragma unroll
        for (int i = 0; i < 16; i++) { if (NTST) st_nt(mine, t + 512 * i, (uint64_t)r + i); else st_def(mine, t + 512 * i, (uint64_t)r + i); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
// and does a line another CU of the same XCD stored come back from L2?  phase 1 stores 64 KiB per workgroup, team barrier, phase 2 loads the
// 64 KiB of the NEXT team member (default or nt loads): FETCH_SIZE should be ~0 if the stored lines are kept.
template <int TEAMS, bool NTLD>
__global__ void __launch_bounds__(512, 4) k_handoff(Ctl* ctl, uint64_t* buf, uint64_t* sink, int rounds) {
    unsigned xcc, team, m;
    if (!join_team<TEAMS>(ctl, xcc, team, m)) return;
    const unsigned T = xcc * TEAMS + team;
    unsigned* ctr = &ctl->bar[T * 32];
    const unsigned t = threadIdx.x;
    uint64_t acc = 0;
    unsigned gen = 0;
    for (int r = 0; r < rounds; r++) {
        uint64_t* mine = buf + ((size_t)(r * 8 * TEAMS + T) * 16 + m) * 8192;
        const uint64_t* other = buf + ((size_t)(r * 8 * TEAMS + T) * 16 + ((m + 5) & 15)) * 8192;
        #pragma unroll
        for (int i = 0; i < 16; i++) st_def(mine, t + 512 * i, (uint64_t)r + i + t);
        team_barrier(ctr, ++gen);
        #pragma unroll
        for (int i = 0; i < 16; i++) acc += NTLD ? ld_nt(other, t + 512 * i) : *(const uint64_t*)((const char*)other + ((t + 512 * i) << 3));
    }
    if (acc == 0x123456789abcull) sink[0] = acc;
}

// read the wide pattern (256 rows at stride 2^16 words, 64-word runs), write either ONE contiguous 128 KiB block per tile (WR = 0)
// or the local pattern (256 runs at stride 2^8 words inside a 512 KiB block, WR = 1): what a plan with [j2][j3][k1] as its
// intermediate layout would do in its first / last pass (no plan can use them without adding a wide stream elsewhere -- the digit order is forced)
template <int WR>
__global__ void __launch_bounds__(512, 4) k_mixed(Cols C) {
    __shared__ uint64_t lds[8192];
    const unsigned t = threadIdx.x, c = t & 63, r0 = t >> 6, b = blockIdx.x;
    const uint64_t* __restrict__ src = C.src[blockIdx.y] + (size_t)b * 64;
    uint64_t* __restrict__ dst = C.dst[blockIdx.y] + (WR == 0 ? (size_t)b * 16384 : (size_t)(b >> 2) * 65536 + (size_t)(b & 3) * 64);
    uint64_t v[32];
    #pragma unroll
    for (int i = 0; i < 32; i++) v[i] = ld_nt(src, ((r0 + 8 * i) << 16) + c);
    tile_work<32, 0>(v, lds);
    #pragma unroll
    for (int i = 0; i < 32; i++) st_nt(dst, WR == 0 ? (r0 + 8 * i) * 64 + c : ((r0 + 8 * i) << 8) + c, v[i]);
}

static uint64_t *IN[NCOL], *SCR[NCOL], *OUT[NCOL], *STAGE;
static Ctl* CTL;
static int g_launches = 10, g_reps = 7;

template <class F>
static double timeit(const char* name, F launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; i++) launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int rep = 0; rep < g_reps; rep++) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < g_launches; i++) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float m; CK(hipEventElapsedTime(&m, e0, e1)); ms.push_back(m / g_launches);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    const double us = ms[ms.size() / 2] * 1000.0 / NCOL;
    Ctl h; CK(hipMemcpy(&h, CTL, 64, hipMemcpyDeviceToHost));
    printf("%-118s %8.2f us/column  %6.2f TB/s%s\n", name, us, 2.0 * 8 * (1 << LOGN) / us * 1e-6, h.bad ? "   ** INCOMPLETE TEAMS: placement was not 1/8 per XCD, result void **" : "");
    fflush(stdout);
    return us;
}

static void reset_ctl() { CK(hipMemsetAsync(CTL, 0, sizeof(Ctl), 0)); }

template <int W, int S1, int S2, int TEAMS, bool INPLACE>
static void teamA() {
    Cols C; for (int c = 0; c < NCOL; c++) { C.src[c] = IN[c]; C.dst[c] = SCR[c]; }
    char nm[200];
    snprintf(nm, sizeof nm, "A team: 4096 rows x %d words (%d KiB super-tile), %d teams/XCD, %s, units %d+%d", W, 4096 * W * 8 >> 10, TEAMS, INPLACE ? "in place" : "slab", 2 * S1, S2);
    timeit(nm, [&] { reset_ctl(); hipLaunchKernelGGL((k_teamA<W, S1, S2, TEAMS, INPLACE>), dim3(8 * TEAMS * 16), dim3(512), 0, 0, C, CTL, STAGE, NCOL); });
}
template <int SPIN, bool NTST, int MAP, int VEC>
static void rowsB() {
    Cols C; for (int c = 0; c < NCOL; c++) { C.src[c] = SCR[c]; C.dst[c] = OUT[c]; }
    char nm[200];
    snprintf(nm, sizeof nm, "B rows: 4 rows/workgroup, %d-byte stores into 32-byte pieces, %s stores, %s, units %d", 8 * VEC, NTST ? "nt" : "default", MAP ? "XCD-grouped" : "launch order", 2 * SPIN);
    timeit(nm, [&] { hipLaunchKernelGGL((k_rowsB<SPIN, NTST, MAP, VEC>), dim3(1024, NCOL), dim3(512), 0, 0, C); });
}
template <int SPIN, int TEAMS>
static void teamB() {
    Cols C; for (int c = 0; c < NCOL; c++) { C.src[c] = SCR[c]; C.dst[c] = OUT[c]; }
    char nm[200];
    snprintf(nm, sizeof nm, "B team: pieces into a 2 MiB slab, whole runs out, %d teams/XCD, units %d", TEAMS, 2 * SPIN);
    timeit(nm, [&] { reset_ctl(); hipLaunchKernelGGL((k_teamB<SPIN, TEAMS>), dim3(8 * TEAMS * 16), dim3(512), 0, 0, C, CTL, STAGE, NCOL); });
}
template <int SPIN>
static void tile256(unsigned log_stride, const char* what) {
    Cols C; for (int c = 0; c < NCOL; c++) { C.src[c] = IN[c]; C.dst[c] = SCR[c]; }
    char nm[200];
    snprintf(nm, sizeof nm, "reference: 256 x 64-word tiles at stride 2^%u words (%s), nt, units %d", log_stride, what, 2 * SPIN);
    timeit(nm, [&] { hipLaunchKernelGGL((k_tile256<SPIN>), dim3(1024, NCOL), dim3(512), 0, 0, C, log_stride); });
}

template <int TIMES, bool NTST>
static void rewrite() {
    char nm[200];
    snprintf(nm, sizeof nm, "rewrite: 2048 workgroups store their own 64 KiB %d times, %s stores (128 MiB distinct bytes; x8 'columns')", TIMES, NTST ? "nt" : "default");
    timeit(nm, [&] { hipLaunchKernelGGL((k_rewrite<TIMES, NTST>), dim3(2048), dim3(512), 0, 0, SCR[0]); });
}
template <int TEAMS, bool NTLD>
static void handoff() {
    char nm[200];
    snprintf(nm, sizeof nm, "handoff: store 64 KiB, team barrier, load a team mate's 64 KiB (%s loads), %d teams/XCD, 8 rounds", NTLD ? "nt" : "default", TEAMS);
    timeit(nm, [&] { reset_ctl(); hipLaunchKernelGGL((k_handoff<TEAMS, NTLD>), dim3(8 * TEAMS * 16), dim3(512), 0, 0, CTL, SCR[0], SCR[1], 8); });
}
template <int WR>
static void mixed() {
    Cols C; for (int c = 0; c < NCOL; c++) { C.src[c] = IN[c]; C.dst[c] = SCR[c]; }
    char nm[200];
    snprintf(nm, sizeof nm, "mixed: wide reads (stride 2^16 words), %s, nt, units 0", WR == 0 ? "one contiguous 128 KiB block written per tile" : "local writes (stride 2^8 words inside 512 KiB)");
    timeit(nm, [&] { hipLaunchKernelGGL((k_mixed<WR>), dim3(1024, NCOL), dim3(512), 0, 0, C); });
}
struct Variant { const char* id; void (*fn)(); };
static void ref16_0() { tile256<0>(16, "pass 1 / 3 of the shipped plan"); }
static void ref16_8() { tile256<8>(16, "pass 1 / 3 of the shipped plan"); }
static void ref8_0() { tile256<0>(8, "pass 2 of the shipped plan"); }
static void ref12_0() { tile256<0>(12, "a pass-A tile without the team"); }
static const Variant VARIANTS[] = {
    {"mixed_contig", mixed<0>}, {"mixed_local", mixed<1>},
    {"ref16_0", ref16_0}, {"ref16_8", ref16_8}, {"ref8_0", ref8_0}, {"ref12_0", ref12_0},
    {"A64_t4_ip_0", teamA<64, 0, 0, 4, true>}, {"A64_t2_ip_0", teamA<64, 0, 0, 2, true>}, {"A64_t3_ip_0", teamA<64, 0, 0, 3, true>},
    {"A32_t4_ip_0", teamA<32, 0, 0, 4, true>}, {"A64_t4_slab_0", teamA<64, 0, 0, 4, false>}, {"A64_t2_slab_0", teamA<64, 0, 0, 2, false>},
    {"A32_t4_slab_0", teamA<32, 0, 0, 4, false>},
    {"A64_t4_ip_w", teamA<64, 8, 7, 4, true>}, {"A64_t2_ip_w", teamA<64, 8, 7, 2, true>}, {"A32_t4_ip_w", teamA<32, 8, 7, 4, true>},
    {"A64_t4_slab_w", teamA<64, 8, 7, 4, false>},
    {"B_def_grp_8", rowsB<0, false, 1, 1>}, {"B_nt_grp_8", rowsB<0, true, 1, 1>}, {"B_def_ord_8", rowsB<0, false, 0, 1>},
    {"B_def_grp_16", rowsB<0, false, 1, 2>}, {"B_nt_grp_16", rowsB<0, true, 1, 2>},
    {"B_def_grp_8_w", rowsB<10, false, 1, 1>}, {"B_def_grp_16_w", rowsB<10, false, 1, 2>},
    {"A64_t1_ip_0", teamA<64, 0, 0, 1, true>}, {"A32_t2_ip_0", teamA<32, 0, 0, 2, true>}, {"A32_t1_ip_0", teamA<32, 0, 0, 1, true>},
    {"A64_t1_slab_0", teamA<64, 0, 0, 1, false>}, {"A32_t1_slab_0", teamA<32, 0, 0, 1, false>},
    {"rewrite1_def", rewrite<1, false>}, {"rewrite8_def", rewrite<8, false>}, {"rewrite8_nt", rewrite<8, true>},
    {"handoff_t1_def", handoff<1, false>}, {"handoff_t1_nt", handoff<1, true>},
    {"Bteam_t4_0", teamB<0, 4>}, {"Bteam_t2_0", teamB<0, 2>}, {"Bteam_t4_w", teamB<10, 4>},
};

int main(int argc, char** argv) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d\n", prop.name, prop.multiProcessorCount);
    const size_t bytes = (size_t)8 << LOGN;
    for (int c = 0; c < NCOL; c++) {
        CK(hipMalloc(&IN[c], bytes)); CK(hipMalloc(&SCR[c], bytes)); CK(hipMalloc(&OUT[c], bytes));
        CK(hipMemset(IN[c], c + 1, bytes)); CK(hipMemset(SCR[c], 3, bytes)); CK(hipMemset(OUT[c], 5, bytes));
    }
    CK(hipMalloc(&STAGE, (size_t)64 * 4096 * 64 * 8)); CK(hipMemset(STAGE, 0, (size_t)64 * 4096 * 64 * 8));
    CK(hipMalloc(&CTL, sizeof(Ctl))); CK(hipMemset(CTL, 0, sizeof(Ctl)));
    if (argc >= 2) {
        if (argc >= 3) { g_launches = atoi(argv[2]); g_reps = 1; }
        for (const Variant& v : VARIANTS) if (!strcmp(v.id, argv[1])) { v.fn(); return 0; }
        printf("unknown variant %s\n", argv[1]); return 1;
    }
    for (int i = 0; i < 30; i++) { Cols C; for (int c = 0; c < NCOL; c++) { C.src[c] = IN[c]; C.dst[c] = SCR[c]; } hipLaunchKernelGGL((k_tile256<8>), dim3(1024, NCOL), dim3(512), 0, 0, C, 16u); }
    CK(hipDeviceSynchronize());
    for (int round = 0; round < 2; round++)
        for (const Variant& v : VARIANTS) { printf("[%-14s] ", v.id); v.fn(); }
    return 0;
}
