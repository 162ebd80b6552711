// Internal declarations shared by the translation units of libministark_hip.so (ms_core / ms_ntt / ms_hash / ms_stage /
// ms_eval / ms_deep / ms_comm .cpp): the context, the plan object, error plumbing, pooled scratch.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ministark_hip.h"
#include "gl.h"
#include "fp252.h"
#include "ntt_kernels.h"

using msntt::MAXC;

int fail(int code, const char* fmt, ...);
int field_words(int field, unsigned* V);
#define HIPCHK(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(MS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define MSCHK(expr)                \
    do {                           \
        int r_ = (expr);           \
        if (r_ != MS_OK) return r_; \
    } while (0)


// what the specialised constraint kernels cost and where they came from (eval_jit.h, jit_cache.h; ms_eval_jit_stats)
struct JitStats { uint64_t compiled = 0, from_disk = 0, failures = 0, damaged_entries = 0; double compile_ms = 0, load_ms = 0; };
JitStats& jit_process_stats();              // the process-wide totals (ms_eval_jit_check has no context)

struct ms_ntt_plan;
struct PlanKey { unsigned V, log_n; bool inverse; uint64_t h; };
struct ms_ctx {
    int device = 0;
    std::vector<std::pair<PlanKey, ms_ntt_plan*>> plan_cache;   // plans used by the fused entry points
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;            // second half of a column group in plan_run (created on first use)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    // columns are processed in groups of about this size (= the scratch buffer).  Measured at 2^24: one column per
    // launch (4096 tiles = exactly two rounds of resident workgroups) loses 5 % of kernel time to the launch tail
    // against 8 columns per launch; the passes are VALU-bound, so there is no cache-locality argument for small groups.
    size_t group_bytes = (size_t)1 << 30;
    std::mutex mu;
    // freed device blocks, by size: GpuVec churn (clone / resize in src/matrix.rs:155-208, the LdeCache of
    // src/eval_gpu.rs:857-898) must not cost a hipMalloc + hipFree pair per column.  All work is ordered on
    // one stream, so a block can be handed out again without synchronising.
    std::multimap<size_t, void*> pool;
    size_t pool_bytes = 0, pool_cap = (size_t)96 << 30;
    std::map<void*, size_t> live;            // size of every block handed out by ms_alloc
    // pinned staging ring for the small host arrays entry points take (query positions, digest indices): the copy to the
    // device is then truly asynchronous and the call does not drain the stream (stage_upload)
    char* stage = nullptr;
    size_t stage_bytes = 0, stage_off = 0;
    // pinned landing buffer of the small downloads (roots, out-of-domain values, the gathers of an opening): a copy into pinned memory
    // + memcpy is 12 us where the same copy into the caller's pageable buffer is 20 us (scripts/download_latency.hip)
    char* land = nullptr;
    std::mutex land_mu;
    void* comm = nullptr;                    // ncclComm_t once ms_comm_init has run
    int comm_rank = 0, comm_size = 1;
    void* prog_buf = nullptr;                // device copy of the current constraint program + constants
    size_t prog_bytes = 0;
    // constraint programs compiled to specialised kernels (eval_jit.h), by hash of the generated source;
    // nullptr = compilation failed once, use the interpreter
    std::map<std::string, hipFunction_t> jit_cache;   // keyed by the full source text, not a hash of it
    std::vector<hipModule_t> jit_modules;
    JitStats jit_stats;
    // optional per-launch timing (ms_profile_*): hipEvent pairs around every kernel launch
    bool profiling = false;
    struct ProfRec { const char* name; hipEvent_t e0, e1; double bytes; };
    std::vector<ProfRec> prof;
};

// RAII: brackets one kernel launch with events on the context's stream when profiling is on
struct ProfScope {
    ms_ctx* ctx; hipEvent_t e0 = nullptr, e1 = nullptr; const char* name; double bytes;
    ProfScope(ms_ctx* c, const char* nm, double algorithmic_bytes) : ctx(c), name(nm), bytes(algorithmic_bytes) {
        if (!ctx->profiling) return;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, ctx->stream);
    }
    ~ProfScope() {
        if (!ctx->profiling) return;
        (void)hipEventRecord(e1, ctx->stream);
        ctx->prof.push_back({name, e0, e1, bytes});
    }
};


int ctx_scratch(ms_ctx* ctx, size_t bytes, void** out);
int pool_alloc(ms_ctx* ctx, size_t bytes, void** d_ptr);          // the caller holds ctx->mu
int pool_free(ms_ctx* ctx, void* d_ptr);
int stage_upload(ms_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);   // the caller holds ctx->mu; h_src may be reused at once
// Pooled temporaries of an entry point, returned to the pool on EVERY exit path (an early return through
// MSCHK / HIPCHK used to strand them in ctx->live until ms_ctx_destroy).  PoolGuard goes through the locking
// public calls and must outlive the function's lock scope; LockedPoolGuard is for code that already holds
// ctx->mu and must be declared after the lock_guard (so that it is destroyed first).
struct PoolGuard {
    ms_ctx* ctx; std::vector<void*> blocks;
    explicit PoolGuard(ms_ctx* c) : ctx(c) {}
    int alloc(size_t bytes, void** p) { const int rc = ms_alloc(ctx, bytes, p); if (rc == MS_OK) blocks.push_back(*p); return rc; }
    ~PoolGuard() { for (void* b : blocks) (void)ms_free(ctx, b); }
    PoolGuard(const PoolGuard&) = delete; PoolGuard& operator=(const PoolGuard&) = delete;
};
struct LockedPoolGuard {
    ms_ctx* ctx; std::vector<void*> blocks;
    explicit LockedPoolGuard(ms_ctx* c) : ctx(c) {}
    int alloc(size_t bytes, void** p) { const int rc = pool_alloc(ctx, bytes, p); if (rc == MS_OK) blocks.push_back(*p); return rc; }
    ~LockedPoolGuard() { for (void* b : blocks) (void)pool_free(ctx, b); }
    LockedPoolGuard(const LockedPoolGuard&) = delete; LockedPoolGuard& operator=(const LockedPoolGuard&) = delete;
};

int stage_view(ms_ctx* ctx, const void* h_src, size_t bytes, const void** d_view, LockedPoolGuard& pooled);   // the caller holds ctx->mu

struct ms_ntt_plan {
    ms_ctx* ctx = nullptr;
    // ms_ntt_plan_create hands out a HANDLE: a copy of the context's cached plan for (field, size, direction, offset)
    // with its own queue; `base` is that cached plan (owner of every table), `refs` counts the handles on a cached plan
    // (the cache never evicts a plan in use).  Building the tables of a 2^22-point plan on the host and uploading them
    // cost 1.3 ms per GpuFft / GpuIfft object -- 2.7 ms of the 16.7 ms prover run -- before plans were shared.
    ms_ntt_plan* base = nullptr;
    int refs = 0;
    unsigned V = 1, log_n = 0;
    bool inverse = false, coset = false;
    // small path (log_n < 12)
    bool small = false;
    bool tiny_fused = false;            // Fp, 2^9 .. 2^11 points: the tables of the (256, n / 256) plan are there too (ntt_fused_tiny)
    uint64_t *d_tw = nullptr, *d_scale_in = nullptr, *d_scale_out = nullptr;
    // multi-pass path
    int npass = 0;
    unsigned lr[4] = {0, 0, 0, 0};      // log2 radix per pass
    unsigned log_s[4] = {0, 0, 0, 0};   // log2 element stride of the pass's digit
    unsigned nfields[4] = {0, 0, 0, 0};
    msntt::DigitField fields[4][3];
    unsigned lo_bits = 0;
    uint64_t *d_tw_lo = nullptr, *d_tw_hi = nullptr, *d_aux_lo = nullptr, *d_aux_hi = nullptr, *d_gtab = nullptr;
    uint64_t* d_wr[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t scale_const = 0;
    int scale_mode = 0;                 // last pass: 0 none, 1 const, 2 table
    uint64_t* d_tables = nullptr;       // one allocation backing every table
    // limb-form passes (ntt2_kernels.h): plain tables of 4 pre-shifted copies per twiddle
    uint64_t* d_wr4[4] = {nullptr, nullptr, nullptr, nullptr};    // radix-256 passes: w_256^e
    uint64_t* d_twu4[4] = {nullptr, nullptr, nullptr, nullptr};   // middle passes: per-tile factor [U][k]
    uint64_t *d_sc4 = nullptr, *d_g4 = nullptr, *d_scu4 = nullptr;
    // three-pass plans (last radix 256 since round 3): pass 1's inter-pass factor from wave-uniform tables, the per-lane
    // remainder applied by pass 2 on its loads (ntt2_first_pass<.., UNI>, ntt2_mid_pass<.., LOADQ>)
    bool uni = false;
    uint64_t *d_tin4 = nullptr, *d_tout4 = nullptr;
    // two-pass coset LDE (lde2_kernels.h), built on first use on the forward plan of the LDE domain: per blow-up
    // [gpl | aux | t2] in one allocation
    struct Lde2 { unsigned log_b = 0; uint64_t *d = nullptr, *gpl = nullptr, *aux = nullptr, *t2 = nullptr, *tin4 = nullptr, *tout4 = nullptr, *c3 = nullptr; };
    std::vector<Lde2> lde2;
    uint64_t* d_oscale = nullptr;       // inverse plans of 2^18 points: n^-1 h^-k, k < n (Montgomery), for the two-pass route (built at first use)
    uint64_t offset_canon = 1;          // the coset offset h (canonical)
    std::vector<void*> queue;
    // Fp252 path (V == 4): plain radix-2 plan, see fp252_kernels.h
    bool is252 = false;
    uint64_t *d252_tw_lo = nullptr, *d252_tw_hi = nullptr, *d252_sc_lo = nullptr, *d252_sc_hi = nullptr;
    int scale_in252 = 0, scale_out252 = 0;
    uint64_t off252[4] = {0, 0, 0, 0};  // the coset offset itself: cache lookups compare it, not just its hash
    // tiled passes of fp252_ntt_kernels.h (2^11 <= n <= 2^30): number of passes (0 = radix-2 sequence only), digit sizes,
    // per-pass tables w_R^e (e < R/2)
    int np252 = 0;
    unsigned lr252[3] = {0, 0, 0};
    uint64_t* d252_twr[3] = {nullptr, nullptr, nullptr};
};


// ms_ntt.cpp
int ctx_plan(ms_ctx* ctx, unsigned V, unsigned log_n, bool inverse, uint64_t h, ms_ntt_plan** out);     // the context's cached plan
int plan252_cached(ms_ctx* ctx, unsigned log_n, bool inverse, const f252::E& h, ms_ntt_plan** out);
int plan_run(ms_ntt_plan* p, const void* const* src, void* const* dst, unsigned ncols, unsigned valid_rows, bool bitrev_out = false);
int bit_reverse_run(ms_ctx* ctx, unsigned V, unsigned log_n, const void* const* src, void* const* dst, unsigned ncols);
unsigned stream_grid(size_t n);
void plans_release_ctx(ms_ctx* ctx);                 // ms_ctx_destroy: plans still held by the caller go with their context
int plan_free_cached(ms_ntt_plan* plan);             // the context's own (cached) plans
