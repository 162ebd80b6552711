// Host side of libministark_hip.so: the C ABI declared in include/ministark_hip.h.
// Compiled as HIP for gfx950 (see ministark_amd/build.py).  Mirrors the reference's
// Planner / GpuFft / GpuIfft / *Stage front-ends (gpu/src/plan.rs, gpu/src/stage.rs).
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
#include "ntt_kernels.h"
#include "ntt2_kernels.h"
#include "sha256_kernels.h"
#include "stage_kernels.h"
#include "fri_kernels.h"
#include "eval_kernels.h"
#include "scan_kernels.h"
#include "eval_opt.h"
#ifndef MS_NO_JIT
#include "eval_jit.h"
#endif
#include "fp252_kernels.h"
#include "fp252_ntt_kernels.h"
#include "lde2_kernels.h"
#include "rpo_kernels.h"
#include "deep_kernels.h"

using msntt::MAXC;

// ---------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
static int fail(int code, const char* fmt, ...) {
    char buf[4096];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
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

extern "C" const char* ms_last_error(void) { return g_last_error.c_str(); }
extern "C" size_t ms_field_bytes(int field) {
    switch (field) {
    case MS_GOLDILOCKS_FP: return 8;
    case MS_GOLDILOCKS_FQ3: return 24;
    case MS_STARK252_FP: return 32;
    default: return 0;
    }
}
static int field_words(int field, unsigned* V) {
    if (field == MS_GOLDILOCKS_FP) { *V = 1; return MS_OK; }
    if (field == MS_GOLDILOCKS_FQ3) { *V = 3; return MS_OK; }
    if (field == MS_STARK252_FP) { *V = 4; return MS_OK; }
    return fail(MS_ERR_INVALID, "unknown field id %d", field);
}

// ---------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------
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
    void* comm = nullptr;                    // ncclComm_t once ms_comm_init has run
    int comm_rank = 0, comm_size = 1;
    void* prog_buf = nullptr;                // device copy of the current constraint program + constants
    size_t prog_bytes = 0;
    // constraint programs compiled to specialised kernels (eval_jit.h), by hash of the generated source;
    // nullptr = compilation failed once, use the interpreter
    std::map<std::string, hipFunction_t> jit_cache;   // keyed by the full source text, not a hash of it
    std::vector<hipModule_t> jit_modules;
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

static int ctx_scratch(ms_ctx* ctx, size_t bytes, void** out) {
    if (ctx->scratch_bytes < bytes) {
        if (ctx->scratch) {
            HIPCHK(hipStreamSynchronize(ctx->stream));
            HIPCHK(hipFree(ctx->scratch));
            ctx->scratch = nullptr;
            ctx->scratch_bytes = 0;
        }
        hipError_t e = hipMalloc(&ctx->scratch, bytes);
        if (e != hipSuccess) return fail(MS_ERR_NOMEM, "scratch allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        ctx->scratch_bytes = bytes;
    }
    *out = ctx->scratch;
    return MS_OK;
}

extern "C" int ms_ctx_create(int device, ms_ctx** out) {
    if (!out) return fail(MS_ERR_INVALID, "ms_ctx_create: out is null");
    int count = 0;
    HIPCHK(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(MS_ERR_INVALID, "device %d out of range (%d visible)", device, count);
    HIPCHK(hipSetDevice(device));
    ms_ctx* ctx = new ms_ctx();
    ctx->device = device;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return fail(MS_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    if (const char* g = getenv("MS_NTT_GROUP_BYTES")) ctx->group_bytes = (size_t)strtoull(g, nullptr, 10);
    if (const char* g = getenv("MS_POOL_BYTES")) ctx->pool_cap = (size_t)strtoull(g, nullptr, 10);
    *out = ctx;
    return MS_OK;
}
extern "C" int ms_ntt_plan_destroy(ms_ntt_plan* plan);
extern "C" int ms_comm_destroy(ms_ctx* ctx);
extern "C" int ms_ctx_destroy(ms_ctx* ctx) {
    if (!ctx) return MS_OK;
    (void)hipStreamSynchronize(ctx->stream);
    (void)ms_comm_destroy(ctx);
    for (auto& kv : ctx->plan_cache) ms_ntt_plan_destroy(kv.second);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    for (auto& kv : ctx->pool) (void)hipFree(kv.second);
    if (ctx->prog_buf) (void)hipFree(ctx->prog_buf);
    for (auto m : ctx->jit_modules) (void)hipModuleUnload(m);
    if (ctx->stream2) { (void)hipStreamDestroy(ctx->stream2); (void)hipEventDestroy(ctx->ev_fork); (void)hipEventDestroy(ctx->ev_join); }
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return MS_OK;
}
extern "C" int ms_sync(ms_ctx* ctx) {
    if (!ctx) return fail(MS_ERR_INVALID, "null context");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return MS_OK;
}
extern "C" void* ms_ctx_stream(ms_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" int ms_profile_enable(ms_ctx* ctx, int on) {
    if (!ctx) return fail(MS_ERR_INVALID, "null context");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (auto& r : ctx->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    ctx->prof.clear();
    ctx->profiling = on != 0;
    return MS_OK;
}
// Writes one line per kernel name: "name calls total_us algorithmic_bytes_per_call\n".
extern "C" int ms_profile_read(ms_ctx* ctx, char* buf, size_t cap) {
    if (!ctx || !buf || !cap) return fail(MS_ERR_INVALID, "ms_profile_read: null argument");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    struct Acc { const char* name; unsigned calls; double us; double bytes; };
    std::vector<Acc> acc;
    for (auto& r : ctx->prof) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, r.e0, r.e1));
        bool found = false;
        for (auto& a : acc) if (!strcmp(a.name, r.name)) { a.calls++; a.us += ms * 1e3; a.bytes += r.bytes; found = true; break; }
        if (!found) acc.push_back({r.name, 1, ms * 1e3, r.bytes});
    }
    std::string out;
    char line[256];
    for (auto& a : acc) { snprintf(line, sizeof line, "%s %u %.3f %.0f\n", a.name, a.calls, a.us, a.bytes / a.calls); out += line; }
    if (out.size() + 1 > cap) return fail(MS_ERR_INVALID, "profile buffer too small (%zu needed)", out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    return MS_OK;
}

// pooled blocks; the caller holds ctx->mu.  A freed block may be handed out again at once: every
// kernel runs on ctx->stream, so the next user queues behind the last one.
static int pool_alloc(ms_ctx* ctx, size_t bytes, void** d_ptr) {
    bytes = (bytes + 255) & ~(size_t)255;
    auto it = ctx->pool.find(bytes);
    if (it != ctx->pool.end()) {
        *d_ptr = it->second;
        ctx->pool_bytes -= bytes;
        ctx->pool.erase(it);
    } else {
        hipError_t e = hipMalloc(d_ptr, bytes);
        if (e != hipSuccess && !ctx->pool.empty()) {          // give cached blocks back and retry
            (void)hipStreamSynchronize(ctx->stream);
            for (auto& kv : ctx->pool) (void)hipFree(kv.second);
            ctx->pool.clear(); ctx->pool_bytes = 0;
            e = hipMalloc(d_ptr, bytes);
        }
        if (e != hipSuccess) return fail(MS_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    }
    ctx->live[*d_ptr] = bytes;
    return MS_OK;
}
static int pool_free(ms_ctx* ctx, void* d_ptr) {
    if (!d_ptr) return MS_OK;
    auto it = ctx->live.find(d_ptr);
    if (it == ctx->live.end()) return fail(MS_ERR_INVALID, "ms_free: pointer was not returned by ms_alloc on this context");
    const size_t bytes = it->second;
    ctx->live.erase(it);
    if (ctx->pool_bytes + bytes <= ctx->pool_cap) { ctx->pool.insert({bytes, d_ptr}); ctx->pool_bytes += bytes; return MS_OK; }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(d_ptr));
    return MS_OK;
}
extern "C" int ms_alloc(ms_ctx* ctx, size_t bytes, void** d_ptr) {
    if (!ctx || !d_ptr) return fail(MS_ERR_INVALID, "ms_alloc: null argument");
    HIPCHK(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> lk(ctx->mu);
    return pool_alloc(ctx, bytes, d_ptr);
}
extern "C" int ms_free(ms_ctx* ctx, void* d_ptr) {
    if (!ctx) return fail(MS_ERR_INVALID, "null context");
    std::lock_guard<std::mutex> lk(ctx->mu);
    return pool_free(ctx, d_ptr);
}
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
extern "C" int ms_copy(ms_ctx* ctx, void* d_dst, const void* d_src, size_t bytes) {
    if (!ctx || (bytes && (!d_dst || !d_src))) return fail(MS_ERR_INVALID, "ms_copy: null argument");
    HIPCHK(hipSetDevice(ctx->device));
    if (bytes && d_dst != d_src) HIPCHK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return MS_OK;
}
extern "C" int ms_upload(ms_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    if (!ctx) return fail(MS_ERR_INVALID, "null context");
    HIPCHK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return MS_OK;
}
extern "C" int ms_download(ms_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
    if (!ctx) return fail(MS_ERR_INVALID, "null context");
    HIPCHK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return MS_OK;
}

// ---------------------------------------------------------------------------------------
// NTT plans
// ---------------------------------------------------------------------------------------
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
    // three-pass plans with a last radix >= 64: pass 1's inter-pass factor from wave-uniform tables, the per-lane
    // remainder applied by pass 2 on its loads (ntt2_first_pass<.., UNI>, ntt2_mid_pass<.., LOADQ>)
    bool uni = false;
    uint64_t *d_tin4 = nullptr, *d_tout4 = nullptr;
    // two-pass coset LDE (lde2_kernels.h), built on first use on the forward plan of the LDE domain: per blow-up
    // [gpl | aux | t2] in one allocation
    struct Lde2 { unsigned log_b = 0; uint64_t *d = nullptr, *gpl = nullptr, *aux = nullptr, *t2 = nullptr, *tin4 = nullptr, *tout4 = nullptr; };
    std::vector<Lde2> lde2;
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

static void powers(std::vector<uint64_t>& out, size_t count, uint64_t base, uint64_t first = 1) {
    out.resize(count);
    uint64_t x = first;
    for (size_t i = 0; i < count; i++) { out[i] = x; x = gl::mul(x, base); }
}

static int plan_build(ms_ctx* ctx, unsigned V, unsigned log_n, bool inverse, uint64_t h, ms_ntt_plan** out);
static int ctx_plan(ms_ctx* ctx, unsigned V, unsigned log_n, bool inverse, uint64_t h, ms_ntt_plan** out);
static int plan_build252(ms_ctx* ctx, unsigned log_n, bool inverse, const void* h_offset, const void* h_group_gen, ms_ntt_plan** out);

extern "C" int ms_ntt_plan_create(ms_ctx* ctx, int field, unsigned log_n, int inverse, const void* h_offset,
                                  const void* h_group_gen, ms_ntt_plan** out) {
    if (!ctx || !out) return fail(MS_ERR_INVALID, "ms_ntt_plan_create: null argument");
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    if (V == 4) return plan_build252(ctx, log_n, inverse != 0, h_offset, h_group_gen, out);
    if (log_n > 32) return fail(MS_ERR_INVALID, "log_n = %u exceeds the field's two-adicity (32)", log_n);
    if (h_group_gen) {
        uint64_t g_m;
        memcpy(&g_m, h_group_gen, 8);
        if (gl::from_mont(g_m) != gl::root_of_unity(log_n))
            return fail(MS_ERR_UNSUPPORTED, "group_gen is not arkworks' get_root_of_unity(2^%u)", log_n);
    }
    uint64_t h = 1;
    if (h_offset) { uint64_t h_m; memcpy(&h_m, h_offset, 8); h = gl::from_mont(h_m); }
    if (h == 0) return fail(MS_ERR_INVALID, "coset offset must be non-zero");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ms_ntt_plan* base = nullptr;
    MSCHK(ctx_plan(ctx, V, log_n, inverse != 0, h, &base));
    ms_ntt_plan* handle = new ms_ntt_plan(*base);
    handle->base = base; handle->refs = 0; handle->queue.clear(); handle->lde2.clear();
    base->refs++;
    *out = handle;
    return MS_OK;
}

// ---- Fp252 plans ------------------------------------------------------------------------
static void powers252(std::vector<uint64_t>& out, size_t count, f252::E base, f252::E first) {
    out.resize(count * 4);
    f252::E x = first;
    for (size_t i = 0; i < count; i++) { memcpy(&out[4 * i], x.l, 32); x = f252::mul(x, base); }
}
static int plan_build252(ms_ctx* ctx, unsigned log_n, bool inverse, const void* h_offset, const void* h_group_gen, ms_ntt_plan** out) {
    if (log_n > 40) return fail(MS_ERR_INVALID, "log_n = %u too large", log_n);
    HIPCHK(hipSetDevice(ctx->device));
    const f252::E gen = f252::root_of_unity(log_n);
    if (h_group_gen) {
        f252::E g; memcpy(g.l, h_group_gen, 32);
        if (!f252::eq(g, gen)) return fail(MS_ERR_UNSUPPORTED, "group_gen is not arkworks' get_root_of_unity(2^%u)", log_n);
    }
    f252::E h = f252::one();
    if (h_offset) memcpy(h.l, h_offset, 32);
    if (f252::is_zero(h) || f252::geq_p(h)) return fail(MS_ERR_INVALID, "coset offset must be a non-zero canonical element");
    const bool coset = !f252::eq(h, f252::one());
    ms_ntt_plan* p = new ms_ntt_plan();
    p->ctx = ctx; p->V = 4; p->log_n = log_n; p->inverse = inverse; p->coset = coset; p->is252 = true;
    memcpy(p->off252, h.l, 32);
    const size_t n = (size_t)1 << log_n;
    const f252::E w = inverse ? f252::inv(gen) : gen;
    // one-level tables up to 2^21 points: every twiddle / scale factor is a single 32-byte load.  (A two-level
    // lookup costs a second Montgomery product per butterfly, and the product -- ~440 VALU instructions -- is
    // what bounds this field.)  Larger domains split the exponent at 2^21.
    p->lo_bits = std::min(21u, log_n);
    std::vector<uint64_t> host, t;
    auto append = [&](const std::vector<uint64_t>& v) { size_t off = host.size(); host.insert(host.end(), v.begin(), v.end()); return off; };
    powers252(t, (size_t)1 << p->lo_bits, w, f252::one()); const size_t o_lo = append(t);
    powers252(t, std::max<size_t>(n >> p->lo_bits, 1), f252::pow_u64(w, (uint64_t)1 << p->lo_bits), f252::one()); const size_t o_hi = append(t);
    size_t o_slo = 0, o_shi = 0;
    const bool scale = inverse || coset;
    if (scale) {
        f252::E g = inverse ? f252::inv(h) : h, c = f252::one();
        if (inverse) { f252::E nn = f252::to_mont(f252::E{{(uint64_t)n, 0, 0, 0}}); c = f252::inv(nn); }
        powers252(t, (size_t)1 << p->lo_bits, g, c); o_slo = append(t);
        powers252(t, std::max<size_t>(n >> p->lo_bits, 1), f252::pow_u64(g, (uint64_t)1 << p->lo_bits), f252::one()); o_shi = append(t);
        if (inverse) p->scale_out252 = 1; else p->scale_in252 = 1;
    }
    size_t o_twr[3] = {0, 0, 0};
    if (log_n >= (unsigned)ms252::TILE_LOG && log_n <= 30) {
        p->np252 = log_n <= 20 ? 2 : 3;
        // MS_NTT252_PASSES=3 forces the three-pass split from 2^17 points on (tests: the emulator cannot hold 2^21 points)
        if (const char* e = getenv("MS_NTT252_PASSES")) if (atoi(e) == 3 && log_n >= 17) p->np252 = 3;
        for (int q = 0; q < p->np252; q++) p->lr252[q] = log_n / p->np252 + ((unsigned)q < log_n % p->np252 ? 1 : 0);
        for (int q = 0; q < p->np252; q++) {
            powers252(t, (size_t)1 << (p->lr252[q] - 1), f252::pow_u64(w, (uint64_t)n >> p->lr252[q]), f252::one());
            o_twr[q] = append(t);
        }
    }
    if (hipMalloc(&p->d_tables, host.size() * 8) != hipSuccess) { delete p; return fail(MS_ERR_NOMEM, "Fp252 plan tables"); }
    if (hipMemcpy(p->d_tables, host.data(), host.size() * 8, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(p->d_tables); delete p; return fail(MS_ERR_HIP, "Fp252 table upload"); }
    p->d252_tw_lo = p->d_tables + o_lo; p->d252_tw_hi = p->d_tables + o_hi;
    for (int q = 0; q < p->np252; q++) p->d252_twr[q] = p->d_tables + o_twr[q];
    if (scale) { p->d252_sc_lo = p->d_tables + o_slo; p->d252_sc_hi = p->d_tables + o_shi; }
    *out = p;
    return MS_OK;
}

// Plans owned by the context, reused by the fused entry points (ms_lde, ms_fri_fold, ...).  The cache is bounded:
// most recently used at the back, and beyond PLAN_CACHE_MAX entries the least recently used plan is destroyed
// (a prover that varies sizes / offsets -- FRI layers, periodic-column cosets -- would otherwise accumulate twiddle
// tables until ms_ctx_destroy).  One call uses at most a handful of plans, so a plan handed out in a call cannot
// be evicted by the same call.
static constexpr size_t PLAN_CACHE_MAX = 32;
static ms_ntt_plan* plan_cache_find(ms_ctx* ctx, unsigned V, unsigned log_n, bool inverse, uint64_t h, const uint64_t* off252 = nullptr) {
    auto& pc = ctx->plan_cache;
    for (size_t i = 0; i < pc.size(); i++) {
        const PlanKey& k = pc[i].first;
        if (k.V != V || k.log_n != log_n || k.inverse != inverse || k.h != h) continue;
        if (off252 && memcmp(pc[i].second->off252, off252, 32) != 0) continue;       // same hash, different offset
        auto hit = pc[i];
        pc.erase(pc.begin() + (long)i);
        pc.push_back(hit);
        return hit.second;
    }
    return nullptr;
}
static void plan_cache_insert(ms_ctx* ctx, const PlanKey& key, ms_ntt_plan* plan) {
    ctx->plan_cache.push_back({key, plan});
    while (ctx->plan_cache.size() > PLAN_CACHE_MAX) {
        size_t victim = 0;
        while (victim + 1 < ctx->plan_cache.size() && ctx->plan_cache[victim].second->refs > 0) victim++;   // least recently used plan without handles
        if (victim + 1 >= ctx->plan_cache.size()) break;       // everything older than the new plan is in use: let the cache grow
        ms_ntt_plan* old = ctx->plan_cache[victim].second;
        ctx->plan_cache.erase(ctx->plan_cache.begin() + (long)victim);
        (void)ms_ntt_plan_destroy(old);                        // synchronises the stream before freeing the tables
    }
}
static int plan252_cached(ms_ctx* ctx, unsigned log_n, bool inverse, const f252::E& h, ms_ntt_plan** out);   // Fp252: keyed by the offset itself
static int ctx_plan(ms_ctx* ctx, unsigned V, unsigned log_n, bool inverse, uint64_t h, ms_ntt_plan** out) {
    if ((*out = plan_cache_find(ctx, V, log_n, inverse, h)) != nullptr) return MS_OK;
    MSCHK(plan_build(ctx, V, log_n, inverse, h, out));
    plan_cache_insert(ctx, PlanKey{V, log_n, inverse, h}, *out);
    return MS_OK;
}

static int plan_build(ms_ctx* ctx, unsigned V, unsigned log_n, bool inverse_b, uint64_t h, ms_ntt_plan** out) {
    const int inverse = inverse_b ? 1 : 0;
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t gen = gl::root_of_unity(log_n);           // plain, arkworks get_root_of_unity
    ms_ntt_plan* p = new ms_ntt_plan();
    p->ctx = ctx; p->V = V; p->log_n = log_n; p->inverse = inverse != 0; p->coset = (h != 1); p->offset_canon = h;
    const size_t n = (size_t)1 << log_n;
    const uint64_t w = p->inverse ? gl::inv(gen) : gen;       // transform root
    const uint64_t hinv = gl::inv(h);
    const uint64_t ninv = gl::inv((uint64_t)(n % gl::P));

    std::vector<uint64_t> host;                                // all tables, concatenated
    auto append = [&](const std::vector<uint64_t>& t) { size_t off = host.size(); host.insert(host.end(), t.begin(), t.end()); return off; };
    std::vector<uint64_t> t;
    size_t off_tw = 0, off_si = 0, off_so = 0, off_lo = 0, off_hi = 0, off_alo = 0, off_ahi = 0, off_g = 0, off_wr[4] = {0, 0, 0, 0};
    bool has_si = false, has_so = false, has_aux = false, has_g = false;

    if (log_n < 12) {
        p->small = true;
        powers(t, std::max<size_t>(n / 2, 1), w); off_tw = append(t);
        if (!p->inverse && p->coset) { powers(t, n, h); off_si = append(t); has_si = true; }
        if (p->inverse) { powers(t, n, hinv, ninv); off_so = append(t); has_so = true; }
    } else {
        // radix decomposition: R1 = 256, the rest split as evenly as possible into radices 16..256
        const unsigned rest = log_n - 8;
        const int extra = (int)((rest + 7) / 8);
        p->npass = 1 + extra;
        p->lr[0] = 8;
        for (int i = 0; i < extra; i++) p->lr[1 + i] = rest / extra + ((unsigned)i < rest % extra ? 1 : 0);
        // three passes: prefer (8, 8, rest - 8) to an even split whenever the last radix is still >= 16 -- two of the
        // three passes are then limb-form radix-256 passes (ntt2_kernels.h), e.g. 2^20 = 256 * 256 * 16 instead of 256 * 64 * 64
        if (extra == 2 && rest >= 12 && rest <= 16) { p->lr[1] = 8; p->lr[2] = rest - 8; }
        unsigned acc = 0;
        for (int q = 0; q < p->npass; q++) { p->log_s[q] = acc; acc += p->lr[q]; }
        // digit fields.  pass 1 maps j' = (j2..jm) [jm least significant] to layout (jm..j2) [j2 least]
        {
            unsigned nf = 0, in_shift = 0;
            for (int q = p->npass - 1; q >= 1; q--) {            // jm first (least significant of j')
                unsigned out_shift = 0;
                for (int r = 1; r < q; r++) out_shift += p->lr[r];
                p->fields[0][nf++] = {in_shift, out_shift, (1u << p->lr[q]) - 1};
                in_shift += p->lr[q];
            }
            p->nfields[0] = nf;
        }
        // pass q (0-based, 1 <= q < npass-1): U = (jm..j_{q+2}) [j_{q+2} least significant in U]
        //   -> j' = (j_{q+2}, ..., jm) [jm least significant]
        for (int q = 1; q < p->npass - 1; q++) {
            unsigned nf = 0, in_shift = 0;
            for (int r = q + 1; r < p->npass; r++) {             // r = digit index (0-based) above q
                unsigned out_shift = 0;
                for (int r2 = r + 1; r2 < p->npass; r2++) out_shift += p->lr[r2];
                p->fields[q][nf++] = {in_shift, out_shift, (1u << p->lr[r]) - 1};
                in_shift += p->lr[r];
            }
            p->nfields[q] = nf;
        }
        p->lo_bits = std::min(12u, log_n);
        powers(t, (size_t)1 << p->lo_bits, w); off_lo = append(t);
        powers(t, n >> p->lo_bits, gl::pow(w, (uint64_t)1 << p->lo_bits)); off_hi = append(t);
        for (int q = 0; q < p->npass; q++) {
            powers(t, (size_t)1 << p->lr[q], gl::pow(w, (uint64_t)n >> p->lr[q])); off_wr[q] = append(t);
        }
        if (!p->inverse && p->coset) {
            powers(t, (size_t)1 << p->lo_bits, h); off_alo = append(t);
            powers(t, std::max<size_t>((n >> 8) >> p->lo_bits, 1), gl::pow(h, (uint64_t)1 << p->lo_bits)); off_ahi = append(t);
            powers(t, 256, gl::pow(h, (uint64_t)(n >> 8))); off_g = append(t);
            has_aux = has_g = true;
        }
        if (p->inverse) {
            if (!p->coset) { p->scale_mode = 1; p->scale_const = ninv; }
            else {
                p->scale_mode = 2;
                powers(t, (size_t)1 << p->lo_bits, hinv, ninv); off_alo = append(t);
                powers(t, n >> p->lo_bits, gl::pow(hinv, (uint64_t)1 << p->lo_bits)); off_ahi = append(t);
                has_aux = true;
            }
        }
    }
    // device tables are in Montgomery form: gld::mmul(data, w * 2^64) = data * w
    for (auto& v : host) v = gl::to_mont(v);
    // ... except the tables of the limb-form passes (ntt2_kernels.h): plain residues, four copies
    // {w, w 2^24, w 2^48, w 2^72} per twiddle, appended after the conversion
    size_t off_wr4[4] = {0, 0, 0, 0}, off_twu4[4] = {0, 0, 0, 0}, off_sc4 = 0, off_gp = 0, off_tin4 = 0, off_tout4 = 0;
    bool has_wr4[4] = {false, false, false, false}, has_twu4[4] = {false, false, false, false}, has_gp = false, has_scu4 = false;
    size_t off_scu4 = 0;
    p->uni = !p->small && p->npass == 3 && p->lr[1] == 8 && p->lr[2] >= 6 && (n * V) % msntt2::TILE == 0;
    if (!p->small) {
        const uint64_t sh[4] = {1, (uint64_t)1 << 24, (uint64_t)1 << 48, gl::pow(2, 72)};
        auto append4 = [&](const std::vector<uint64_t>& plain) {
            const size_t off = host.size();
            host.reserve(off + 4 * plain.size());
            for (uint64_t v : plain) for (int i = 0; i < 4; i++) host.push_back(gl::mul(v, sh[i]));
            return off;
        };
        for (int q = 0; q < p->npass; q++) {
            if (p->lr[q] != 8) continue;
            powers(t, 256, gl::pow(w, (uint64_t)n >> 8)); off_wr4[q] = append4(t); has_wr4[q] = true;
            if (q >= 1 && q < p->npass - 1) {
                // w_U^k = w_n^((rev(U) k) << log_s): the factor ntt_mid_pass builds per tile in LDS (twl[])
                const size_t nU = n >> (8 + p->log_s[q]);
                const uint64_t ws = gl::pow(w, (uint64_t)1 << p->log_s[q]);
                t.resize(nU * 256);
                for (size_t U = 0; U < nU; U++) {
                    unsigned rU = 0;
                    for (unsigned f = 0; f < p->nfields[q]; f++)
                        rU |= (((unsigned)U >> p->fields[q][f].in_shift) & p->fields[q][f].mask) << p->fields[q][f].out_shift;
                    const uint64_t wu = gl::pow(ws, rU);
                    // UNI plans: pass 2 also carries h^j3 of the inter-pass factor (h w_n^k1)^(R3 j2 + j3), j3 = rev(U)
                    uint64_t x = (p->uni && q == 1 && !p->inverse && p->coset) ? gl::pow(h, rU) : 1;
                    for (unsigned k = 0; k < 256; k++) { t[U * 256 + k] = x; x = gl::mul(x, wu); }
                }
                off_twu4[q] = append4(t); has_twu4[q] = true;
            }
        }
        t.assign(1, ninv); off_sc4 = append4(t);
        if (p->scale_mode == 2 && p->lr[p->npass - 1] == 8) {      // inverse coset, last radix 256: h^-(k 2^log_s) per output row
            powers(t, 256, gl::pow(hinv, (uint64_t)1 << p->log_s[p->npass - 1])); off_scu4 = append4(t); has_scu4 = true;
        }
        if (p->uni) {
            // pass 1: tin4[j2][b][a'] = w_256^(a' b) w_n^(a' R3 j2) = w_n^(a' (b n/256 + R3 j2));
            //         tout4[j2][b'] = h^(R3 j2) w_n^(16 b' R3 j2)      (h = 1 unless this is a forward coset transform)
            const unsigned r3 = p->lr[2];
            const uint64_t hh = (!p->inverse && p->coset) ? h : 1;
            t.resize((size_t)256 * 256);
            for (unsigned j2 = 0; j2 < 256; j2++)
                for (unsigned b = 0; b < 16; b++) {
                    const uint64_t m = ((uint64_t)b * (n >> 8) + ((uint64_t)j2 << r3)) & (n - 1);
                    const uint64_t wm = gl::pow(w, m);
                    uint64_t x = 1;
                    for (unsigned a = 0; a < 16; a++) { t[((size_t)j2 * 16 + b) * 16 + a] = x; x = gl::mul(x, wm); }
                }
            off_tin4 = append4(t);
            t.resize((size_t)256 * 16);
            for (unsigned j2 = 0; j2 < 256; j2++) {
                const uint64_t wm = gl::pow(w, (((uint64_t)j2 << r3) * 16) & (n - 1));
                uint64_t x = gl::pow(hh, (uint64_t)j2 << r3);
                for (unsigned bp = 0; bp < 16; bp++) { t[(size_t)j2 * 16 + bp] = x; x = gl::mul(x, wm); }
            }
            off_tout4 = append4(t);
        }
        if (!p->inverse && p->coset) { powers(t, 256, gl::pow(h, (uint64_t)(n >> 8))); off_gp = append4(t); has_gp = true; }
    }
    p->scale_const = gl::to_mont(p->scale_const);
    hipError_t e = hipMalloc(&p->d_tables, host.size() * 8);
    if (e != hipSuccess) { delete p; return fail(MS_ERR_NOMEM, "plan tables (%zu bytes): %s", host.size() * 8, hipGetErrorString(e)); }
    e = hipMemcpy(p->d_tables, host.data(), host.size() * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(p->d_tables); delete p; return fail(MS_ERR_HIP, "plan table upload: %s", hipGetErrorString(e)); }
    if (p->small) {
        p->d_tw = p->d_tables + off_tw;
        p->d_scale_in = has_si ? p->d_tables + off_si : nullptr;
        p->d_scale_out = has_so ? p->d_tables + off_so : nullptr;
    } else {
        p->d_tw_lo = p->d_tables + off_lo; p->d_tw_hi = p->d_tables + off_hi;
        for (int q = 0; q < p->npass; q++) p->d_wr[q] = p->d_tables + off_wr[q];
        if (has_aux) { p->d_aux_lo = p->d_tables + off_alo; p->d_aux_hi = p->d_tables + off_ahi; }
        if (has_g) p->d_gtab = p->d_tables + off_g;
        for (int q = 0; q < p->npass; q++) {
            if (has_wr4[q]) p->d_wr4[q] = p->d_tables + off_wr4[q];
            if (has_twu4[q]) p->d_twu4[q] = p->d_tables + off_twu4[q];
        }
        p->d_sc4 = p->d_tables + off_sc4;
        if (has_scu4) p->d_scu4 = p->d_tables + off_scu4;
        if (p->uni) { p->d_tin4 = p->d_tables + off_tin4; p->d_tout4 = p->d_tables + off_tout4; }
        if (has_gp) p->d_g4 = p->d_tables + off_gp;
    }
    *out = p;
    return MS_OK;
}

extern "C" int ms_ntt_plan_destroy(ms_ntt_plan* plan) {
    if (!plan) return MS_OK;
    if (plan->base) {                                              // a handle: the cached plan keeps the tables
        { std::lock_guard<std::mutex> lk(plan->ctx->mu); plan->base->refs--; }
        delete plan;
        return MS_OK;
    }
    (void)hipStreamSynchronize(plan->ctx->stream);
    if (plan->d_tables) (void)hipFree(plan->d_tables);
    for (auto& l : plan->lde2) if (l.d) (void)hipFree(l.d);
    delete plan;
    return MS_OK;
}

template <int RB, bool INV, bool LAST>
static void launch_mid_scale(int scale, bool bitrev, dim3 grid, hipStream_t st, const msntt::PassParams& P) {
    if constexpr (LAST) {
        if (bitrev) {      // fused bit-reversed store (LDE): forward transforms only carry scale 0
            if (scale == 1) hipLaunchKernelGGL((msntt::ntt_mid_pass<RB, INV, true, 1, true>), grid, dim3(msntt::NT), 0, st, P);
            else if (scale == 2) hipLaunchKernelGGL((msntt::ntt_mid_pass<RB, INV, true, 2, true>), grid, dim3(msntt::NT), 0, st, P);
            else hipLaunchKernelGGL((msntt::ntt_mid_pass<RB, INV, true, 0, true>), grid, dim3(msntt::NT), 0, st, P);
            return;
        }
        if (scale == 1) hipLaunchKernelGGL((msntt::ntt_mid_pass<RB, INV, true, 1>), grid, dim3(msntt::NT), 0, st, P);
        else if (scale == 2) hipLaunchKernelGGL((msntt::ntt_mid_pass<RB, INV, true, 2>), grid, dim3(msntt::NT), 0, st, P);
        else hipLaunchKernelGGL((msntt::ntt_mid_pass<RB, INV, true, 0>), grid, dim3(msntt::NT), 0, st, P);
    } else {
        hipLaunchKernelGGL((msntt::ntt_mid_pass<RB, INV, false, 0>), grid, dim3(msntt::NT), 0, st, P);
    }
}
template <int RB>
static void launch_mid(bool inv, bool last, int scale, bool bitrev, dim3 grid, hipStream_t st, const msntt::PassParams& P) {
    if (inv) { if (last) launch_mid_scale<RB, true, true>(scale, bitrev, grid, st, P); else launch_mid_scale<RB, true, false>(scale, false, grid, st, P); }
    else     { if (last) launch_mid_scale<RB, false, true>(scale, bitrev, grid, st, P); else launch_mid_scale<RB, false, false>(scale, false, grid, st, P); }
}

static int bit_reverse_run(ms_ctx* ctx, unsigned V, unsigned log_n, const void* const* src, void* const* dst, unsigned ncols);
static unsigned stream_grid(size_t n);

// Tiled passes (fp252_ntt_kernels.h).  log_zero_ext: the source holds only the first n >> log_zero_ext elements, the rest
// of the domain is implicit zeros (needs 2^log_zero_ext <= R_0); bitrev_out: bit-reversed order, fused into the last pass.
static int plan_run252_tiled(ms_ntt_plan* p, const void* const* src, void* const* dst, unsigned ncols, unsigned log_zero_ext, bool bitrev_out) {
    ms_ctx* ctx = p->ctx;
    hipStream_t st = ctx->stream;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << p->log_n, col_bytes = n * 32;
    const int np = p->np252;
    if (log_zero_ext > p->lr252[0]) return fail(MS_ERR_INVALID, "internal: zero extension 2^%u beyond the first radix", log_zero_ext);
    unsigned group = (unsigned)std::max<size_t>(1, std::min<size_t>(msntt::MAXC, ctx->group_bytes / col_bytes));
    group = std::min(group, ncols);
    void* scratch = nullptr;
    MSCHK(ctx_scratch(ctx, (size_t)group * col_bytes, &scratch));
    static const char* const names[3] = {"ntt252_pass1", "ntt252_pass2", "ntt252_pass3"};
    if (getenv("MS_NTT_DEBUG"))
        fprintf(stderr, "[ms_ntt] Fp252 log_n=%u tiled: %d passes, radices 2^%u 2^%u 2^%u, zero extension 2^%u, bitrev %d\n", p->log_n, np,
                p->lr252[0], p->lr252[1], p->lr252[2], log_zero_ext, (int)bitrev_out);
    for (unsigned c0 = 0; c0 < ncols; c0 += group) {
        const unsigned nc = std::min(group, ncols - c0);
        unsigned done = 0;                                        // log2 of R_0 .. R_(q-1)
        for (int q = 0; q < np; q++) {
            ms252::PassParams P;
            memset(&P, 0, sizeof P);
            const bool last = q == np - 1;
            for (unsigned c = 0; c < nc; c++) {
                uint64_t* scr = (uint64_t*)((char*)scratch + (size_t)c * col_bytes);
                P.src[c] = q == 0 ? (const uint64_t*)src[c0 + c] : scr;
                P.dst[c] = last ? (uint64_t*)dst[c0 + c] : scr;
            }
            P.twr = p->d252_twr[q]; P.tw_lo = p->d252_tw_lo; P.tw_hi = p->d252_tw_hi; P.sc_lo = p->d252_sc_lo; P.sc_hi = p->d252_sc_hi;
            P.log_n = p->log_n; P.lo_bits = p->lo_bits;
            P.log_r = p->lr252[q]; P.log_c = ms252::TILE_LOG - P.log_r;
            P.log_s = p->log_n - done - P.log_r; P.log_tw = done;
            P.valid_rows = (1u << p->lr252[0]) >> log_zero_ext;
            P.log_r0 = p->lr252[0]; P.log_r1 = np == 3 ? p->lr252[1] : 0;
            P.scale_in = p->scale_in252; P.scale_out = p->scale_out252; P.bitrev_out = bitrev_out ? 1 : 0;
            const dim3 grid((unsigned)(n >> ms252::TILE_LOG), nc), block(ms252::NT2);
            ProfScope ps(ctx, names[q], 2.0 * col_bytes * nc);
            if (q == 0) hipLaunchKernelGGL((ms252::ntt252_strided_pass<ms252::NT2, true>), grid, block, 0, st, P);
            else if (!last) hipLaunchKernelGGL((ms252::ntt252_strided_pass<ms252::NT2, false>), grid, block, 0, st, P);
            else hipLaunchKernelGGL((ms252::ntt252_last_pass<ms252::NT2>), grid, block, 0, st, P);
            done += P.log_r;
        }
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}

static int plan_run252(ms_ntt_plan* p, const void* const* src, void* const* dst, unsigned ncols) {
    static const bool radix2_only = getenv("MS_NTT252_RADIX2") != nullptr && atoi(getenv("MS_NTT252_RADIX2")) != 0;   // A/B measurements
    if (p->np252 && !radix2_only) return plan_run252_tiled(p, src, dst, ncols, 0, false);
    ms_ctx* ctx = p->ctx;
    hipStream_t st = ctx->stream;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << p->log_n;
    for (unsigned c = 0; c < ncols; c++)
        if (src[c] != dst[c]) HIPCHK(hipMemcpyAsync(dst[c], src[c], n * 32, hipMemcpyDeviceToDevice, st));
    MSCHK(bit_reverse_run(ctx, 4, p->log_n, (const void* const*)dst, dst, ncols));
    for (unsigned c = 0; c < ncols; c++) {
        ms252::Params P;
        memset(&P, 0, sizeof P);
        P.col = (uint64_t*)dst[c]; P.tw_lo = p->d252_tw_lo; P.tw_hi = p->d252_tw_hi; P.sc_lo = p->d252_sc_lo; P.sc_hi = p->d252_sc_hi;
        P.log_n = p->log_n; P.lo_bits = p->lo_bits; P.scale_in = p->scale_in252; P.scale_out = p->scale_out252;
        const unsigned clog = std::min<unsigned>(p->log_n, ms252::CHUNK_LOG);
        {
            ProfScope ps(ctx, "ntt252_local", 64.0 * n);
            hipLaunchKernelGGL(ms252::ntt252_local, dim3((unsigned)(n >> clog)), dim3(ms252::NT), 0, st, P);
        }
        for (unsigned s = clog; s < p->log_n;) {                 // stages s+1 .. s+R per launch
            const unsigned R = std::min(ms252::MAX_FUSED_STAGES, p->log_n - s);
            P.stage = s;
            const dim3 g((unsigned)(((n >> R) + ms252::NT - 1) / ms252::NT));
            ProfScope ps(ctx, "ntt252_stages", 64.0 * n);
            if (R == 1) hipLaunchKernelGGL(ms252::ntt252_stages<1>, g, dim3(ms252::NT), 0, st, P);
            else hipLaunchKernelGGL(ms252::ntt252_stages<2>, g, dim3(ms252::NT), 0, st, P);
            s += R;
        }
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}

// Transform `ncols` columns: src[c] -> dst[c] (may alias).  valid_rows < 256 means the
// source only holds the first valid_rows/256 of the domain, the rest is implicit zeros.
static int plan_run(ms_ntt_plan* p, const void* const* src, void* const* dst, unsigned ncols, unsigned valid_rows, bool bitrev_out = false) {
    if (p->is252) {
        if (valid_rows != 256 || bitrev_out) return fail(MS_ERR_INVALID, "internal: Fp252 zero extension / fused bit reversal go through plan_run252_tiled");
        return plan_run252(p, src, dst, ncols);
    }
    if (bitrev_out && p->small) return fail(MS_ERR_INVALID, "internal: fused bit reversal needs the multi-pass path");
    ms_ctx* ctx = p->ctx;
    hipStream_t st = ctx->stream;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << p->log_n;
    const size_t col_bytes = n * p->V * 8;
    if (p->small) {
        if (valid_rows != 256) return fail(MS_ERR_INVALID, "zero-extended input needs a domain of at least 4096 points");
        for (unsigned c0 = 0; c0 < ncols; c0 += MAXC) {
            unsigned nc = std::min<unsigned>(MAXC, ncols - c0);
            msntt::SmallParams S;
            memset(&S, 0, sizeof S);
            for (unsigned c = 0; c < nc; c++) { S.src[c] = (const uint64_t*)src[c0 + c]; S.dst[c] = (uint64_t*)dst[c0 + c]; }
            S.tw = p->d_tw; S.scale_in = p->d_scale_in; S.scale_out = p->d_scale_out; S.log_n = p->log_n; S.V = p->V;
            ProfScope ps(ctx, "ntt_small", 2.0 * col_bytes * nc);
            hipLaunchKernelGGL(msntt::ntt_small, dim3(1, nc), dim3(msntt::NT), 0, st, S);
        }
        HIPCHK(hipGetLastError());
        return MS_OK;
    }
    unsigned group = (unsigned)std::max<size_t>(1, std::min<size_t>(MAXC, ctx->group_bytes / col_bytes));
    group = std::min(group, ncols);
    // uniform-factor plans on Fp columns: pass 1 stores whole lines in a row order that permutes the words inside every run of
    // 64; pass 2 un-permutes while it reads, in place on the scratch column (its tile owns those 64 words), and the last pass
    // goes scratch -> dst (natural order, or bit-reversed for the LDE).
    const bool perm = p->uni && p->V == 1;
    void* scratch = nullptr;
    MSCHK(ctx_scratch(ctx, (size_t)group * col_bytes, &scratch));
    const unsigned tiles = (unsigned)(n * p->V / msntt::TILE);
    // MS_NTT_STREAMS=2: the two halves of a group on two streams -- kernels of different passes then overlap (a pass
    // alternates between a memory phase and an arithmetic phase per workgroup); measured 162 -> 159 us per 2^24 column
    // over 8 columns.  Off by default: with concurrent kernels the per-kernel durations of a trace no longer add up to the
    // wall time, and the gain is under 2 %.  Never while per-launch profiling is on (its events sit on one stream).
    static const bool two_streams = getenv("MS_NTT_STREAMS") != nullptr && atoi(getenv("MS_NTT_STREAMS")) == 2;
    const bool two = two_streams && !ctx->profiling && ncols >= 2 && group >= 2 && p->log_n >= 20;
    if (two) {
        if (!ctx->stream2) {
            HIPCHK(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
        }
        HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
        HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    }
    void* const scratch_all = scratch;
    const hipStream_t st_main = st;
    for (unsigned g0 = 0; g0 < ncols; g0 += group) {
      const unsigned gnc = std::min(group, ncols - g0);
      const unsigned half = (two && gnc >= 2) ? gnc / 2 : gnc;
      for (unsigned s0 = 0; s0 < gnc; s0 = (s0 == 0) ? half : gnc) {     // at most two ranges: [0, half), [half, gnc)
        const unsigned c0 = g0 + s0, nc = (s0 == 0) ? half : gnc - half;
        const hipStream_t st = (s0 == 0) ? st_main : ctx->stream2;
        void* const scratch = (char*)scratch_all + (size_t)s0 * col_bytes;
        for (int q = 0; q < p->npass; q++) {
            msntt::PassParams P;
            memset(&P, 0, sizeof P);
            const bool last = (q == p->npass - 1);
            for (unsigned c = 0; c < nc; c++) {
                uint64_t* scr = (uint64_t*)((char*)scratch + (size_t)c * col_bytes);
                P.src[c] = (q == 0) ? (const uint64_t*)src[c0 + c] : scr;
                P.dst[c] = last ? (uint64_t*)dst[c0 + c] : scr;
            }
            P.tw_lo = p->d_tw_lo; P.tw_hi = p->d_tw_hi; P.wr = p->d_wr[q];
            P.aux_lo = p->d_aux_lo; P.aux_hi = p->d_aux_hi; P.gtab = p->d_gtab;
            P.log_n = p->log_n; P.V = p->V; P.valid_rows = valid_rows; P.lo_bits = p->lo_bits; P.log_s = p->log_s[q];
            P.nfields = p->nfields[q];
            for (unsigned f = 0; f < P.nfields; f++) P.fields[f] = p->fields[q][f];
            P.scale_const = p->scale_const;
            dim3 grid(tiles, nc);
            static const char* const pass_names[4] = {"ntt_pass1", "ntt_pass2", "ntt_pass3", "ntt_pass4"};
            ProfScope ps(ctx, pass_names[q], 2.0 * col_bytes * nc);
            // limb-form radix-256 passes (ntt2_kernels.h) wherever a pass has radix 256 and rows of >= 64 words
            const size_t pass_sw = ((size_t)1 << p->log_s[q]) * p->V;
            // (the per-element scale walk of an inverse coset transform stays with the round-1 last pass: the walk is two table
            // loads and a Montgomery product per word, which the 4-wave limb kernel hides worse -- 91 vs 75 us per 2^24
            // column; so does the fused bit-reversed store of Fq3 columns, whose runs interleave three words)
            const bool v2_ok = p->lr[q] == 8 && (n * p->V) % msntt2::TILE == 0 &&
                               (q == 0 ? ((n >> 8) * p->V) % msntt2::TW == 0
                                       : (pass_sw % msntt2::TW == 0 && !(last && p->scale_mode == 2 && p->d_scu4 == nullptr) &&
                                          !(last && bitrev_out && (p->V != 1 || p->inverse || p->scale_mode != 0))));
            static const bool dbg = getenv("MS_NTT_DEBUG") != nullptr;
            if (dbg) fprintf(stderr, "[ms_ntt] log_n=%u V=%u pass %d/%d radix 2^%u: %s kernel\n", p->log_n, p->V, q + 1, p->npass, p->lr[q], v2_ok ? (p->uni && q < 2 ? "limb-form (ntt2), uniform inter-pass factor" : "limb-form (ntt2)") : "round-1");
            if (p->uni && q < 2 && !v2_ok) return fail(MS_ERR_INVALID, "internal: uniform inter-pass plan without its limb-form passes");
            if (v2_ok) {
                msntt2::Params Q;
                memset(&Q, 0, sizeof Q);
                for (unsigned c = 0; c < nc; c++) { Q.src[c] = P.src[c]; Q.dst[c] = P.dst[c]; }
                Q.wr4 = p->d_wr4[q]; Q.twu4 = p->d_twu4[q]; Q.sc4 = p->d_sc4; Q.scu4 = p->d_scu4; Q.g4 = p->d_g4;
                Q.tw_lo = p->d_tw_lo; Q.tw_hi = p->d_tw_hi; Q.aux_lo = p->d_aux_lo; Q.aux_hi = p->d_aux_hi;
                Q.log_n = p->log_n; Q.V = p->V; Q.valid_rows = valid_rows; Q.lo_bits = p->lo_bits; Q.log_s = p->log_s[q];
                Q.tin4 = p->d_tin4; Q.tout4 = p->d_tout4; Q.r3 = p->lr[2];
                Q.nfields = P.nfields;
                for (unsigned f = 0; f < P.nfields; f++) Q.fields[f] = P.fields[f];
                const dim3 g2((unsigned)(n * p->V / msntt2::TILE), nc), b2(msntt2::NT);
                if (q == 0) {
                    const bool cos = (!p->inverse && p->coset);
                    const int na = valid_rows == 64 ? 4 : valid_rows == 32 ? 2 : valid_rows == 16 ? 1 : 16;
#define MS_P1(INV, COS, NA) do { if (perm) hipLaunchKernelGGL((msntt2::ntt2_first_pass<INV, COS, NA, true, true>), g2, b2, 0, st, Q); \
                                 else if (p->uni) hipLaunchKernelGGL((msntt2::ntt2_first_pass<INV, COS, NA, true>), g2, b2, 0, st, Q); \
                                 else hipLaunchKernelGGL((msntt2::ntt2_first_pass<INV, COS, NA, false>), g2, b2, 0, st, Q); } while (0)
                    if (p->inverse) MS_P1(true, false, 16);
                    else if (cos) {
                        if (na == 4) MS_P1(false, true, 4);
                        else if (na == 2) MS_P1(false, true, 2);
                        else if (na == 1) MS_P1(false, true, 1);
                        else MS_P1(false, true, 16);
                    } else {
                        if (na == 4) MS_P1(false, false, 4);
                        else if (na == 2) MS_P1(false, false, 2);
                        else if (na == 1) MS_P1(false, false, 1);
                        else MS_P1(false, false, 16);
                    }
#undef MS_P1
                } else if (!last) {
                    if (perm) {             // ... and reads pass 1's permuted rows, writes the natural order (in place)
                        if (p->inverse) hipLaunchKernelGGL((msntt2::ntt2_mid_pass<true, false, 0, true, true>), g2, b2, 0, st, Q);
                        else hipLaunchKernelGGL((msntt2::ntt2_mid_pass<false, false, 0, true, true>), g2, b2, 0, st, Q);
                    } else if (p->uni) {    // q == 1 of three: applies the per-lane remainder of pass 1's factor on its loads
                        if (p->inverse) hipLaunchKernelGGL((msntt2::ntt2_mid_pass<true, false, 0, true>), g2, b2, 0, st, Q);
                        else hipLaunchKernelGGL((msntt2::ntt2_mid_pass<false, false, 0, true>), g2, b2, 0, st, Q);
                    } else if (p->inverse) hipLaunchKernelGGL((msntt2::ntt2_mid_pass<true, false, 0>), g2, b2, 0, st, Q);
                    else hipLaunchKernelGGL((msntt2::ntt2_mid_pass<false, false, 0>), g2, b2, 0, st, Q);
                } else if (bitrev_out) {
                    hipLaunchKernelGGL(msntt2::ntt2_last_pass_bitrev, g2, b2, 0, st, Q);
                } else {
                    const int scale = p->scale_mode;
                    if (p->inverse) {
                        if (scale == 2) hipLaunchKernelGGL((msntt2::ntt2_mid_pass<true, true, 2>), g2, b2, 0, st, Q);
                        else if (scale == 1) hipLaunchKernelGGL((msntt2::ntt2_mid_pass<true, true, 1>), g2, b2, 0, st, Q);
                        else hipLaunchKernelGGL((msntt2::ntt2_mid_pass<true, true, 0>), g2, b2, 0, st, Q);
                    } else hipLaunchKernelGGL((msntt2::ntt2_mid_pass<false, true, 0>), g2, b2, 0, st, Q);
                }
                continue;
            }
            if (q == 0) {
                const bool cos = (!p->inverse && p->coset);
                if (p->inverse) hipLaunchKernelGGL((msntt::ntt_first_pass<true, false>), grid, dim3(msntt::NT), 0, st, P);
                else if (cos) {
                    if (valid_rows == 64) hipLaunchKernelGGL((msntt::ntt_first_pass<false, true, 4>), grid, dim3(msntt::NT), 0, st, P);
                    else if (valid_rows == 32) hipLaunchKernelGGL((msntt::ntt_first_pass<false, true, 2>), grid, dim3(msntt::NT), 0, st, P);
                    else if (valid_rows == 16) hipLaunchKernelGGL((msntt::ntt_first_pass<false, true, 1>), grid, dim3(msntt::NT), 0, st, P);
                    else hipLaunchKernelGGL((msntt::ntt_first_pass<false, true>), grid, dim3(msntt::NT), 0, st, P);
                } else {
                    if (valid_rows == 64) hipLaunchKernelGGL((msntt::ntt_first_pass<false, false, 4>), grid, dim3(msntt::NT), 0, st, P);
                    else if (valid_rows == 32) hipLaunchKernelGGL((msntt::ntt_first_pass<false, false, 2>), grid, dim3(msntt::NT), 0, st, P);
                    else if (valid_rows == 16) hipLaunchKernelGGL((msntt::ntt_first_pass<false, false, 1>), grid, dim3(msntt::NT), 0, st, P);
                    else hipLaunchKernelGGL((msntt::ntt_first_pass<false, false>), grid, dim3(msntt::NT), 0, st, P);
                }
            } else {
                const int scale = last ? p->scale_mode : 0;
                switch (p->lr[q]) {
                case 4: launch_mid<1>(p->inverse, last, scale, bitrev_out, grid, st, P); break;
                case 5: launch_mid<2>(p->inverse, last, scale, bitrev_out, grid, st, P); break;
                case 6: launch_mid<4>(p->inverse, last, scale, bitrev_out, grid, st, P); break;
                case 7: launch_mid<8>(p->inverse, last, scale, bitrev_out, grid, st, P); break;
                case 8: launch_mid<16>(p->inverse, last, scale, bitrev_out, grid, st, P); break;
                default: return fail(MS_ERR_INVALID, "internal: bad pass radix 2^%u", p->lr[q]);
                }
            }
        }
      }
    }
    if (two) {
        HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}

// ---- two-pass coset LDE (lde2_kernels.h) ------------------------------------------------------------------------------
// fwd = the forward plan of the LDE domain (N = n << log_b points, offset h).  Applies to Fp columns of 2^17..2^20 rows.
static bool lde2_applicable(const ms_ntt_plan* fwd, unsigned V, unsigned log_n, unsigned log_b) {
    static const bool off = getenv("MS_LDE2") != nullptr && atoi(getenv("MS_LDE2")) == 0;       // A/B measurements
    return !off && V == 1 && log_n >= 17 && log_n <= 20 && log_b >= 1 && log_b <= 6 && !fwd->small && fwd->d_wr4[0] != nullptr;
}
static int lde2_tables(ms_ntt_plan* fwd, unsigned log_n, unsigned log_b, ms_ntt_plan::Lde2** out) {
    for (auto& l : fwd->lde2) if (l.log_b == log_b) { *out = &l; return MS_OK; }
    const size_t n = (size_t)1 << log_n, L = n >> 8, T = L >> 8, beta = (size_t)1 << log_b;
    const uint64_t wN = gl::root_of_unity(log_n + log_b), wL = gl::root_of_unity(log_n - 8), h = fwd->offset_canon;
    const bool uni = T >= 4;                                      // lde2_kernels.h: uniform split of pass A's factor
    const size_t nt = L >> 6, n_tin = uni ? nt * 256 * 4 : 0, n_tout = uni ? beta * nt * 16 * 4 : 0;
    std::vector<uint64_t> host(beta * 256 * 4 + beta * L + 256 * T * 4 + n_tin + n_tout);
    uint64_t* gpl = host.data();
    uint64_t* aux = gpl + beta * 256 * 4;
    uint64_t* t2 = aux + beta * L;
    uint64_t* tin4 = t2 + 256 * T * 4;
    uint64_t* tout4 = tin4 + n_tin;
    const uint64_t sh[4] = {1, (uint64_t)1 << 24, (uint64_t)1 << 48, gl::pow(2, 72)};
    if (uni) {
        const uint64_t wn = gl::root_of_unity(log_n);
        for (size_t i0h = 0; i0h < nt; i0h++) {
            const uint64_t E = 64 * i0h;
            for (size_t b = 0; b < 16; b++) {
                const uint64_t wm = gl::pow(wn, (b * L + E) & (n - 1));          // w_256^b w_n^E
                uint64_t x = 1;
                for (size_t a = 0; a < 16; a++, x = gl::mul(x, wm))
                    for (int c = 0; c < 4; c++) tin4[((i0h * 16 + b) * 16 + a) * 4 + c] = gl::mul(x, sh[c]);
            }
        }
        uint64_t G = h;
        for (size_t j = 0; j < beta; j++, G = gl::mul(G, wN))
            for (size_t i0h = 0; i0h < nt; i0h++) {
                const uint64_t E = 64 * i0h, wm = gl::pow(wn, (16 * E) & (n - 1));
                uint64_t x = gl::pow(G, E);
                for (size_t bp = 0; bp < 16; bp++, x = gl::mul(x, wm))
                    for (int c = 0; c < 4; c++) tout4[((j * nt + i0h) * 16 + bp) * 4 + c] = gl::mul(x, sh[c]);
            }
    }
    uint64_t G = h;                                               // G_j = h w_N^j
    for (size_t j = 0; j < beta; j++, G = gl::mul(G, wN)) {
        const uint64_t GL = gl::pow(G, (uint64_t)L);
        uint64_t x = 1;
        for (size_t i = 0; i < 256; i++) { for (int c = 0; c < 4; c++) gpl[(j * 256 + i) * 4 + c] = gl::mul(x, sh[c]); x = gl::mul(x, GL); }   // plain, 4 copies
        x = 1;
        for (size_t i = 0; i < L; i++) { aux[j * L + i] = gl::to_mont(x); x = gl::mul(x, G); }
    }
    for (size_t k = 0; k < 256; k++) {
        const uint64_t wk = gl::pow(wL, (uint64_t)k);
        uint64_t x = 1;
        for (size_t t = 0; t < T; t++) { for (int c = 0; c < 4; c++) t2[(k * T + t) * 4 + c] = gl::mul(x, sh[c]); x = gl::mul(x, wk); }          // plain, 4 copies
    }
    ms_ntt_plan::Lde2 l;
    l.log_b = log_b;
    if (hipMalloc(&l.d, host.size() * 8) != hipSuccess) return fail(MS_ERR_NOMEM, "LDE tables (%zu bytes)", host.size() * 8);
    if (hipMemcpy(l.d, host.data(), host.size() * 8, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(l.d); return fail(MS_ERR_HIP, "LDE table upload"); }
    l.gpl = l.d; l.aux = l.d + beta * 256 * 4; l.t2 = l.aux + beta * L;
    if (uni) { l.tin4 = l.t2 + 256 * T * 4; l.tout4 = l.tin4 + n_tin; }
    fwd->lde2.push_back(l);
    *out = &fwd->lde2.back();
    return MS_OK;
}
// coefficients (2^log_n words per column, src) -> bit-reversed evaluations on the coset of N points (dst, N words per column)
static int lde2_run(ms_ntt_plan* fwd, unsigned log_n, unsigned log_b, const void* const* src, void* const* dst, unsigned ncols) {
    ms_ctx* ctx = fwd->ctx;
    hipStream_t st = ctx->stream;
    HIPCHK(hipSetDevice(ctx->device));
    ms_ntt_plan::Lde2* tb = nullptr;
    MSCHK(lde2_tables(fwd, log_n, log_b, &tb));
    const size_t n = (size_t)1 << log_n, N = n << log_b, col_bytes = N * 8;
    const unsigned T = (unsigned)(n >> 16);
    unsigned group = (unsigned)std::max<size_t>(1, std::min<size_t>(MAXC, ctx->group_bytes / col_bytes));
    group = std::min(group, ncols);
    void* scratch = nullptr;
    MSCHK(ctx_scratch(ctx, (size_t)group * col_bytes, &scratch));
    for (unsigned c0 = 0; c0 < ncols; c0 += group) {
        const unsigned nc = std::min(group, ncols - c0);
        mslde2::Params P;
        memset(&P, 0, sizeof P);
        P.wr4 = fwd->d_wr4[0]; P.gpl = tb->gpl; P.aux = tb->aux; P.t2 = tb->t2; P.tin4 = tb->tin4; P.tout4 = tb->tout4;
        static const bool no_uni = getenv("MS_LDE2_PERLANE") != nullptr && atoi(getenv("MS_LDE2_PERLANE")) != 0;    // A/B measurements
        const bool uni = tb->tin4 != nullptr && !no_uni;
        P.tw_lo = fwd->d_tw_lo; P.tw_hi = fwd->d_tw_hi; P.lo_bits = fwd->lo_bits; P.log_n = log_n; P.log_b = log_b;
        for (unsigned c = 0; c < nc; c++) { P.src[c] = (const uint64_t*)src[c0 + c]; P.dst[c] = (uint64_t*)((char*)scratch + (size_t)c * col_bytes); }
        {
            ProfScope ps(ctx, "lde2_pass_a", (double)(n * 8 + col_bytes) * nc);
            const dim3 ga((unsigned)(n >> 14), 1u << log_b, nc);
            if (uni) hipLaunchKernelGGL(mslde2::lde2_strided_pass<true>, ga, dim3(msntt2::NT), 0, st, P);
            else hipLaunchKernelGGL(mslde2::lde2_strided_pass<false>, ga, dim3(msntt2::NT), 0, st, P);
        }
        for (unsigned c = 0; c < nc; c++) { P.src[c] = (const uint64_t*)((char*)scratch + (size_t)c * col_bytes); P.dst[c] = (uint64_t*)dst[c0 + c]; }
        {
            ProfScope ps(ctx, "lde2_pass_b", 2.0 * col_bytes * nc);
            const dim3 g(4 * T, nc, 1u << log_b), b(msntt2::NT);
            switch (T) {
            case 16: if (uni) hipLaunchKernelGGL((mslde2::lde2_rows_pass<16, true>), g, b, 0, st, P); else hipLaunchKernelGGL((mslde2::lde2_rows_pass<16, false>), g, b, 0, st, P); break;
            case 8: if (uni) hipLaunchKernelGGL((mslde2::lde2_rows_pass<8, true>), g, b, 0, st, P); else hipLaunchKernelGGL((mslde2::lde2_rows_pass<8, false>), g, b, 0, st, P); break;
            case 4: if (uni) hipLaunchKernelGGL((mslde2::lde2_rows_pass<4, true>), g, b, 0, st, P); else hipLaunchKernelGGL((mslde2::lde2_rows_pass<4, false>), g, b, 0, st, P); break;
            default: hipLaunchKernelGGL((mslde2::lde2_rows_pass<2, false>), g, b, 0, st, P); break;
            }
        }
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}

extern "C" int ms_ntt_encode(ms_ntt_plan* plan, void* d_column) {
    if (!plan || !d_column) return fail(MS_ERR_INVALID, "ms_ntt_encode: null argument");
    std::lock_guard<std::mutex> lk(plan->ctx->mu);
    plan->queue.push_back(d_column);
    return MS_OK;
}
extern "C" int ms_ntt_enqueue(ms_ntt_plan* plan, void* const* d_columns, unsigned ncols) {
    if (!plan || (!d_columns && ncols)) return fail(MS_ERR_INVALID, "ms_ntt_enqueue: null argument");
    if (ncols == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(plan->ctx->mu);
    return plan_run(plan, (const void* const*)d_columns, d_columns, ncols, 256);
}
extern "C" int ms_ntt_execute(ms_ntt_plan* plan) {
    if (!plan) return fail(MS_ERR_INVALID, "ms_ntt_execute: null plan");
    std::vector<void*> q;
    { std::lock_guard<std::mutex> lk(plan->ctx->mu); q.swap(plan->queue); }
    if (!q.empty()) MSCHK(ms_ntt_enqueue(plan, q.data(), (unsigned)q.size()));
    HIPCHK(hipStreamSynchronize(plan->ctx->stream));
    return MS_OK;
}

// ---------------------------------------------------------------------------------------
// bit reversal
// ---------------------------------------------------------------------------------------
static int bit_reverse_run(ms_ctx* ctx, unsigned V, unsigned log_n, const void* const* src, void* const* dst, unsigned ncols) {
    hipStream_t st = ctx->stream;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << log_n;
    if (log_n >= 10 && V != 4) {
        for (unsigned c0 = 0; c0 < ncols; c0 += MAXC) {
            unsigned nc = std::min<unsigned>(MAXC, ncols - c0);
            msntt::BitrevParams B;
            memset(&B, 0, sizeof B);
            for (unsigned c = 0; c < nc; c++) { B.src[c] = (const uint64_t*)src[c0 + c]; B.dst[c] = (uint64_t*)dst[c0 + c]; }
            B.log_n = log_n;
            dim3 grid((unsigned)(n >> 10), nc);
            ProfScope ps(ctx, "bit_reverse", 2.0 * n * V * 8 * nc);
            if (V == 1) hipLaunchKernelGGL(msntt::bit_reverse_tiled<1>, grid, dim3(msntt::NT), 0, st, B);
            else hipLaunchKernelGGL(msntt::bit_reverse_tiled<3>, grid, dim3(msntt::NT), 0, st, B);
        }
    } else {
        // tiny: out of place through scratch when aliased
        const size_t col_bytes = n * V * 8;
        void* scratch = nullptr;
        const unsigned grp = (unsigned)std::max<size_t>(1, std::min<size_t>(MAXC, ((size_t)256 << 20) / col_bytes));
        MSCHK(ctx_scratch(ctx, (size_t)grp * col_bytes, &scratch));
        for (unsigned c0 = 0; c0 < ncols; c0 += grp) {
            unsigned nc = std::min<unsigned>(grp, ncols - c0);
            msntt::BitrevParams B;
            memset(&B, 0, sizeof B);
            for (unsigned c = 0; c < nc; c++) { B.src[c] = (const uint64_t*)src[c0 + c]; B.dst[c] = (uint64_t*)((char*)scratch + c * col_bytes); }
            B.log_n = log_n;
            dim3 grid((unsigned)((n + msntt::NT - 1) / msntt::NT), nc);
            if (V == 1) hipLaunchKernelGGL(msntt::bit_reverse_simple<1>, grid, dim3(msntt::NT), 0, st, B);
            else if (V == 3) hipLaunchKernelGGL(msntt::bit_reverse_simple<3>, grid, dim3(msntt::NT), 0, st, B);
            else hipLaunchKernelGGL(msntt::bit_reverse_simple<4>, grid, dim3(msntt::NT), 0, st, B);
            for (unsigned c = 0; c < nc; c++)
                HIPCHK(hipMemcpyAsync(dst[c0 + c], (char*)scratch + c * col_bytes, col_bytes, hipMemcpyDeviceToDevice, st));
        }
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_bit_reverse(ms_ctx* ctx, int field, unsigned log_n, void* const* d_columns, unsigned ncols) {
    if (!ctx || (!d_columns && ncols)) return fail(MS_ERR_INVALID, "ms_bit_reverse: null argument");
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    if (log_n > 40) return fail(MS_ERR_INVALID, "log_n too large");
    std::lock_guard<std::mutex> lk(ctx->mu);
    return bit_reverse_run(ctx, V, log_n, (const void* const*)d_columns, d_columns, ncols);
}

// ---------------------------------------------------------------------------------------
// fused LDE
// ---------------------------------------------------------------------------------------
extern "C" int ms_lde(ms_ctx* ctx, int field, unsigned log_n, unsigned log_blowup, const void* h_offset,
                      const void* const* d_in, void* const* d_out, unsigned ncols, int bit_reversed) {
    if (!ctx || !d_in || !d_out) return fail(MS_ERR_INVALID, "ms_lde: null argument");
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    const unsigned log_N = log_n + log_blowup;
    if (V == 4) {
        // compute-bound field: iNTT into the head of the output column, explicit zero padding, coset NTT,
        // bit reversal -- the plain sequence (the fused / pruned passes are Goldilocks kernels)
        if (log_N > 40) return fail(MS_ERR_INVALID, "LDE domain 2^%u too large", log_N);
        f252::E h252 = f252::one();
        if (h_offset) memcpy(h252.l, h_offset, 32);
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (f252::is_zero(h252) || f252::geq_p(h252)) return fail(MS_ERR_INVALID, "coset offset must be a non-zero canonical element");
        ms_ntt_plan *inv = nullptr, *fwd = nullptr;
        MSCHK(plan252_cached(ctx, log_n, true, f252::one(), &inv));
        MSCHK(plan252_cached(ctx, log_N, false, h252, &fwd));
        MSCHK(plan_run252(inv, d_in, d_out, ncols));
        // tiled passes: the coefficients are read straight from the head of the output column (zeros implicit), the
        // bit reversal is part of the last pass
        if (fwd->np252 && log_blowup <= fwd->lr252[0] && !getenv("MS_NTT252_RADIX2"))
            return plan_run252_tiled(fwd, (const void* const*)d_out, d_out, ncols, log_blowup, bit_reversed != 0);
        const size_t n = (size_t)1 << log_n, N = (size_t)1 << log_N;
        if (N > n)
            for (unsigned c = 0; c < ncols; c++) HIPCHK(hipMemsetAsync((char*)d_out[c] + n * 32, 0, (N - n) * 32, ctx->stream));
        MSCHK(plan_run252(fwd, (const void* const*)d_out, d_out, ncols));
        if (bit_reversed) MSCHK(bit_reverse_run(ctx, 4, log_N, (const void* const*)d_out, d_out, ncols));
        return MS_OK;
    }
    if (log_N > 32) return fail(MS_ERR_INVALID, "LDE domain 2^%u exceeds the two-adicity", log_N);
    uint64_t h = 1;
    if (h_offset) { uint64_t h_m; memcpy(&h_m, h_offset, 8); h = gl::from_mont(h_m); }
    if (h == 0) return fail(MS_ERR_INVALID, "coset offset must be non-zero");
    ms_ntt_plan *inv = nullptr, *fwd = nullptr;
    int rc;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        MSCHK(ctx_plan(ctx, V, log_n, true, 1, &inv));
        MSCHK(ctx_plan(ctx, V, log_N, false, h, &fwd));
        // coefficients land in the first 2^log_n elements of the output column
        rc = plan_run(inv, d_in, d_out, ncols, 256);
        const size_t n = (size_t)1 << log_n, N = (size_t)1 << log_N;
        if (rc == MS_OK && lde2_applicable(fwd, V, log_n, log_blowup)) {
            // beta coset transforms of size n in two passes each, blocks land in the bit-reversed order
            rc = lde2_run(fwd, log_n, log_blowup, (const void* const*)d_out, d_out, ncols);
            if (rc == MS_OK && !bit_reversed) rc = bit_reverse_run(ctx, V, log_N, (const void* const*)d_out, d_out, ncols);
            return rc;
        }
        if (rc == MS_OK) {
            if (!fwd->small && log_blowup <= 4) {
                // zero padding is implicit in pass 1, the bit reversal is fused into the last pass
                rc = plan_run(fwd, (const void* const*)d_out, d_out, ncols, 256u >> log_blowup, bit_reversed != 0);
                bit_reversed = 0;
            } else {
                for (unsigned c = 0; c < ncols && rc == MS_OK; c++)
                    if (hipMemsetAsync((char*)d_out[c] + n * V * 8, 0, (N - n) * V * 8, ctx->stream) != hipSuccess)
                        rc = fail(MS_ERR_HIP, "hipMemsetAsync failed");
                if (rc == MS_OK) rc = plan_run(fwd, (const void* const*)d_out, d_out, ncols, 256);
            }
        }
        if (rc == MS_OK && bit_reversed) rc = bit_reverse_run(ctx, V, log_N, (const void* const*)d_out, d_out, ncols);
    }
    return rc;
}

// Matrix::into_evaluations / bit_reversed_evaluate on columns shorter than the domain (src/matrix.rs:193-251:
// "resize the column to the domain size", i.e. zero-extend the coefficient vector): the second half of ms_lde.
extern "C" int ms_evaluate(ms_ctx* ctx, int field, unsigned log_n, unsigned log_domain, const void* h_offset,
                           const void* const* d_in, void* const* d_out, unsigned ncols, int bit_reversed) {
    if (!ctx || !d_in || !d_out) return fail(MS_ERR_INVALID, "ms_evaluate: null argument");
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    if (log_n > log_domain) return fail(MS_ERR_INVALID, "more coefficients (2^%u) than domain points (2^%u)", log_n, log_domain);
    const unsigned log_blowup = log_domain - log_n;
    const size_t n = (size_t)1 << log_n, N = (size_t)1 << log_domain;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    ms_ntt_plan* fwd = nullptr;
    if (V == 4) {
        if (log_domain > 40) return fail(MS_ERR_INVALID, "domain 2^%u too large", log_domain);
        f252::E h252 = f252::one();
        if (h_offset) memcpy(h252.l, h_offset, 32);
        if (f252::is_zero(h252) || f252::geq_p(h252)) return fail(MS_ERR_INVALID, "coset offset must be a non-zero canonical element");
        MSCHK(plan252_cached(ctx, log_domain, false, h252, &fwd));
    } else {
        if (log_domain > 32) return fail(MS_ERR_INVALID, "domain 2^%u exceeds the two-adicity", log_domain);
        uint64_t h = 1;
        if (h_offset) { uint64_t h_m; memcpy(&h_m, h_offset, 8); h = gl::from_mont(h_m); }
        if (h == 0) return fail(MS_ERR_INVALID, "coset offset must be non-zero");
        MSCHK(ctx_plan(ctx, V, log_domain, false, h, &fwd));
    }
    if (V != 4 && lde2_applicable(fwd, V, log_n, log_blowup)) {
        MSCHK(lde2_run(fwd, log_n, log_blowup, d_in, d_out, ncols));
        if (!bit_reversed) MSCHK(bit_reverse_run(ctx, V, log_domain, (const void* const*)d_out, d_out, ncols));
        return MS_OK;
    }
    if (V == 4 && fwd->np252 && log_blowup <= fwd->lr252[0] && !getenv("MS_NTT252_RADIX2"))
        return plan_run252_tiled(fwd, d_in, d_out, ncols, log_blowup, bit_reversed != 0);
    if (V != 4 && !fwd->small && log_blowup >= 2 && log_blowup <= 4) {
        // pass 1 reads only the rows that hold coefficients (straight from d_in), zero padding is implicit,
        // the bit reversal is fused into the last pass
        return plan_run(fwd, d_in, d_out, ncols, 256u >> log_blowup, bit_reversed != 0);
    }
    for (unsigned c = 0; c < ncols; c++) {
        if (d_in[c] != d_out[c]) HIPCHK(hipMemcpyAsync(d_out[c], d_in[c], n * V * 8, hipMemcpyDeviceToDevice, ctx->stream));
        if (N > n) HIPCHK(hipMemsetAsync((char*)d_out[c] + n * V * 8, 0, (N - n) * V * 8, ctx->stream));
    }
    MSCHK(plan_run(fwd, (const void* const*)d_out, d_out, ncols, 256));
    if (bit_reversed) MSCHK(bit_reverse_run(ctx, V, log_domain, (const void* const*)d_out, d_out, ncols));
    return MS_OK;
}

// composition_poly.chunks(k) -> k columns (src/prover.rs:113-121): out[c][j] = in[j*k + c]
extern "C" int ms_deinterleave(ms_ctx* ctx, int field, size_t n_out, unsigned k, const void* d_in, void* const* d_out) {
    if (!ctx || !d_in || !d_out) return fail(MS_ERR_INVALID, "ms_deinterleave: null argument");
    const size_t fb = ms_field_bytes(field);
    if (!fb) return fail(MS_ERR_UNSUPPORTED, "unknown field %d", field);
    if (k == 0 || k > (unsigned)msstage::MAXCOLS) return fail(MS_ERR_UNSUPPORTED, "1..%d columns", msstage::MAXCOLS);
    if (n_out == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    msscan::DeinterleaveParams P;
    memset(&P, 0, sizeof P);
    for (unsigned c = 0; c < k; c++) { if (!d_out[c]) return fail(MS_ERR_INVALID, "null column %u", c); P.out[c] = (uint64_t*)d_out[c]; }
    P.in = (const uint64_t*)d_in; P.n_out = n_out; P.k = k; P.V = (unsigned)(fb / 8);
    const size_t total = n_out * k * P.V;
    ProfScope ps(ctx, "deinterleave", 16.0 * total);
    hipLaunchKernelGGL(msscan::deinterleave, dim3(stream_grid(total)), dim3(msscan::NT), 0, ctx->stream, P);
    HIPCHK(hipGetLastError());
    return MS_OK;
}

// ---------------------------------------------------------------------------------------
// SHA-256 commitments
// ---------------------------------------------------------------------------------------
extern "C" int ms_sha256_rows(ms_ctx* ctx, int field, size_t nrows, const void* const* d_cols, unsigned ncols, void* d_leaves) {
    if (!ctx || (!d_cols && ncols) || !d_leaves) return fail(MS_ERR_INVALID, "ms_sha256_rows: null argument");
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    if (ncols > (unsigned)mssha::MAXCOLS) return fail(MS_ERR_UNSUPPORTED, "at most %d columns per commitment", mssha::MAXCOLS);
    if (nrows == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    mssha::RowsParams P;
    memset(&P, 0, sizeof P);
    for (unsigned c = 0; c < ncols; c++) P.cols[c] = (const uint64_t*)d_cols[c];
    P.leaves = (uint8_t*)d_leaves; P.nrows = nrows; P.ncols = ncols; P.V = V; P.row_stride = V;
    if (ncols && (ncols * V) % 8 == 0) { P.fold_last = 1; mssha::sha256_fold_pad_block((uint64_t)ncols * V * 64, P.kw_last); }
    {
        ProfScope ps(ctx, "sha256_rows", (double)nrows * ncols * V * 8 + 32.0 * nrows);
        hipLaunchKernelGGL(mssha::sha256_rows, dim3((unsigned)((nrows + mssha::NT - 1) / mssha::NT)), dim3(mssha::NT), 0, ctx->stream, P);
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_sha256_rows_row_major(ms_ctx* ctx, int field, size_t nrows, unsigned ncols, const void* d_matrix, void* d_leaves) {
    if (!ctx || !d_matrix || !d_leaves) return fail(MS_ERR_INVALID, "ms_sha256_rows_row_major: null argument");
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    if (ncols == 0 || ncols > (unsigned)mssha::MAXCOLS) return fail(MS_ERR_UNSUPPORTED, "1..%d columns per row", mssha::MAXCOLS);
    if (nrows == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    mssha::RowsParams P;
    memset(&P, 0, sizeof P);
    for (unsigned c = 0; c < ncols; c++) P.cols[c] = (const uint64_t*)d_matrix + (size_t)c * V;
    P.leaves = (uint8_t*)d_leaves; P.nrows = nrows; P.ncols = ncols; P.V = V; P.row_stride = ncols * V;
    if ((ncols * V) % 8 == 0) { P.fold_last = 1; mssha::sha256_fold_pad_block((uint64_t)ncols * V * 64, P.kw_last); }
    {
        ProfScope ps(ctx, "sha256_rows", (double)nrows * ncols * V * 8 + 32.0 * nrows);
        hipLaunchKernelGGL(mssha::sha256_rows, dim3((unsigned)((nrows + mssha::NT - 1) / mssha::NT)), dim3(mssha::NT), 0, ctx->stream, P);
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_sha256_merkle(ms_ctx* ctx, size_t nleaves, const void* d_leaves, void* d_nodes) {
    if (!ctx || !d_leaves || !d_nodes) return fail(MS_ERR_INVALID, "ms_sha256_merkle: null argument");
    if (nleaves < 2 || (nleaves & (nleaves - 1))) return fail(MS_ERR_INVALID, "number of leaves must be a power of two >= 2");
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    uint8_t* nodes = (uint8_t*)d_nodes;
    HIPCHK(hipMemsetAsync(nodes, 0, 32, ctx->stream));
    const uint8_t* src = (const uint8_t*)d_leaves;
    for (size_t count = nleaves / 2; count >= 1; count >>= 1) {
        uint8_t* dst = nodes + count * 32;
        if (count <= (size_t)mssha::NT) {                        // the remaining levels in one launch
            ProfScope ps(ctx, "sha256_merkle_top", 96.0 * (2 * count - 1));
            hipLaunchKernelGGL(mssha::sha256_merkle_top, dim3(1), dim3(mssha::NT), 0, ctx->stream, src, nodes, (unsigned)count);
            break;
        }
        ProfScope ps(ctx, "sha256_merkle_level", 96.0 * count);
        hipLaunchKernelGGL(mssha::sha256_merge_level, dim3((unsigned)((count + mssha::NT - 1) / mssha::NT)), dim3(mssha::NT), 0, ctx->stream, src, dst, count);
        src = dst;
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}

// ---------------------------------------------------------------------------------------
// element-wise stages
// ---------------------------------------------------------------------------------------
static unsigned stream_grid(size_t n) { return (unsigned)std::max<size_t>(1, std::min<size_t>((n + msstage::NT - 1) / msstage::NT, 256 * 16)); }
static int field_pair(int lf, int rf, unsigned* VL, unsigned* VR) {
    MSCHK(field_words(lf, VL));
    MSCHK(field_words(rf, VR));
    if (*VR > *VL || ((*VL == 4) != (*VR == 4)))
        return fail(MS_ERR_UNSUPPORTED, "rhs field must embed into the lhs field (Fp,Fp / Fq3,Fq3 / Fq3,Fp / Fp252,Fp252)");
    return MS_OK;
}
static size_t norm_shift(long shift, size_t n) {
    if (n == 0) return 0;
    long long m = (long long)shift % (long long)n;
    if (m < 0) m += (long long)n;
    return (size_t)m;
}
extern "C" int ms_binary(ms_ctx* ctx, int op, int lf, int rf, size_t n, void* d_dst, const void* d_lhs, const void* d_rhs, long shift) {
    if (!ctx || !d_dst || !d_lhs || !d_rhs) return fail(MS_ERR_INVALID, "ms_binary: null argument");
    if (op != MS_ADD && op != MS_MUL) return fail(MS_ERR_INVALID, "unknown binary op %d", op);
    unsigned VL = 0, VR = 0;
    MSCHK(field_pair(lf, rf, &VL, &VR));
    if (n == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    const size_t sh = norm_shift(shift, n);
    uint64_t* dst = (uint64_t*)d_dst; const uint64_t* l = (const uint64_t*)d_lhs; const uint64_t* r = (const uint64_t*)d_rhs;
    dim3 g(stream_grid(n)), b(msstage::NT);
    ProfScope ps(ctx, op == MS_ADD ? "stage_add" : "stage_mul", 8.0 * n * (2 * VL + VR));
    using namespace msstage;
    if (VL == 3 && VR == 3 && op == MS_ADD) {
        // component-wise: an Fq3 + Fq3 column add is an Fp add over 3n consecutive words (fully coalesced)
        hipLaunchKernelGGL((k_binary<FpT, FpT, 0>), dim3(stream_grid(3 * n)), b, 0, ctx->stream, dst, l, r, 3 * n, 3 * sh);
    }
    else if (VL == 4) { if (op == MS_ADD) hipLaunchKernelGGL((k_binary<Fp252T, Fp252T, 0>), g, b, 0, ctx->stream, dst, l, r, n, sh); else hipLaunchKernelGGL((k_binary<Fp252T, Fp252T, 1>), g, b, 0, ctx->stream, dst, l, r, n, sh); }
    else if (VL == 1) { if (op == MS_ADD) hipLaunchKernelGGL((k_binary<FpT, FpT, 0>), g, b, 0, ctx->stream, dst, l, r, n, sh); else hipLaunchKernelGGL((k_binary<FpT, FpT, 1>), g, b, 0, ctx->stream, dst, l, r, n, sh); }
    else if (VR == 3) { if (op == MS_ADD) hipLaunchKernelGGL((k_binary<Fq3T, Fq3T, 0>), g, b, 0, ctx->stream, dst, l, r, n, sh); else hipLaunchKernelGGL((k_binary<Fq3T, Fq3T, 1>), g, b, 0, ctx->stream, dst, l, r, n, sh); }
    else { if (op == MS_ADD) hipLaunchKernelGGL((k_binary<Fq3T, FpT, 0>), g, b, 0, ctx->stream, dst, l, r, n, sh); else hipLaunchKernelGGL((k_binary<Fq3T, FpT, 1>), g, b, 0, ctx->stream, dst, l, r, n, sh); }
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_binary_const(ms_ctx* ctx, int op, int lf, int rf, size_t n, void* d_dst, const void* d_lhs, const void* h_const) {
    if (!ctx || !d_dst || !d_lhs || !h_const) return fail(MS_ERR_INVALID, "ms_binary_const: null argument");
    if (op != MS_ADD && op != MS_MUL) return fail(MS_ERR_INVALID, "unknown binary op %d", op);
    unsigned VL = 0, VR = 0;
    MSCHK(field_pair(lf, rf, &VL, &VR));
    if (n == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    msstage::Const3 c = {{0, 0, 0, 0}};
    memcpy(c.w, h_const, VR * 8);
    uint64_t* dst = (uint64_t*)d_dst; const uint64_t* l = (const uint64_t*)d_lhs;
    dim3 g(stream_grid(n)), b(msstage::NT);
    ProfScope ps(ctx, op == MS_ADD ? "stage_add_const" : "stage_mul_const", 16.0 * n * VL);
    using namespace msstage;
    if (VL == 3 && VR == 1 && op == MS_MUL) {
        // scaling an Fq3 column by an Fp constant acts on every word alike
        hipLaunchKernelGGL((k_binary_const<FpT, FpT, 1>), dim3(stream_grid(3 * n)), b, 0, ctx->stream, dst, l, c, 3 * n);
    }
    else if (VL == 4) { if (op == MS_ADD) hipLaunchKernelGGL((k_binary_const<Fp252T, Fp252T, 0>), g, b, 0, ctx->stream, dst, l, c, n); else hipLaunchKernelGGL((k_binary_const<Fp252T, Fp252T, 1>), g, b, 0, ctx->stream, dst, l, c, n); }
    else if (VL == 1) { if (op == MS_ADD) hipLaunchKernelGGL((k_binary_const<FpT, FpT, 0>), g, b, 0, ctx->stream, dst, l, c, n); else hipLaunchKernelGGL((k_binary_const<FpT, FpT, 1>), g, b, 0, ctx->stream, dst, l, c, n); }
    else if (VR == 3) { if (op == MS_ADD) hipLaunchKernelGGL((k_binary_const<Fq3T, Fq3T, 0>), g, b, 0, ctx->stream, dst, l, c, n); else hipLaunchKernelGGL((k_binary_const<Fq3T, Fq3T, 1>), g, b, 0, ctx->stream, dst, l, c, n); }
    else { if (op == MS_ADD) hipLaunchKernelGGL((k_binary_const<Fq3T, FpT, 0>), g, b, 0, ctx->stream, dst, l, c, n); else hipLaunchKernelGGL((k_binary_const<Fq3T, FpT, 1>), g, b, 0, ctx->stream, dst, l, c, n); }
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_mul_pow(ms_ctx* ctx, int lf, int rf, size_t n, void* d_dst, const void* d_lhs, const void* d_rhs, unsigned power, long shift) {
    if (!ctx || !d_dst || !d_lhs || !d_rhs) return fail(MS_ERR_INVALID, "ms_mul_pow: null argument");
    unsigned VL = 0, VR = 0;
    MSCHK(field_pair(lf, rf, &VL, &VR));
    if (n == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    const size_t sh = norm_shift(shift, n);
    uint64_t* dst = (uint64_t*)d_dst; const uint64_t* l = (const uint64_t*)d_lhs; const uint64_t* r = (const uint64_t*)d_rhs;
    dim3 g(stream_grid(n)), b(msstage::NT);
    ProfScope ps(ctx, "stage_mul_pow", 8.0 * n * (2 * VL + VR));
    using namespace msstage;
    if (VL == 4) hipLaunchKernelGGL((k_mul_pow<Fp252T, Fp252T>), g, b, 0, ctx->stream, dst, l, r, n, sh, power);
    else if (VL == 1) hipLaunchKernelGGL((k_mul_pow<FpT, FpT>), g, b, 0, ctx->stream, dst, l, r, n, sh, power);
    else if (VR == 3) hipLaunchKernelGGL((k_mul_pow<Fq3T, Fq3T>), g, b, 0, ctx->stream, dst, l, r, n, sh, power);
    else hipLaunchKernelGGL((k_mul_pow<Fq3T, FpT>), g, b, 0, ctx->stream, dst, l, r, n, sh, power);
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_unary(ms_ctx* ctx, int op, int field, size_t n, void* d_dst, const void* d_src, unsigned exponent) {
    if (!ctx || !d_dst || !d_src) return fail(MS_ERR_INVALID, "ms_unary: null argument");
    if (op != MS_NEG && op != MS_INV && op != MS_EXP) return fail(MS_ERR_INVALID, "unknown unary op %d", op);
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    if (n == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    uint64_t* dst = (uint64_t*)d_dst; const uint64_t* src = (const uint64_t*)d_src;
    dim3 g(stream_grid(n)), b(msstage::NT);
    ProfScope ps(ctx, op == MS_NEG ? "stage_neg" : op == MS_INV ? "stage_inverse" : "stage_exp", 16.0 * n * V);
    using namespace msstage;
    if (V == 4) {
        if (op == MS_NEG) hipLaunchKernelGGL((k_unary<Fp252T, 0>), g, b, 0, ctx->stream, dst, src, n, exponent);
        else if (op == MS_INV) hipLaunchKernelGGL((k_unary<Fp252T, 1>), g, b, 0, ctx->stream, dst, src, n, exponent);
        else hipLaunchKernelGGL((k_unary<Fp252T, 2>), g, b, 0, ctx->stream, dst, src, n, exponent);
    } else if (V == 1) {
        if (op == MS_NEG) hipLaunchKernelGGL((k_unary<FpT, 0>), g, b, 0, ctx->stream, dst, src, n, exponent);
        else if (op == MS_INV) hipLaunchKernelGGL((k_unary<FpT, 1>), g, b, 0, ctx->stream, dst, src, n, exponent);
        else hipLaunchKernelGGL((k_unary<FpT, 2>), g, b, 0, ctx->stream, dst, src, n, exponent);
    } else {
        if (op == MS_NEG) hipLaunchKernelGGL((k_unary<FpT, 0>), dim3(stream_grid(3 * n)), b, 0, ctx->stream, dst, src, 3 * n, exponent);   // component-wise
        else if (op == MS_INV) hipLaunchKernelGGL((k_unary<Fq3T, 1>), g, b, 0, ctx->stream, dst, src, n, exponent);
        else hipLaunchKernelGGL((k_unary<Fq3T, 2>), g, b, 0, ctx->stream, dst, src, n, exponent);
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_convert(ms_ctx* ctx, int dst_field, int src_field, size_t n, void* d_dst, const void* d_src) {
    if (!ctx || !d_dst || !d_src) return fail(MS_ERR_INVALID, "ms_convert: null argument");
    unsigned VD = 0, VS = 0;
    MSCHK(field_pair(dst_field, src_field, &VD, &VS));
    if (n == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    if (VD == VS) {
        if (d_dst != d_src) HIPCHK(hipMemcpyAsync(d_dst, d_src, n * VD * 8, hipMemcpyDeviceToDevice, ctx->stream));
        return MS_OK;
    }
    if (VD != 3 || VS != 1) return fail(MS_ERR_UNSUPPORTED, "only the Fp -> Fq3 embedding exists");
    ProfScope ps(ctx, "stage_convert", 8.0 * n * (VD + VS));
    hipLaunchKernelGGL(msstage::k_convert_fp_fq3, dim3(stream_grid(n)), dim3(msstage::NT), 0, ctx->stream, (uint64_t*)d_dst, (const uint64_t*)d_src, n);
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_fill(ms_ctx* ctx, int field, size_t n, void* d_dst, const void* h_value) {
    if (!ctx || !d_dst || !h_value) return fail(MS_ERR_INVALID, "ms_fill: null argument");
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    if (n == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    msstage::Const3 c = {{0, 0, 0, 0}};
    memcpy(c.w, h_value, V * 8);
    ProfScope ps(ctx, "stage_fill", 8.0 * n * V);
    hipLaunchKernelGGL(msstage::k_fill, dim3(stream_grid(n * V)), dim3(msstage::NT), 0, ctx->stream, (uint64_t*)d_dst, c, n * V, V);
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_sum_columns(ms_ctx* ctx, int field, size_t n, const void* const* d_cols, unsigned ncols, void* d_dst) {
    if (!ctx || !d_cols || !d_dst) return fail(MS_ERR_INVALID, "ms_sum_columns: null argument");
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    if (ncols == 0) return fail(MS_ERR_INVALID, "sum of zero columns");
    if (ncols > (unsigned)msstage::MAXCOLS) return fail(MS_ERR_UNSUPPORTED, "at most %d columns per call", msstage::MAXCOLS);
    if (n == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    msstage::SumParams P;
    memset(&P, 0, sizeof P);
    for (unsigned c = 0; c < ncols; c++) P.cols[c] = (const uint64_t*)d_cols[c];
    P.dst = (uint64_t*)d_dst; P.nwords = n * V; P.ncols = ncols;
    ProfScope ps(ctx, "sum_columns", 8.0 * n * V * (ncols + 1));
    if (V == 4) { P.nwords = n; hipLaunchKernelGGL(msstage::k_sum_columns252, dim3(stream_grid(n)), dim3(msstage::NT), 0, ctx->stream, P); }
    else hipLaunchKernelGGL(msstage::k_sum_columns, dim3(stream_grid(n * V)), dim3(msstage::NT), 0, ctx->stream, P);
    HIPCHK(hipGetLastError());
    return MS_OK;
}

// ---------------------------------------------------------------------------------------
// FRI fold
// ---------------------------------------------------------------------------------------
template <int V>
static void launch_fold(unsigned ff, dim3 g, hipStream_t st, const msfri::FoldParams& P) {
    switch (ff) {
    case 2: hipLaunchKernelGGL((msfri::fri_fold<2, V>), g, dim3(msfri::NT), 0, st, P); break;
    case 4: hipLaunchKernelGGL((msfri::fri_fold<4, V>), g, dim3(msfri::NT), 0, st, P); break;
    case 8: hipLaunchKernelGGL((msfri::fri_fold<8, V>), g, dim3(msfri::NT), 0, st, P); break;
    default: hipLaunchKernelGGL((msfri::fri_fold<16, V>), g, dim3(msfri::NT), 0, st, P); break;
    }
}
extern "C" int ms_fri_fold(ms_ctx* ctx, int field, unsigned log_n, unsigned folding_factor, const void* h_alpha,
                           const void* h_offset, const void* d_evals, void* d_out) {
    if (!ctx || !h_alpha || !d_evals || !d_out) return fail(MS_ERR_INVALID, "ms_fri_fold: null argument");
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    if (folding_factor != 2 && folding_factor != 4 && folding_factor != 8 && folding_factor != 16)
        return fail(MS_ERR_UNSUPPORTED, "folding factor %u not supported (2, 4, 8, 16)", folding_factor);   // src/fri.rs:186-192
    unsigned log_ff = 0;
    while ((1u << log_ff) < folding_factor) log_ff++;
    if (log_n < log_ff || log_n > 32) return fail(MS_ERR_INVALID, "bad layer size 2^%u for folding factor %u", log_n, folding_factor);
    {   // lane c reads d_evals[c*ff .. c*ff + ff) and writes d_out[c]: overlapping buffers would corrupt the next layer
        const size_t in_bytes = ((size_t)1 << log_n) * V * 8, out_bytes = in_bytes / folding_factor;
        const char *a = (const char*)d_evals, *b = (const char*)d_out;
        if (a < b + out_bytes && b < a + in_bytes) return fail(MS_ERR_INVALID, "ms_fri_fold: d_out overlaps d_evals (the fold is not an in-place operation)");
    }
    if (V == 4) {
        f252::E h252 = f252::one();
        if (h_offset) memcpy(h252.l, h_offset, 32);
        if (f252::is_zero(h252) || f252::geq_p(h252)) return fail(MS_ERR_INVALID, "domain offset must be a non-zero canonical element");
        std::lock_guard<std::mutex> lk(ctx->mu);
        HIPCHK(hipSetDevice(ctx->device));
        ms_ntt_plan* plan = nullptr;
        MSCHK(plan252_cached(ctx, log_n, true, f252::one(), &plan));
        ms252::Fold252Params P;
        memset(&P, 0, sizeof P);
        P.src = (const uint64_t*)d_evals; P.dst = (uint64_t*)d_out;
        P.tw_lo = plan->d252_tw_lo; P.tw_hi = plan->d252_tw_hi; P.lo_bits = plan->lo_bits; P.log_m = log_n - log_ff;
        const f252::E hinv = f252::inv(h252);
        memcpy(P.hinv, hinv.l, 32);
        memcpy(P.alpha, h_alpha, 32);
        const f252::E zinv = f252::pow_u64(f252::inv(f252::root_of_unity(log_n)), (uint64_t)1 << (log_n - log_ff));
        f252::E zp = f252::one();
        for (unsigned k = 0; k < folding_factor / 2; k++) { memcpy(P.zinv[k], zp.l, 32); zp = f252::mul(zp, zinv); }
        const size_t m = (size_t)1 << (log_n - log_ff);
        dim3 g((unsigned)((m + ms252::NT - 1) / ms252::NT));
        ProfScope ps(ctx, "fri_fold252", 32.0 * (((size_t)1 << log_n) + m));
        switch (folding_factor) {
        case 2: hipLaunchKernelGGL(ms252::fri_fold252<2>, g, dim3(ms252::NT), 0, ctx->stream, P); break;
        case 4: hipLaunchKernelGGL(ms252::fri_fold252<4>, g, dim3(ms252::NT), 0, ctx->stream, P); break;
        case 8: hipLaunchKernelGGL(ms252::fri_fold252<8>, g, dim3(ms252::NT), 0, ctx->stream, P); break;
        default: hipLaunchKernelGGL(ms252::fri_fold252<16>, g, dim3(ms252::NT), 0, ctx->stream, P); break;
        }
        HIPCHK(hipGetLastError());
        return MS_OK;
    }
    uint64_t h = 1;
    if (h_offset) { uint64_t h_m; memcpy(&h_m, h_offset, 8); h = gl::from_mont(h_m); }
    if (h == 0) return fail(MS_ERR_INVALID, "domain offset must be non-zero");
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    // powers of w_n^-1: the tables of the size-n inverse plan (multi-pass layout needs log_n >= 12;
    // smaller layers get a dedicated two-level table through a plan of size max(n, 4096))
    ms_ntt_plan* plan = nullptr;
    const unsigned tl = std::max(log_n, 12u);
    MSCHK(ctx_plan(ctx, 1, tl, true, 1, &plan));
    msfri::FoldParams P;
    memset(&P, 0, sizeof P);
    P.src = (const uint64_t*)d_evals; P.dst = (uint64_t*)d_out;
    P.tw_lo = plan->d_tw_lo; P.tw_hi = plan->d_tw_hi; P.lo_bits = plan->lo_bits;
    P.log_m = log_n - log_ff;
    P.hinv = gl::to_mont(gl::inv(h));
    memcpy(P.alpha, h_alpha, V * 8);
    // table exponent scale: w_n = w_(2^tl)^(2^(tl-log_n)); fold it into the index below
    P.log_m |= (tl - log_n) << 8;
    const size_t m = (size_t)1 << (log_n - log_ff);
    dim3 g((unsigned)((m + msfri::NT - 1) / msfri::NT));
    ProfScope ps(ctx, "fri_fold", 8.0 * V * (((size_t)1 << log_n) + m));
    if (V == 1) launch_fold<1>(folding_factor, g, ctx->stream, P); else launch_fold<3>(folding_factor, g, ctx->stream, P);
    HIPCHK(hipGetLastError());
    return MS_OK;
}

// ---------------------------------------------------------------------------------------
// fused constraint evaluation
// ---------------------------------------------------------------------------------------
extern "C" int ms_eval_program(ms_ctx* ctx, const uint32_t* h_prog, unsigned ninstr, const void* h_consts, unsigned nconst_words,
                               unsigned log_n, unsigned lde_step, const void* h_domain_offset, const void* d_x_lde,
                               const void* const* d_base_cols, unsigned nbase, const void* const* d_ext_cols, unsigned next,
                               const void* const* d_periodic, const unsigned* periodic_len, unsigned nperiodic,
                               int out_field, void* d_out) {
    return ms_eval_program_ex(ctx, h_prog, ninstr, h_consts, nconst_words, log_n, lde_step, h_domain_offset, d_x_lde, d_base_cols, nbase,
                              d_ext_cols, next, d_periodic, periodic_len, nperiodic, out_field, d_out, 0u);
}
extern "C" int ms_eval_program_ex(ms_ctx* ctx, const uint32_t* h_prog, unsigned ninstr, const void* h_consts, unsigned nconst_words,
                                  unsigned log_n, unsigned lde_step, const void* h_domain_offset, const void* d_x_lde,
                                  const void* const* d_base_cols, unsigned nbase, const void* const* d_ext_cols, unsigned next,
                                  const void* const* d_periodic, const unsigned* periodic_len, unsigned nperiodic,
                                  int out_field, void* d_out, unsigned flags) {
    using namespace mseval;
    if (flags & ~(unsigned)MS_EVAL_BIT_REVERSED) return fail(MS_ERR_INVALID, "ms_eval_program_ex: unknown flags 0x%x", flags);
    if (!ctx || !h_prog || !d_out || (nconst_words && !h_consts)) return fail(MS_ERR_INVALID, "ms_eval_program: null argument");
    if (nbase > (unsigned)MAXCOLS || next > (unsigned)MAXCOLS) return fail(MS_ERR_UNSUPPORTED, "at most %d base and %d extension columns", MAXCOLS, MAXCOLS);
    if (nperiodic > 16u) return fail(MS_ERR_UNSUPPORTED, "at most 16 periodic columns");      // the other slots hold hoisted tables
    if ((nbase && !d_base_cols) || (next && !d_ext_cols) || (nperiodic && (!d_periodic || !periodic_len))) return fail(MS_ERR_INVALID, "ms_eval_program: null column table");
    if (log_n > 32) return fail(MS_ERR_INVALID, "log_n too large");
    if (lde_step == 0) return fail(MS_ERR_INVALID, "lde_step must be positive");
    const bool is252 = out_field == MS_STARK252_FP;          // Fq = Fp = Fp252: P-typed opcodes only, 4-word elements
    if (out_field != MS_GOLDILOCKS_FP && out_field != MS_GOLDILOCKS_FQ3 && !is252) return fail(MS_ERR_UNSUPPORTED, "unknown output field");
    const unsigned PW = is252 ? 4 : 1;
    // ---- validate: every register is written before it is read, all operands are in range
    unsigned maxp = 0, maxq = 0;
    std::vector<char> pw(256, 0), qw(128, 0);
    bool stored = false;
    const Instr* prog = (const Instr*)h_prog;
    auto P_ok = [&](uint32_t r) { return r < 256 && pw[r]; };
    auto Q_ok = [&](uint32_t r) { return r < 128 && qw[r]; };
    for (unsigned k = 0; k < ninstr; k++) {
        const Instr I = prog[k];
        bool ok = true, dp = false, dq = false;
        switch (I.op) {
        case OP_X_P: dp = true; break;
        case OP_CONST_P: ok = (uint64_t)I.a + PW <= nconst_words; dp = true; break;
        case OP_CONST_Q: ok = (uint64_t)I.a + 3 <= nconst_words; dq = true; break;
        case OP_TRACE_P: ok = I.a < nbase; dp = true; break;
        case OP_TRACE_Q: ok = I.a < next; dq = true; break;
        case OP_PERIODIC_P: ok = I.a < nperiodic && periodic_len[I.a] > 0; dp = true; break;
        case OP_PERIODIC_Q: ok = I.a < nperiodic && periodic_len[I.a] > 0; dq = true; break;
        case OP_NEG_P: case OP_INV_P: case OP_POW_P: ok = P_ok(I.a); dp = true; break;
        case OP_NEG_Q: case OP_INV_Q: case OP_POW_Q: ok = Q_ok(I.a); dq = true; break;
        case OP_ADD_PP: case OP_MUL_PP: ok = P_ok(I.a) && P_ok(I.b); dp = true; break;
        case OP_ADD_QQ: case OP_MUL_QQ: ok = Q_ok(I.a) && Q_ok(I.b); dq = true; break;
        case OP_ADD_QP: case OP_MUL_QP: ok = Q_ok(I.a) && P_ok(I.b); dq = true; break;
        case OP_EMBED: ok = P_ok(I.a); dq = true; break;
        case OP_STORE_Q: ok = Q_ok(I.a) && I.b == 0 && out_field == MS_GOLDILOCKS_FQ3; stored = true; break;
        case OP_STORE_P: ok = P_ok(I.a) && I.b == 0 && (out_field == MS_GOLDILOCKS_FP || is252); stored = true; break;
        default: ok = false;
        }
        if (is252 && (dq || I.op == OP_STORE_Q)) ok = false;
        if (dp) { if (I.dst >= 256) ok = false; else { pw[I.dst] = 1; maxp = std::max(maxp, I.dst + 1); } }
        if (dq) { if (I.dst >= 128) ok = false; else { qw[I.dst] = 1; maxq = std::max(maxq, I.dst + 1); } }
        if (!ok) return fail(MS_ERR_INVALID, "constraint program: invalid instruction %u (op %u dst %u a %u b %u)", k, I.op, I.dst, I.a, I.b);
    }
    if (!stored) return fail(MS_ERR_INVALID, "constraint program never stores a result");
    const size_t n = (size_t)1 << log_n;
    uint64_t h = 1;
    if (h_domain_offset && !is252) { uint64_t h_m; memcpy(&h_m, h_domain_offset, 8); h = gl::from_mont(h_m); }
    f252::E h252 = f252::one();
    if (h_domain_offset && is252) memcpy(h252.l, h_domain_offset, 32);
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    // ---- rewrite: short-period sub-expressions -> tables, long x^e chains -> twiddle lookups (eval_opt.h)
    static const bool no_opt = getenv("MS_EVAL_NO_HOIST") != nullptr;
    SplitProgram split;
    if (!no_opt) {
        split = split_periodic(prog, ninstr, log_n, d_x_lde == nullptr, periodic_len, nperiodic, (unsigned)MAXPERIODIC - nperiodic, PW);
        size_t words = 0;
        for (unsigned w : split.table_words) words += (size_t)w << split.log_period;
        if (words * 8 > ((size_t)64 << 20))                   // keep the tables cache-sized: fall back to short periods only
            split = split_periodic(prog, ninstr, log_n, d_x_lde == nullptr, periodic_len, nperiodic, (unsigned)MAXPERIODIC - nperiodic, PW, 12);
    }
    std::vector<uint64_t> consts((const uint64_t*)h_consts, (const uint64_t*)h_consts + nconst_words);
    for (auto& xp : split.xpows) {
        split.main[xp.instr].a = (uint32_t)consts.size();
        if (is252) { const f252::E v = f252::pow_u64(h252, xp.e); consts.insert(consts.end(), v.l, v.l + 4); }
        else consts.push_back(gl::to_mont(gl::pow(h, xp.e)));
    }
    if (getenv("MS_EVAL_DEBUG")) {
        fprintf(stderr, "split active=%d log_period=%u tables=%zu\n", (int)split.active, split.log_period, split.table_words.size());
        for (unsigned k = 0; k < ninstr; k++) fprintf(stderr, "  orig %3u: op %2u dst %u a %u b %u\n", k, prog[k].op, prog[k].dst, prog[k].a, prog[k].b);
        for (auto& I : split.prologue) fprintf(stderr, "  pro : op %2u dst %u a %u b %u\n", I.op, I.dst, I.a, I.b);
        for (auto& I : split.main) fprintf(stderr, "  main: op %2u dst %u a %u b %u\n", I.op, I.dst, I.a, I.b);
    }
    const unsigned h252_slot = (unsigned)consts.size();       // the Fp252 domain offset travels as one more constant
    if (is252) consts.insert(consts.end(), h252.l, h252.l + 4);
    const Instr* main_prog = split.active ? split.main.data() : prog;
    unsigned main_n = split.active ? (unsigned)split.main.size() : ninstr;
    const unsigned pro_n = (unsigned)split.prologue.size();
    // ---- rewrite 3: divisions by x-only denominators -> full-length tables, inverted in batches (eval_opt.h split_inversions)
    const unsigned short_tables = (unsigned)split.table_words.size();
    InvSplit isplit;
    if (log_n >= 12) isplit = split_inversions(main_prog, main_n, nperiodic + short_tables, (unsigned)MAXPERIODIC - nperiodic - short_tables, PW);
    if (isplit.active) { main_prog = isplit.main.data(); main_n = (unsigned)isplit.main.size(); }
    const unsigned den_n = (unsigned)isplit.denom.size();
    // ---- program(s) + constants -> device
    const size_t mbytes = (size_t)main_n * sizeof(Instr), pbytes = (size_t)pro_n * sizeof(Instr), dbytes = (size_t)den_n * sizeof(Instr), cbytes = consts.size() * 8;
    const size_t poff = (mbytes + 15) & ~(size_t)15, doff = (poff + pbytes + 15) & ~(size_t)15, coff = (doff + dbytes + 15) & ~(size_t)15, total = coff + cbytes + 64;
    HIPCHK(hipStreamSynchronize(ctx->stream));               // a previous evaluation may still read the buffer
    if (ctx->prog_bytes < total) {
        if (ctx->prog_buf) HIPCHK(hipFree(ctx->prog_buf));
        ctx->prog_buf = nullptr; ctx->prog_bytes = 0;
        if (hipMalloc(&ctx->prog_buf, total) != hipSuccess) return fail(MS_ERR_NOMEM, "program buffer");
        ctx->prog_bytes = total;
    }
    HIPCHK(hipMemcpy(ctx->prog_buf, main_prog, mbytes, hipMemcpyHostToDevice));
    if (pbytes) HIPCHK(hipMemcpy((char*)ctx->prog_buf + poff, split.prologue.data(), pbytes, hipMemcpyHostToDevice));
    if (dbytes) HIPCHK(hipMemcpy((char*)ctx->prog_buf + doff, isplit.denom.data(), dbytes, hipMemcpyHostToDevice));
    if (cbytes) HIPCHK(hipMemcpy((char*)ctx->prog_buf + coff, consts.data(), cbytes, hipMemcpyHostToDevice));
    EvalParams E;
    memset(&E, 0, sizeof E);
    E.consts = (const uint64_t*)((char*)ctx->prog_buf + coff);
    for (unsigned c = 0; c < nbase; c++) E.base_cols[c] = (const uint64_t*)d_base_cols[c];
    for (unsigned c = 0; c < next; c++) E.ext_cols[c] = (const uint64_t*)d_ext_cols[c];
    for (unsigned c = 0; c < nperiodic; c++) { E.periodic[c] = (const uint64_t*)d_periodic[c]; E.periodic_len[c] = periodic_len[c]; }
    E.out = (uint64_t*)d_out; E.x_lde = (const uint64_t*)d_x_lde;
    E.h_mont = is252 ? h252_slot : gl::to_mont(h); E.lde_step = lde_step;
    unsigned table_log = log_n;                               // domain the w table was built for
    if (!d_x_lde) {
        ms_ntt_plan* plan = nullptr;
        if (is252) {
            MSCHK(plan252_cached(ctx, log_n, false, f252::one(), &plan));
            E.tw_lo = plan->d252_tw_lo; E.tw_hi = plan->d252_tw_hi; E.lo_bits = plan->lo_bits;
        } else {
            table_log = std::max(log_n, 12u);
            MSCHK(ctx_plan(ctx, 1, table_log, false, 1, &plan));
            E.tw_lo = plan->d_tw_lo; E.tw_hi = plan->d_tw_hi; E.lo_bits = plan->lo_bits;
        }
    }
    // specialised kernel for a program (compiled on first use), or nullptr -> interpreter
    auto specialised = [&](const Instr* pr, unsigned cnt) -> hipFunction_t {
#ifndef MS_NO_JIT
        static const bool off = getenv("MS_EVAL_JIT") && !strcmp(getenv("MS_EVAL_JIT"), "0");
        if (off) return nullptr;
        const std::string src = jit_source(pr, cnt, is252, maxp, maxq);
        const std::string& key = src;
        auto it = ctx->jit_cache.find(key);
        if (it != ctx->jit_cache.end()) return it->second;
        hipFunction_t fn = nullptr;
        std::vector<char> code;
        std::string log;
        if (jit_compile(src, code, log)) {
            hipModule_t mod = nullptr;
            if (hipModuleLoadData(&mod, code.data()) == hipSuccess && hipModuleGetFunction(&fn, mod, "ms_eval_jit") == hipSuccess) ctx->jit_modules.push_back(mod);
            else { fn = nullptr; (void)hipGetLastError(); }
        } else if (getenv("MS_EVAL_DEBUG")) fprintf(stderr, "[ministark_hip] constraint kernel compilation failed, using the interpreter:\n%s\n", log.c_str());
        ctx->jit_cache[key] = fn;
        return fn;
#else
        (void)pr; (void)cnt;
        return nullptr;
#endif
    };
    auto launch = [&](const EvalParams& Q, hipFunction_t fn) {
        dim3 g((unsigned)((Q.n + NT - 1) / NT));
        if (fn) {
            EvalParams A = Q;
            void* args[] = {&A};
            if (hipModuleLaunchKernel(fn, g.x, 1, 1, 256, 1, 1, 0, ctx->stream, args, nullptr) == hipSuccess) return;
            // a module-API launch error does not reliably surface in hipGetLastError(): do not leave d_out unwritten,
            // run the interpreter instead and stop offering this kernel
            (void)hipGetLastError();
            for (auto& kv : ctx->jit_cache) if (kv.second == fn) kv.second = nullptr;
        }
        if (is252) {
            if (maxp <= 16) hipLaunchKernelGGL((eval_program252<16>), g, dim3(NT), 0, ctx->stream, Q);
            else if (maxp <= 64) hipLaunchKernelGGL((eval_program252<64>), g, dim3(NT), 0, ctx->stream, Q);
            else hipLaunchKernelGGL((eval_program252<256>), g, dim3(NT), 0, ctx->stream, Q);
        } else {
            if (maxp <= 16 && maxq <= 8) hipLaunchKernelGGL((eval_program<16, 8>), g, dim3(NT), 0, ctx->stream, Q);
            else if (maxp <= 64 && maxq <= 32) hipLaunchKernelGGL((eval_program<64, 32>), g, dim3(NT), 0, ctx->stream, Q);
            else hipLaunchKernelGGL((eval_program<256, 128>), g, dim3(NT), 0, ctx->stream, Q);
        }
    };
    // ---- prologue: the short-period values on the first 2^log_period points -> tables
    void* tables = nullptr;
    LockedPoolGuard pooled(ctx);                              // stream-ordered: the next user of a block queues behind these kernels
    if (pro_n) {
        const size_t period = (size_t)1 << split.log_period;
        size_t words = 0;
        for (unsigned w : split.table_words) words += w * period;
        MSCHK(pooled.alloc(words * 8, &tables));
        uint64_t* tp = (uint64_t*)tables;
        for (size_t t = 0; t < split.table_words.size(); t++) {
            E.periodic[nperiodic + t] = tp; E.periodic_len[nperiodic + t] = (uint32_t)period;
            tp += split.table_words[t] * period;
        }
        // Fp252 with a handful of points: one lane running the 252-bit Fermat inverse is ~0.6 ms of pure latency on
        // the device and microseconds on a host core -- the same fp252.h functions, so the same values
        bool on_host = false;
        if (is252 && period <= 64 && !d_x_lde) {
            on_host = true;
            for (auto& I : split.prologue) if (I.op == OP_PERIODIC_P) on_host = false;       // caller tables live on the device
        }
        if (on_host) {
            std::vector<uint64_t> host_tabs(words, 0);
            const f252::E w = f252::root_of_unity(log_n);
            f252::E xi = h252;                                                                // x_i = h * w^i
            std::vector<f252::E> rp(256);
            for (size_t i = 0; i < period; i++) {
                for (auto& I : split.prologue) {
                    switch (I.op) {
                    case OP_X_P: rp[I.dst] = xi; break;
                    case OP_CONST_P: memcpy(rp[I.dst].l, &consts[I.a], 32); break;
                    case OP_NEG_P: rp[I.dst] = f252::neg(rp[I.a]); break;
                    case OP_ADD_PP: rp[I.dst] = f252::add(rp[I.a], rp[I.b]); break;
                    case OP_MUL_PP: rp[I.dst] = f252::mul(rp[I.a], rp[I.b]); break;
                    case OP_INV_P: rp[I.dst] = f252::inv(rp[I.a]); break;
                    case OP_POW_P: rp[I.dst] = f252::pow_u64(rp[I.a], I.b); break;
                    case OP_STORE_P: {
                        size_t off = 0;
                        for (unsigned t = 0; t + nperiodic + 1 < I.b; t++) off += split.table_words[t] * period;
                        memcpy(&host_tabs[off + 4 * i], rp[I.a].l, 32);
                    } break;
                    default: break;
                    }
                }
                xi = f252::mul(xi, w);
            }
            HIPCHK(hipMemcpyAsync(tables, host_tabs.data(), words * 8, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));           // host_tabs is pageable and about to go out of scope
        } else {
            EvalParams Q = E;
            Q.prog = (const Instr*)((char*)ctx->prog_buf + poff); Q.ninstr = pro_n;
            Q.n = period; Q.log_n = split.log_period; Q.xshift = table_log - log_n;      // the first points of the same domain
        Q.bitrev = 0;
            ProfScope ps(ctx, "eval_prologue", 0.0);
            launch(Q, nullptr);                                   // runs on a few points: not worth a compilation
        }
    }
    E.prog = (const Instr*)ctx->prog_buf; E.ninstr = main_n; E.n = n; E.log_n = log_n; E.xshift = table_log - log_n;
    E.bitrev = (flags & MS_EVAL_BIT_REVERSED) ? 1 : 0;
    // ---- the x-only denominators of every point (in the launch's own layout), inverted in place
    void* inv_tables = nullptr;
    if (den_n) {
        size_t words = 0;
        for (unsigned w : isplit.table_words) words += (size_t)w * n;
        MSCHK(pooled.alloc(words * 8, &inv_tables));
        uint64_t* tp = (uint64_t*)inv_tables;
        for (size_t t = 0; t < isplit.table_words.size(); t++) {
            E.periodic[nperiodic + short_tables + t] = tp; E.periodic_len[nperiodic + short_tables + t] = (uint32_t)std::min<size_t>(n, 0xFFFFFFFFu);
            tp += (size_t)isplit.table_words[t] * n;
        }
        EvalParams Q = E;
        Q.prog = (const Instr*)((char*)ctx->prog_buf + doff); Q.ninstr = den_n;
        {
            hipFunction_t fn = n >= ((size_t)1 << 16) ? specialised(isplit.denom.data(), den_n) : nullptr;
            ProfScope ps(ctx, "eval_denominators", 0.0);
            launch(Q, fn);
        }
        tp = (uint64_t*)inv_tables;
        for (size_t t = 0; t < isplit.table_words.size(); t++) {
            const unsigned w = isplit.table_words[t];
            ProfScope ps(ctx, "eval_batch_inverse", 16.0 * w * n);
            if (w == 1) hipLaunchKernelGGL((batch_inverse<msstage::FpT, 16>), dim3((unsigned)((n + NT * 16 - 1) / (NT * 16))), dim3(NT), 0, ctx->stream, tp, n);
            else if (w == 3) hipLaunchKernelGGL((batch_inverse<msstage::Fq3T, 8>), dim3((unsigned)((n + NT * 8 - 1) / (NT * 8))), dim3(NT), 0, ctx->stream, tp, n);
            else if (n < ((size_t)1 << 16)) hipLaunchKernelGGL((batch_inverse<msstage::Fp252T, 8>), dim3((unsigned)((n + NT * 8 - 1) / (NT * 8))), dim3(NT), 0, ctx->stream, tp, n);
            else {                                             // two levels: one 252-bit Fermat inverse per 64 elements
                const unsigned blocks = (unsigned)(n / (NT * 8));
                const size_t m = (size_t)blocks * NT;          // lanes of the sweep = entries of the product array
                void* prod = nullptr;
                MSCHK(pooled.alloc(m * 32, &prod));
                hipLaunchKernelGGL((batch_inverse_up<msstage::Fp252T, 8>), dim3(blocks), dim3(NT), 0, ctx->stream, tp, n, (uint64_t*)prod);
                hipLaunchKernelGGL((batch_inverse<msstage::Fp252T, 8>), dim3((unsigned)((m + NT * 8 - 1) / (NT * 8))), dim3(NT), 0, ctx->stream, (uint64_t*)prod, m);
                hipLaunchKernelGGL((batch_inverse_down<msstage::Fp252T, 8>), dim3(blocks), dim3(NT), 0, ctx->stream, tp, n, (const uint64_t*)prod);
            }
            tp += (size_t)w * n;
        }
    }
    {
        hipFunction_t fn = n >= ((size_t)1 << 16) ? specialised(main_prog, main_n) : nullptr;   // small domains: the interpreter is quicker than a compilation
        ProfScope ps(ctx, fn ? (is252 ? "eval_program252_jit" : "eval_program_jit") : (is252 ? "eval_program252" : "eval_program"),
                     is252 ? 32.0 * n * (nbase + 1) : 8.0 * n * (nbase + 3.0 * next + (out_field == MS_GOLDILOCKS_FQ3 ? 3 : 1)));
        launch(E, fn);
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}

extern "C" int ms_eval_jit_check(const uint32_t* h_prog, unsigned ninstr, int out_field, size_t* code_bytes) {
#ifndef MS_NO_JIT
    using namespace mseval;
    if (!h_prog || !code_bytes) return fail(MS_ERR_INVALID, "ms_eval_jit_check: null argument");
    const bool is252 = out_field == MS_STARK252_FP;
    const Instr* prog = (const Instr*)h_prog;
    unsigned maxp = 0, maxq = 0;
    for (unsigned k = 0; k < ninstr; k++) {
        if (prog[k].op >= OP_COUNT || prog[k].dst >= 256) return fail(MS_ERR_INVALID, "invalid instruction %u", k);
        if (op_is_store(prog[k].op)) continue;
        if (op_is_q_dst(prog[k].op)) maxq = std::max(maxq, prog[k].dst + 1); else maxp = std::max(maxp, prog[k].dst + 1);
    }
    std::vector<char> code;
    std::string log;
    if (!jit_compile(jit_source(prog, ninstr, is252, maxp, maxq), code, log)) return fail(MS_ERR_UNSUPPORTED, "hiprtc: %s", log.c_str());
    *code_bytes = code.size();
    return MS_OK;
#else
    (void)h_prog; (void)ninstr; (void)out_field; (void)code_bytes;
    return fail(MS_ERR_UNSUPPORTED, "built without hiprtc");
#endif
}

// ---------------------------------------------------------------------------------------
// running products / evaluations, query gathers
// ---------------------------------------------------------------------------------------
template <class F, bool HAS_A, bool HAS_B, int PER>
static void scan_launch_per(ms_ctx* ctx, const msscan::ScanParams& P) {
    using namespace msscan;
    { ProfScope ps(ctx, "scan_reduce", 8.0 * P.n * F::V * ((HAS_A ? 1 : 0) + (HAS_B ? 1 : 0)));
      hipLaunchKernelGGL((scan_reduce<F, HAS_A, HAS_B, PER>), dim3(P.nblocks), dim3(NT), 0, ctx->stream, P); }
    { ProfScope ps(ctx, "scan_blocks", 0.0);
      hipLaunchKernelGGL((scan_blocks<F, HAS_A, HAS_B>), dim3(1), dim3(NT), 0, ctx->stream, P); }
    { ProfScope ps(ctx, "scan_apply", 8.0 * P.n * F::V * (1 + (HAS_A ? 1 : 0) + (HAS_B ? 1 : 0)));
      hipLaunchKernelGGL((scan_apply<F, HAS_A, HAS_B, PER>), dim3(P.nblocks), dim3(NT), 0, ctx->stream, P); }
}
static unsigned scan_rows_per_lane(size_t n) { return n < ((size_t)1 << 20) ? 4 : 16; }
template <class F, bool HAS_A, bool HAS_B>
static void scan_launch(ms_ctx* ctx, const msscan::ScanParams& P) {
    if (scan_rows_per_lane(P.n) == 4) scan_launch_per<F, HAS_A, HAS_B, 4>(ctx, P); else scan_launch_per<F, HAS_A, HAS_B, 16>(ctx, P);
}
extern "C" int ms_scan_affine(ms_ctx* ctx, int field, size_t n, const void* d_a, const void* d_b, const void* h_init, int inclusive, void* d_out) {
    if (!ctx || !d_out || !h_init) return fail(MS_ERR_INVALID, "ms_scan_affine: null argument");
    if (!d_a && !d_b) return fail(MS_ERR_INVALID, "ms_scan_affine: neither multipliers nor addends given");
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    if (n == 0) return MS_OK;
    const size_t tile = (size_t)msscan::NT * scan_rows_per_lane(n);
    if ((n + tile - 1) / tile > 0xFFFFFFFFull) return fail(MS_ERR_UNSUPPORTED, "column too long");
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    msscan::ScanParams P;
    memset(&P, 0, sizeof P);
    P.a = (const uint64_t*)d_a; P.b = (const uint64_t*)d_b; P.out = (uint64_t*)d_out;
    memcpy(P.init, h_init, V * 8);
    P.n = n; P.nblocks = (unsigned)((n + tile - 1) / tile); P.inclusive = inclusive != 0;
    void* tmp = nullptr;
    LockedPoolGuard pooled(ctx);
    MSCHK(pooled.alloc((size_t)P.nblocks * 3 * V * 8, &tmp));
    P.agg = (uint64_t*)tmp; P.block_state = (uint64_t*)tmp + (size_t)P.nblocks * 2 * V;
    using msstage::FpT; using msstage::Fq3T; using msstage::Fp252T;
    if (V == 1) {
        if (d_a && d_b) scan_launch<FpT, true, true>(ctx, P); else if (d_a) scan_launch<FpT, true, false>(ctx, P); else scan_launch<FpT, false, true>(ctx, P);
    } else if (V == 4) {
        if (d_a && d_b) scan_launch<Fp252T, true, true>(ctx, P); else if (d_a) scan_launch<Fp252T, true, false>(ctx, P); else scan_launch<Fp252T, false, true>(ctx, P);
    } else {
        if (d_a && d_b) scan_launch<Fq3T, true, true>(ctx, P); else if (d_a) scan_launch<Fq3T, true, false>(ctx, P); else scan_launch<Fq3T, false, true>(ctx, P);
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_gather_rows(ms_ctx* ctx, int field, size_t nrows, const void* const* d_cols, unsigned ncols,
                              const uint64_t* h_positions, size_t npos, void* d_out) {
    if (!ctx || !d_cols || !d_out || (npos && !h_positions)) return fail(MS_ERR_INVALID, "ms_gather_rows: null argument");
    const size_t fb = ms_field_bytes(field);
    if (!fb) return fail(MS_ERR_UNSUPPORTED, "unknown field %d", field);
    if (ncols == 0 || ncols > (unsigned)msstage::MAXCOLS) return fail(MS_ERR_UNSUPPORTED, "1..%d columns", msstage::MAXCOLS);
    for (size_t p = 0; p < npos; p++) if (h_positions[p] >= nrows) return fail(MS_ERR_INVALID, "row %llu out of range", (unsigned long long)h_positions[p]);
    if (npos == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    void* d_pos = nullptr;
    LockedPoolGuard pooled(ctx);
    MSCHK(pooled.alloc(npos * 8, &d_pos));
    HIPCHK(hipMemcpyAsync(d_pos, h_positions, npos * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));               // h_positions is pageable caller memory
    msscan::GatherRowsParams P;
    memset(&P, 0, sizeof P);
    for (unsigned c = 0; c < ncols; c++) { if (!d_cols[c]) return fail(MS_ERR_INVALID, "null column %u", c); P.cols[c] = (const uint64_t*)d_cols[c]; }
    P.pos = (const uint64_t*)d_pos; P.out = (uint64_t*)d_out; P.npos = npos; P.ncols = ncols; P.V = (unsigned)(fb / 8);
    const size_t total = npos * ncols * P.V;
    { ProfScope ps(ctx, "gather_rows", 16.0 * total);
      hipLaunchKernelGGL(msscan::gather_rows, dim3(stream_grid(total)), dim3(msscan::NT), 0, ctx->stream, P); }
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_gather_digests(ms_ctx* ctx, size_t ndigests, const void* d_digests, const uint64_t* h_indices, size_t count, void* d_out) {
    if (!ctx || !d_digests || !d_out || (count && !h_indices)) return fail(MS_ERR_INVALID, "ms_gather_digests: null argument");
    for (size_t k = 0; k < count; k++) if (h_indices[k] >= ndigests) return fail(MS_ERR_INVALID, "digest %llu out of range", (unsigned long long)h_indices[k]);
    if (count == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    void* d_idx = nullptr;
    LockedPoolGuard pooled(ctx);
    MSCHK(pooled.alloc(count * 8, &d_idx));
    HIPCHK(hipMemcpyAsync(d_idx, h_indices, count * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    { ProfScope ps(ctx, "gather_digests", 64.0 * count);
      hipLaunchKernelGGL(msscan::gather_records, dim3(stream_grid(count * 4)), dim3(msscan::NT), 0, ctx->stream,
                         (const uint64_t*)d_digests, (const uint64_t*)d_idx, (uint64_t*)d_out, count, 4u); }
    HIPCHK(hipGetLastError());
    return MS_OK;
}

// ---------------------------------------------------------------------------------------
// RPO-256 commitments
// ---------------------------------------------------------------------------------------
static int rpo_rows(ms_ctx* ctx, size_t nrows, const uint64_t* const* cols, unsigned ncols, unsigned stride, void* d_digests) {
    if (ncols == 0) return fail(MS_ERR_INVALID, "the zero-length input is not allowed");          // plan.rs:72
    if (ncols > (unsigned)msrpo::MAXCOLS) return fail(MS_ERR_UNSUPPORTED, "at most %d columns per commitment", msrpo::MAXCOLS);
    if (nrows == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    msrpo::RowsParams P;
    memset(&P, 0, sizeof P);
    for (unsigned c = 0; c < ncols; c++) P.cols[c] = cols[c];
    P.digests = (uint64_t*)d_digests; P.nrows = nrows; P.ncols = ncols; P.row_stride = stride;
    ProfScope ps(ctx, "rpo256_rows", 8.0 * nrows * (ncols + 4));
    hipLaunchKernelGGL(msrpo::rpo256_rows, dim3((unsigned)((nrows + msrpo::NT - 1) / msrpo::NT)), dim3(msrpo::NT), 0, ctx->stream, P);
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_rpo256_rows(ms_ctx* ctx, size_t nrows, const void* const* d_cols, unsigned ncols, void* d_digests) {
    if (!ctx || !d_cols || !d_digests) return fail(MS_ERR_INVALID, "ms_rpo256_rows: null argument");
    std::vector<const uint64_t*> cols(ncols);
    for (unsigned c = 0; c < ncols; c++) cols[c] = (const uint64_t*)d_cols[c];
    return rpo_rows(ctx, nrows, cols.data(), ncols, 1, d_digests);
}
// rows of a column-major matrix of `field`: an Fq3 column contributes its components c0, c1, c2 in the order
// the SHA-256 leaves serialise them (src/hash.rs:93-98) -- the column pointers are simply taken at word stride 3
extern "C" int ms_rpo256_rows_field(ms_ctx* ctx, int field, size_t nrows, const void* const* d_cols, unsigned ncols, void* d_digests) {
    if (!ctx || !d_cols || !d_digests) return fail(MS_ERR_INVALID, "ms_rpo256_rows_field: null argument");
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    if (V != 1 && V != 3) return fail(MS_ERR_UNSUPPORTED, "RPO-256 absorbs Goldilocks elements (Fp or Fq3 columns)");
    std::vector<const uint64_t*> cols;
    for (unsigned c = 0; c < ncols; c++) {
        if (!d_cols[c]) return fail(MS_ERR_INVALID, "null column %u", c);
        for (unsigned k = 0; k < V; k++) cols.push_back((const uint64_t*)d_cols[c] + k);
    }
    return rpo_rows(ctx, nrows, cols.data(), (unsigned)cols.size(), V, d_digests);
}
extern "C" int ms_rpo256_rows_row_major(ms_ctx* ctx, size_t nrows, unsigned ncols, const void* d_matrix, void* d_digests) {
    if (!ctx || !d_matrix || !d_digests) return fail(MS_ERR_INVALID, "ms_rpo256_rows_row_major: null argument");
    std::vector<const uint64_t*> cols(ncols);
    for (unsigned c = 0; c < ncols; c++) cols[c] = (const uint64_t*)d_matrix + c;
    return rpo_rows(ctx, nrows, cols.data(), ncols, ncols, d_digests);
}
extern "C" int ms_rpo256_merkle(ms_ctx* ctx, size_t nleaves, const void* d_leaves, void* d_nodes) {
    if (!ctx || !d_leaves || !d_nodes) return fail(MS_ERR_INVALID, "ms_rpo256_merkle: null argument");
    if (nleaves < 2 || (nleaves & (nleaves - 1))) return fail(MS_ERR_INVALID, "number of leaves must be a power of two >= 2");
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    uint64_t* nodes = (uint64_t*)d_nodes;
    HIPCHK(hipMemsetAsync(nodes, 0, 32, ctx->stream));
    const uint64_t* src = (const uint64_t*)d_leaves;
    for (size_t count = nleaves / 2; count >= 1; count >>= 1) {
        uint64_t* dst = nodes + count * 4;
        ProfScope ps(ctx, "rpo256_merkle_level", 96.0 * count);
        hipLaunchKernelGGL(msrpo::rpo256_merge_level, dim3((unsigned)((count + msrpo::NT - 1) / msrpo::NT)), dim3(msrpo::NT), 0, ctx->stream, src, dst, count);
        src = dst;
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}

// ---------------------------------------------------------------------------------------
// DEEP composition
// ---------------------------------------------------------------------------------------
static int point_words(int point_field, unsigned* PW) {
    if (point_field == MS_GOLDILOCKS_FP) { *PW = 1; return MS_OK; }
    if (point_field == MS_GOLDILOCKS_FQ3) { *PW = 3; return MS_OK; }
    return fail(MS_ERR_UNSUPPORTED, "point field must be Goldilocks Fp or Fq3");
}
static gl::Fq3 q3_load(const uint64_t* p, unsigned PW) { return PW == 3 ? gl::Fq3{p[0], p[1], p[2]} : gl::Fq3{p[0], 0, 0}; }

// ---- the 252-bit instantiations (Fq = Fp = Fp252)
static int horner_eval252(ms_ctx* ctx, size_t n, const void* const* d_cols, unsigned ncols, const unsigned* h_qcol, const uint64_t* h_qpoints,
                          unsigned nq, uint64_t* h_out) {
    if (ncols > (unsigned)msdeep::MAXCOLS) return fail(MS_ERR_UNSUPPORTED, "at most %d columns per call", msdeep::MAXCOLS);
    if (nq == 0) return MS_OK;
    for (unsigned q = 0; q < nq; q++) if (h_qcol[q] >= ncols) return fail(MS_ERR_INVALID, "query %u names column %u of %u", q, h_qcol[q], ncols);
    const unsigned nblocks = (unsigned)std::max<size_t>(1, (n + 4095) / 4096);
    void *d_qcol = nullptr, *d_pts = nullptr, *d_part = nullptr;
    PoolGuard pooled(ctx);                                 // temporaries go back to the pool on every exit path
    MSCHK(pooled.alloc((size_t)nq * 4, &d_qcol));
    MSCHK(pooled.alloc((size_t)nq * 32, &d_pts));
    MSCHK(pooled.alloc((size_t)nq * nblocks * 32, &d_part));
    std::vector<uint64_t> part((size_t)nq * nblocks * 4);
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        HIPCHK(hipSetDevice(ctx->device));
        HIPCHK(hipMemcpyAsync(d_qcol, h_qcol, (size_t)nq * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(d_pts, h_qpoints, (size_t)nq * 32, hipMemcpyHostToDevice, ctx->stream));
        msdeep252::HornerParams H;
        memset(&H, 0, sizeof H);
        for (unsigned c = 0; c < ncols; c++) H.cols[c] = (const uint64_t*)d_cols[c];
        H.qcol = (const uint32_t*)d_qcol; H.qpoint = (const uint64_t*)d_pts; H.partial = (uint64_t*)d_part; H.n = n; H.nblocks = nblocks;
        {
            ProfScope ps(ctx, "horner_blocks252", 32.0 * n * nq);
            hipLaunchKernelGGL(msdeep252::horner_blocks, dim3(nblocks, nq), dim3(msdeep252::NT), 0, ctx->stream, H);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(part.data(), d_part, part.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    for (unsigned q = 0; q < nq; q++) {                       // sum_b E_b * (x^4096)^b
        f252::E xb;
        memcpy(xb.l, h_qpoints + 4 * (size_t)q, 32);
        for (int sq = 0; sq < 12; sq++) xb = f252::mul(xb, xb);
        f252::E acc = f252::zero();
        for (unsigned b = nblocks; b-- > 0;) {
            f252::E e;
            memcpy(e.l, &part[((size_t)q * nblocks + b) * 4], 32);
            acc = f252::add(f252::mul(acc, xb), e);
        }
        memcpy(h_out + 4 * (size_t)q, acc.l, 32);
    }
    return MS_OK;
}
static uint64_t offset_key252(const f252::E& h) {
    uint64_t k = 1469598103934665603ull;
    for (int w = 0; w < 4; w++) { k ^= h.l[w]; k *= 1099511628211ull; }
    return k | ((uint64_t)1 << 63);
}
static int plan252_cached(ms_ctx* ctx, unsigned log_n, bool inverse, const f252::E& h, ms_ntt_plan** out) {
    const bool coset = !f252::eq(h, f252::one());
    const uint64_t key = coset ? offset_key252(h) : 1;
    if ((*out = plan_cache_find(ctx, 4, log_n, inverse, key, h.l)) != nullptr) return MS_OK;
    MSCHK(plan_build252(ctx, log_n, inverse, h.l, nullptr, out));
    plan_cache_insert(ctx, PlanKey{4, log_n, inverse, key}, *out);
    return MS_OK;
}
static int deep_compose252(ms_ctx* ctx, unsigned log_n, const void* h_offset, const void* const* d_polys, unsigned ncols,
                           const uint64_t* h_points, unsigned npoints, const unsigned* h_term_col, const unsigned* h_term_point,
                           const uint64_t* h_term_alpha, const uint64_t* h_term_ood, unsigned nterms,
                           const uint64_t* h_degree_alpha, const uint64_t* h_degree_beta, void* d_out) {
    if (ncols > (unsigned)msdeep::MAXCOLS) return fail(MS_ERR_UNSUPPORTED, "at most %d columns", msdeep::MAXCOLS);
    if (npoints == 0 || npoints > (unsigned)msdeep::MAXPOINTS) return fail(MS_ERR_UNSUPPORTED, "1..%d distinct out-of-domain points", msdeep::MAXPOINTS);
    if (log_n > 40) return fail(MS_ERR_INVALID, "log_n too large");
    for (unsigned t = 0; t < nterms; t++)
        if (h_term_col[t] >= ncols || h_term_point[t] >= npoints) return fail(MS_ERR_INVALID, "term %u out of range", t);
    f252::E h = f252::to_mont(f252::E{{3, 0, 0, 0}});          // the field's generator (gpu/src/fields.rs:241)
    if (h_offset) memcpy(h.l, h_offset, 32);
    if (f252::is_zero(h) || f252::geq_p(h)) return fail(MS_ERR_INVALID, "coset offset must be a non-zero canonical element");
    const size_t n = (size_t)1 << log_n;
    std::vector<void*> ev(ncols, nullptr);
    void *d_terms = nullptr, *d_q = nullptr;
    PoolGuard pooled(ctx);                                 // temporaries go back to the pool on every exit path
    for (unsigned c = 0; c < ncols; c++) MSCHK(pooled.alloc(n * 32, &ev[c]));
    MSCHK(pooled.alloc(std::max<size_t>(1, nterms) * sizeof(msdeep252::Term), &d_terms));
    MSCHK(pooled.alloc(n * 32, &d_q));
    std::vector<msdeep252::Term> terms(nterms);
    for (unsigned t = 0; t < nterms; t++) {
        terms[t].col = h_term_col[t]; terms[t].point = h_term_point[t];
        memcpy(terms[t].alpha, h_term_alpha + 4 * (size_t)t, 32);
        memcpy(terms[t].ood, h_term_ood + 4 * (size_t)t, 32);
    }
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        HIPCHK(hipSetDevice(ctx->device));
        if (nterms) HIPCHK(hipMemcpyAsync(d_terms, terms.data(), nterms * sizeof(msdeep252::Term), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        ms_ntt_plan *fwd = nullptr, *inv = nullptr, *sub = nullptr;
        MSCHK(plan252_cached(ctx, log_n, false, h, &fwd));
        MSCHK(plan252_cached(ctx, log_n, true, h, &inv));
        MSCHK(plan252_cached(ctx, log_n, false, f252::one(), &sub));           // its tables hold w_n^i
        if (ncols) MSCHK(plan_run252(fwd, d_polys, ev.data(), ncols));
        msdeep252::DeepParams D;
        memset(&D, 0, sizeof D);
        for (unsigned c = 0; c < ncols; c++) D.cols[c] = (const uint64_t*)ev[c];
        D.terms = (const msdeep252::Term*)d_terms; D.tw_lo = sub->d252_tw_lo; D.tw_hi = sub->d252_tw_hi; D.lo_bits = sub->lo_bits;
        memcpy(D.points, h_points, (size_t)npoints * 32);
        memcpy(D.h, h.l, 32);
        D.out = (uint64_t*)d_q; D.n = n; D.nterms = nterms; D.npoints = npoints;
        const dim3 g((unsigned)((n + msdeep252::NT - 1) / msdeep252::NT));
        { ProfScope ps(ctx, "deep_points252", 32.0 * n * (ncols + 1));
          hipLaunchKernelGGL(msdeep252::deep_points, g, dim3(msdeep252::NT), 0, ctx->stream, D); }
        HIPCHK(hipGetLastError());
        const void* qsrc[1] = {d_q};
        void* qdst[1] = {d_q};
        MSCHK(plan_run252(inv, qsrc, qdst, 1));
        f252::E da, db;
        memcpy(da.l, h_degree_alpha, 32);
        memcpy(db.l, h_degree_beta, 32);
        { ProfScope ps(ctx, "deep_degree_adjust252", 64.0 * n);
          hipLaunchKernelGGL(msdeep252::deep_degree_adjust, g, dim3(msdeep252::NT), 0, ctx->stream, (uint64_t*)d_out, (const uint64_t*)d_q, n, da, db); }
        HIPCHK(hipGetLastError());
    }
    return MS_OK;
}

extern "C" int ms_horner_eval(ms_ctx* ctx, int coeff_field, int point_field, size_t n, const void* const* d_cols, unsigned ncols,
                              const unsigned* h_qcol, const void* h_qpoints, unsigned nq, void* h_out) {
    if (!ctx || !d_cols || !h_qcol || !h_qpoints || !h_out) return fail(MS_ERR_INVALID, "ms_horner_eval: null argument");
    if (coeff_field == MS_STARK252_FP || point_field == MS_STARK252_FP) {
        if (coeff_field != point_field) return fail(MS_ERR_UNSUPPORTED, "the 252-bit field has no extension: coefficients and points must both be Fp252");
        return horner_eval252(ctx, n, d_cols, ncols, h_qcol, (const uint64_t*)h_qpoints, nq, (uint64_t*)h_out);
    }
    unsigned CW = 0, PW = 0;
    MSCHK(point_words(coeff_field, &CW));
    MSCHK(point_words(point_field, &PW));
    if (CW > PW) return fail(MS_ERR_UNSUPPORTED, "coefficients must embed into the point field");
    if (ncols > (unsigned)msdeep::MAXCOLS) return fail(MS_ERR_UNSUPPORTED, "at most %d columns per call", msdeep::MAXCOLS);
    if (nq == 0) return MS_OK;
    for (unsigned q = 0; q < nq; q++) if (h_qcol[q] >= ncols) return fail(MS_ERR_INVALID, "query %u names column %u of %u", q, h_qcol[q], ncols);
    const unsigned nblocks = (unsigned)std::max<size_t>(1, (n + 4095) / 4096);
    // device staging: qcol (u32), qpoints (3 words), partials
    std::vector<uint64_t> pts((size_t)nq * 3, 0);
    for (unsigned q = 0; q < nq; q++) memcpy(&pts[3 * q], (const uint64_t*)h_qpoints + (size_t)q * PW, PW * 8);
    void *d_qcol = nullptr, *d_pts = nullptr, *d_part = nullptr;
    PoolGuard pooled(ctx);                                 // temporaries go back to the pool on every exit path
    MSCHK(pooled.alloc((size_t)nq * 4, &d_qcol));
    MSCHK(pooled.alloc((size_t)nq * 24, &d_pts));
    MSCHK(pooled.alloc((size_t)nq * nblocks * 24, &d_part));
    std::vector<uint64_t> part((size_t)nq * nblocks * 3);
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        HIPCHK(hipSetDevice(ctx->device));
        HIPCHK(hipMemcpyAsync(d_qcol, h_qcol, (size_t)nq * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(d_pts, pts.data(), (size_t)nq * 24, hipMemcpyHostToDevice, ctx->stream));
        msdeep::HornerParams H;
        memset(&H, 0, sizeof H);
        for (unsigned c = 0; c < ncols; c++) H.cols[c] = (const uint64_t*)d_cols[c];
        H.qcol = (const uint32_t*)d_qcol; H.qpoint = (const uint64_t*)d_pts; H.partial = (uint64_t*)d_part; H.n = n; H.nblocks = nblocks;
        dim3 g(nblocks, nq);
        {
            ProfScope ps(ctx, "horner_blocks", 8.0 * CW * n * nq);
            if (CW == 1 && PW == 1) hipLaunchKernelGGL((msdeep::horner_blocks<1, 1>), g, dim3(msdeep::NT), 0, ctx->stream, H);
            else if (CW == 1) hipLaunchKernelGGL((msdeep::horner_blocks<1, 3>), g, dim3(msdeep::NT), 0, ctx->stream, H);
            else hipLaunchKernelGGL((msdeep::horner_blocks<3, 3>), g, dim3(msdeep::NT), 0, ctx->stream, H);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(part.data(), d_part, part.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    // combine the block values on the host: sum_b E_b * (x^4096)^b
    for (unsigned q = 0; q < nq; q++) {
        gl::Fq3 x = q3_load(&pts[3 * q], 3), xb = x;
        for (int sq = 0; sq < 12; sq++) xb = gl::mont_mul(xb, xb);
        gl::Fq3 acc = {0, 0, 0};
        for (unsigned b = nblocks; b-- > 0;) acc = gl::add(gl::mont_mul(acc, xb), q3_load(&part[((size_t)q * nblocks + b) * 3], 3));
        uint64_t* o = (uint64_t*)h_out + (size_t)q * PW;
        o[0] = acc.c0;
        if (PW == 3) { o[1] = acc.c1; o[2] = acc.c2; }
    }
    return MS_OK;
}

extern "C" int ms_deep_compose(ms_ctx* ctx, int point_field, unsigned log_n, const void* h_offset,
                               const void* const* d_base_polys, unsigned nbase, const void* const* d_ext_polys, unsigned next,
                               const void* h_points, unsigned npoints, const unsigned* h_term_col, const unsigned* h_term_point,
                               const void* h_term_alpha, const void* h_term_ood, unsigned nterms,
                               const void* h_degree_alpha, const void* h_degree_beta, void* d_out) {
    if (!ctx || !h_points || !h_term_col || !h_term_point || !h_term_alpha || !h_term_ood || !h_degree_alpha || !h_degree_beta || !d_out)
        return fail(MS_ERR_INVALID, "ms_deep_compose: null argument");
    if (point_field == MS_STARK252_FP) {
        if (next) return fail(MS_ERR_INVALID, "the 252-bit field has no extension columns: pass every polynomial as a base column");
        if (nbase && !d_base_polys) return fail(MS_ERR_INVALID, "ms_deep_compose: null column table");
        return deep_compose252(ctx, log_n, h_offset, d_base_polys, nbase, (const uint64_t*)h_points, npoints, h_term_col, h_term_point,
                               (const uint64_t*)h_term_alpha, (const uint64_t*)h_term_ood, nterms, (const uint64_t*)h_degree_alpha,
                               (const uint64_t*)h_degree_beta, d_out);
    }
    unsigned PW = 0;
    MSCHK(point_words(point_field, &PW));
    if (PW == 1 && next) return fail(MS_ERR_INVALID, "extension columns need point_field = Fq3");
    if (nbase > (unsigned)msdeep::MAXCOLS || next > (unsigned)msdeep::MAXCOLS) return fail(MS_ERR_UNSUPPORTED, "at most %d columns of each kind", msdeep::MAXCOLS);
    if (npoints == 0 || npoints > (unsigned)msdeep::MAXPOINTS) return fail(MS_ERR_UNSUPPORTED, "1..%d distinct out-of-domain points", msdeep::MAXPOINTS);
    if ((nbase && !d_base_polys) || (next && !d_ext_polys)) return fail(MS_ERR_INVALID, "ms_deep_compose: null column table");
    if (log_n > 32) return fail(MS_ERR_INVALID, "log_n too large");
    for (unsigned t = 0; t < nterms; t++)
        if (h_term_col[t] >= nbase + next || h_term_point[t] >= npoints) return fail(MS_ERR_INVALID, "term %u out of range", t);
    uint64_t h = gl::GENERATOR;
    if (h_offset) { uint64_t h_m; memcpy(&h_m, h_offset, 8); h = gl::from_mont(h_m); }
    if (h == 0) return fail(MS_ERR_INVALID, "coset offset must be non-zero");
    const size_t n = (size_t)1 << log_n;
    // scratch: coset evaluations of every polynomial + the evaluation/coefficient column of Q
    std::vector<void*> ev(nbase + next, nullptr);
    void *d_terms = nullptr, *d_q = nullptr;
    PoolGuard pooled(ctx);                                 // temporaries go back to the pool on every exit path
    for (unsigned c = 0; c < nbase; c++) MSCHK(pooled.alloc(n * 8, &ev[c]));
    for (unsigned c = 0; c < next; c++) MSCHK(pooled.alloc(n * 24, &ev[nbase + c]));
    MSCHK(pooled.alloc(std::max<size_t>(1, nterms) * sizeof(msdeep::Term), &d_terms));
    MSCHK(pooled.alloc(n * PW * 8, &d_q));
    std::vector<msdeep::Term> terms(nterms);
    for (unsigned t = 0; t < nterms; t++) {
        memset(&terms[t], 0, sizeof(msdeep::Term));
        terms[t].col = h_term_col[t]; terms[t].point = h_term_point[t];
        memcpy(terms[t].alpha, (const uint64_t*)h_term_alpha + (size_t)t * PW, PW * 8);
        memcpy(terms[t].ood, (const uint64_t*)h_term_ood + (size_t)t * PW, PW * 8);
    }
    int rc = MS_OK;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        HIPCHK(hipSetDevice(ctx->device));
        if (nterms) HIPCHK(hipMemcpyAsync(d_terms, terms.data(), nterms * sizeof(msdeep::Term), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));            // `terms` is a host temporary
        ms_ntt_plan *f1 = nullptr, *f3 = nullptr, *inv = nullptr, *tw = nullptr;
        if (nbase) { MSCHK(ctx_plan(ctx, 1, log_n, false, h, &f1)); rc = plan_run(f1, d_base_polys, ev.data(), nbase, 256); }
        if (rc == MS_OK && next) { MSCHK(ctx_plan(ctx, 3, log_n, false, h, &f3)); rc = plan_run(f3, d_ext_polys, ev.data() + nbase, next, 256); }
        if (rc != MS_OK) return rc;
        const unsigned tl = std::max(log_n, 12u);
        MSCHK(ctx_plan(ctx, 1, tl, false, 1, &tw));
        msdeep::DeepParams D;
        memset(&D, 0, sizeof D);
        for (unsigned c = 0; c < nbase; c++) D.base[c] = (const uint64_t*)ev[c];
        for (unsigned c = 0; c < next; c++) D.ext[c] = (const uint64_t*)ev[nbase + c];
        D.terms = (const msdeep::Term*)d_terms; D.tw_lo = tw->d_tw_lo; D.tw_hi = tw->d_tw_hi; D.lo_bits = tw->lo_bits; D.xshift = tl - log_n;
        for (unsigned k = 0; k < npoints; k++) memcpy(D.points[k], (const uint64_t*)h_points + (size_t)k * PW, PW * 8);
        D.out = (uint64_t*)d_q; D.h_mont = gl::to_mont(h); D.n = n; D.nbase = nbase; D.nterms = nterms; D.npoints = npoints;
        dim3 g((unsigned)((n + msdeep::NT - 1) / msdeep::NT));
        {
            ProfScope ps(ctx, "deep_points", 8.0 * n * (nbase + 3.0 * next + PW));
            if (PW == 1) hipLaunchKernelGGL((msdeep::deep_points<1>), g, dim3(msdeep::NT), 0, ctx->stream, D);
            else hipLaunchKernelGGL((msdeep::deep_points<3>), g, dim3(msdeep::NT), 0, ctx->stream, D);
        }
        HIPCHK(hipGetLastError());
        MSCHK(ctx_plan(ctx, PW, log_n, true, h, &inv));
        const void* qsrc[1] = {d_q};
        void* qdst[1] = {d_q};
        MSCHK(plan_run(inv, qsrc, qdst, 1, 256));
        msdeep::Q da = {{0, 0, 0}}, db = {{0, 0, 0}};
        memcpy(da.w, h_degree_alpha, PW * 8);
        memcpy(db.w, h_degree_beta, PW * 8);
        {
            ProfScope ps(ctx, "deep_degree_adjust", 16.0 * n * PW);
            if (PW == 1) hipLaunchKernelGGL((msdeep::deep_degree_adjust<1>), g, dim3(msdeep::NT), 0, ctx->stream, (uint64_t*)d_out, (const uint64_t*)d_q, n, da, db);
            else hipLaunchKernelGGL((msdeep::deep_degree_adjust<3>), g, dim3(msdeep::NT), 0, ctx->stream, (uint64_t*)d_out, (const uint64_t*)d_q, n, da, db);
        }
        HIPCHK(hipGetLastError());
    }
    return MS_OK;
}

// ---------------------------------------------------------------------------------------
// proof-of-work grinding
// ---------------------------------------------------------------------------------------
extern "C" int ms_sha256_pow_grind(ms_ctx* ctx, const void* h_seed32, unsigned bits, uint64_t max_nonce, uint64_t* nonce) {
    if (!ctx || !h_seed32 || !nonce) return fail(MS_ERR_INVALID, "ms_sha256_pow_grind: null argument");
    if (bits > 64) return fail(MS_ERR_INVALID, "proof-of-work bits must be <= 64");
    void* d_found = nullptr;
    PoolGuard pooled(ctx);                                 // temporaries go back to the pool on every exit path
    MSCHK(pooled.alloc(8, &d_found));
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    mssha::PowParams P;
    const uint8_t* sb = (const uint8_t*)h_seed32;
    for (int q = 0; q < 8; q++) P.seed[q] = ((uint32_t)sb[4 * q] << 24) | ((uint32_t)sb[4 * q + 1] << 16) | ((uint32_t)sb[4 * q + 2] << 8) | sb[4 * q + 3];
    P.bits = bits; P.found = (unsigned long long*)d_found;
    unsigned long long window = 1ull << 12;             // grows to 2^24 nonces per launch
    unsigned long long none = ~0ull, found = ~0ull;
    int rc = MS_OK;
    for (unsigned long long base = 1; base <= max_nonce && rc == MS_OK; base += P.count, window = std::min(window * 4, 1ull << 24)) {
        P.base = base; P.count = std::min<unsigned long long>(window, max_nonce - base + 1);
        if (hipMemcpyAsync(d_found, &none, 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { rc = fail(MS_ERR_HIP, "pow: memcpy"); break; }
        {
            ProfScope ps(ctx, "sha256_pow_grind", 0.0);
            hipLaunchKernelGGL(mssha::sha256_pow_grind, dim3((unsigned)((P.count + mssha::NT - 1) / mssha::NT)), dim3(mssha::NT), 0, ctx->stream, P);
        }
        if (hipMemcpyAsync(&found, d_found, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = fail(MS_ERR_HIP, "pow: readback"); break; }
        if (found != none) break;
    }
    if (rc != MS_OK) return rc;
    if (found == none) return fail(MS_ERR_INVALID, "no nonce below %llu has %u leading zero bits", (unsigned long long)max_nonce, bits);
    *nonce = found;
    return MS_OK;
}

// ---------------------------------------------------------------------------------------
// multi-GPU exchange over RCCL.  librccl is loaded on first use: a single-GPU user never
// touches it, and a host that already carries an RCCL (torch does) gets that same copy by soname.
// ---------------------------------------------------------------------------------------
// The exchange schedule as data (both builds): ms_cols_to_rows_alltoall below issues exactly these operations.
extern "C" int ms_cols_to_rows_schedule(unsigned nranks, unsigned rank, unsigned my_ncols, unsigned total_cols, size_t blk_bytes,
                                        ms_xchg_op* ops, size_t cap, size_t* count) {
    if (!count || !nranks || rank >= nranks) return fail(MS_ERR_INVALID, "ms_cols_to_rows_schedule: rank %u of %u", rank, nranks);
    const unsigned G = nranks, me = rank;
    const unsigned mine = total_cols > me ? (total_cols - me + G - 1) / G : 0;       // columns c = me, me + G, ...
    if (my_ncols != mine) return fail(MS_ERR_INVALID, "rank %u of %u owns %u of %u columns, %u given", me, G, mine, total_cols, my_ncols);
    size_t k = 0;
    auto put = [&](uint32_t kind, uint32_t peer, uint32_t src_col, uint32_t dst_col, uint64_t off) {
        if (ops && k < cap) ops[k] = ms_xchg_op{kind, peer, src_col, dst_col, off, (uint64_t)blk_bytes};
        k++;
    };
    for (unsigned peer = 0; peer < G; peer++) {
        if (peer == me) continue;
        for (unsigned j = 0; j < my_ncols; j++) put(MS_XCHG_SEND, peer, j, 0, (uint64_t)peer * blk_bytes);      // my columns, the peer's rows
        for (unsigned c = peer; c < total_cols; c += G) put(MS_XCHG_RECV, peer, 0, c, 0);                        // the peer's columns, my rows
    }
    for (unsigned j = 0; j < my_ncols; j++) put(MS_XCHG_COPY, me, j, me + j * G, (uint64_t)me * blk_bytes);    // my own block never leaves the device
    *count = k;
    if (ops && k > cap) return fail(MS_ERR_INVALID, "ms_cols_to_rows_schedule: %zu operations, room for %zu", k, cap);
    return MS_OK;
}

#ifndef MS_EMU
#include <dlfcn.h>
#include <rccl/rccl.h>
namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;
int rccl_load() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return MS_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* nm : names) if ((h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
    if (!h) return fail(MS_ERR_UNSUPPORTED, "librccl.so.1 not found: %s", dlerror());
    RcclApi a;
    a.lib = h;
#define MS_SYM(field, name) do { *(void**)(&a.field) = dlsym(h, name); if (!a.field) return fail(MS_ERR_UNSUPPORTED, "librccl: missing symbol %s", name); } while (0)
    MS_SYM(GetUniqueId, "ncclGetUniqueId"); MS_SYM(CommInitRank, "ncclCommInitRank"); MS_SYM(CommDestroy, "ncclCommDestroy");
    MS_SYM(Send, "ncclSend"); MS_SYM(Recv, "ncclRecv"); MS_SYM(AllGather, "ncclAllGather");
    MS_SYM(GroupStart, "ncclGroupStart"); MS_SYM(GroupEnd, "ncclGroupEnd"); MS_SYM(GetErrorString, "ncclGetErrorString");
#undef MS_SYM
    g_rccl = a;
    return MS_OK;
}
}  // namespace
#define NCCLCHK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return fail(MS_ERR_HIP, "%s: %s", #call, g_rccl.GetErrorString(r_)); } while (0)

extern "C" int ms_comm_unique_id(void* h_id128) {
    if (!h_id128) return fail(MS_ERR_INVALID, "ms_comm_unique_id: null argument");
    static_assert(sizeof(ncclUniqueId) == MS_COMM_ID_BYTES, "RCCL unique id size");
    MSCHK(rccl_load());
    ncclUniqueId id;
    NCCLCHK(g_rccl.GetUniqueId(&id));
    memcpy(h_id128, &id, sizeof id);
    return MS_OK;
}
extern "C" int ms_comm_init(ms_ctx* ctx, int nranks, int rank, const void* h_id128) {
    if (!ctx || !h_id128) return fail(MS_ERR_INVALID, "ms_comm_init: null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks || (nranks & (nranks - 1))) return fail(MS_ERR_INVALID, "ms_comm_init: rank %d of %d (a power of two)", rank, nranks);
    MSCHK(rccl_load());
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->comm) return fail(MS_ERR_INVALID, "ms_comm_init: this context already has a communicator");
    HIPCHK(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, h_id128, sizeof id);
    ncclComm_t comm = nullptr;
    NCCLCHK(g_rccl.CommInitRank(&comm, nranks, id, rank));
    ctx->comm = comm; ctx->comm_rank = rank; ctx->comm_size = nranks;
    return MS_OK;
}
extern "C" int ms_comm_destroy(ms_ctx* ctx) {
    if (!ctx) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);                    // the other communicator entry points hold it too
    if (!ctx->comm) return MS_OK;
    (void)hipStreamSynchronize(ctx->stream);
    (void)g_rccl.CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr; ctx->comm_rank = 0; ctx->comm_size = 1;
    return MS_OK;
}
extern "C" int ms_comm_rank(ms_ctx* ctx, int* rank, int* nranks) {
    if (!ctx) return fail(MS_ERR_INVALID, "null context");
    if (rank) *rank = ctx->comm_rank;
    if (nranks) *nranks = ctx->comm_size;
    return MS_OK;
}
extern "C" int ms_cols_to_rows_alltoall(ms_ctx* ctx, int field, size_t nrows, const void* const* d_my_cols, unsigned my_ncols,
                                        unsigned total_cols, void* const* d_shard_cols) {
    if (!ctx || (my_ncols && !d_my_cols) || (total_cols && !d_shard_cols)) return fail(MS_ERR_INVALID, "ms_cols_to_rows_alltoall: null argument");
    if (!ctx->comm) return fail(MS_ERR_INVALID, "ms_cols_to_rows_alltoall: no communicator (ms_comm_init)");
    const size_t fb = ms_field_bytes(field);
    if (!fb) return fail(MS_ERR_UNSUPPORTED, "unknown field %d", field);
    const unsigned G = (unsigned)ctx->comm_size, me = (unsigned)ctx->comm_rank;
    if (nrows % G) return fail(MS_ERR_INVALID, "%zu rows do not split over %u ranks", nrows, G);
    for (unsigned j = 0; j < my_ncols; j++) if (!d_my_cols[j]) return fail(MS_ERR_INVALID, "null column %u", j);
    for (unsigned c = 0; c < total_cols; c++) if (!d_shard_cols[c]) return fail(MS_ERR_INVALID, "null shard column %u", c);
    const size_t blk = nrows / G * fb;                                                // bytes of one rank's rows of one column
    size_t nops = 0;
    MSCHK(ms_cols_to_rows_schedule(G, me, my_ncols, total_cols, blk, nullptr, 0, &nops));      // also checks the ownership count
    std::vector<ms_xchg_op> ops(nops);
    MSCHK(ms_cols_to_rows_schedule(G, me, my_ncols, total_cols, blk, ops.data(), nops, &nops));
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    ProfScope ps(ctx, "cols_to_rows_alltoall", (double)blk * total_cols * 2.0);
    // one group for all point-to-point operations; a failing call must not leave the group open (the communicator and this
    // thread would stay in group mode): remember the first error, always close the group, then report
    ncclResult_t first = ncclSuccess;
    const char* what = "";
    NCCLCHK(g_rccl.GroupStart());
    for (const ms_xchg_op& op : ops) {
        ncclResult_t r = ncclSuccess;
        if (op.kind == MS_XCHG_SEND) r = g_rccl.Send((const char*)d_my_cols[op.src_col] + op.src_offset, op.bytes, ncclUint8, (int)op.peer, comm, ctx->stream);
        else if (op.kind == MS_XCHG_RECV) r = g_rccl.Recv(d_shard_cols[op.dst_col], op.bytes, ncclUint8, (int)op.peer, comm, ctx->stream);
        if (r != ncclSuccess && first == ncclSuccess) { first = r; what = op.kind == MS_XCHG_SEND ? "ncclSend" : "ncclRecv"; }
        if (first != ncclSuccess) break;
    }
    const ncclResult_t rend = g_rccl.GroupEnd();
    if (first != ncclSuccess) return fail(MS_ERR_HIP, "%s: %s", what, g_rccl.GetErrorString(first));
    if (rend != ncclSuccess) return fail(MS_ERR_HIP, "ncclGroupEnd: %s", g_rccl.GetErrorString(rend));
    for (const ms_xchg_op& op : ops)
        if (op.kind == MS_XCHG_COPY)
            HIPCHK(hipMemcpyAsync(d_shard_cols[op.dst_col], (const char*)d_my_cols[op.src_col] + op.src_offset, op.bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return MS_OK;
}
extern "C" int ms_p2p_batch(ms_ctx* ctx, const ms_p2p_op* ops, size_t count) {
    if (!ctx || (count && !ops)) return fail(MS_ERR_INVALID, "ms_p2p_batch: null argument");
    if (!ctx->comm) return fail(MS_ERR_INVALID, "ms_p2p_batch: no communicator (ms_comm_init)");
    for (size_t k = 0; k < count; k++)
        if (ops[k].kind > MS_XCHG_RECV || (int)ops[k].peer >= ctx->comm_size || (int)ops[k].peer == ctx->comm_rank || !ops[k].d_ptr)
            return fail(MS_ERR_INVALID, "ms_p2p_batch: operation %zu (kind %u, peer %u)", k, ops[k].kind, ops[k].peer);
    if (!count) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    double bytes = 0;
    for (size_t k = 0; k < count; k++) bytes += (double)ops[k].bytes;
    ProfScope ps(ctx, "p2p_batch", bytes);
    ncclResult_t first = ncclSuccess;
    NCCLCHK(g_rccl.GroupStart());
    for (size_t k = 0; k < count && first == ncclSuccess; k++)
        first = ops[k].kind == MS_XCHG_SEND ? g_rccl.Send(ops[k].d_ptr, ops[k].bytes, ncclUint8, (int)ops[k].peer, comm, ctx->stream)
                                            : g_rccl.Recv(ops[k].d_ptr, ops[k].bytes, ncclUint8, (int)ops[k].peer, comm, ctx->stream);
    const ncclResult_t rend = g_rccl.GroupEnd();                 // always closed, see ms_cols_to_rows_alltoall
    if (first != ncclSuccess) return fail(MS_ERR_HIP, "ncclSend/ncclRecv: %s", g_rccl.GetErrorString(first));
    if (rend != ncclSuccess) return fail(MS_ERR_HIP, "ncclGroupEnd: %s", g_rccl.GetErrorString(rend));
    return MS_OK;
}
extern "C" int ms_allgather_digests(ms_ctx* ctx, const void* d_my_digest32, void* d_all_digests) {
    if (!ctx || !d_my_digest32 || !d_all_digests) return fail(MS_ERR_INVALID, "ms_allgather_digests: null argument");
    if (!ctx->comm) return fail(MS_ERR_INVALID, "ms_allgather_digests: no communicator (ms_comm_init)");
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    ProfScope ps(ctx, "allgather_digests", 32.0 * ctx->comm_size);
    NCCLCHK(g_rccl.AllGather(d_my_digest32, d_all_digests, 32, ncclUint8, (ncclComm_t)ctx->comm, ctx->stream));
    return MS_OK;
}
#else   // the execution-model simulator of tests/emu has no RCCL: the exchange is exercised there through the mirror's test hook
extern "C" int ms_comm_unique_id(void*) { return fail(MS_ERR_UNSUPPORTED, "RCCL is not part of the simulator build"); }
extern "C" int ms_comm_init(ms_ctx*, int, int, const void*) { return fail(MS_ERR_UNSUPPORTED, "RCCL is not part of the simulator build"); }
extern "C" int ms_comm_destroy(ms_ctx*) { return MS_OK; }
extern "C" int ms_comm_rank(ms_ctx* ctx, int* rank, int* nranks) { if (rank) *rank = 0; if (nranks) *nranks = 1; (void)ctx; return MS_OK; }
extern "C" int ms_cols_to_rows_alltoall(ms_ctx*, int, size_t, const void* const*, unsigned, unsigned, void* const*) { return fail(MS_ERR_UNSUPPORTED, "RCCL is not part of the simulator build"); }
extern "C" int ms_allgather_digests(ms_ctx*, const void*, void*) { return fail(MS_ERR_UNSUPPORTED, "RCCL is not part of the simulator build"); }
extern "C" int ms_p2p_batch(ms_ctx*, const ms_p2p_op*, size_t) { return fail(MS_ERR_UNSUPPORTED, "RCCL is not part of the simulator build"); }
#endif
