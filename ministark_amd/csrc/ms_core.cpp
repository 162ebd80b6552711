// Context, errors, pooled device memory, per-launch profiling (include/ministark_hip.h "runtime" / "memory").
#include "ms_internal.h"

// ---------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
int fail(int code, const char* fmt, ...) {
    char buf[4096];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
extern "C" const char* ms_last_error(void) { return g_last_error.c_str(); }
extern "C" size_t ms_field_bytes(int field) {
    switch (field) {
    case MS_GOLDILOCKS_FP: return 8;
    case MS_GOLDILOCKS_FQ3: return 24;
    case MS_STARK252_FP: return 32;
    default: return 0;
    }
}
int field_words(int field, unsigned* V) {
    if (field == MS_GOLDILOCKS_FP) { *V = 1; return MS_OK; }
    if (field == MS_GOLDILOCKS_FQ3) { *V = 3; return MS_OK; }
    if (field == MS_STARK252_FP) { *V = 4; return MS_OK; }
    return fail(MS_ERR_INVALID, "unknown field id %d", field);
}

// ---------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------
int ctx_scratch(ms_ctx* ctx, size_t bytes, void** out) {
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
extern "C" int ms_comm_destroy(ms_ctx* ctx);
extern "C" int ms_ctx_destroy(ms_ctx* ctx) {
    if (!ctx) return MS_OK;
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    (void)ms_comm_destroy(ctx);
    plans_release_ctx(ctx);                                    // handles the caller still holds (they refer to this context)
    for (auto& kv : ctx->plan_cache) plan_free_cached(kv.second);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    for (auto& kv : ctx->pool) (void)hipFree(kv.second);
    if (ctx->prog_buf) (void)hipFree(ctx->prog_buf);
    if (ctx->stage) (void)hipHostFree(ctx->stage);
    { std::lock_guard<std::mutex> lk(ctx->land_mu); if (ctx->land) (void)hipHostFree(ctx->land); ctx->land = nullptr; }
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
int pool_alloc(ms_ctx* ctx, size_t bytes, void** d_ptr) {
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
int pool_free(ms_ctx* ctx, void* d_ptr) {
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
static constexpr size_t STAGE_RING = (size_t)1 << 20;
static void stage_init(ms_ctx* ctx) {
    if (ctx->stage) return;
    void* p = nullptr;
    if (hipHostMalloc(&p, STAGE_RING, 0) == hipSuccess) { ctx->stage = (char*)p; ctx->stage_bytes = STAGE_RING; } else (void)hipGetLastError();
}
// a slot of the ring for `need` bytes; on wrap everything that still reads the ring (copies, kernels) has to be done first
static int stage_slot(ms_ctx* ctx, size_t need, char** slot) {
    if (ctx->stage_off + need > ctx->stage_bytes) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        ctx->stage_off = 0;
    }
    *slot = ctx->stage + ctx->stage_off;
    ctx->stage_off += need;
    return MS_OK;
}
int stage_upload(ms_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    if (!bytes) return MS_OK;
    stage_init(ctx);
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (!ctx->stage || need > ctx->stage_bytes / 2) {            // large or no pinned memory: pageable copy, wait for it
        HIPCHK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return MS_OK;
    }
    char* slot = nullptr;
    MSCHK(stage_slot(ctx, need, &slot));
    memcpy(slot, h_src, bytes);
    HIPCHK(hipMemcpyAsync(d_dst, slot, bytes, hipMemcpyHostToDevice, ctx->stream));
    return MS_OK;
}
// Small index lists (query positions, digest indices): the kernel reads them where they are -- pinned host memory is mapped into
// the device's address space -- so the call costs no copy command at all (a hipMemcpyAsync is 6-10 us of host time, more than the
// gather kernel it feeds; twenty of them per proof).  *d_view is valid for kernels enqueued on the context's stream before the
// ring wraps (stage_slot waits for the stream then).  Falls back to a pooled device copy for lists that do not fit.
int stage_view(ms_ctx* ctx, const void* h_src, size_t bytes, const void** d_view, LockedPoolGuard& pooled) {
    stage_init(ctx);
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (!ctx->stage || need > ctx->stage_bytes / 8) {
        void* d = nullptr;
        MSCHK(pooled.alloc(bytes, &d));
        MSCHK(stage_upload(ctx, d, h_src, bytes));
        *d_view = d;
        return MS_OK;
    }
    char* slot = nullptr;
    MSCHK(stage_slot(ctx, need, &slot));
    memcpy(slot, h_src, bytes);
    *d_view = slot;
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
extern "C" int ms_copy(ms_ctx* ctx, void* d_dst, const void* d_src, size_t bytes) {
    if (!ctx || (bytes && (!d_dst || !d_src))) return fail(MS_ERR_INVALID, "ms_copy: null argument");
    HIPCHK(hipSetDevice(ctx->device));
    if (bytes && d_dst != d_src) HIPCHK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return MS_OK;
}
extern "C" int ms_upload(ms_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    if (!ctx) return fail(MS_ERR_INVALID, "null context");
    if (bytes && (!d_dst || !h_src)) return fail(MS_ERR_INVALID, "ms_upload: null argument");
    if (!bytes) return MS_OK;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return MS_OK;
}
extern "C" int ms_download(ms_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
    if (!ctx) return fail(MS_ERR_INVALID, "null context");
    if (bytes && (!h_dst || !d_src)) return fail(MS_ERR_INVALID, "ms_download: null argument");
    if (!bytes) return MS_OK;
    HIPCHK(hipSetDevice(ctx->device));                            // several contexts (devices) may live in one process
    static constexpr size_t LAND = (size_t)256 << 10;
    if (bytes <= LAND) {                                          // small: through the pinned landing buffer
        std::lock_guard<std::mutex> lk(ctx->land_mu);
        if (!ctx->land) {
            void* p = nullptr;
            if (hipHostMalloc(&p, LAND, hipHostMallocPortable) == hipSuccess) ctx->land = (char*)p; else (void)hipGetLastError();
        }
        if (ctx->land) {
            HIPCHK(hipMemcpyAsync(ctx->land, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            memcpy(h_dst, ctx->land, bytes);
            return MS_OK;
        }
    }
    HIPCHK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return MS_OK;
}
