// Multi-GPU exchange over RCCL.
#include "ms_internal.h"

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

#include <dlfcn.h>
#if !defined(MS_EMU) && __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// A build host without the RCCL development headers (a single-GPU box) still builds the library: the few types and
// constants of the NCCL API that the lazily loaded entry points need, as RCCL 2.x defines them.
extern "C" {
typedef struct ncclComm* ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
}
#endif
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
    // MS_RCCL_LIB: a library exporting the same nine entry points (test hook: tests/emu/fake_rccl.cpp runs them between CPU
    // processes over shared memory, so that the code below executes with more than one rank in the GPU-less container)
    void* h = nullptr;
    if (const char* override_path = getenv("MS_RCCL_LIB")) {
        if (!(h = dlopen(override_path, RTLD_NOW | RTLD_LOCAL))) return fail(MS_ERR_UNSUPPORTED, "MS_RCCL_LIB=%s: %s", override_path, dlerror());
    } else {
#ifdef MS_EMU
        return fail(MS_ERR_UNSUPPORTED, "RCCL is not part of the simulator build (MS_RCCL_LIB names a stand-in)");
#else
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) if ((h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
        if (!h) return fail(MS_ERR_UNSUPPORTED, "librccl.so.1 not found: %s", dlerror());
#endif
    }
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

