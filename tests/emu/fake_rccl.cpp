// TEST INFRASTRUCTURE ONLY.  The nine NCCL entry points ministark_amd/csrc/ms_comm.cpp binds with dlsym, implemented between the
// PROCESSES of a CPU test over POSIX shared memory, so that the product's own multi-GPU code -- ms_comm_init, the schedule-driven
// ms_cols_to_rows_alltoall, ms_p2p_batch, ms_allgather_digests -- executes with world sizes > 1 in the GPU-less container
// (the simulator's "device" memory is host memory).  Selected with MS_RCCL_LIB; never part of the product.
//
// Semantics kept from NCCL: point-to-point operations between one pair of ranks match in issue order; operations issued between
// ncclGroupStart and ncclGroupEnd make progress together (no ordering between different peers, so "everybody sends first" does not
// deadlock); ncclAllGather is a collective over all ranks.  One byte FIFO per ordered pair of ranks.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

namespace {
constexpr size_t CAP = 1 << 20;                      // bytes in flight per ordered pair
struct Channel { std::atomic<uint64_t> head, tail; char pad[48]; char data[CAP]; };
struct Header { std::atomic<uint32_t> ready, left; char pad[56]; };
struct Comm { int rank, n; char* base; size_t size; std::string name; };
struct Op { bool send; char* ptr; size_t bytes, done; int peer; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local Comm* g_comm = nullptr;

Channel* chan(Comm* c, int src, int dst) { return (Channel*)(c->base + sizeof(Header)) + ((size_t)src * c->n + dst); }

size_t progress(Comm* c, Op& op) {                   // move what fits right now; returns bytes moved
    Channel* ch = op.send ? chan(c, c->rank, op.peer) : chan(c, op.peer, c->rank);
    const uint64_t head = ch->head.load(std::memory_order_acquire), tail = ch->tail.load(std::memory_order_acquire);
    size_t room = op.send ? CAP - (size_t)(head - tail) : (size_t)(head - tail);
    size_t nb = op.bytes - op.done < room ? op.bytes - op.done : room;
    size_t moved = 0;
    while (moved < nb) {
        const size_t pos = (size_t)((op.send ? head : tail) + moved) % CAP;
        const size_t piece = nb - moved < CAP - pos ? nb - moved : CAP - pos;
        if (op.send) memcpy(ch->data + pos, op.ptr + op.done + moved, piece); else memcpy(op.ptr + op.done + moved, ch->data + pos, piece);
        moved += piece;
    }
    if (op.send) ch->head.store(head + nb, std::memory_order_release); else ch->tail.store(tail + nb, std::memory_order_release);
    op.done += nb;
    return nb;
}
int run(Comm* c, std::vector<Op>& ops) {
    size_t left = ops.size();
    timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    while (left) {
        size_t moved = 0;
        for (size_t k = 0; k < ops.size(); k++) {
            Op& op = ops[k];
            if (op.done == op.bytes) continue;
            bool first = true;                       // per (peer, direction) only the oldest unfinished operation may move: issue order
            for (size_t j = 0; j < k; j++) if (ops[j].done != ops[j].bytes && ops[j].peer == op.peer && ops[j].send == op.send) { first = false; break; }
            if (!first) continue;
            moved += progress(c, op);
            if (op.done == op.bytes) left--;
        }
        if (!moved) {
            sched_yield();
            timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
            if (t.tv_sec - t0.tv_sec > 120) { fprintf(stderr, "fake_rccl: rank %d stuck with %zu operations\n", c->rank, left); return 1; }
        }
    }
    ops.clear();
    return 0;
}
}  // namespace

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
int ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    timespec t; clock_gettime(CLOCK_REALTIME, &t);
    snprintf(id->internal, sizeof id->internal, "/msfake_%d_%ld", (int)getpid(), (long)t.tv_nsec);
    return 0;
}
int ncclCommInitRank(void** out, int nranks, ncclUniqueId id, int rank) {
    Comm* c = new Comm{rank, nranks, nullptr, sizeof(Header) + (size_t)nranks * nranks * sizeof(Channel), std::string(id.internal)};
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->size) != 0) { perror("fake_rccl: shm_open"); return 2; }
    } else {
        for (int tries = 0; tries < 60000; tries++) {          // wait for rank 0 to create and size the region
            fd = shm_open(c->name.c_str(), O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size == c->size) break;
            if (fd >= 0) { close(fd); fd = -1; }
            usleep(1000);
        }
        if (fd < 0) { fprintf(stderr, "fake_rccl: rank %d never saw %s\n", rank, c->name.c_str()); return 2; }
    }
    c->base = (char*)mmap(nullptr, c->size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->base == (char*)MAP_FAILED) { perror("fake_rccl: mmap"); return 2; }
    Header* h = (Header*)c->base;
    h->ready.fetch_add(1);
    for (int tries = 0; h->ready.load() < (uint32_t)nranks; tries++) { if (tries > 60000) return 2; usleep(1000); }
    *out = c;
    return 0;
}
int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    Header* h = (Header*)c->base;
    const uint32_t gone = h->left.fetch_add(1) + 1;
    if (gone == (uint32_t)c->n) shm_unlink(c->name.c_str());   // the last one out removes the name
    munmap(c->base, c->size);
    delete c;
    return 0;
}
int ncclGroupStart() { g_depth++; return 0; }
static int flush() { return g_comm && !g_ops.empty() ? run(g_comm, g_ops) : 0; }
int ncclGroupEnd() { if (--g_depth == 0) return flush(); return 0; }
int ncclSend(const void* buf, size_t count, int, int peer, void* comm, void*) {
    g_comm = (Comm*)comm;
    g_ops.push_back(Op{true, (char*)buf, count, 0, peer});
    return g_depth ? 0 : flush();
}
int ncclRecv(void* buf, size_t count, int, int peer, void* comm, void*) {
    g_comm = (Comm*)comm;
    g_ops.push_back(Op{false, (char*)buf, count, 0, peer});
    return g_depth ? 0 : flush();
}
int ncclAllGather(const void* send, void* recv, size_t count, int, void* comm, void*) {
    Comm* c = (Comm*)comm;
    std::vector<Op> ops;
    for (int r = 0; r < c->n; r++) {
        if (r == c->rank) { memmove((char*)recv + (size_t)r * count, send, count); continue; }
        ops.push_back(Op{true, (char*)send, count, 0, r});
        ops.push_back(Op{false, (char*)recv + (size_t)r * count, count, 0, r});
    }
    return run(c, ops);
}
const char* ncclGetErrorString(int rc) { return rc == 0 ? "no error" : rc == 1 ? "fake_rccl: no progress for 120 s" : "fake_rccl: shared memory set-up failed"; }
}
