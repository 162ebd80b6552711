// TEST INFRASTRUCTURE ONLY -- fiber scheduler behind tests/emu/hip/hip_runtime.h.
#include <hip/hip_runtime.h>

uint3_emu threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace emu {

static constexpr size_t STACK = 256 * 1024;

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = false;
    uint3_emu tid;
};

static ucontext_t g_sched;
static Fiber* g_cur = nullptr;
static const std::function<void()>* g_body = nullptr;
static std::vector<char*> g_stack_pool;

static void trampoline() {
    (*g_body)();
    g_cur->done = true;
    swapcontext(&g_cur->ctx, &g_sched);
}

void yield_barrier() {
    Fiber* f = g_cur;
    swapcontext(&f->ctx, &g_sched);
}

void run_block_threads(unsigned nthreads, const std::function<void()>& body) {
    g_body = &body;
    while (g_stack_pool.size() < nthreads) g_stack_pool.push_back((char*)malloc(STACK));
    std::vector<Fiber> fibers(nthreads);
    for (unsigned i = 0; i < nthreads; i++) {
        Fiber& f = fibers[i];
        f.stack = g_stack_pool[i];
        unsigned x = i % blockDim.x, y = (i / blockDim.x) % blockDim.y, z = i / (blockDim.x * blockDim.y);
        f.tid = {x, y, z};
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STACK;
        f.ctx.uc_link = &g_sched;
        makecontext(&f.ctx, trampoline, 0);
    }
    unsigned remaining = nthreads;
    while (remaining) {
        unsigned finished_this_sweep = 0;
        for (unsigned i = 0; i < nthreads; i++) {
            Fiber& f = fibers[i];
            if (f.done) continue;
            g_cur = &f;
            threadIdx = f.tid;
            swapcontext(&g_sched, &f.ctx);
            if (f.done) finished_this_sweep++;
        }
        remaining -= finished_this_sweep;
        if (finished_this_sweep != 0 && remaining != 0) {
            // some threads exited while others wait at a barrier: legal in HIP only
            // if the exited ones never reach another barrier; keep sweeping.
        }
    }
}

}  // namespace emu
