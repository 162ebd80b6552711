// Specialised constraint kernels: the register program of one AIR is turned into straight-line HIP
// (one call of an eval_kernels.h helper per instruction, registers as named locals) and compiled for
// gfx950 with hiprtc the first time it is evaluated.  Compared with the interpreter in eval_kernels.h
// there is no instruction fetch / decode, the register file is allocated by the compiler in VGPRs, trace
// loads are scheduled together ahead of the arithmetic and constant operands fold.  The arithmetic is
// the same device code (gl.h / gl_dev.h / stage_kernels.h / fp252.h are handed to hiprtc as in-memory
// headers), so results are bit-identical to the interpreter; the interpreter remains the path when
// hiprtc is unavailable or MS_EVAL_JIT=0.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "eval_kernels.h"
#include "eval_jit_source.h"
#include "jit_cache.h"
#include "ms_internal.h"

namespace mseval {

// {name, text} of every header the generated source includes (written by ministark_amd/build.py)
static const char* const kJitHeaders[][2] = {
#include "_embedded_headers.inc"
};
static constexpr int kJitNumHeaders = (int)(sizeof(kJitHeaders) / sizeof(kJitHeaders[0]));


static const char* const kJitOpts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-uninitialized", "-Wno-unused-value"};
static constexpr int kJitNumOpts = (int)(sizeof(kJitOpts) / sizeof(kJitOpts[0]));

// source -> gfx950 code object.  Returns false (with the compiler log) on failure.
static inline bool jit_compile(const std::string& src, std::vector<char>& code, std::string& log) {
    const char* hdr_src[kJitNumHeaders];
    const char* hdr_name[kJitNumHeaders];
    for (int h = 0; h < kJitNumHeaders; h++) { hdr_name[h] = kJitHeaders[h][0]; hdr_src[h] = kJitHeaders[h][1]; }
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "ms_eval_jit.hip", kJitNumHeaders, hdr_src, hdr_name) != HIPRTC_SUCCESS) { log = "hiprtcCreateProgram failed"; return false; }
    const hiprtcResult r = hiprtcCompileProgram(prog, kJitNumOpts, kJitOpts);
    size_t ls = 0;
    if (hiprtcGetProgramLogSize(prog, &ls) == HIPRTC_SUCCESS && ls > 1) { log.resize(ls); hiprtcGetProgramLog(prog, &log[0]); }
    bool ok = r == HIPRTC_SUCCESS;
    if (ok) {
        size_t cs = 0;
        ok = hiprtcGetCodeSize(prog, &cs) == HIPRTC_SUCCESS && cs > 0;
        if (ok) { code.resize(cs); ok = hiprtcGetCode(prog, code.data()) == HIPRTC_SUCCESS; }
    }
    hiprtcDestroyProgram(&prog);
    return ok;
}

// Key of a generated source in the on-disk cache (jit_cache.h): everything the code object depends on.  The prefix (options, compiler
// version, every embedded header) is hashed once per process.
static inline std::string jit_key(const std::string& src) {
    static const msjit::Sha256 prefix = [] {
        msjit::Sha256 s;
        s.field(std::string("ministark_hip specialised constraint kernel, cache format 1"));
        for (int o = 0; o < kJitNumOpts; o++) s.field(std::string(kJitOpts[o]));
        int ver[2] = {0, 0};
        (void)hiprtcVersion(&ver[0], &ver[1]);
        s.field(ver, sizeof ver);
        for (int h = 0; h < kJitNumHeaders; h++) { s.field(std::string(kJitHeaders[h][0])); s.field(kJitHeaders[h][1], strlen(kJitHeaders[h][1])); }
        return s;
    }();
    msjit::Sha256 s = prefix;
    s.field(src);
    uint8_t dg[32];
    s.final(dg);
    return msjit::hex(dg, 32);
}

static inline double jit_now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// The code object of a generated source: from the disk cache when a whole entry is there, else compiled (and stored).  `st` collects
// what happened (ms_eval_jit_stats); `force_compile` skips the read (an entry that hipModuleLoadData refused).
static inline bool jit_obtain(const std::string& src, std::vector<char>& code, std::string& log, JitStats& st, bool force_compile = false) {
    const std::string dir = msjit::cache_dir();
    const std::string key = dir.empty() ? std::string() : jit_key(src);
    if (!dir.empty() && !force_compile) {
        const double t0 = jit_now_ms();
        bool damaged = false;
        const bool hit = msjit::disk_load(dir, key, code, &damaged);
        if (damaged) st.damaged_entries++;
        if (hit) { st.from_disk++; st.load_ms += jit_now_ms() - t0; return true; }
    }
    const double t0 = jit_now_ms();
    const bool ok = jit_compile(src, code, log);
    st.compile_ms += jit_now_ms() - t0;
    if (!ok) { st.failures++; return false; }
    st.compiled++;
    if (!dir.empty()) (void)msjit::disk_store(dir, key, code);
    return true;
}

// A compilation that fails is a 5-10x slower evaluation (the interpreter), never a wrong one: said once per process on stderr,
// counted in ms_eval_jit_stats, the log kept in ms_last_error-style text by the caller.
static inline void jit_warn_failure(const std::string& log) {
    static std::atomic<bool> said{false};
    if (said.exchange(true)) return;
    fprintf(stderr, "[ministark_hip] WARNING: hiprtc could not compile a constraint kernel; this process evaluates that program with the interpreter "
                    "(same results, several times slower).  ms_eval_jit_stats() counts such programs.  Compiler log:\n%.2000s\n", log.c_str());
}

}  // namespace mseval
