// TEST INFRASTRUCTURE ONLY -- the simulator's stand-in for hiprtc: the generated source of a specialised constraint kernel
// (ministark_amd/csrc/eval_jit_source.h, the SAME text the product hands to hiprtc) is compiled by g++ against the simulator's
// hip_runtime.h into a shared object, loaded, and launched through the fiber scheduler like every other kernel.  What this checks without a
// GPU: that the generator's text compiles against eval_kernels.h and computes what the interpreter (and the oracle) computes for the same
// register program -- every opcode's line, operand order, the accumulator opcodes of the regrouping pass, signed row offsets.
//
// Switched on by MS_EMU_JIT=1 (off: the simulator runs the interpreter, as the product does without hiprtc).  Objects are kept in
// tests/emu/_build/jit/<hash of source + the simulator library's build time>.so and reused.
#pragma once
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "jit_cache.h"

#ifndef MS_EMU_DIR
#error "build_emu.py passes -DMS_EMU_DIR and -DMS_CSRC_DIR"
#endif

namespace emu_jit {

typedef void (*entry_t)(void**);

static inline bool enabled() { const char* e = getenv("MS_EMU_JIT"); return e && !strcmp(e, "1"); }

// -> the kernel's entry (hipFunction_t of the simulator = a function taking hipModuleLaunchKernel's argument array), or nullptr with `log`
static inline void* obtain(const std::string& src, std::string& log, bool* compiled) {
    *compiled = false;
    Dl_info self;
    if (!dladdr((const void*)&enabled, &self) || !self.dli_fname) { log = "dladdr failed"; return nullptr; }
    struct stat st;
    if (stat(self.dli_fname, &st) != 0) { log = "stat of the simulator library failed"; return nullptr; }
    const std::string full = src + "\nextern \"C\" void ms_eval_jit_emu_entry(void** args) { ms_eval_jit(*(mseval::EvalParams*)args[0]); }\n";
    msjit::Sha256 h;
    h.field(full);
    const long long stamp[2] = {(long long)st.st_mtim.tv_sec, (long long)st.st_mtim.tv_nsec};
    h.field(stamp, sizeof stamp);
    uint8_t dg[32];
    h.final(dg);
    const std::string dir = std::string(MS_EMU_DIR) + "/_build/jit";
    if (!msjit::mkdirs(dir)) { log = "cannot create " + dir; return nullptr; }
    const std::string stem = dir + "/" + msjit::hex(dg, 16), so = stem + ".so";
    if (access(so.c_str(), R_OK) != 0) {
        char tag[64];
        snprintf(tag, sizeof tag, ".%ld", (long)getpid());
        const std::string cpp = stem + tag + ".cpp", tmp = stem + tag + ".so", err = stem + tag + ".log";
        FILE* f = fopen(cpp.c_str(), "w");
        if (!f) { log = "cannot write " + cpp; return nullptr; }
        fputs(full.c_str(), f);
        fclose(f);
        const std::string cmd = std::string("g++ -O1 -std=c++17 -fPIC -shared -w -DMS_NO_JIT -I") + MS_EMU_DIR + " -I" + MS_CSRC_DIR + " " + cpp + " -o " + tmp + " " + self.dli_fname + " 2> " + err;
        const int rc = system(cmd.c_str());
        if (rc != 0) {
            log = "g++ failed on the generated source (" + cpp + "):\n";
            if (FILE* e = fopen(err.c_str(), "r")) { char buf[4096]; size_t k = fread(buf, 1, sizeof buf - 1, e); buf[k] = 0; log += buf; fclose(e); }
            return nullptr;
        }
        (void)unlink(err.c_str());
        (void)unlink(cpp.c_str());
        if (rename(tmp.c_str(), so.c_str()) != 0) { log = "rename failed"; return nullptr; }
        *compiled = true;
    }
    void* lib = dlopen(so.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!lib) { log = std::string("dlopen: ") + dlerror(); return nullptr; }
    void* fn = dlsym(lib, "ms_eval_jit_emu_entry");
    if (!fn) { log = "the entry point is missing"; return nullptr; }
    return fn;
}

}  // namespace emu_jit
