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
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "eval_kernels.h"

namespace mseval {

// {name, text} of every header the generated source includes (written by ministark_amd/build.py)
static const char* const kJitHeaders[][2] = {
#include "_embedded_headers.inc"
};
static constexpr int kJitNumHeaders = (int)(sizeof(kJitHeaders) / sizeof(kJitHeaders[0]));

static inline std::string jit_source(const Instr* prog, unsigned ninstr, bool is252, unsigned maxp, unsigned maxq) {
    std::string s;
    s.reserve(4096 + (size_t)ninstr * 64);
    s += "#include \"eval_kernels.h\"\nusing namespace mseval;\n";
    s += "extern \"C\" __global__ void __launch_bounds__(256) ms_eval_jit(EvalParams P) {\n";
    s += "    using F3 = msstage::Fq3T; using F1 = msstage::FpT; using F4 = msstage::Fp252T;\n";
    s += "    const size_t R = (size_t)blockIdx.x * 256 + threadIdx.x;\n    if (R >= P.n) return;\n    const size_t i = ev_point(P, R);\n";
    char b[256];
    for (unsigned r = 0; r < maxp; r++) { snprintf(b, sizeof b, is252 ? "    f252::E p%u;\n" : "    uint64_t p%u;\n", r); s += b; }
    for (unsigned r = 0; r < maxq; r++) { snprintf(b, sizeof b, "    gl::Fq3 q%u;\n", r); s += b; }
    for (int r = 0; r < NACC; r++) { snprintf(b, sizeof b, is252 ? "    Acc19 acc%d;\n" : "    Acc6 acc%d;\n", r); s += b; }
    if (!is252) s += "    AccQ accq;\n";
    for (unsigned k = 0; k < ninstr; k++) {
        const Instr I = prog[k];
        const unsigned d = I.dst, x = I.a, y = I.b;
        b[0] = 0;
        if (is252) {
            switch (I.op) {
            case OP_X_P: snprintf(b, sizeof b, "p%u = ev252_x(P, i);", d); break;
            case OP_CONST_P: snprintf(b, sizeof b, "p%u = ev252_const(P, %uu);", d, x); break;
            case OP_TRACE_P: snprintf(b, sizeof b, "p%u = ev252_trace(P, i, %uu, %uu);", d, x, y); break;
            case OP_PERIODIC_P: snprintf(b, sizeof b, "p%u = ev252_periodic(P, i, %uu);", d, x); break;
            case OP_NEG_P: snprintf(b, sizeof b, "p%u = f252::neg(p%u);", d, x); break;
            case OP_ADD_PP: snprintf(b, sizeof b, "p%u = f252::add(p%u, p%u);", d, x, y); break;
            case OP_MUL_PP: snprintf(b, sizeof b, "p%u = f252::mul(p%u, p%u);", d, x, y); break;
            case OP_INV_P: snprintf(b, sizeof b, "p%u = f252::inv(p%u);", d, x); break;
            case OP_POW_P: snprintf(b, sizeof b, "p%u = msstage::powu<F4>(p%u, %uu);", d, x, y); break;
            case OP_STORE_P: snprintf(b, sizeof b, "ev252_store(P, R, %uu, p%u);", y, x); break;
            case OP_XPOW_P: snprintf(b, sizeof b, "p%u = ev252_xpow(P, i, %uu, %uu);", d, x, y); break;
            case OP_TABLE_P: snprintf(b, sizeof b, "p%u = ev252_table(P, R, i, %uu, %uu);", d, x, y); break;
            case OP_ACC_ZERO: snprintf(b, sizeof b, "acc_zero(acc%u);", d & (NACC - 1)); break;
            case OP_ACC_MACC: snprintf(b, sizeof b, "acc_macc(acc%u, p%u, P.consts, %uu);", d & (NACC - 1), x, y); break;
            case OP_ACC_MACP: snprintf(b, sizeof b, "acc_macp(acc%u, p%u, p%u);", d & (NACC - 1), x, y); break;
            case OP_ACC_RED: snprintf(b, sizeof b, "p%u = acc_reduce(acc%u);", d, x & (NACC - 1)); break;
            default: break;
            }
        } else {
            switch (I.op) {
            case OP_X_P: snprintf(b, sizeof b, "p%u = ev_x(P, i);", d); break;
            case OP_CONST_P: snprintf(b, sizeof b, "p%u = P.consts[%u];", d, x); break;
            case OP_CONST_Q: snprintf(b, sizeof b, "q%u = ev_const_q(P, %uu);", d, x); break;
            case OP_TRACE_P: snprintf(b, sizeof b, "p%u = ev_trace_p(P, i, %uu, %uu);", d, x, y); break;
            case OP_TRACE_Q: snprintf(b, sizeof b, "q%u = ev_trace_q(P, i, %uu, %uu);", d, x, y); break;
            case OP_PERIODIC_P: snprintf(b, sizeof b, "p%u = ev_periodic_p(P, i, %uu);", d, x); break;
            case OP_PERIODIC_Q: snprintf(b, sizeof b, "q%u = ev_periodic_q(P, i, %uu);", d, x); break;
            case OP_NEG_P: snprintf(b, sizeof b, "p%u = gl::neg(p%u);", d, x); break;
            case OP_NEG_Q: snprintf(b, sizeof b, "q%u = gl::neg(q%u);", d, x); break;
            case OP_ADD_PP: snprintf(b, sizeof b, "p%u = gl::add(p%u, p%u);", d, x, y); break;
            case OP_ADD_QQ: snprintf(b, sizeof b, "q%u = gl::add(q%u, q%u);", d, x, y); break;
            case OP_ADD_QP: snprintf(b, sizeof b, "q%u = msstage::Mix<F3, F1>::add(q%u, p%u);", d, x, y); break;
            case OP_MUL_PP: snprintf(b, sizeof b, "p%u = gld::mmul(p%u, p%u);", d, x, y); break;
            case OP_MUL_QQ: snprintf(b, sizeof b, "q%u = F3::mul(q%u, q%u);", d, x, y); break;
            case OP_MUL_QP: snprintf(b, sizeof b, "q%u = msstage::Mix<F3, F1>::mul(q%u, p%u);", d, x, y); break;
            case OP_INV_P: snprintf(b, sizeof b, "p%u = F1::inv(p%u);", d, x); break;
            case OP_INV_Q: snprintf(b, sizeof b, "q%u = F3::inv(q%u);", d, x); break;
            case OP_POW_P: snprintf(b, sizeof b, "p%u = msstage::powu<F1>(p%u, %uu);", d, x, y); break;
            case OP_POW_Q: snprintf(b, sizeof b, "q%u = msstage::powu<F3>(q%u, %uu);", d, x, y); break;
            case OP_EMBED: snprintf(b, sizeof b, "q%u = gl::Fq3{p%u, 0, 0};", d, x); break;
            case OP_STORE_Q: snprintf(b, sizeof b, "ev_store_q(P, R, %uu, q%u);", y, x); break;
            case OP_STORE_P: snprintf(b, sizeof b, "ev_store_p(P, R, %uu, p%u);", y, x); break;
            case OP_XPOW_P: snprintf(b, sizeof b, "p%u = ev_xpow(P, i, %uu, %uu);", d, x, y); break;
            case OP_TABLE_P: snprintf(b, sizeof b, "p%u = ev_table_p(P, R, i, %uu, %uu);", d, x, y); break;
            case OP_TABLE_Q: snprintf(b, sizeof b, "q%u = ev_table_q(P, R, %uu);", d, x); break;
            case OP_ACC_ZERO: snprintf(b, sizeof b, "acc_zero(acc%u);", d & (NACC - 1)); break;
            case OP_ACC_MACC: snprintf(b, sizeof b, "acc_macc(acc%u, p%u, P.consts, %uu);", d & (NACC - 1), x, y); break;
            case OP_ACC_MACP: snprintf(b, sizeof b, "acc_macp(acc%u, p%u, p%u);", d & (NACC - 1), x, y); break;
            case OP_ACC_RED: snprintf(b, sizeof b, "p%u = acc_reduce(acc%u);", d, x & (NACC - 1)); break;
            case OP_ACCQ_ZERO: snprintf(b, sizeof b, "acc_zero(accq);"); break;
            case OP_ACCQ_MACC: {
                static const char* const fn[4] = {"accq_macc_p_cp(accq, p%u, P.consts, %uu);", "accq_macc_q_cp(accq, q%u, P.consts, %uu);",
                                                  "accq_macc_p_cq(accq, p%u, P.consts, %uu);", "accq_macc_q_cq(accq, q%u, P.consts, %uu);"};
                snprintf(b, sizeof b, fn[d & 3], x, y);
            } break;
            case OP_ACCQ_MACP: snprintf(b, sizeof b, (d & 1) ? "accq_macp_q_p(accq, q%u, p%u);" : "accq_macp_p_p(accq, p%u, p%u);", x, y); break;
            case OP_ACCQ_RED: snprintf(b, sizeof b, "q%u = accq_reduce(accq);", d); break;
            default: break;
            }
        }
        s += "    "; s += b; s += "\n";
    }
    s += "}\n";
    return s;
}

static inline uint64_t jit_hash(const std::string& s) {
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return h;
}

// source -> gfx950 code object.  Returns false (with the compiler log) on failure.
static inline bool jit_compile(const std::string& src, std::vector<char>& code, std::string& log) {
    const char* hdr_src[kJitNumHeaders];
    const char* hdr_name[kJitNumHeaders];
    for (int h = 0; h < kJitNumHeaders; h++) { hdr_name[h] = kJitHeaders[h][0]; hdr_src[h] = kJitHeaders[h][1]; }
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "ms_eval_jit.hip", kJitNumHeaders, hdr_src, hdr_name) != HIPRTC_SUCCESS) { log = "hiprtcCreateProgram failed"; return false; }
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-uninitialized", "-Wno-unused-value"};
    const hiprtcResult r = hiprtcCompileProgram(prog, (int)(sizeof(opts) / sizeof(opts[0])), opts);
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

}  // namespace mseval
