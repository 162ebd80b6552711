// The generated source of a specialised constraint kernel (host only, no runtime-compiler dependency): one call of an eval_kernels.h helper
// per instruction of the register program, registers as named locals.  eval_jit.h hands the text to hiprtc; the execution-model
// simulator of tests/emu compiles the SAME text with g++ (tests/emu/emu_jit.h), so that a mismatch between this generator and the
// interpreter of eval_kernels.h shows in the GPU-less test run too.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "eval_kernels.h"

namespace mseval {

// lde_step: the row stride of the launch (252-bit programs stage multiply-read columns in LDS for it: ev252_stage; part of the source)
static inline std::string jit_source(const Instr* prog, unsigned ninstr, bool is252, unsigned maxp, unsigned maxq, unsigned lde_step = 1) {
    // 252-bit programs: the columns / full-length tables read at two or more row offsets, their window and place in the staging buffer
    struct Staged { bool table; unsigned id; int lo, hi; unsigned words_at; };
    std::vector<Staged> staged;
    // MS_EVAL_STAGE=1 switches the staging ON.  Measured on configs[3] (iii) (round 6, profiles/r06_c4iii_staging.txt): the kernel's HBM
    // fetches fall from 4.83 to 3.01 GB (the whole evaluation 2.9 x -> 1.95 x its algorithmic bytes) and the kernel gets SLOWER, 2.28 -> 2.42 ms:
    // it is bound by instruction issue at two waves per SIMD, not by memory, and the staging prologue (nine loads, a barrier) plus two LDS
    // reads per value cost more issue slots than the second-hand fetches cost latency.  Off by default: the time is the figure of merit.
    static const bool stage_on = getenv("MS_EVAL_STAGE") && !strcmp(getenv("MS_EVAL_STAGE"), "1");
    if (is252 && stage_on) {
        std::vector<Staged> seen;
        std::vector<std::vector<int>> offs;
        for (unsigned k = 0; k < ninstr; k++) {
            const Instr I = prog[k];
            if (I.op != OP_TRACE_P && I.op != OP_TABLE_P) continue;
            const bool tb = I.op == OP_TABLE_P;
            size_t at = 0;
            while (at < seen.size() && !(seen[at].table == tb && seen[at].id == I.a)) at++;
            if (at == seen.size()) { seen.push_back(Staged{tb, I.a, 0, 0, 0}); offs.emplace_back(); }
            const int o = (int)(int32_t)I.b;
            if (std::find(offs[at].begin(), offs[at].end(), o) == offs[at].end()) offs[at].push_back(o);
        }
        unsigned words = 0;
        for (size_t at = 0; at < seen.size(); at++) {
            if (offs[at].size() < 2) continue;                                 // read once: nothing to share
            long long lo = 0, hi = 0;
            for (int o : offs[at]) { lo = std::min(lo, (long long)o * lde_step); hi = std::max(hi, (long long)o * lde_step); }
            if (hi - lo > 64) continue;                                        // the window would be mostly halo
            const unsigned W = 256 + (unsigned)(hi - lo);
            if ((words + 4 * W) * 8 > 76 * 1024) continue;                     // two workgroups per CU (160 KiB of LDS)
            Staged st = seen[at];
            st.lo = (int)lo; st.hi = (int)hi; st.words_at = words;
            staged.push_back(st);
            words += 4 * W;
        }
        if (staged.empty()) words = 0;
    }
    auto staged_of = [&](bool tb, unsigned id) -> const Staged* {
        for (auto& st : staged) if (st.table == tb && st.id == id) return &st;
        return nullptr;
    };
    std::string s;
    s.reserve(4096 + (size_t)ninstr * 64);
    s += "#include \"eval_kernels.h\"\nusing namespace mseval;\n";
    // (MS_EVAL_JIT_WAVES: a minimum number of waves per SIMD for the generated kernel -- an experiment knob; part of the source, hence of the cache key)
    const char* waves = getenv("MS_EVAL_JIT_WAVES");
    s += "extern \"C\" __global__ void __launch_bounds__(256";
    if (waves && atoi(waves) > 0) { s += ", "; s += std::to_string(atoi(waves)); }
    s += ") ms_eval_jit(EvalParams P) {\n";
    s += "    using F3 = msstage::Fq3T; using F1 = msstage::FpT; using F4 = msstage::Fp252T;\n";
    s += is252 ? "    const size_t R = ev252_pos(P);\n" : "    const size_t R = (size_t)blockIdx.x * 256 + threadIdx.x;\n";
    char b[256];
    if (!staged.empty()) {
        // (the launches of a specialised kernel cover n >= 2^16 points exactly: no thread leaves before the barrier)
        unsigned total = 0;
        for (auto& st : staged) total = st.words_at + 4 * (256 + (unsigned)(st.hi - st.lo));
        snprintf(b, sizeof b, "    __shared__ uint64_t stg[%u];\n    const bool staged = !P.bitrev;\n    const size_t R0 = (size_t)blockIdx.x * 256;\n    const unsigned rl = (unsigned)(R - R0);\n    if (staged) {\n", total);
        s += b;
        for (auto& st : staged) {
            snprintf(b, sizeof b, "        ev252_stage(stg + %u, %s[%u], R0, %d, %uu, P.n);\n", st.words_at, st.table ? "P.periodic" : "P.base_cols", st.id, st.lo, 256 + (unsigned)(st.hi - st.lo));
            s += b;
        }
        s += "        __syncthreads();\n    }\n";
    }
    s += "    if (R >= P.n) return;\n    const size_t i = ev_point(P, R);\n";
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
            case OP_TRACE_P:
                if (const Staged* st = staged_of(false, x))
                    snprintf(b, sizeof b, "p%u = staged ? ev252_lds(stg + %u, rl + %d) : ev252_trace(P, i, %uu, %uu);", d, st->words_at, (int)(int32_t)y * (int)lde_step - st->lo, x, y);
                else snprintf(b, sizeof b, "p%u = ev252_trace(P, i, %uu, %uu);", d, x, y);
                break;
            case OP_PERIODIC_P: snprintf(b, sizeof b, "p%u = ev252_periodic(P, i, %uu);", d, x); break;
            case OP_NEG_P: snprintf(b, sizeof b, "p%u = f252::neg(p%u);", d, x); break;
            case OP_ADD_PP: snprintf(b, sizeof b, "p%u = f252::add(p%u, p%u);", d, x, y); break;
            case OP_MUL_PP: snprintf(b, sizeof b, "p%u = f252::mul(p%u, p%u);", d, x, y); break;
            case OP_INV_P: snprintf(b, sizeof b, "p%u = f252::inv(p%u);", d, x); break;
            case OP_POW_P: snprintf(b, sizeof b, "p%u = msstage::powu<F4>(p%u, %uu);", d, x, y); break;
            case OP_STORE_P: snprintf(b, sizeof b, "ev252_store(P, R, %uu, p%u);", y, x); break;
            case OP_XPOW_P: snprintf(b, sizeof b, "p%u = ev252_xpow(P, i, %uu, %uu);", d, x, y); break;
            case OP_TABLE_P:
                if (const Staged* st = staged_of(true, x))
                    snprintf(b, sizeof b, "p%u = staged ? ev252_lds(stg + %u, rl + %d) : ev252_table(P, R, i, %uu, %uu);", d, st->words_at, (int)(int32_t)y * (int)lde_step - st->lo, x, y);
                else snprintf(b, sizeof b, "p%u = ev252_table(P, R, i, %uu, %uu);", d, x, y);
                break;
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

}  // namespace mseval
