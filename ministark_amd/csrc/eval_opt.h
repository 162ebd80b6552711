// Host-side rewriting of a constraint program before it is launched (no field arithmetic on data:
// this only reorganises the program the caller lowered from the reference's expression DAG).
//
// 1. Periodic hoisting.  On the evaluation domain x_i = h*w_n^i a value built only from constants,
//    periodic columns and powers x^e repeats with period n / gcd(n, e) in i.  AIR zerofiers are of that
//    kind: 1 / (x^N - 1) on an LDE domain of n = N * blowup points takes `blowup` distinct values.  The
//    reference's evaluators recompute them at every point (src/eval_cpu.rs:306-428 with a batched
//    inversion per 512-point chunk, src/eval_gpu.rs: one full-array inverse kernel).  Here the
//    sub-program of short-period values (the "prologue") is evaluated ONCE on the first 2^k points into
//    small tables, and the per-point program reads them back as periodic columns.  Exact field
//    arithmetic: the tables hold exactly the values the per-point evaluation would produce.
// 2. x^e for a large exponent is h^e * w_n^(e*i mod n): one table lookup (the forward plan's twiddle
//    table) instead of a square-and-multiply chain (OP_XPOW_P).
#pragma once
#include <cstdint>
#include <vector>
#include "eval_kernels.h"

namespace mseval {

static constexpr unsigned CLS_FULL = 255;       // depends on the trace: not periodic
static constexpr unsigned HOIST_MAX_LOG = 18;   // tables of at most 2^18 points (and 1/8 of the domain), L2-resident

struct SplitProgram {
    bool active = false;
    std::vector<Instr> prologue, main;
    unsigned log_period = 0;                 // the prologue runs on 2^log_period points
    std::vector<unsigned> table_words;       // element words of every table (1 Fp, 3 Fq3, 4 Fp252)
    struct XPow { unsigned instr; uint32_t e; };   // main-program instructions rewritten to OP_XPOW_P (a = constant slot, filled by the caller)
    std::vector<XPow> xpows;
};

static inline bool op_is_q_dst(uint32_t op) {
    switch (op) {
    case OP_CONST_Q: case OP_TRACE_Q: case OP_PERIODIC_Q: case OP_NEG_Q: case OP_ADD_QQ: case OP_ADD_QP:
    case OP_MUL_QQ: case OP_MUL_QP: case OP_INV_Q: case OP_POW_Q: case OP_EMBED: case OP_TABLE_Q: case OP_ACCQ_RED: return true;
    default: return false;
    }
}
// register operands of an instruction: (file, index) pairs; file 0 = P, 1 = Q
static inline int op_operands(const Instr& I, unsigned (*out)[2]) {
    switch (I.op) {
    case OP_NEG_P: case OP_INV_P: case OP_POW_P: case OP_EMBED: case OP_STORE_P: out[0][0] = 0; out[0][1] = I.a; return 1;
    case OP_NEG_Q: case OP_INV_Q: case OP_POW_Q: case OP_STORE_Q: out[0][0] = 1; out[0][1] = I.a; return 1;
    case OP_ADD_PP: case OP_MUL_PP: out[0][0] = 0; out[0][1] = I.a; out[1][0] = 0; out[1][1] = I.b; return 2;
    case OP_ADD_QQ: case OP_MUL_QQ: out[0][0] = 1; out[0][1] = I.a; out[1][0] = 1; out[1][1] = I.b; return 2;
    case OP_ADD_QP: case OP_MUL_QP: out[0][0] = 1; out[0][1] = I.a; out[1][0] = 0; out[1][1] = I.b; return 2;
    default: return 0;
    }
}
static inline bool op_is_leaf(uint32_t op) {
    return op == OP_X_P || op == OP_CONST_P || op == OP_CONST_Q || op == OP_TRACE_P || op == OP_TRACE_Q || op == OP_PERIODIC_P || op == OP_PERIODIC_Q;
}
static inline bool op_is_store(uint32_t op) { return op == OP_STORE_P || op == OP_STORE_Q; }

// `prog` has been validated.  x_generated: x_i = h*w^i is produced by the kernel (no caller-supplied x array).
// first_table: index of the first free periodic slot; max_tables: free slots.  elem_words_p: words of a P register.
static inline SplitProgram split_periodic(const Instr* prog, unsigned ninstr, unsigned log_n, bool x_generated,
                                          const unsigned* periodic_len, unsigned first_table, unsigned max_tables, unsigned elem_words_p,
                                          unsigned max_log = HOIST_MAX_LOG) {
    SplitProgram S;
    if (!x_generated || log_n < 2) return S;
    if (log_n < 4) return S;
    const unsigned lim = max_log < log_n - 3 ? max_log : log_n - 3;
    std::vector<unsigned> cls(ninstr, CLS_FULL);
    std::vector<int> defp(256, -1), defq(128, -1);
    std::vector<char> x_valued(ninstr, 0);
    auto def_of = [&](unsigned file, unsigned r) { return file ? defq[r] : defp[r]; };
    // ---- period class of every value
    for (unsigned k = 0; k < ninstr; k++) {
        const Instr I = prog[k];
        unsigned opnd[2][2];
        const int nop = op_operands(I, opnd);
        unsigned c = 0;
        switch (I.op) {
        case OP_X_P: c = log_n; x_valued[k] = 1; break;
        case OP_CONST_P: case OP_CONST_Q: c = 0; break;
        case OP_TRACE_P: case OP_TRACE_Q: c = CLS_FULL; break;
        case OP_PERIODIC_P: case OP_PERIODIC_Q: {
            const unsigned len = periodic_len[I.a];
            c = CLS_FULL;
            if (len && (len & (len - 1)) == 0) { unsigned l = 0; while ((1u << l) < len) l++; if (l <= log_n) c = l; }
        } break;
        case OP_POW_P: case OP_POW_Q: {
            const int d = def_of(opnd[0][0], opnd[0][1]);
            c = cls[d];
            if (I.op == OP_POW_P && x_valued[d]) {
                if (I.b == 0) c = 0;
                else { unsigned tz = 0; while (!((I.b >> tz) & 1)) tz++; c = tz >= log_n ? 0 : log_n - tz; }
            }
        } break;
        default:
            for (int o = 0; o < nop; o++) { const unsigned cc = cls[def_of(opnd[o][0], opnd[o][1])]; if (cc > c) c = cc; }
        }
        if (op_is_store(I.op)) continue;
        cls[k] = c;
        if (op_is_q_dst(I.op)) defq[I.dst] = (int)k; else defp[I.dst] = (int)k;
    }
    // ---- the slice of short-period values and the values that leave it
    // cost of recomputing a value per point (its whole operand tree), in rough Fp-multiplication thirds
    std::vector<unsigned> cost(ninstr, 0);
    std::fill(defp.begin(), defp.end(), -1);
    std::fill(defq.begin(), defq.end(), -1);
    for (unsigned k = 0; k < ninstr; k++) {
        const Instr I = prog[k];
        unsigned opnd[2][2];
        const int nop = op_operands(I, opnd);
        if (op_is_store(I.op)) continue;
        unsigned c = 0;
        switch (I.op) {
        case OP_NEG_P: case OP_ADD_PP: c = 1; break;
        case OP_NEG_Q: case OP_ADD_QQ: case OP_ADD_QP: c = 2; break;
        case OP_MUL_PP: c = 3; break;
        case OP_MUL_QP: c = 8; break;
        case OP_MUL_QQ: c = 25; break;
        case OP_INV_P: c = 220; break;
        case OP_INV_Q: c = 300; break;
        case OP_POW_P: case OP_POW_Q: { unsigned bits = 0; for (uint32_t e = I.b; e; e >>= 1) bits += 1 + (e & 1); c = 3 * bits * (I.op == OP_POW_Q ? 8 : 1); } break;
        case OP_PERIODIC_P: case OP_PERIODIC_Q: c = 1; break;
        default: c = 0;
        }
        for (int o = 0; o < nop; o++) c += cost[def_of(opnd[o][0], opnd[o][1])];
        cost[k] = c > 100000 ? 100000 : c;
        if (op_is_q_dst(I.op)) defq[I.dst] = (int)k; else defp[I.dst] = (int)k;
    }
    static constexpr unsigned HOIST_MIN_COST = 6;   // a table read is about two multiplications' worth
    std::vector<char> in_slice(ninstr, 0), boundary(ninstr, 0), in_pro(ninstr, 0);
    for (unsigned k = 0; k < ninstr; k++)
        if (!op_is_store(prog[k].op) && cls[k] <= lim) in_slice[k] = 1;
    for (bool changed = true; changed;) {
        changed = false;
        std::fill(boundary.begin(), boundary.end(), 0);
        std::fill(defp.begin(), defp.end(), -1);
        std::fill(defq.begin(), defq.end(), -1);
        for (unsigned k = 0; k < ninstr; k++) {
            const Instr I = prog[k];
            unsigned opnd[2][2];
            const int nop = op_operands(I, opnd);
            if (!in_slice[k])
                for (int o = 0; o < nop; o++) { const int d = def_of(opnd[o][0], opnd[o][1]); if (in_slice[d]) boundary[d] = 1; }
            if (op_is_store(I.op)) continue;
            if (op_is_q_dst(I.op)) defq[I.dst] = (int)k; else defp[I.dst] = (int)k;
        }
        // cheap values are not worth a table: they (and, next round, their operands) go back to the per-point program
        for (unsigned k = 0; k < ninstr; k++)
            if (boundary[k] && !op_is_leaf(prog[k].op) && cost[k] < HOIST_MIN_COST) { in_slice[k] = 0; changed = true; }
    }
    unsigned ntab = 0, maxc = 0;
    for (unsigned k = 0; k < ninstr; k++)
        if (boundary[k] && !op_is_leaf(prog[k].op)) { ntab++; if (cls[k] > maxc) maxc = cls[k]; }
    const bool hoist = ntab > 0 && ntab <= max_tables;
    // the prologue = everything the table values depend on (x itself is not short-period, x^e can be)
    if (hoist) {
        std::vector<int> dp(256, -1), dq(128, -1);
        std::vector<std::vector<int>> deps(ninstr);
        for (unsigned k = 0; k < ninstr; k++) {
            const Instr I = prog[k];
            unsigned opnd[2][2];
            const int nop = op_operands(I, opnd);
            for (int o = 0; o < nop; o++) deps[k].push_back(opnd[o][0] ? dq[opnd[o][1]] : dp[opnd[o][1]]);
            if (op_is_store(I.op)) continue;
            if (op_is_q_dst(I.op)) dq[I.dst] = (int)k; else dp[I.dst] = (int)k;
        }
        for (unsigned k = 0; k < ninstr; k++) if (boundary[k] && !op_is_leaf(prog[k].op)) in_pro[k] = 1;
        for (int k = (int)ninstr - 1; k >= 0; k--) if (in_pro[k]) for (int d : deps[k]) in_pro[d] = 1;
    }
    // ---- emit
    std::fill(defp.begin(), defp.end(), -1);
    std::fill(defq.begin(), defq.end(), -1);
    unsigned tab = 0;
    std::vector<Instr> main;
    std::vector<uint32_t> main_e;                   // exponent of a rewritten x^e (0: not rewritten)
    for (unsigned k = 0; k < ninstr; k++) {
        Instr I = prog[k];
        const bool q = op_is_q_dst(I.op);
        if (hoist && in_pro[k]) S.prologue.push_back(I);
        if (hoist && boundary[k] && !op_is_leaf(I.op)) {
            S.prologue.push_back(Instr{q ? (uint32_t)OP_STORE_Q : (uint32_t)OP_STORE_P, 0, I.dst, first_table + tab + 1});
            main.push_back(Instr{q ? (uint32_t)OP_PERIODIC_Q : (uint32_t)OP_PERIODIC_P, I.dst, first_table + tab, 0});
            main_e.push_back(0);
            S.table_words.push_back(q ? 3u : elem_words_p);
            tab++;
        } else {
            // x^e by table lookup when the chain would be long (and x itself is what is raised)
            uint32_t e = 0;
            if (I.op == OP_POW_P && I.b >= 8 && x_valued[def_of(0, I.a)]) { e = I.b; I.op = OP_XPOW_P; }
            main.push_back(I);
            main_e.push_back(e);
        }
        if (op_is_store(I.op)) continue;
        if (q) defq[I.dst] = (int)k; else defp[I.dst] = (int)k;
    }
    // dead-code elimination of the per-point program (hoisted sub-trees, x when only x^e was needed)
    {
        std::vector<char> lp(256, 0), lq(128, 0), keep(main.size(), 0);
        for (int k = (int)main.size() - 1; k >= 0; k--) {
            const Instr I = main[k];
            unsigned opnd[2][2];
            int nop = op_operands(I, opnd);
            if (I.op == OP_XPOW_P) nop = 0;                       // its `a` is a constant slot
            bool live = op_is_store(I.op);
            if (!live) {
                char& l = op_is_q_dst(I.op) ? lq[I.dst] : lp[I.dst];
                live = l; l = 0;
            }
            if (!live) continue;
            keep[k] = 1;
            for (int o = 0; o < nop; o++) (opnd[o][0] ? lq[opnd[o][1]] : lp[opnd[o][1]]) = 1;
        }
        for (size_t k = 0; k < main.size(); k++) {
            if (!keep[k]) continue;
            if (main_e[k]) S.xpows.push_back({(unsigned)S.main.size(), main_e[k]});
            S.main.push_back(main[k]);
        }
    }
    S.log_period = maxc;
    S.active = hoist || !S.xpows.empty();
    if (!hoist) { S.prologue.clear(); S.table_words.clear(); }
    return S;
}

// 3. Batch inversion of x-only denominators.  A division by something built from x, constants and periodic values only --
//    the (X - 1), (X - g^-1) of boundary / terminal constraints, a zerofier too long to hoist -- costs one Fermat inverse per
//    point in a per-point program (72 multiplications over Goldilocks, ~370 over the 252-bit field; the reference's CPU
//    evaluator batch-inverts per 512-point chunk instead, eval_cpu.rs:101-107).  Such denominators are computed for all
//    points by a small program of their own into full-length tables, inverted there with Montgomery's trick
//    (eval_kernels.h batch_inverse), and the per-point program reads the inverse back at its position (OP_TABLE_*).
struct InvSplit {
    bool active = false;
    std::vector<Instr> denom, main;          // denom: stores denominator t into table slot first_table + t (STORE b = slot + 1)
    std::vector<unsigned> table_words;
};
static inline InvSplit split_inversions(const Instr* prog, unsigned ninstr, unsigned first_table, unsigned max_tables, unsigned elem_words_p) {
    InvSplit S;
    std::vector<char> xo(ninstr, 0), in_den(ninstr, 0), hoisted(ninstr, 0);
    std::vector<int> defp(256, -1), defq(128, -1);
    std::vector<std::vector<int>> deps(ninstr);
    unsigned ntab = 0;
    for (unsigned k = 0; k < ninstr; k++) {
        const Instr I = prog[k];
        unsigned opnd[2][2];
        const int nop = I.op == OP_XPOW_P ? 0 : op_operands(I, opnd);
        bool x_only;
        switch (I.op) {
        case OP_TRACE_P: case OP_TRACE_Q: case OP_TABLE_P: case OP_TABLE_Q: x_only = false; break;
        case OP_X_P: case OP_CONST_P: case OP_CONST_Q: case OP_PERIODIC_P: case OP_PERIODIC_Q: case OP_XPOW_P: x_only = true; break;
        default:
            x_only = nop > 0;
            for (int o = 0; o < nop; o++) { const int d = opnd[o][0] ? defq[opnd[o][1]] : defp[opnd[o][1]]; deps[k].push_back(d); if (d < 0 || !xo[d]) x_only = false; }
        }
        if (op_is_store(I.op)) continue;
        xo[k] = x_only;
        if ((I.op == OP_INV_P || I.op == OP_INV_Q) && x_only && ntab < max_tables) { hoisted[k] = 1; ntab++; for (int d : deps[k]) in_den[d] = 1; }
        if (op_is_q_dst(I.op)) defq[I.dst] = (int)k; else defp[I.dst] = (int)k;
    }
    if (!ntab) return S;
    for (int k = (int)ninstr - 1; k >= 0; k--) if (in_den[k]) for (int d : deps[k]) if (d >= 0) in_den[d] = 1;
    unsigned tab = 0;
    std::vector<Instr> main;
    for (unsigned k = 0; k < ninstr; k++) {
        const Instr I = prog[k];
        if (hoisted[k]) {
            // the store of the denominator comes BEFORE the inversion itself, which stays in the denominators' program when a later
            // denominator is built on this inverse (a / (b / (x - c))) and may write the register it reads (dst == a)
            const bool q = I.op == OP_INV_Q;
            S.denom.push_back(Instr{q ? (uint32_t)OP_STORE_Q : (uint32_t)OP_STORE_P, 0, I.a, first_table + tab + 1});
            main.push_back(Instr{q ? (uint32_t)OP_TABLE_Q : (uint32_t)OP_TABLE_P, I.dst, first_table + tab, 0});
            S.table_words.push_back(q ? 3u : elem_words_p);
            tab++;
        } else main.push_back(I);
        if (in_den[k]) S.denom.push_back(I);
    }
    // dead-code elimination of the per-point program (the denominators' own sub-trees, x when nothing else needs it)
    std::vector<char> lp(256, 0), lq(128, 0), keep(main.size(), 0);
    for (int k = (int)main.size() - 1; k >= 0; k--) {
        const Instr I = main[k];
        unsigned opnd[2][2];
        const int nop = I.op == OP_XPOW_P ? 0 : op_operands(I, opnd);
        bool live = op_is_store(I.op);
        if (!live) { char& l = op_is_q_dst(I.op) ? lq[I.dst] : lp[I.dst]; live = l; l = 0; }
        if (!live) continue;
        keep[k] = 1;
        for (int o = 0; o < nop; o++) (opnd[o][0] ? lq[opnd[o][1]] : lp[opnd[o][1]]) = 1;
    }
    for (size_t k = 0; k < main.size(); k++) if (keep[k]) S.main.push_back(main[k]);
    S.active = true;
    return S;
}

}  // namespace mseval
