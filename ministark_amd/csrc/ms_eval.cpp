// Fused constraint evaluation: program validation, rewriting (eval_opt.h), specialisation (eval_jit.h), launches
// (src/eval_gpu.rs, parity with src/eval_cpu.rs:33-150).
#include <algorithm>
#include <chrono>
#include "ms_internal.h"
#include "stage_kernels.h"
#include "eval_kernels.h"
#include "eval_opt.h"
#include "eval_regroup.h"
#include "eval_shift.h"
#ifndef MS_NO_JIT
#include "eval_jit.h"
#elif defined(MS_EMU)
#include "eval_jit_source.h"      // the execution-model simulator of tests/emu: the same generated source, compiled by g++ (tests/emu/emu_jit.h)
#include "emu_jit.h"
#endif

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
static int eval_locked(ms_ctx* ctx, const uint32_t* h_prog, unsigned ninstr, const void* h_consts, unsigned nconst_words,
                       unsigned log_n, unsigned lde_step, const void* h_domain_offset, const void* d_x_lde,
                       const void* const* d_base_cols, unsigned nbase, const void* const* d_ext_cols, unsigned next,
                       const void* const* d_periodic, const unsigned* periodic_len, unsigned nperiodic,
                       int out_field, void* d_out, unsigned flags);

extern "C" int ms_eval_program_ex(ms_ctx* ctx, const uint32_t* h_prog, unsigned ninstr, const void* h_consts, unsigned nconst_words,
                                  unsigned log_n, unsigned lde_step, const void* h_domain_offset, const void* d_x_lde,
                                  const void* const* d_base_cols, unsigned nbase, const void* const* d_ext_cols, unsigned next,
                                  const void* const* d_periodic, const unsigned* periodic_len, unsigned nperiodic,
                                  int out_field, void* d_out, unsigned flags) {
    if (!ctx) return fail(MS_ERR_INVALID, "ms_eval_program: null argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    MSCHK(eval_locked(ctx, h_prog, ninstr, h_consts, nconst_words, log_n, lde_step, h_domain_offset, d_x_lde, d_base_cols, nbase, d_ext_cols, next,
                      d_periodic, periodic_len, nperiodic, out_field, d_out, flags));
    // MS_EVAL_SELFCHECK=1 (tests, bug hunts): the five rewriting passes and the specialised kernels must not change a single output word --
    // the ORIGINAL program runs once more on the plain interpreter and every word is compared (blocks; two downloads of the result)
    static const bool selfcheck = getenv("MS_EVAL_SELFCHECK") && strcmp(getenv("MS_EVAL_SELFCHECK"), "0");
    if (selfcheck && !(flags & MS_EVAL_PLAIN)) {
        const size_t bytes = ((size_t)1 << log_n) * ms_field_bytes(out_field);
        LockedPoolGuard pooled(ctx);
        void* plain = nullptr;
        MSCHK(pooled.alloc(bytes, &plain));
        MSCHK(eval_locked(ctx, h_prog, ninstr, h_consts, nconst_words, log_n, lde_step, h_domain_offset, d_x_lde, d_base_cols, nbase, d_ext_cols, next,
                          d_periodic, periodic_len, nperiodic, out_field, plain, flags | MS_EVAL_PLAIN));
        std::vector<uint64_t> a(bytes / 8), b(bytes / 8);
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipMemcpy(a.data(), d_out, bytes, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(b.data(), plain, bytes, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < a.size(); i++)
            if (a[i] != b[i]) return fail(MS_ERR_INTERNAL, "constraint evaluation self-check: word %zu of %zu differs between the rewritten program (%016llx) and the "
                                          "original on the interpreter (%016llx) -- a bug in a rewriting pass or a specialised kernel", i, a.size(), (unsigned long long)a[i], (unsigned long long)b[i]);
    }
    return MS_OK;
}

static int eval_locked(ms_ctx* ctx, const uint32_t* h_prog, unsigned ninstr, const void* h_consts, unsigned nconst_words,
                       unsigned log_n, unsigned lde_step, const void* h_domain_offset, const void* d_x_lde,
                       const void* const* d_base_cols, unsigned nbase, const void* const* d_ext_cols, unsigned next,
                       const void* const* d_periodic, const unsigned* periodic_len, unsigned nperiodic,
                       int out_field, void* d_out, unsigned flags) {
    using namespace mseval;
    // MS_EVAL_TIMING=1: the host's share of a call on stderr -- validation + rewriting passes | uploads | source generation + kernel lookup + launches
    static const bool timing = getenv("MS_EVAL_TIMING") != nullptr;
    const auto t_entry = std::chrono::steady_clock::now();
    auto us_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t).count(); };
    if (flags & ~(unsigned)(MS_EVAL_BIT_REVERSED | MS_EVAL_PLAIN)) return fail(MS_ERR_INVALID, "ms_eval_program_ex: unknown flags 0x%x", flags);
    const bool plain = (flags & MS_EVAL_PLAIN) != 0;           // the program as given: no rewriting pass, no specialised kernel
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
    HIPCHK(hipSetDevice(ctx->device));
    // ---- rewrite: short-period sub-expressions -> tables, long x^e chains -> twiddle lookups (eval_opt.h)
    SplitProgram split;
    if (!plain) {
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
    // (MS_EVAL_SPLIT_MIN_LOG_N: the exhaustive structural test runs the table passes on 64-point domains)
    static const unsigned split_min_log_n = getenv("MS_EVAL_SPLIT_MIN_LOG_N") ? (unsigned)atoi(getenv("MS_EVAL_SPLIT_MIN_LOG_N")) : 12u;
    if (log_n >= split_min_log_n && !plain) isplit = split_inversions(main_prog, main_n, nperiodic + short_tables, (unsigned)MAXPERIODIC - nperiodic - short_tables, PW);
    // ---- rewrite 3b: denominators X - a whose roots differ by a power of the trace generator share one table (eval_shift.h; MS_EVAL_SHARE_TABLES=0: off)
    if (isplit.active && !d_x_lde && isplit.table_words.size() >= 2) {
        static const bool off = getenv("MS_EVAL_SHARE_TABLES") && !strcmp(getenv("MS_EVAL_SHARE_TABLES"), "0");
        const bool dbg = getenv("MS_EVAL_DEBUG") != nullptr;
        if (off) {}
        else if (is252) share_shifted_tables<Host252>(isplit, nperiodic + short_tables, PW, consts, f252::pow_u64(f252::root_of_unity(log_n), lde_step), maxp, dbg);
        else share_shifted_tables<HostGL>(isplit, nperiodic + short_tables, PW, consts, gl::to_mont(gl::pow(gl::root_of_unity(log_n), lde_step)), maxp, dbg);
    }
    // ---- Goldilocks tables whose denominator is X + c are generated inside the inversion kernel (eval_kernels.h batch_inverse_x_plus_c):
    // their stores leave the denominators' program, which disappears when nothing else is left in it
    std::vector<char> fused;
    std::vector<uint64_t> fused_c;
    if (isplit.active && !is252 && !d_x_lde && log_n >= 12) {
        static const bool off = getenv("MS_EVAL_FUSE_DENOMINATORS") && !strcmp(getenv("MS_EVAL_FUSE_DENOMINATORS"), "0");
        std::vector<int> store_of;
        if (!off) tables_x_plus_c<HostGL>(isplit, nperiodic + short_tables, PW, consts, fused, fused_c, store_of);
        bool any = false;
        for (char f : fused) any = any || f;
        if (any) {
            std::vector<Instr> denom;
            bool stores_left = false;
            for (size_t k = 0; k < isplit.denom.size(); k++) {
                bool drop = false;
                for (size_t t = 0; t < fused.size(); t++) if (fused[t] && store_of[t] == (int)k) drop = true;
                if (drop) continue;
                if (op_is_store(isplit.denom[k].op)) stores_left = true;
                denom.push_back(isplit.denom[k]);
            }
            if (!stores_left) denom.clear();
            isplit.denom.swap(denom);
            if (getenv("MS_EVAL_DEBUG")) fprintf(stderr, "fused denominators: %zu of %zu tables are X + c%s\n", (size_t)std::count(fused.begin(), fused.end(), 1), fused.size(), stores_left ? "" : " (no denominators' program left)");
        } else fused.clear();
    }
    if (isplit.active) { main_prog = isplit.main.data(); main_n = (unsigned)isplit.main.size(); }
    const unsigned den_n = (unsigned)isplit.denom.size();
    // ---- rewrite 4: the result as sums of products with one reduction per sum (eval_regroup.h; Fq = Fp programs; MS_EVAL_REGROUP=0: off)
    Regrouped regrouped;
    {
        static const bool off = getenv("MS_EVAL_REGROUP") && !strcmp(getenv("MS_EVAL_REGROUP"), "0");
        if (!off && !plain) {
            static const bool force = getenv("MS_EVAL_REGROUP") && !strcmp(getenv("MS_EVAL_REGROUP"), "force");     // the fuzzers: whenever it CAN be applied
            if (is252) regrouped = regroup_sums_of_products<Host252>(main_prog, main_n, consts, force);
            else if (maxq == 0 && out_field != MS_GOLDILOCKS_FQ3) regrouped = regroup_sums_of_products<HostGL>(main_prog, main_n, consts, force);
            else regrouped = regroup_sums_of_products_q(main_prog, main_n, consts, force);
            if (getenv("MS_EVAL_DEBUG")) fprintf(stderr, "regroup: %s, estimated vector instructions per point %u -> %u\n", regrouped.active ? "applied" : "not applied", regrouped.old_cost, regrouped.new_cost);
            if (regrouped.active && getenv("MS_EVAL_DEBUG")) for (auto& I : regrouped.prog) fprintf(stderr, "  regr: op %2u dst %u a %u b %u\n", I.op, I.dst, I.a, I.b);
            if (regrouped.active) { main_prog = regrouped.prog.data(); main_n = (unsigned)regrouped.prog.size(); maxp = std::max(maxp, regrouped.maxp); maxq = std::max(maxq, regrouped.maxq); }
        }
    }
    const double us_rewrite = us_since(t_entry);
    // ---- program(s) + constants -> device
    const size_t mbytes = (size_t)main_n * sizeof(Instr), pbytes = (size_t)pro_n * sizeof(Instr), dbytes = (size_t)den_n * sizeof(Instr), cbytes = consts.size() * 8;
    const size_t poff = (mbytes + 15) & ~(size_t)15, doff = (poff + pbytes + 15) & ~(size_t)15, coff = (doff + dbytes + 15) & ~(size_t)15, total = coff + cbytes + 64;
    if (ctx->prog_bytes < total) {
        HIPCHK(hipStreamSynchronize(ctx->stream));           // a previous evaluation may still read the buffer that is about to go
        if (ctx->prog_buf) HIPCHK(hipFree(ctx->prog_buf));
        ctx->prog_buf = nullptr; ctx->prog_bytes = 0;
        if (hipMalloc(&ctx->prog_buf, total) != hipSuccess) return fail(MS_ERR_NOMEM, "program buffer");
        ctx->prog_bytes = total;
    }
    // stream-ordered copies out of the pinned ring: they queue behind a previous evaluation that still reads the buffer, no drain
    // (ONE copy command for the four pieces: each is a blit launch of its own on the stream, 5-7 us in front of a 130 us evaluation)
    {
        std::vector<char> image(coff + cbytes, 0);
        memcpy(image.data(), main_prog, mbytes);
        if (pbytes) memcpy(image.data() + poff, split.prologue.data(), pbytes);
        if (dbytes) memcpy(image.data() + doff, isplit.denom.data(), dbytes);
        if (cbytes) memcpy(image.data() + coff, consts.data(), cbytes);
        MSCHK(stage_upload(ctx, ctx->prog_buf, image.data(), image.size()));
    }
    const double us_upload = us_since(t_entry) - us_rewrite;
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
    // (MS_EVAL_JIT_MIN_LOG_N: the tests send 256-point domains through the generated kernels; the default is 2^16 points)
    static const unsigned jit_min_log_n = getenv("MS_EVAL_JIT_MIN_LOG_N") ? (unsigned)atoi(getenv("MS_EVAL_JIT_MIN_LOG_N")) : 16u;
    auto specialised = [&](const Instr* pr, unsigned cnt) -> hipFunction_t {
#ifndef MS_NO_JIT
        static const bool off = getenv("MS_EVAL_JIT") && !strcmp(getenv("MS_EVAL_JIT"), "0");
        if (off || plain) return nullptr;
        const std::string src = jit_source(pr, cnt, is252, maxp, maxq, lde_step);
        if (const char* dump = getenv("MS_EVAL_DUMP")) { if (FILE* f = fopen(dump, "a")) { fputs(src.c_str(), f); fputs("\n// ----\n", f); fclose(f); } }
        const std::string& key = src;
        auto it = ctx->jit_cache.find(key);
        if (it != ctx->jit_cache.end()) return it->second;
        hipFunction_t fn = nullptr;
        std::vector<char> code;
        std::string log;
        JitStats st;
        // a cached entry the loader refuses (it passed its digest: a foreign or stale code object) is compiled afresh once
        for (int attempt = 0; attempt < 2 && !fn; attempt++) {
            const uint64_t disk_before = st.from_disk;
            if (!jit_obtain(src, code, log, st, attempt == 1)) break;
            const double t0 = jit_now_ms();
            hipModule_t mod = nullptr;
            if (hipModuleLoadData(&mod, code.data()) == hipSuccess && hipModuleGetFunction(&fn, mod, "ms_eval_jit") == hipSuccess) ctx->jit_modules.push_back(mod);
            else { fn = nullptr; (void)hipGetLastError(); if (mod) (void)hipModuleUnload(mod); }
            st.load_ms += jit_now_ms() - t0;
            if (!fn && st.from_disk == disk_before) break;       // a freshly compiled object that does not load: no second try
            if (!fn) { st.damaged_entries++; st.from_disk--; }
        }
        if (!fn) {
            if (!st.failures) st.failures++;
            jit_warn_failure(log.empty() ? std::string("(the code object was produced but hipModuleLoadData / hipModuleGetFunction refused it)") : log);
        } else if (getenv("MS_EVAL_DEBUG")) {
            int regs = 0, spill = 0;
            (void)hipFuncGetAttribute(&regs, HIP_FUNC_ATTRIBUTE_NUM_REGS, fn);
            (void)hipFuncGetAttribute(&spill, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, fn);
            fprintf(stderr, "[ministark_hip] specialised constraint kernel: %u instructions, %d vector registers, %d bytes of scratch per lane (%s)\n", cnt, regs, spill,
                    st.from_disk ? "from the disk cache" : "compiled");
        }
        for (JitStats* t : {&ctx->jit_stats, &jit_process_stats()}) {
            t->compiled += st.compiled; t->from_disk += st.from_disk; t->failures += st.failures; t->damaged_entries += st.damaged_entries;
            t->compile_ms += st.compile_ms; t->load_ms += st.load_ms;
        }
        ctx->jit_cache[key] = fn;
        return fn;
#elif defined(MS_EMU)
        if (!emu_jit::enabled() || plain) return nullptr;
        const std::string src = jit_source(pr, cnt, is252, maxp, maxq, lde_step);
        if (const char* dump = getenv("MS_EVAL_DUMP")) { if (FILE* f = fopen(dump, "a")) { fputs(src.c_str(), f); fputs("\n// ----\n", f); fclose(f); } }
        auto it = ctx->jit_cache.find(src);
        if (it != ctx->jit_cache.end()) return it->second;
        std::string log;
        bool compiled = false;
        hipFunction_t fn = emu_jit::obtain(src, log, &compiled);
        if (!fn) { ctx->jit_stats.failures++; fprintf(stderr, "[ministark_hip simulator] specialised constraint kernel: %s\n", log.c_str()); }
        else if (compiled) ctx->jit_stats.compiled++;
        else ctx->jit_stats.from_disk++;
        ctx->jit_cache[src] = fn;
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
        static const bool host_off = getenv("MS_EVAL_HOST_TABLES") && !strcmp(getenv("MS_EVAL_HOST_TABLES"), "0");
        if (period <= 64 && !d_x_lde && !host_off) {
            on_host = true;
            for (auto& I : split.prologue) if (I.op == OP_PERIODIC_P || I.op == OP_PERIODIC_Q || I.op == OP_TRACE_P || I.op == OP_TRACE_Q || I.op == OP_XPOW_P || I.op >= OP_TABLE_P) on_host = false;   // caller tables live on the device
        }
        if (on_host && !is252) {
            // Goldilocks (round 5): the same few points on the host too -- the launch was 16 us of latency in front of every evaluation
            // (the interpreter's register arrays, one wave); gl.h's host functions are the kernels' own formulas, so the same words
            std::vector<uint64_t> host_tabs(words, 0);
            const uint64_t w = gl::root_of_unity(log_n);
            uint64_t xi = h;                                                                  // x_i = h * w^i, canonical
            std::vector<uint64_t> rp(256);
            std::vector<gl::Fq3> rq(128);
            for (size_t i = 0; i < period; i++) {
                for (auto& I : split.prologue) {
                    switch (I.op) {
                    case OP_X_P: rp[I.dst] = gl::to_mont(xi); break;
                    case OP_CONST_P: rp[I.dst] = consts[I.a]; break;
                    case OP_CONST_Q: rq[I.dst] = gl::Fq3{consts[I.a], consts[I.a + 1], consts[I.a + 2]}; break;
                    case OP_NEG_P: rp[I.dst] = gl::neg(rp[I.a]); break;
                    case OP_NEG_Q: rq[I.dst] = gl::neg(rq[I.a]); break;
                    case OP_ADD_PP: rp[I.dst] = gl::add(rp[I.a], rp[I.b]); break;
                    case OP_ADD_QQ: rq[I.dst] = gl::add(rq[I.a], rq[I.b]); break;
                    case OP_ADD_QP: rq[I.dst] = gl::Fq3{gl::add(rq[I.a].c0, rp[I.b]), rq[I.a].c1, rq[I.a].c2}; break;
                    case OP_MUL_PP: rp[I.dst] = gl::mont_mul(rp[I.a], rp[I.b]); break;
                    case OP_MUL_QQ: rq[I.dst] = gl::mont_mul(rq[I.a], rq[I.b]); break;
                    case OP_MUL_QP: rq[I.dst] = gl::mont_mul_fp(rq[I.a], rp[I.b]); break;
                    case OP_INV_P: rp[I.dst] = gl::mont_inv(rp[I.a]); break;
                    case OP_INV_Q: rq[I.dst] = gl::mont_inv(rq[I.a]); break;
                    case OP_POW_P: rp[I.dst] = gl::mont_pow(rp[I.a], (uint64_t)I.b); break;
                    case OP_POW_Q: rq[I.dst] = gl::mont_pow(rq[I.a], (uint64_t)I.b); break;
                    case OP_EMBED: rq[I.dst] = gl::Fq3{rp[I.a], 0, 0}; break;
                    case OP_STORE_P: case OP_STORE_Q: {
                        size_t off = 0;
                        for (unsigned t = 0; t + nperiodic + 1 < I.b; t++) off += split.table_words[t] * period;
                        if (I.op == OP_STORE_P) host_tabs[off + i] = rp[I.a];
                        else { host_tabs[off + 3 * i] = rq[I.a].c0; host_tabs[off + 3 * i + 1] = rq[I.a].c1; host_tabs[off + 3 * i + 2] = rq[I.a].c2; }
                    } break;
                    default: break;
                    }
                }
                xi = gl::mul(xi, w);
            }
            MSCHK(stage_upload(ctx, tables, host_tabs.data(), words * 8));                    // stream-ordered, out of the pinned ring
        } else if (on_host) {
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
            MSCHK(stage_upload(ctx, tables, host_tabs.data(), words * 8));                    // stream-ordered, out of the pinned ring
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
    if (isplit.active && !isplit.table_words.empty()) {
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
        if (den_n) {
            hipFunction_t fn = log_n >= jit_min_log_n ? specialised(isplit.denom.data(), den_n) : nullptr;
            ProfScope ps(ctx, "eval_denominators", 0.0);
            launch(Q, fn);
        }
        tp = (uint64_t*)inv_tables;
        // 252-bit tables of a large domain are inverted TOGETHER (they are adjacent in memory): the middle level of the two-level scheme is
        // one serial Fermat chain per lane at one or two waves per SIMD -- pure latency, paid once for all tables instead of once per table
        bool all252 = !isplit.table_words.empty() && n >= ((size_t)1 << 16);
        for (unsigned w : isplit.table_words) if (w != 4) all252 = false;
        const size_t ntab_merged = all252 ? isplit.table_words.size() : 0;
        if (ntab_merged) {
            const size_t nn = n * ntab_merged;
            ProfScope ps(ctx, "eval_batch_inverse", 16.0 * 4 * nn);
            const unsigned blocks = (unsigned)(nn / (NT * 8));
            const size_t m = (size_t)blocks * NT;
            void* prod = nullptr;
            MSCHK(pooled.alloc(m * 32, &prod));
            hipLaunchKernelGGL((batch_inverse_up<msstage::Fp252T, 8>), dim3(blocks), dim3(NT), 0, ctx->stream, tp, nn, (uint64_t*)prod);
            // the middle level: one Fermat inverse (55 000 instructions) per lane -- 32 products per lane while that still leaves a wave per SIMD
            size_t min_lanes = 65536;                            // (MS_EVAL_INV_MIN_LANES: the tests reach the wide variant on a small domain)
            if (const char* e = getenv("MS_EVAL_INV_MIN_LANES")) min_lanes = (size_t)std::max(1L, atol(e));
            if (m >= 32 * min_lanes) hipLaunchKernelGGL((batch_inverse<msstage::Fp252T, 32>), dim3((unsigned)((m + NT * 32 - 1) / (NT * 32))), dim3(NT), 0, ctx->stream, (uint64_t*)prod, m);
            else hipLaunchKernelGGL((batch_inverse<msstage::Fp252T, 16>), dim3((unsigned)((m + NT * 16 - 1) / (NT * 16))), dim3(NT), 0, ctx->stream, (uint64_t*)prod, m);
            hipLaunchKernelGGL((batch_inverse_down<msstage::Fp252T, 8>), dim3(blocks), dim3(NT), 0, ctx->stream, tp, nn, (const uint64_t*)prod);
        }
        // (Fp tables stay one launch each at 16 elements per inverse: merged at 32 per inverse the kernel holds 128 registers and was slower,
        // 95 -> 106 us for the fib AIR's two tables at 2^23 points)
        const size_t t_first = ntab_merged;
        for (size_t t = t_first; t < isplit.table_words.size(); t++) {
            const unsigned w = isplit.table_words[t];
            ProfScope ps(ctx, "eval_batch_inverse", 16.0 * w * n);
            // Fp / Fq3: the stage's kernel, in place (it keeps the K values in registers: one read and one write of the table)
            if (w == 1 && t < fused.size() && fused[t]) {      // X + c: generated in the kernel (n is a multiple of 16 NT: log_n >= 12)
                XcParams X;
                memset(&X, 0, sizeof X);
                X.c = fused_c[t];
                const uint64_t wn = gl::root_of_unity(log_n);
                for (unsigned j = 0; j < 16; j++) {
                    const uint64_t e = E.bitrev ? (uint64_t)(((j & 1) << 3) | ((j & 2) << 1) | ((j & 4) >> 1) | ((j & 8) >> 3)) : (uint64_t)j * (n / 16);
                    X.m[j] = gl::to_mont(gl::pow(wn, e));
                }
                hipLaunchKernelGGL((batch_inverse_x_plus_c<16>), dim3((unsigned)(n / ((size_t)NT * 16))), dim3(NT), 0, ctx->stream, E, tp, X);
            }
            else if (w == 1) hipLaunchKernelGGL((msstage::k_batch_inverse<msstage::FpT, 16>), dim3((unsigned)((n + msstage::NT * 16 - 1) / (msstage::NT * 16))), dim3(msstage::NT), 0, ctx->stream, tp, (const uint64_t*)tp, n);
            else if (w == 3) hipLaunchKernelGGL((msstage::k_batch_inverse<msstage::Fq3T, 8>), dim3((unsigned)((n + msstage::NT * 8 - 1) / (msstage::NT * 8))), dim3(msstage::NT), 0, ctx->stream, tp, (const uint64_t*)tp, n);
            else if (n < ((size_t)1 << 16)) hipLaunchKernelGGL((batch_inverse<msstage::Fp252T, 8>), dim3((unsigned)((n + NT * 8 - 1) / (NT * 8))), dim3(NT), 0, ctx->stream, tp, n);
            else {                                             // two levels: one 252-bit Fermat inverse per 128 elements
                const unsigned blocks = (unsigned)(n / (NT * 8));
                const size_t m = (size_t)blocks * NT;          // lanes of the sweep = entries of the product array
                void* prod = nullptr;
                MSCHK(pooled.alloc(m * 32, &prod));
                hipLaunchKernelGGL((batch_inverse_up<msstage::Fp252T, 8>), dim3(blocks), dim3(NT), 0, ctx->stream, tp, n, (uint64_t*)prod);
                hipLaunchKernelGGL((batch_inverse<msstage::Fp252T, 16>), dim3((unsigned)((m + NT * 16 - 1) / (NT * 16))), dim3(NT), 0, ctx->stream, (uint64_t*)prod, m);   // one Fermat inverse per 128 elements
                hipLaunchKernelGGL((batch_inverse_down<msstage::Fp252T, 8>), dim3(blocks), dim3(NT), 0, ctx->stream, tp, n, (const uint64_t*)prod);
            }
            tp += (size_t)w * n;
        }
    }
    {
        hipFunction_t fn = log_n >= jit_min_log_n ? specialised(main_prog, main_n) : nullptr;    // small domains: the interpreter is quicker than a compilation
        ProfScope ps(ctx, fn ? (is252 ? "eval_program252_jit" : "eval_program_jit") : (is252 ? "eval_program252" : "eval_program"),
                     is252 ? 32.0 * n * (nbase + 1) : 8.0 * n * (nbase + 3.0 * next + (out_field == MS_GOLDILOCKS_FQ3 ? 3 : 1)));
        launch(E, fn);
    }
    HIPCHK(hipGetLastError());
    if (timing) fprintf(stderr, "[ministark_hip] evaluation, host side: rewriting %.1f us, upload %.1f us, tables + sources + launches %.1f us\n", us_rewrite, us_upload, us_since(t_entry) - us_rewrite - us_upload);
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
    const std::string src = jit_source(prog, ninstr, is252, maxp, maxq);
    if (const char* dump = getenv("MS_EVAL_DUMP")) { if (FILE* f = fopen(dump, "a")) { fputs(src.c_str(), f); fputs("\n// ----\n", f); fclose(f); } }
    // through the on-disk cache like an evaluation (jit_cache.h): the first call for a program compiles and stores, later processes load
    static std::mutex mu;                                     // the process totals have no context lock of their own
    std::lock_guard<std::mutex> lk(mu);
    JitStats st;
    const bool ok = jit_obtain(src, code, log, st);
    JitStats& T = jit_process_stats();
    T.compiled += st.compiled; T.from_disk += st.from_disk; T.failures += st.failures; T.damaged_entries += st.damaged_entries; T.compile_ms += st.compile_ms; T.load_ms += st.load_ms;
    if (!ok) return fail(MS_ERR_UNSUPPORTED, "hiprtc: %s", log.c_str());
    *code_bytes = code.size();
    return MS_OK;
#else
    (void)h_prog; (void)ninstr; (void)out_field; (void)code_bytes;
    return fail(MS_ERR_UNSUPPORTED, "built without hiprtc");
#endif
}

JitStats& jit_process_stats() { static JitStats s; return s; }

extern "C" int ms_eval_jit_stats(ms_ctx* ctx, ms_jit_stats* out) {
    if (!out) return fail(MS_ERR_INVALID, "ms_eval_jit_stats: null argument");
    JitStats st;
    if (ctx) { std::lock_guard<std::mutex> lk(ctx->mu); st = ctx->jit_stats; }
    else st = jit_process_stats();
    out->kernels_compiled = st.compiled; out->kernels_from_disk = st.from_disk; out->compile_failures = st.failures;
    out->damaged_entries = st.damaged_entries; out->compile_ms = st.compile_ms; out->load_ms = st.load_ms;
    return MS_OK;
}
