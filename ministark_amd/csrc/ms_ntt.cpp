// NTT / iNTT plans and passes, bit reversal, the fused LDE and evaluate entry points (GpuFft / GpuIfft / Matrix methods:
// gpu/src/plan.rs:186-462, src/matrix.rs:102-251).
#include "ms_internal.h"
#include "ntt2_kernels.h"
#include "fp252_kernels.h"
#include "fp252_ntt_kernels.h"
#include "lde2_kernels.h"
#include "stage_kernels.h"
#include "scan_kernels.h"

// ---------------------------------------------------------------------------------------
// NTT plans
// ---------------------------------------------------------------------------------------
static void powers(std::vector<uint64_t>& out, size_t count, uint64_t base, uint64_t first = 1) {
    out.resize(count);
    uint64_t x = first;
    for (size_t i = 0; i < count; i++) { out[i] = x; x = gl::mul(x, base); }
}

static int plan_build(ms_ctx* ctx, unsigned V, unsigned log_n, bool inverse, uint64_t h, ms_ntt_plan** out);
int ctx_plan(ms_ctx* ctx, unsigned V, unsigned log_n, bool inverse, uint64_t h, ms_ntt_plan** out);
static int plan_build252(ms_ctx* ctx, unsigned log_n, bool inverse, const void* h_offset, const void* h_group_gen, ms_ntt_plan** out);

// Plan objects handed to the caller (ms_ntt_plan_create): a plan refers to its context (lock, stream, cached tables), so a
// context that is destroyed first releases them itself and takes them out of this registry; every entry point that receives
// a plan looks it up before touching it -- a plan destroyed after its context (a GpuFft collected after Planner.close()) is a
// no-op, a transform on it an error, never a read of freed memory -- for calls that do not RACE with the teardown: the lookup
// and the use that follows are two steps, so destroying a plan or its context on one thread while another thread is inside a
// call on that plan is outside the contract (include/ministark_hip.h, "Threads"), as it is for the reference's plans.
#include <set>
static std::mutex g_user_plans_mu;
static std::set<ms_ntt_plan*> g_user_plans;
static bool user_plan_alive(ms_ntt_plan* p) { std::lock_guard<std::mutex> lk(g_user_plans_mu); return g_user_plans.count(p) != 0; }
static int plan_free(ms_ntt_plan* plan);

static int plan_create_impl(ms_ctx* ctx, int field, unsigned log_n, int inverse, const void* h_offset, const void* h_group_gen, ms_ntt_plan** out);
extern "C" int ms_ntt_plan_create(ms_ctx* ctx, int field, unsigned log_n, int inverse, const void* h_offset,
                                  const void* h_group_gen, ms_ntt_plan** out) {
    if (!ctx || !out) return fail(MS_ERR_INVALID, "ms_ntt_plan_create: null argument");
    MSCHK(plan_create_impl(ctx, field, log_n, inverse, h_offset, h_group_gen, out));
    std::lock_guard<std::mutex> lk(g_user_plans_mu);
    g_user_plans.insert(*out);
    return MS_OK;
}
void plans_release_ctx(ms_ctx* ctx) {
    std::vector<ms_ntt_plan*> mine;
    {
        std::lock_guard<std::mutex> lk(g_user_plans_mu);
        for (auto it = g_user_plans.begin(); it != g_user_plans.end();)
            if ((*it)->ctx == ctx) { mine.push_back(*it); it = g_user_plans.erase(it); } else ++it;
    }
    for (ms_ntt_plan* p : mine) (void)plan_free(p);
}
static int plan_create_impl(ms_ctx* ctx, int field, unsigned log_n, int inverse, const void* h_offset, const void* h_group_gen, ms_ntt_plan** out) {
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    if (V == 4) {
        // the 252-bit plans are cached per context like the Goldilocks ones (round 5: a handle used to build its own tables -- 2^18 powers
        // of a 252-bit root on the host, 6-16 ms per `GpuFft::from(domain)` in the reference's criterion harness, gpu/benches/fft.rs)
        if (log_n > 40) return fail(MS_ERR_INVALID, "log_n = %u too large", log_n);
        if (h_group_gen) {
            f252::E g; memcpy(g.l, h_group_gen, 32);
            if (!f252::eq(g, f252::root_of_unity(log_n))) return fail(MS_ERR_UNSUPPORTED, "group_gen is not arkworks' get_root_of_unity(2^%u)", log_n);
        }
        f252::E h = f252::one();
        if (h_offset) memcpy(h.l, h_offset, 32);
        if (f252::is_zero(h) || f252::geq_p(h)) return fail(MS_ERR_INVALID, "coset offset must be a non-zero canonical element");
        std::lock_guard<std::mutex> lk(ctx->mu);
        ms_ntt_plan* base = nullptr;
        MSCHK(plan252_cached(ctx, log_n, inverse != 0, h, &base));
        ms_ntt_plan* handle = new ms_ntt_plan(*base);
        handle->base = base; handle->refs = 0; handle->queue.clear(); handle->lde2.clear();
        base->refs++;
        *out = handle;
        return MS_OK;
    }
    if (log_n > 32) return fail(MS_ERR_INVALID, "log_n = %u exceeds the field's two-adicity (32)", log_n);
    if (h_group_gen) {
        uint64_t g_m;
        memcpy(&g_m, h_group_gen, 8);
        if (gl::from_mont(g_m) != gl::root_of_unity(log_n))
            return fail(MS_ERR_UNSUPPORTED, "group_gen is not arkworks' get_root_of_unity(2^%u)", log_n);
    }
    uint64_t h = 1;
    if (h_offset) { uint64_t h_m; memcpy(&h_m, h_offset, 8); h = gl::from_mont(h_m); }
    if (h == 0) return fail(MS_ERR_INVALID, "coset offset must be non-zero");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ms_ntt_plan* base = nullptr;
    MSCHK(ctx_plan(ctx, V, log_n, inverse != 0, h, &base));
    ms_ntt_plan* handle = new ms_ntt_plan(*base);
    handle->base = base; handle->refs = 0; handle->queue.clear(); handle->lde2.clear();
    base->refs++;
    *out = handle;
    return MS_OK;
}

// ---- Fp252 plans ------------------------------------------------------------------------
static void powers252(std::vector<uint64_t>& out, size_t count, f252::E base, f252::E first) {
    out.resize(count * 4);
    f252::E x = first;
    for (size_t i = 0; i < count; i++) { memcpy(&out[4 * i], x.l, 32); x = f252::mul(x, base); }
}
static int plan_build252(ms_ctx* ctx, unsigned log_n, bool inverse, const void* h_offset, const void* h_group_gen, ms_ntt_plan** out) {
    if (log_n > 40) return fail(MS_ERR_INVALID, "log_n = %u too large", log_n);
    HIPCHK(hipSetDevice(ctx->device));
    const f252::E gen = f252::root_of_unity(log_n);
    if (h_group_gen) {
        f252::E g; memcpy(g.l, h_group_gen, 32);
        if (!f252::eq(g, gen)) return fail(MS_ERR_UNSUPPORTED, "group_gen is not arkworks' get_root_of_unity(2^%u)", log_n);
    }
    f252::E h = f252::one();
    if (h_offset) memcpy(h.l, h_offset, 32);
    if (f252::is_zero(h) || f252::geq_p(h)) return fail(MS_ERR_INVALID, "coset offset must be a non-zero canonical element");
    const bool coset = !f252::eq(h, f252::one());
    ms_ntt_plan* p = new ms_ntt_plan();
    p->ctx = ctx; p->V = 4; p->log_n = log_n; p->inverse = inverse; p->coset = coset; p->is252 = true;
    memcpy(p->off252, h.l, 32);
    const size_t n = (size_t)1 << log_n;
    const f252::E w = inverse ? f252::inv(gen) : gen;
    // one-level tables up to 2^21 points: every twiddle / scale factor is a single 32-byte load.  (A two-level
    // lookup costs a second Montgomery product per butterfly, and the product -- ~440 VALU instructions -- is
    // what bounds this field.)  Larger domains split the exponent at 2^21.
    p->lo_bits = std::min(21u, log_n);
    std::vector<uint64_t> host, t;
    auto append = [&](const std::vector<uint64_t>& v) { size_t off = host.size(); host.insert(host.end(), v.begin(), v.end()); return off; };
    powers252(t, (size_t)1 << p->lo_bits, w, f252::one()); const size_t o_lo = append(t);
    powers252(t, std::max<size_t>(n >> p->lo_bits, 1), f252::pow_u64(w, (uint64_t)1 << p->lo_bits), f252::one()); const size_t o_hi = append(t);
    size_t o_slo = 0, o_shi = 0;
    const bool scale = inverse || coset;
    if (scale) {
        f252::E g = inverse ? f252::inv(h) : h, c = f252::one();
        if (inverse) { f252::E nn = f252::to_mont(f252::E{{(uint64_t)n, 0, 0, 0}}); c = f252::inv(nn); }
        powers252(t, (size_t)1 << p->lo_bits, g, c); o_slo = append(t);
        powers252(t, std::max<size_t>(n >> p->lo_bits, 1), f252::pow_u64(g, (uint64_t)1 << p->lo_bits), f252::one()); o_shi = append(t);
        if (inverse) p->scale_out252 = 1; else p->scale_in252 = 1;
    }
    size_t o_twr[3] = {0, 0, 0};
    if (log_n >= (unsigned)ms252::TILE_LOG && log_n <= 30) {
        p->np252 = log_n <= 20 ? 2 : 3;
        // MS_NTT252_PASSES=3 forces the three-pass split from 2^17 points on (tests: the emulator cannot hold 2^21 points)
        if (const char* e = getenv("MS_NTT252_PASSES")) if (atoi(e) == 3 && log_n >= 17) p->np252 = 3;
        for (int q = 0; q < p->np252; q++) p->lr252[q] = log_n / p->np252 + ((unsigned)q < log_n % p->np252 ? 1 : 0);
        for (int q = 0; q < p->np252; q++) {
            powers252(t, (size_t)1 << (p->lr252[q] - 1), f252::pow_u64(w, (uint64_t)n >> p->lr252[q]), f252::one());
            o_twr[q] = append(t);
        }
    }
    if (hipMalloc(&p->d_tables, host.size() * 8) != hipSuccess) { delete p; return fail(MS_ERR_NOMEM, "Fp252 plan tables"); }
    if (hipMemcpy(p->d_tables, host.data(), host.size() * 8, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(p->d_tables); delete p; return fail(MS_ERR_HIP, "Fp252 table upload"); }
    p->d252_tw_lo = p->d_tables + o_lo; p->d252_tw_hi = p->d_tables + o_hi;
    for (int q = 0; q < p->np252; q++) p->d252_twr[q] = p->d_tables + o_twr[q];
    if (scale) { p->d252_sc_lo = p->d_tables + o_slo; p->d252_sc_hi = p->d_tables + o_shi; }
    *out = p;
    return MS_OK;
}

// Plans owned by the context, reused by the fused entry points (ms_lde, ms_fri_fold, ...).  The cache is bounded:
// most recently used at the back, and beyond PLAN_CACHE_MAX entries the least recently used plan is destroyed
// (a prover that varies sizes / offsets -- FRI layers, periodic-column cosets -- would otherwise accumulate twiddle
// tables until ms_ctx_destroy).  One call uses at most a handful of plans, so a plan handed out in a call cannot
// be evicted by the same call.
static constexpr size_t PLAN_CACHE_MAX = 32;
static ms_ntt_plan* plan_cache_find(ms_ctx* ctx, unsigned V, unsigned log_n, bool inverse, uint64_t h, const uint64_t* off252 = nullptr) {
    auto& pc = ctx->plan_cache;
    for (size_t i = 0; i < pc.size(); i++) {
        const PlanKey& k = pc[i].first;
        if (k.V != V || k.log_n != log_n || k.inverse != inverse || k.h != h) continue;
        if (off252 && memcmp(pc[i].second->off252, off252, 32) != 0) continue;       // same hash, different offset
        auto hit = pc[i];
        pc.erase(pc.begin() + (long)i);
        pc.push_back(hit);
        return hit.second;
    }
    return nullptr;
}
static void plan_cache_insert(ms_ctx* ctx, const PlanKey& key, ms_ntt_plan* plan) {
    ctx->plan_cache.push_back({key, plan});
    while (ctx->plan_cache.size() > PLAN_CACHE_MAX) {
        size_t victim = 0;
        while (victim + 1 < ctx->plan_cache.size() && ctx->plan_cache[victim].second->refs > 0) victim++;   // least recently used plan without handles
        if (victim + 1 >= ctx->plan_cache.size()) break;       // everything older than the new plan is in use: let the cache grow
        ms_ntt_plan* old = ctx->plan_cache[victim].second;
        ctx->plan_cache.erase(ctx->plan_cache.begin() + (long)victim);
        (void)plan_free(old);                                  // synchronises the stream before freeing the tables
    }
}
int plan252_cached(ms_ctx* ctx, unsigned log_n, bool inverse, const f252::E& h, ms_ntt_plan** out);   // Fp252: keyed by the offset itself
int ctx_plan(ms_ctx* ctx, unsigned V, unsigned log_n, bool inverse, uint64_t h, ms_ntt_plan** out) {
    if ((*out = plan_cache_find(ctx, V, log_n, inverse, h)) != nullptr) return MS_OK;
    MSCHK(plan_build(ctx, V, log_n, inverse, h, out));
    plan_cache_insert(ctx, PlanKey{V, log_n, inverse, h}, *out);
    return MS_OK;
}

static int plan_build(ms_ctx* ctx, unsigned V, unsigned log_n, bool inverse_b, uint64_t h, ms_ntt_plan** out) {
    const int inverse = inverse_b ? 1 : 0;
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t gen = gl::root_of_unity(log_n);           // plain, arkworks get_root_of_unity
    ms_ntt_plan* p = new ms_ntt_plan();
    p->ctx = ctx; p->V = V; p->log_n = log_n; p->inverse = inverse != 0; p->coset = (h != 1); p->offset_canon = h;
    const size_t n = (size_t)1 << log_n;
    const uint64_t w = p->inverse ? gl::inv(gen) : gen;       // transform root
    const uint64_t hinv = gl::inv(h);
    const uint64_t ninv = gl::inv((uint64_t)(n % gl::P));

    std::vector<uint64_t> host;                                // all tables, concatenated
    auto append = [&](const std::vector<uint64_t>& t) { size_t off = host.size(); host.insert(host.end(), t.begin(), t.end()); return off; };
    std::vector<uint64_t> t;
    size_t off_tw = 0, off_si = 0, off_so = 0, off_lo = 0, off_hi = 0, off_alo = 0, off_ahi = 0, off_g = 0, off_wr[4] = {0, 0, 0, 0};
    bool has_si = false, has_so = false, has_aux = false, has_g = false;

    if (log_n < 12) {
        p->small = true;
        powers(t, std::max<size_t>(n / 2, 1), w); off_tw = append(t);
        if (!p->inverse && p->coset) { powers(t, n, h); off_si = append(t); has_si = true; }
        if (p->inverse) { powers(t, n, hinv, ninv); off_so = append(t); has_so = true; }
    }
    // Fp columns of 2^11 points ALSO get the tables of the (256, 8) plan: their dense transforms run as ntt_fused_tiny (ntt_kernels.h: two columns
    // per workgroup), everything else of a small plan (Fq3 columns, zero-extended inputs) stays with ntt_small.  Measured, 512 columns, forward /
    // inverse (profiles/r06_c2_sweep_small.json): 2^11 0.103 / 0.101 -> 0.135 / 0.127 of HBM and 29.4 -> 27.3 us for one column.  At 2^10 and 2^9
    // the same kernel (four and eight columns per workgroup) LOSES to ntt_small, 0.085 -> 0.072 and 0.055 -> 0.035: a batch of such columns is a
    // handful of workgroups whose time is their own latency, and ntt_small's ten cheap stages are the shorter chain there.  Not used below 2^11.
    p->tiny_fused = log_n == 11 && V == 1;
    if (log_n >= 12 || p->tiny_fused) {
        // radix decomposition: R1 = 256, the rest split as evenly as possible into radices 16..256
        const unsigned rest = log_n - 8;
        const int extra = (int)((rest + 7) / 8);
        p->npass = 1 + extra;
        p->lr[0] = 8;
        for (int i = 0; i < extra; i++) p->lr[1 + i] = rest / extra + ((unsigned)i < rest % extra ? 1 : 0);
        // three passes (2^17..2^24): (256, R, 256) with R = n / 2^16 -- both outer passes are the limb-form radix-256 kernels with
        // the uniform inter-pass factor (ntt2_kernels.h), so the last pass always has its limb-form variants (scale of an inverse
        // coset transform, fused bit-reversed store); the small radix sits in the middle: R <= 16 holds its whole network in
        // registers (ntt2_small_mid_pass), R = 32..128 is ntt2_mid_pass_r, R = 256 ntt2_mid_pass
        if (rest >= 9 && rest <= 16) { p->npass = 3; p->lr[1] = rest - 8; p->lr[2] = 8; }
        unsigned acc = 0;
        for (int q = 0; q < p->npass; q++) { p->log_s[q] = acc; acc += p->lr[q]; }
        // digit fields.  pass 1 maps j' = (j2..jm) [jm least significant] to layout (jm..j2) [j2 least]
        {
            unsigned nf = 0, in_shift = 0;
            for (int q = p->npass - 1; q >= 1; q--) {            // jm first (least significant of j')
                unsigned out_shift = 0;
                for (int r = 1; r < q; r++) out_shift += p->lr[r];
                p->fields[0][nf++] = {in_shift, out_shift, (1u << p->lr[q]) - 1};
                in_shift += p->lr[q];
            }
            p->nfields[0] = nf;
        }
        // pass q (0-based, 1 <= q < npass-1): U = (jm..j_{q+2}) [j_{q+2} least significant in U]
        //   -> j' = (j_{q+2}, ..., jm) [jm least significant]
        for (int q = 1; q < p->npass - 1; q++) {
            unsigned nf = 0, in_shift = 0;
            for (int r = q + 1; r < p->npass; r++) {             // r = digit index (0-based) above q
                unsigned out_shift = 0;
                for (int r2 = r + 1; r2 < p->npass; r2++) out_shift += p->lr[r2];
                p->fields[q][nf++] = {in_shift, out_shift, (1u << p->lr[r]) - 1};
                in_shift += p->lr[r];
            }
            p->nfields[q] = nf;
        }
        p->lo_bits = std::min(12u, log_n);
        powers(t, (size_t)1 << p->lo_bits, w); off_lo = append(t);
        powers(t, n >> p->lo_bits, gl::pow(w, (uint64_t)1 << p->lo_bits)); off_hi = append(t);
        for (int q = 0; q < p->npass; q++) {
            powers(t, (size_t)1 << p->lr[q], gl::pow(w, (uint64_t)n >> p->lr[q])); off_wr[q] = append(t);
        }
        if (!p->inverse && p->coset) {
            powers(t, (size_t)1 << p->lo_bits, h); off_alo = append(t);
            powers(t, std::max<size_t>((n >> 8) >> p->lo_bits, 1), gl::pow(h, (uint64_t)1 << p->lo_bits)); off_ahi = append(t);
            powers(t, 256, gl::pow(h, (uint64_t)(n >> 8))); off_g = append(t);
            has_aux = has_g = true;
        }
        if (p->inverse) {
            if (!p->coset) { p->scale_mode = 1; p->scale_const = ninv; }
            else {
                p->scale_mode = 2;
                powers(t, (size_t)1 << p->lo_bits, hinv, ninv); off_alo = append(t);
                powers(t, n >> p->lo_bits, gl::pow(hinv, (uint64_t)1 << p->lo_bits)); off_ahi = append(t);
                has_aux = true;
            }
        }
    }
    // device tables are in Montgomery form: gld::mmul(data, w * 2^64) = data * w
    for (auto& v : host) v = gl::to_mont(v);
    // ... except the tables of the limb-form passes (ntt2_kernels.h): plain residues, four copies
    // {w, w 2^24, w 2^48, w 2^72} per twiddle, appended after the conversion
    size_t off_wr4[4] = {0, 0, 0, 0}, off_twu4[4] = {0, 0, 0, 0}, off_sc4 = 0, off_gp = 0, off_tin4 = 0, off_tout4 = 0;
    bool has_wr4[4] = {false, false, false, false}, has_twu4[4] = {false, false, false, false}, has_gp = false, has_scu4 = false;
    size_t off_scu4 = 0;
    p->uni = !p->small && p->npass == 3 && (p->lr[1] == 8 || p->lr[2] == 8) && p->lr[2] >= 6 && (n * V) % msntt2::TILE == 0;
    if (!p->small) {
        const uint64_t sh[4] = {1, (uint64_t)1 << 24, (uint64_t)1 << 48, gl::pow(2, 72)};
        auto append4 = [&](const std::vector<uint64_t>& plain) {
            const size_t off = host.size();
            host.reserve(off + 4 * plain.size());
            for (uint64_t v : plain) for (int i = 0; i < 4; i++) host.push_back(gl::mul(v, sh[i]));
            return off;
        };
        for (int q = 0; q < p->npass; q++) {
            const bool small_mid = p->uni && q == 1 && p->lr[q] < 8;       // ntt2_small_mid_pass / ntt2_mid_pass_r: [U][k2], k2 < R
            if (p->lr[q] != 8 && !small_mid) continue;
            if (!small_mid) { powers(t, 256, gl::pow(w, (uint64_t)n >> 8)); off_wr4[q] = append4(t); has_wr4[q] = true; }
            else if (p->lr[q] > 4) {                                       // ntt2_mid_pass_r: w_R^e, e < R = 16 T2 (it needs e = a' b < 16 T2)
                powers(t, (size_t)1 << p->lr[q], gl::pow(w, (uint64_t)n >> p->lr[q]));
                off_wr4[q] = append4(t); has_wr4[q] = true;
            }
            if (q >= 1 && q < p->npass - 1) {
                // w_U^k = w_n^((rev(U) k) << log_s): the factor ntt_mid_pass builds per tile in LDS (twl[])
                const size_t R = (size_t)1 << p->lr[q];
                const size_t nU = n >> (p->lr[q] + p->log_s[q]);
                const uint64_t ws = gl::pow(w, (uint64_t)1 << p->log_s[q]);
                t.resize(nU * R);
                for (size_t U = 0; U < nU; U++) {
                    unsigned rU = 0;
                    for (unsigned f = 0; f < p->nfields[q]; f++)
                        rU |= (((unsigned)U >> p->fields[q][f].in_shift) & p->fields[q][f].mask) << p->fields[q][f].out_shift;
                    const uint64_t wu = gl::pow(ws, rU);
                    // UNI plans: pass 2 also carries h^j3 of the inter-pass factor (h w_n^k1)^(R3 j2 + j3), j3 = rev(U)
                    uint64_t x = (p->uni && q == 1 && !p->inverse && p->coset) ? gl::pow(h, rU) : 1;
                    for (size_t k = 0; k < R; k++) { t[U * R + k] = x; x = gl::mul(x, wu); }
                }
                off_twu4[q] = append4(t); has_twu4[q] = true;
            }
        }
        t.assign(1, ninv); off_sc4 = append4(t);
        if (p->scale_mode == 2 && p->lr[p->npass - 1] == 8) {      // inverse coset, last radix 256: h^-(k 2^log_s) per output row
            powers(t, 256, gl::pow(hinv, (uint64_t)1 << p->log_s[p->npass - 1])); off_scu4 = append4(t); has_scu4 = true;
        }
        if (p->uni) {
            // pass 1: tin4[j2][b][a'] = w_256^(a' b) w_n^(a' R3 j2) = w_n^(a' (b n/256 + R3 j2));
            //         tout4[j2][b'] = h^(R3 j2) w_n^(16 b' R3 j2)      (h = 1 unless this is a forward coset transform)
            const unsigned r3 = p->lr[2];
            const unsigned R2 = 1u << p->lr[1];
            const uint64_t hh = (!p->inverse && p->coset) ? h : 1;
            t.resize((size_t)R2 * 256);
            for (unsigned j2 = 0; j2 < R2; j2++)
                for (unsigned b = 0; b < 16; b++) {
                    const uint64_t m = ((uint64_t)b * (n >> 8) + ((uint64_t)j2 << r3)) & (n - 1);
                    const uint64_t wm = gl::pow(w, m);
                    uint64_t x = 1;
                    for (unsigned a = 0; a < 16; a++) { t[((size_t)j2 * 16 + b) * 16 + a] = x; x = gl::mul(x, wm); }
                }
            off_tin4 = append4(t);
            t.resize((size_t)R2 * 16);
            for (unsigned j2 = 0; j2 < R2; j2++) {
                const uint64_t wm = gl::pow(w, (((uint64_t)j2 << r3) * 16) & (n - 1));
                uint64_t x = gl::pow(hh, (uint64_t)j2 << r3);
                for (unsigned bp = 0; bp < 16; bp++) { t[(size_t)j2 * 16 + bp] = x; x = gl::mul(x, wm); }
            }
            off_tout4 = append4(t);
        }
        if (!p->inverse && p->coset) { powers(t, 256, gl::pow(h, (uint64_t)(n >> 8))); off_gp = append4(t); has_gp = true; }
    }
    p->scale_const = gl::to_mont(p->scale_const);
    hipError_t e = hipMalloc(&p->d_tables, host.size() * 8);
    if (e != hipSuccess) { delete p; return fail(MS_ERR_NOMEM, "plan tables (%zu bytes): %s", host.size() * 8, hipGetErrorString(e)); }
    e = hipMemcpy(p->d_tables, host.data(), host.size() * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(p->d_tables); delete p; return fail(MS_ERR_HIP, "plan table upload: %s", hipGetErrorString(e)); }
    if (p->small) {
        p->d_tw = p->d_tables + off_tw;
        p->d_scale_in = has_si ? p->d_tables + off_si : nullptr;
        p->d_scale_out = has_so ? p->d_tables + off_so : nullptr;
    }
    if (!p->small || p->tiny_fused) {
        p->d_tw_lo = p->d_tables + off_lo; p->d_tw_hi = p->d_tables + off_hi;
        for (int q = 0; q < p->npass; q++) p->d_wr[q] = p->d_tables + off_wr[q];
        if (has_aux) { p->d_aux_lo = p->d_tables + off_alo; p->d_aux_hi = p->d_tables + off_ahi; }
        if (has_g) p->d_gtab = p->d_tables + off_g;
        for (int q = 0; q < p->npass; q++) {
            if (has_wr4[q]) p->d_wr4[q] = p->d_tables + off_wr4[q];
            if (has_twu4[q]) p->d_twu4[q] = p->d_tables + off_twu4[q];
        }
        p->d_sc4 = p->d_tables + off_sc4;
        if (has_scu4) p->d_scu4 = p->d_tables + off_scu4;
        if (p->uni) { p->d_tin4 = p->d_tables + off_tin4; p->d_tout4 = p->d_tables + off_tout4; }
        if (has_gp) p->d_g4 = p->d_tables + off_gp;
    }
    *out = p;
    return MS_OK;
}

extern "C" int ms_ntt_plan_destroy(ms_ntt_plan* plan) {
    if (!plan) return MS_OK;
    {
        std::lock_guard<std::mutex> lk(g_user_plans_mu);
        if (!g_user_plans.erase(plan)) return MS_OK;               // released together with its context
    }
    return plan_free(plan);
}
int plan_free_cached(ms_ntt_plan* plan) { return plan_free(plan); }
// cached plans (owned by the context) and plans leaving the registry
static int plan_free(ms_ntt_plan* plan) {
    if (!plan) return MS_OK;
    if (plan->base) {                                              // a handle: the cached plan keeps the tables
        { std::lock_guard<std::mutex> lk(plan->ctx->mu); plan->base->refs--; }
        delete plan;
        return MS_OK;
    }
    (void)hipStreamSynchronize(plan->ctx->stream);
    if (plan->ctx->stream2) (void)hipStreamSynchronize(plan->ctx->stream2);
    if (plan->d_tables) (void)hipFree(plan->d_tables);
    if (plan->d_oscale) (void)hipFree(plan->d_oscale);
    for (auto& l : plan->lde2) if (l.d) (void)hipFree(l.d);
    delete plan;
    return MS_OK;
}

template <int RB, bool INV, bool LAST>
static void launch_mid_scale(int scale, bool bitrev, dim3 grid, hipStream_t st, const msntt::PassParams& P) {
    if constexpr (LAST) {
        if (bitrev) {      // fused bit-reversed store (LDE): forward transforms only carry scale 0
            if (scale == 1) hipLaunchKernelGGL((msntt::ntt_mid_pass<RB, INV, true, 1, true>), grid, dim3(msntt::NT), 0, st, P);
            else if (scale == 2) hipLaunchKernelGGL((msntt::ntt_mid_pass<RB, INV, true, 2, true>), grid, dim3(msntt::NT), 0, st, P);
            else hipLaunchKernelGGL((msntt::ntt_mid_pass<RB, INV, true, 0, true>), grid, dim3(msntt::NT), 0, st, P);
            return;
        }
        if (scale == 1) hipLaunchKernelGGL((msntt::ntt_mid_pass<RB, INV, true, 1>), grid, dim3(msntt::NT), 0, st, P);
        else if (scale == 2) hipLaunchKernelGGL((msntt::ntt_mid_pass<RB, INV, true, 2>), grid, dim3(msntt::NT), 0, st, P);
        else hipLaunchKernelGGL((msntt::ntt_mid_pass<RB, INV, true, 0>), grid, dim3(msntt::NT), 0, st, P);
    } else {
        hipLaunchKernelGGL((msntt::ntt_mid_pass<RB, INV, false, 0>), grid, dim3(msntt::NT), 0, st, P);
    }
}
template <int RB>
static void launch_mid(bool inv, bool last, int scale, bool bitrev, dim3 grid, hipStream_t st, const msntt::PassParams& P) {
    if (inv) { if (last) launch_mid_scale<RB, true, true>(scale, bitrev, grid, st, P); else launch_mid_scale<RB, true, false>(scale, false, grid, st, P); }
    else     { if (last) launch_mid_scale<RB, false, true>(scale, bitrev, grid, st, P); else launch_mid_scale<RB, false, false>(scale, false, grid, st, P); }
}

int bit_reverse_run(ms_ctx* ctx, unsigned V, unsigned log_n, const void* const* src, void* const* dst, unsigned ncols);

static uint64_t offset_key252(const f252::E& h) {
    uint64_t k = 1469598103934665603ull;
    for (int w = 0; w < 4; w++) { k ^= h.l[w]; k *= 1099511628211ull; }
    return k | ((uint64_t)1 << 63);
}
int plan252_cached(ms_ctx* ctx, unsigned log_n, bool inverse, const f252::E& h, ms_ntt_plan** out) {
    const bool coset = !f252::eq(h, f252::one());
    const uint64_t key = coset ? offset_key252(h) : 1;
    if ((*out = plan_cache_find(ctx, 4, log_n, inverse, key, h.l)) != nullptr) return MS_OK;
    MSCHK(plan_build252(ctx, log_n, inverse, h.l, nullptr, out));
    plan_cache_insert(ctx, PlanKey{4, log_n, inverse, key}, *out);
    return MS_OK;
}

// Tiled passes (fp252_ntt_kernels.h).  log_zero_ext: the source holds only the first n >> log_zero_ext elements, the rest
// of the domain is implicit zeros (needs 2^log_zero_ext <= R_0); bitrev_out: bit-reversed order, fused into the last pass.
static int plan_run252_tiled(ms_ntt_plan* p, const void* const* src, void* const* dst, unsigned ncols, unsigned log_zero_ext, bool bitrev_out) {
    ms_ctx* ctx = p->ctx;
    hipStream_t st = ctx->stream;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << p->log_n, col_bytes = n * 32;
    const int np = p->np252;
    if (log_zero_ext > p->lr252[0]) return fail(MS_ERR_INVALID, "internal: zero extension 2^%u beyond the first radix", log_zero_ext);
    unsigned group = (unsigned)std::max<size_t>(1, std::min<size_t>(msntt::MAXC, ctx->group_bytes / col_bytes));
    group = std::min(group, ncols);
    void* scratch = nullptr;
    MSCHK(ctx_scratch(ctx, (size_t)group * col_bytes, &scratch));
    static const char* const names[3] = {"ntt252_pass1", "ntt252_pass2", "ntt252_pass3"};
    if (getenv("MS_NTT_DEBUG"))
        fprintf(stderr, "[ms_ntt] Fp252 log_n=%u tiled: %d passes, radices 2^%u 2^%u 2^%u, zero extension 2^%u, bitrev %d\n", p->log_n, np,
                p->lr252[0], p->lr252[1], p->lr252[2], log_zero_ext, (int)bitrev_out);
    for (unsigned c0 = 0; c0 < ncols; c0 += group) {
        const unsigned nc = std::min(group, ncols - c0);
        unsigned done = 0;                                        // log2 of R_0 .. R_(q-1)
        for (int q = 0; q < np; q++) {
            ms252::PassParams P;
            memset(&P, 0, sizeof P);
            const bool last = q == np - 1;
            for (unsigned c = 0; c < nc; c++) {
                uint64_t* scr = (uint64_t*)((char*)scratch + (size_t)c * col_bytes);
                P.src[c] = q == 0 ? (const uint64_t*)src[c0 + c] : scr;
                P.dst[c] = last ? (uint64_t*)dst[c0 + c] : scr;
            }
            P.twr = p->d252_twr[q]; P.tw_lo = p->d252_tw_lo; P.tw_hi = p->d252_tw_hi; P.sc_lo = p->d252_sc_lo; P.sc_hi = p->d252_sc_hi;
            P.log_n = p->log_n; P.lo_bits = p->lo_bits;
            P.log_r = p->lr252[q]; P.log_c = ms252::TILE_LOG - P.log_r;
            P.log_s = p->log_n - done - P.log_r; P.log_tw = done;
            P.valid_rows = (1u << p->lr252[0]) >> log_zero_ext;
            P.log_r0 = p->lr252[0]; P.log_r1 = np == 3 ? p->lr252[1] : 0;
            P.scale_in = p->scale_in252; P.scale_out = p->scale_out252; P.bitrev_out = bitrev_out ? 1 : 0;
            const dim3 grid((unsigned)(n >> ms252::TILE_LOG), nc), block(ms252::NT2);
            ProfScope ps(ctx, names[q], 2.0 * col_bytes * nc);
            if (q == 0) hipLaunchKernelGGL((ms252::ntt252_strided_pass<ms252::NT2, true>), grid, block, 0, st, P);
            else if (!last) hipLaunchKernelGGL((ms252::ntt252_strided_pass<ms252::NT2, false>), grid, block, 0, st, P);
            else hipLaunchKernelGGL((ms252::ntt252_last_pass<ms252::NT2>), grid, block, 0, st, P);
            done += P.log_r;
        }
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}

static int plan_run252(ms_ntt_plan* p, const void* const* src, void* const* dst, unsigned ncols) {
    if (p->np252) return plan_run252_tiled(p, src, dst, ncols, 0, false);
    ms_ctx* ctx = p->ctx;
    hipStream_t st = ctx->stream;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << p->log_n;
    for (unsigned c = 0; c < ncols; c++)
        if (src[c] != dst[c]) HIPCHK(hipMemcpyAsync(dst[c], src[c], n * 32, hipMemcpyDeviceToDevice, st));
    MSCHK(bit_reverse_run(ctx, 4, p->log_n, (const void* const*)dst, dst, ncols));
    for (unsigned c = 0; c < ncols; c++) {
        ms252::Params P;
        memset(&P, 0, sizeof P);
        P.col = (uint64_t*)dst[c]; P.tw_lo = p->d252_tw_lo; P.tw_hi = p->d252_tw_hi; P.sc_lo = p->d252_sc_lo; P.sc_hi = p->d252_sc_hi;
        P.log_n = p->log_n; P.lo_bits = p->lo_bits; P.scale_in = p->scale_in252; P.scale_out = p->scale_out252;
        const unsigned clog = std::min<unsigned>(p->log_n, ms252::CHUNK_LOG);
        {
            ProfScope ps(ctx, "ntt252_local", 64.0 * n);
            hipLaunchKernelGGL(ms252::ntt252_local, dim3((unsigned)(n >> clog)), dim3(ms252::NT), 0, st, P);
        }
        for (unsigned s = clog; s < p->log_n;) {                 // stages s+1 .. s+R per launch
            const unsigned R = std::min(ms252::MAX_FUSED_STAGES, p->log_n - s);
            P.stage = s;
            const dim3 g((unsigned)(((n >> R) + ms252::NT - 1) / ms252::NT));
            ProfScope ps(ctx, "ntt252_stages", 64.0 * n);
            if (R == 1) hipLaunchKernelGGL(ms252::ntt252_stages<1>, g, dim3(ms252::NT), 0, st, P);
            else hipLaunchKernelGGL(ms252::ntt252_stages<2>, g, dim3(ms252::NT), 0, st, P);
            s += R;
        }
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}

// Transform `ncols` columns: src[c] -> dst[c] (may alias).  valid_rows < 256 means the
// source only holds the first valid_rows/256 of the domain, the rest is implicit zeros.
static int lde2_run(ms_ntt_plan* fwd, unsigned log_n, unsigned log_b, const void* const* src, void* const* dst, unsigned ncols, bool natural, unsigned V = 1, ms_ntt_plan* inv = nullptr);
int plan_run(ms_ntt_plan* p, const void* const* src, void* const* dst, unsigned ncols, unsigned valid_rows, bool bitrev_out) {
    if (p->is252) {
        if (valid_rows != 256 || bitrev_out) return fail(MS_ERR_INVALID, "internal: Fp252 zero extension / fused bit reversal go through plan_run252_tiled");
        return plan_run252(p, src, dst, ncols);
    }
    if (bitrev_out && p->small) return fail(MS_ERR_INVALID, "internal: fused bit reversal needs the multi-pass path");
    ms_ctx* ctx = p->ctx;
    hipStream_t st = ctx->stream;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << p->log_n;
    const size_t col_bytes = n * p->V * 8;
    static const bool fused_off = getenv("MS_NTT_FUSED_SMALL") && !strcmp(getenv("MS_NTT_FUSED_SMALL"), "0");
    // the tables every fused launch takes (ntt_fused_small / ntt_fused_tiny): the column pointers travel as a table in the staging ring
    auto fused_params = [&](const void* d_tab) {
        msntt::FusedParams F;
        memset(&F, 0, sizeof F);
        F.cols = (const uint64_t* const*)d_tab;
        F.tw_lo = p->d_tw_lo; F.tw_hi = p->d_tw_hi; F.wr = p->d_wr[0]; F.wr2 = p->d_wr[1];
        F.aux_lo = p->d_aux_lo; F.aux_hi = p->d_aux_hi; F.gtab = p->d_gtab;
        F.log_n = p->log_n; F.lo_bits = p->lo_bits;
        F.nfields = p->nfields[0];
        for (unsigned f = 0; f < F.nfields; f++) F.fields[f] = p->fields[0][f];
        F.scale_const = p->scale_const;
        return F;
    };
    if (p->small && p->tiny_fused && !fused_off && valid_rows == 256 && !bitrev_out) {
        constexpr unsigned PER_LAUNCH = 4096;
        for (unsigned c0 = 0; c0 < ncols; c0 += PER_LAUNCH) {
            const unsigned nc = std::min<unsigned>(PER_LAUNCH, ncols - c0);
            std::vector<const void*> tab(2 * (size_t)nc);
            for (unsigned c = 0; c < nc; c++) { tab[2 * c] = src[c0 + c]; tab[2 * c + 1] = dst[c0 + c]; }
            LockedPoolGuard pooled(ctx);
            const void* d_tab = nullptr;
            MSCHK(stage_view(ctx, tab.data(), tab.size() * sizeof(void*), &d_tab, pooled));
            msntt::FusedParams P = fused_params(d_tab);
            P.ncols = nc;
            const unsigned pack = 16u >> (p->log_n - 8);          // columns per workgroup: 2 / 4 / 8 at 2^11 / 2^10 / 2^9 points
            const dim3 grid((nc + pack - 1) / pack), block(msntt::NT);
            ProfScope ps(ctx, "ntt_fused_tiny", 2.0 * col_bytes * nc);
#define MS_TINY(LOGN_) do { \
            if (p->inverse) { \
                if (p->scale_mode == 2) hipLaunchKernelGGL((msntt::ntt_fused_tiny<LOGN_, true, false, 2>), grid, block, 0, st, P); \
                else if (p->scale_mode == 1) hipLaunchKernelGGL((msntt::ntt_fused_tiny<LOGN_, true, false, 1>), grid, block, 0, st, P); \
                else hipLaunchKernelGGL((msntt::ntt_fused_tiny<LOGN_, true, false, 0>), grid, block, 0, st, P); \
            } else if (p->coset) hipLaunchKernelGGL((msntt::ntt_fused_tiny<LOGN_, false, true, 0>), grid, block, 0, st, P); \
            else hipLaunchKernelGGL((msntt::ntt_fused_tiny<LOGN_, false, false, 0>), grid, block, 0, st, P); } while (0)
            MS_TINY(11);
#undef MS_TINY
        }
        HIPCHK(hipGetLastError());
        return MS_OK;
    }
    if (p->small) {
        if (valid_rows != 256) return fail(MS_ERR_INVALID, "zero-extended input needs a domain of at least 4096 points");
        constexpr unsigned PER_LAUNCH = 4096;                  // 64 KiB of pointers, read in place from the staging ring (stage_view)
        for (unsigned c0 = 0; c0 < ncols; c0 += PER_LAUNCH) {
            const unsigned nc = std::min<unsigned>(PER_LAUNCH, ncols - c0);
            std::vector<const void*> tab(2 * (size_t)nc);
            for (unsigned c = 0; c < nc; c++) { tab[2 * c] = src[c0 + c]; tab[2 * c + 1] = dst[c0 + c]; }
            LockedPoolGuard pooled(ctx);
            const void* d_tab = nullptr;
            MSCHK(stage_view(ctx, tab.data(), tab.size() * sizeof(void*), &d_tab, pooled));
            msntt::SmallParams S;
            memset(&S, 0, sizeof S);
            S.cols = (const uint64_t* const*)d_tab;
            S.tw = p->d_tw; S.scale_in = p->d_scale_in; S.scale_out = p->d_scale_out; S.log_n = p->log_n; S.V = p->V;
            ProfScope ps(ctx, "ntt_small", 2.0 * col_bytes * nc);
            hipLaunchKernelGGL(msntt::ntt_small, dim3(1, nc), dim3(msntt::NT), 0, st, S);
        }
        HIPCHK(hipGetLastError());
        return MS_OK;
    }
    // Forward transforms of 2^18-point Fp columns: TWO passes -- the coset-LDE kernels with a single coset and natural-order stores
    // (256 x 1024: runs of 16 words in the second pass's stores; from 2^19 on the runs would be 64 bytes and the three-pass plan wins).
    // Measured (profiles/r04_c2_sweep.json): 2.48 -> 2.25 us per column over 64 columns.  At 2^17 the rows are too short for the uniform
    // split of pass A's factor (T = 2) and its per-lane running product loses: 1.50 against 1.37 us -- stays on three passes.
    if (!p->inverse && p->V == 1 && valid_rows == 256 && !bitrev_out && p->log_n == 18 && p->d_wr4[0] != nullptr)
        return lde2_run(p->base ? p->base : p, p->log_n, 0, src, dst, ncols, true);   // the CACHED plan owns (and frees) the tables: a handle is a copy
    // ... and the INVERSE transforms of that size through the same two kernels (round 6): the column read backwards is the inverse's sum, the
    // forward plan on the subgroup supplies every table, n^-1 h^-k multiplies pass B's natural-order output (lde2_kernels.h Params::rev).
    // Measured, 64 columns: 2.65 -> 2.35 us per column (forward: 2.15); MS_NTT_INV18_TWO_PASS=0: the three passes.
    static const bool inv18_off = getenv("MS_NTT_INV18_TWO_PASS") && !strcmp(getenv("MS_NTT_INV18_TWO_PASS"), "0");
    if (p->inverse && !inv18_off && p->V == 1 && valid_rows == 256 && !bitrev_out && p->log_n == 18) {
        ms_ntt_plan* owner = p->base ? p->base : p;
        ms_ntt_plan* fwd1 = nullptr;
        MSCHK(ctx_plan(ctx, 1, 18, false, 1, &fwd1));
        if (fwd1->d_wr4[0] != nullptr) {
            if (owner->coset && !owner->d_oscale) {               // n^-1 h^-k, k < n: built once per (cached) plan
                const size_t n18 = (size_t)1 << 18;
                std::vector<uint64_t> sc(n18);
                const uint64_t hinv = gl::inv(owner->offset_canon);
                uint64_t x = gl::inv((uint64_t)n18);
                for (size_t k = 0; k < n18; k++) { sc[k] = gl::to_mont(x); x = gl::mul(x, hinv); }
                uint64_t* d = nullptr;
                if (hipMalloc(&d, n18 * 8) != hipSuccess) return fail(MS_ERR_NOMEM, "inverse scale table");
                if (hipMemcpy(d, sc.data(), n18 * 8, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return fail(MS_ERR_HIP, "inverse scale table upload"); }
                owner->d_oscale = d;
            }
            return lde2_run(fwd1, 18, 0, src, dst, ncols, true, 1, owner);   // (the owner's d_oscale: a handle's copy of the plan may predate the table)
        }
    }
    // 2^12- and 2^13-point Fp columns (the (256, 16) / (256, 32) plans): both passes in ONE launch, the column stays in LDS in between, and one
    // launch takes any number of columns (ntt_kernels.h ntt_fused_small; MS_NTT_FUSED_SMALL=0: the two launches, for before / after timings)
    static const unsigned fused_max = getenv("MS_NTT_FUSED_MAX_LOG") ? (unsigned)atoi(getenv("MS_NTT_FUSED_MAX_LOG")) : 14u;
    if (!fused_off && p->V == 1 && p->log_n >= 12 && p->log_n <= std::min(14u, fused_max) && p->npass == 2 && p->lr[0] == 8 && p->lr[1] == p->log_n - 8 && valid_rows == 256 && !bitrev_out) {
        constexpr unsigned PER_LAUNCH = 4096;                  // 64 KiB of pointers: an eighth of the staging ring's half (stage_view reads them in place)
        for (unsigned c0 = 0; c0 < ncols; c0 += PER_LAUNCH) {
            const unsigned nc = std::min<unsigned>(PER_LAUNCH, ncols - c0);
            std::vector<const void*> tab(2 * (size_t)nc);
            for (unsigned c = 0; c < nc; c++) { tab[2 * c] = src[c0 + c]; tab[2 * c + 1] = dst[c0 + c]; }
            LockedPoolGuard pooled(ctx);
            const void* d_tab = nullptr;
            MSCHK(stage_view(ctx, tab.data(), tab.size() * sizeof(void*), &d_tab, pooled));
            const msntt::FusedParams P = fused_params(d_tab);
            const dim3 grid(nc), block(msntt::NT << (p->log_n - 12));
            ProfScope ps(ctx, "ntt_fused_small", 2.0 * col_bytes * nc);
#define MS_FUSED(LOGN_) do { \
            if (p->inverse) { \
                if (p->scale_mode == 2) hipLaunchKernelGGL((msntt::ntt_fused_small<LOGN_, true, false, 2>), grid, block, 0, st, P); \
                else if (p->scale_mode == 1) hipLaunchKernelGGL((msntt::ntt_fused_small<LOGN_, true, false, 1>), grid, block, 0, st, P); \
                else hipLaunchKernelGGL((msntt::ntt_fused_small<LOGN_, true, false, 0>), grid, block, 0, st, P); \
            } else if (p->coset) hipLaunchKernelGGL((msntt::ntt_fused_small<LOGN_, false, true, 0>), grid, block, 0, st, P); \
            else hipLaunchKernelGGL((msntt::ntt_fused_small<LOGN_, false, false, 0>), grid, block, 0, st, P); } while (0)
            if (p->log_n == 12) MS_FUSED(12); else if (p->log_n == 13) MS_FUSED(13); else MS_FUSED(14);
#undef MS_FUSED
        }
        HIPCHK(hipGetLastError());
        return MS_OK;
    }
    unsigned group = (unsigned)std::max<size_t>(1, std::min<size_t>(MAXC, ctx->group_bytes / col_bytes));
    group = std::min(group, ncols);
    // uniform-factor plans on Fp columns: pass 1 stores whole lines in a row order that permutes the words inside every run of
    // 64; pass 2 un-permutes while it reads, in place on the scratch column (its tile owns those 64 words), and the last pass
    // goes scratch -> dst (natural order, or bit-reversed for the LDE).
    const bool perm = p->uni && p->V == 1;
    void* scratch = nullptr;
    MSCHK(ctx_scratch(ctx, (size_t)group * col_bytes, &scratch));
    const unsigned tiles = (unsigned)(n * p->V / msntt::TILE);
    // MS_NTT_STREAMS=2: the two halves of a group on two streams -- kernels of different passes then overlap (a pass
    // alternates between a memory phase and an arithmetic phase per workgroup); measured 162 -> 159 us per 2^24 column
    // over 8 columns.  Off by default: with concurrent kernels the per-kernel durations of a trace no longer add up to the
    // wall time, and the gain is under 2 %.  Never while per-launch profiling is on (its events sit on one stream).
    // Columns below 2^20 points lose with it (round 4, MS_NTT_STREAMS_MIN_LOG=12: 2^16 x 128 0.67 -> 0.88 us per column, 2^17 x 64 1.40 -> 1.60).
    static const bool two_streams = getenv("MS_NTT_STREAMS") != nullptr && atoi(getenv("MS_NTT_STREAMS")) == 2;
    static const unsigned two_min_log = getenv("MS_NTT_STREAMS_MIN_LOG") ? (unsigned)atoi(getenv("MS_NTT_STREAMS_MIN_LOG")) : 20u;
    const bool two = two_streams && !ctx->profiling && ncols >= 2 && group >= 2 && p->log_n >= two_min_log;
    if (two) {
        if (!ctx->stream2) {
            HIPCHK(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
        }
        HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
        HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    }
    void* const scratch_all = scratch;
    const hipStream_t st_main = st;
    for (unsigned g0 = 0; g0 < ncols; g0 += group) {
      const unsigned gnc = std::min(group, ncols - g0);
      const unsigned half = (two && gnc >= 2) ? gnc / 2 : gnc;
      for (unsigned s0 = 0; s0 < gnc; s0 = (s0 == 0) ? half : gnc) {     // at most two ranges: [0, half), [half, gnc)
        const unsigned c0 = g0 + s0, nc = (s0 == 0) ? half : gnc - half;
        const hipStream_t st = (s0 == 0) ? st_main : ctx->stream2;
        void* const scratch = (char*)scratch_all + (size_t)s0 * col_bytes;
        for (int q = 0; q < p->npass; q++) {
            msntt::PassParams P;
            memset(&P, 0, sizeof P);
            const bool last = (q == p->npass - 1);
            for (unsigned c = 0; c < nc; c++) {
                uint64_t* scr = (uint64_t*)((char*)scratch + (size_t)c * col_bytes);
                P.src[c] = (q == 0) ? (const uint64_t*)src[c0 + c] : scr;
                P.dst[c] = last ? (uint64_t*)dst[c0 + c] : scr;
            }
            P.tw_lo = p->d_tw_lo; P.tw_hi = p->d_tw_hi; P.wr = p->d_wr[q];
            P.aux_lo = p->d_aux_lo; P.aux_hi = p->d_aux_hi; P.gtab = p->d_gtab;
            P.log_n = p->log_n; P.V = p->V; P.valid_rows = valid_rows; P.lo_bits = p->lo_bits; P.log_s = p->log_s[q];
            P.nfields = p->nfields[q];
            for (unsigned f = 0; f < P.nfields; f++) P.fields[f] = p->fields[q][f];
            P.scale_const = p->scale_const;
            dim3 grid(tiles, nc);
            static const char* const pass_names[4] = {"ntt_pass1", "ntt_pass2", "ntt_pass3", "ntt_pass4"};
            ProfScope ps(ctx, pass_names[q], 2.0 * col_bytes * nc);
            // limb-form radix-256 passes (ntt2_kernels.h) wherever a pass has radix 256 and rows of >= 64 words
            const size_t pass_sw = ((size_t)1 << p->log_s[q]) * p->V;
            // (the per-element scale walk of an inverse coset transform stays with the round-1 last pass: the walk is two table
            // loads and a Montgomery product per word, which the 4-wave limb kernel hides worse -- 91 vs 75 us per 2^24
            // column; so does the fused bit-reversed store of Fq3 columns, whose runs interleave three words)
            const bool small_mid = p->uni && q == 1 && p->lr[q] < 8;
            const bool v2_ok = small_mid || (p->lr[q] == 8 && (n * p->V) % msntt2::TILE == 0 &&
                               (q == 0 ? ((n >> 8) * p->V) % msntt2::TW == 0
                                       : (pass_sw % msntt2::TW == 0 && !(last && p->scale_mode == 2 && p->d_scu4 == nullptr) &&
                                          !(last && bitrev_out && (p->V != 1 || p->inverse || p->scale_mode != 0)))));
            static const bool dbg = getenv("MS_NTT_DEBUG") != nullptr;
            if (dbg) fprintf(stderr, "[ms_ntt] log_n=%u V=%u pass %d/%d radix 2^%u: %s kernel\n", p->log_n, p->V, q + 1, p->npass, p->lr[q], v2_ok ? (p->uni && q < 2 ? "limb-form (ntt2), uniform inter-pass factor" : "limb-form (ntt2)") : "round-1");
            if (p->uni && q < 2 && !v2_ok) return fail(MS_ERR_INVALID, "internal: uniform inter-pass plan without its limb-form passes");
            if (v2_ok) {
                msntt2::Params Q;
                memset(&Q, 0, sizeof Q);
                for (unsigned c = 0; c < nc; c++) { Q.src[c] = P.src[c]; Q.dst[c] = P.dst[c]; }
                Q.wr4 = p->d_wr4[q]; Q.twu4 = p->d_twu4[q]; Q.sc4 = p->d_sc4; Q.scu4 = p->d_scu4; Q.g4 = p->d_g4;
                Q.tw_lo = p->d_tw_lo; Q.tw_hi = p->d_tw_hi; Q.aux_lo = p->d_aux_lo; Q.aux_hi = p->d_aux_hi;
                Q.log_n = p->log_n; Q.V = p->V; Q.valid_rows = valid_rows; Q.lo_bits = p->lo_bits; Q.log_s = p->log_s[q];
                Q.tin4 = p->d_tin4; Q.tout4 = p->d_tout4; Q.r3 = p->lr[2];
                // pass 1 of a uniform-factor plan with last radix 256 (four tiles per block of a row): XCD-aware tile order (ntt2_kernels.h)
                Q.xcd_map = (q == 0 && p->uni && p->V == 1 && p->lr[2] == 8 && (n / msntt2::TILE) % 8 == 0) ? 1u : 0u;
                Q.nfields = P.nfields;
                for (unsigned f = 0; f < P.nfields; f++) Q.fields[f] = P.fields[f];
                const dim3 g2((unsigned)(n * p->V / msntt2::TILE), nc), b2(msntt2::NT);
                // non-temporal tile accesses once the columns of this launch and their scratch exceed the Infinity Cache (ntt2_kernels.h)
                const bool stream_hint = 2 * (size_t)nc * col_bytes > ((size_t)256 << 20);
#define MS_K2(NAME, ...) do { if (stream_hint) hipLaunchKernelGGL((msntt2::NAME<true, __VA_ARGS__>), g2, b2, 0, st, Q); \
                              else hipLaunchKernelGGL((msntt2::NAME<false, __VA_ARGS__>), g2, b2, 0, st, Q); } while (0)
                if (q == 0) {
                    const bool cos = (!p->inverse && p->coset);
                    const int na = valid_rows == 64 ? 4 : valid_rows == 32 ? 2 : valid_rows == 16 ? 1 : 16;
#define MS_P1(INV, COS, NA) do { if (perm) MS_K2(ntt2_first_pass, INV, COS, NA, true, true); \
                                 else if (p->uni) MS_K2(ntt2_first_pass, INV, COS, NA, true); \
                                 else MS_K2(ntt2_first_pass, INV, COS, NA, false); } while (0)
                    if (p->inverse) MS_P1(true, false, 16);
                    else if (cos) {
                        if (na == 4) MS_P1(false, true, 4);
                        else if (na == 2) MS_P1(false, true, 2);
                        else if (na == 1) MS_P1(false, true, 1);
                        else MS_P1(false, true, 16);
                    } else {
                        if (na == 4) MS_P1(false, false, 4);
                        else if (na == 2) MS_P1(false, false, 2);
                        else if (na == 1) MS_P1(false, false, 1);
                        else MS_P1(false, false, 16);
                    }
#undef MS_P1
                } else if (small_mid) {     // (256, R, 256) plans: the whole radix-R network in registers
#define MS_SM(LOGR) do { if (p->inverse) { if (perm) MS_K2(ntt2_small_mid_pass, true, LOGR, true); \
                                           else MS_K2(ntt2_small_mid_pass, true, LOGR, false); } \
                         else { if (perm) MS_K2(ntt2_small_mid_pass, false, LOGR, true); \
                                else MS_K2(ntt2_small_mid_pass, false, LOGR, false); } } while (0)
#define MS_MR(LOGT2) do { if (p->inverse) { if (perm) MS_K2(ntt2_mid_pass_r, true, LOGT2, true); \
                                            else MS_K2(ntt2_mid_pass_r, true, LOGT2, false); } \
                          else { if (perm) MS_K2(ntt2_mid_pass_r, false, LOGT2, true); \
                                 else MS_K2(ntt2_mid_pass_r, false, LOGT2, false); } } while (0)
                    switch (p->lr[q]) {
                    case 1: MS_SM(1); break; case 2: MS_SM(2); break; case 3: MS_SM(3); break; case 4: MS_SM(4); break;
                    case 5: MS_MR(1); break; case 6: MS_MR(2); break; default: MS_MR(3); break;
                    }
#undef MS_MR
#undef MS_SM
                } else if (!last) {
                    if (perm) {             // ... and reads pass 1's permuted rows, writes the natural order (in place)
                        if (p->inverse) MS_K2(ntt2_mid_pass, true, false, 0, true, true);
                        else MS_K2(ntt2_mid_pass, false, false, 0, true, true);
                    } else if (p->uni) {    // q == 1 of three: applies the per-lane remainder of pass 1's factor on its loads
                        if (p->inverse) MS_K2(ntt2_mid_pass, true, false, 0, true);
                        else MS_K2(ntt2_mid_pass, false, false, 0, true);
                    } else if (p->inverse) MS_K2(ntt2_mid_pass, true, false, 0);
                    else MS_K2(ntt2_mid_pass, false, false, 0);
                } else if (bitrev_out) {
                    { if (stream_hint) hipLaunchKernelGGL(msntt2::ntt2_last_pass_bitrev<true>, g2, b2, 0, st, Q); else hipLaunchKernelGGL(msntt2::ntt2_last_pass_bitrev<false>, g2, b2, 0, st, Q); }
                } else {
                    const int scale = p->scale_mode;
                    if (p->inverse) {
                        if (scale == 2) MS_K2(ntt2_mid_pass, true, true, 2);
                        else if (scale == 1) MS_K2(ntt2_mid_pass, true, true, 1);
                        else MS_K2(ntt2_mid_pass, true, true, 0);
                    } else MS_K2(ntt2_mid_pass, false, true, 0);
                }
                continue;
            }
            if (q == 0) {
                const bool cos = (!p->inverse && p->coset);
                if (p->inverse) hipLaunchKernelGGL((msntt::ntt_first_pass<true, false>), grid, dim3(msntt::NT), 0, st, P);
                else if (cos) {
                    if (valid_rows == 64) hipLaunchKernelGGL((msntt::ntt_first_pass<false, true, 4>), grid, dim3(msntt::NT), 0, st, P);
                    else if (valid_rows == 32) hipLaunchKernelGGL((msntt::ntt_first_pass<false, true, 2>), grid, dim3(msntt::NT), 0, st, P);
                    else if (valid_rows == 16) hipLaunchKernelGGL((msntt::ntt_first_pass<false, true, 1>), grid, dim3(msntt::NT), 0, st, P);
                    else hipLaunchKernelGGL((msntt::ntt_first_pass<false, true>), grid, dim3(msntt::NT), 0, st, P);
                } else {
                    if (valid_rows == 64) hipLaunchKernelGGL((msntt::ntt_first_pass<false, false, 4>), grid, dim3(msntt::NT), 0, st, P);
                    else if (valid_rows == 32) hipLaunchKernelGGL((msntt::ntt_first_pass<false, false, 2>), grid, dim3(msntt::NT), 0, st, P);
                    else if (valid_rows == 16) hipLaunchKernelGGL((msntt::ntt_first_pass<false, false, 1>), grid, dim3(msntt::NT), 0, st, P);
                    else hipLaunchKernelGGL((msntt::ntt_first_pass<false, false>), grid, dim3(msntt::NT), 0, st, P);
                }
            } else {
                const int scale = last ? p->scale_mode : 0;
                switch (p->lr[q]) {
                case 4: launch_mid<1>(p->inverse, last, scale, bitrev_out, grid, st, P); break;
                case 5: launch_mid<2>(p->inverse, last, scale, bitrev_out, grid, st, P); break;
                case 6: launch_mid<4>(p->inverse, last, scale, bitrev_out, grid, st, P); break;
                case 7: launch_mid<8>(p->inverse, last, scale, bitrev_out, grid, st, P); break;
                case 8: launch_mid<16>(p->inverse, last, scale, bitrev_out, grid, st, P); break;
                default: return fail(MS_ERR_INVALID, "internal: bad pass radix 2^%u", p->lr[q]);
                }
            }
        }
      }
    }
    if (two) {
        HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}

// ---- two-pass coset LDE (lde2_kernels.h) ------------------------------------------------------------------------------
// fwd = the forward plan of the LDE domain (N = n << log_b points, offset h).  Applies to Fp columns of 2^17..2^22 rows
// (rows of pass B: n / 256 <= 16384 words = one workgroup's tile).
#ifndef MS_LDE2_MAX_LOG_N            // (a build with -DMS_LDE2_MAX_LOG_N=20 is the round-3 dispatch, for before / after timings)
#define MS_LDE2_MAX_LOG_N 22
#endif
// Fq3 columns (V = 3, round 6): 2^18..2^20 rows (rows of pass B: at most 4096 elements -- the three word planes of a row share a workgroup;
// at 2^17 rows the (256, 2, 256) plan is the faster one: 0.37 against 0.49 ms for 9 columns at blow-up 8, profiles/r06_lde_fq3_probe.txt).
static bool lde2_applicable(const ms_ntt_plan* fwd, unsigned V, unsigned log_n, unsigned log_b) {
    static const bool fq3_off = getenv("MS_LDE2_FQ3") && !strcmp(getenv("MS_LDE2_FQ3"), "0");      // (before / after timings, tests of the three-pass route)
    static const bool fq3_all = getenv("MS_LDE2_FQ3") && !strcmp(getenv("MS_LDE2_FQ3"), "all");    // (the tests reach the T = 2 instantiation)
    if (V == 3 && (fq3_off || log_n > 20 || (log_n < 18 && !fq3_all))) return false;
    return (V == 1 || V == 3) && log_n >= 17 && log_n <= MS_LDE2_MAX_LOG_N && log_b >= 1 && log_b <= 6 && !fwd->small && fwd->d_wr4[0] != nullptr;
}
static int lde2_tables(ms_ntt_plan* fwd, unsigned log_n, unsigned log_b, ms_ntt_plan::Lde2** out) {
    for (auto& l : fwd->lde2) if (l.log_b == log_b) { *out = &l; return MS_OK; }
    const size_t n = (size_t)1 << log_n, L = n >> 8, T = L >> 8, beta = (size_t)1 << log_b;
    const uint64_t wN = gl::root_of_unity(log_n + log_b), wL = gl::root_of_unity(log_n - 8), h = fwd->offset_canon;
    const bool uni = T >= 4;                                      // lde2_kernels.h: uniform split of pass A's factor
    const size_t nt = L >> 6, n_tin = uni ? nt * 256 * 4 : 0, n_tout = uni ? beta * nt * 16 * 4 : 0;
    const size_t T1 = T > 16 ? T / 16 : 0, n_c3 = T1 * 16 * 4;      // pass B, rows of 8192 / 16384 words: w_T^(t1 s0)
    std::vector<uint64_t> host(beta * 256 * 4 + beta * L + 256 * T * 4 + n_tin + n_tout + n_c3);
    uint64_t* gpl = host.data();
    uint64_t* aux = gpl + beta * 256 * 4;
    uint64_t* t2 = aux + beta * L;
    uint64_t* tin4 = t2 + 256 * T * 4;
    uint64_t* tout4 = tin4 + n_tin;
    uint64_t* c3 = tout4 + n_tout;
    const uint64_t sh[4] = {1, (uint64_t)1 << 24, (uint64_t)1 << 48, gl::pow(2, 72)};
    if (uni) {
        const uint64_t wn = gl::root_of_unity(log_n);
        for (size_t i0h = 0; i0h < nt; i0h++) {
            const uint64_t E = 64 * i0h;
            for (size_t b = 0; b < 16; b++) {
                const uint64_t wm = gl::pow(wn, (b * L + E) & (n - 1));          // w_256^b w_n^E
                uint64_t x = 1;
                for (size_t a = 0; a < 16; a++, x = gl::mul(x, wm))
                    for (int c = 0; c < 4; c++) tin4[((i0h * 16 + b) * 16 + a) * 4 + c] = gl::mul(x, sh[c]);
            }
        }
        uint64_t G = h;
        for (size_t j = 0; j < beta; j++, G = gl::mul(G, wN))
            for (size_t i0h = 0; i0h < nt; i0h++) {
                const uint64_t E = 64 * i0h, wm = gl::pow(wn, (16 * E) & (n - 1));
                uint64_t x = gl::pow(G, E);
                for (size_t bp = 0; bp < 16; bp++, x = gl::mul(x, wm))
                    for (int c = 0; c < 4; c++) tout4[((j * nt + i0h) * 16 + bp) * 4 + c] = gl::mul(x, sh[c]);
            }
    }
    uint64_t G = h;                                               // G_j = h w_N^j
    for (size_t j = 0; j < beta; j++, G = gl::mul(G, wN)) {
        const uint64_t GL = gl::pow(G, (uint64_t)L);
        uint64_t x = 1;
        for (size_t i = 0; i < 256; i++) { for (int c = 0; c < 4; c++) gpl[(j * 256 + i) * 4 + c] = gl::mul(x, sh[c]); x = gl::mul(x, GL); }   // plain, 4 copies
        x = 1;
        for (size_t i = 0; i < L; i++) { aux[j * L + i] = gl::to_mont(x); x = gl::mul(x, G); }
    }
    for (size_t k = 0; k < 256; k++) {
        const uint64_t wk = gl::pow(wL, (uint64_t)k);
        uint64_t x = 1;
        for (size_t t = 0; t < T; t++) { for (int c = 0; c < 4; c++) t2[(k * T + t) * 4 + c] = gl::mul(x, sh[c]); x = gl::mul(x, wk); }          // plain, 4 copies
    }
    if (T1) {
        unsigned log_t = 0;
        while (((size_t)1 << log_t) < T) log_t++;
        const uint64_t wT = gl::root_of_unity(log_t);
        for (size_t t1 = 0; t1 < T1; t1++)
            for (size_t s0 = 0; s0 < 16; s0++) {
                const uint64_t x = gl::pow(wT, (uint64_t)(t1 * s0));
                for (int c = 0; c < 4; c++) c3[(t1 * 16 + s0) * 4 + c] = gl::mul(x, sh[c]);
            }
    }
    ms_ntt_plan::Lde2 l;
    l.log_b = log_b;
    if (hipMalloc(&l.d, host.size() * 8) != hipSuccess) return fail(MS_ERR_NOMEM, "LDE tables (%zu bytes)", host.size() * 8);
    if (hipMemcpy(l.d, host.data(), host.size() * 8, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(l.d); return fail(MS_ERR_HIP, "LDE table upload"); }
    l.gpl = l.d; l.aux = l.d + beta * 256 * 4; l.t2 = l.aux + beta * L;
    if (uni) { l.tin4 = l.t2 + 256 * T * 4; l.tout4 = l.tin4 + n_tin; }
    if (T1) l.c3 = l.t2 + 256 * T * 4 + n_tin + n_tout;
    fwd->lde2.push_back(l);
    *out = &fwd->lde2.back();
    return MS_OK;
}
// coefficients (2^log_n words per column, src) -> bit-reversed evaluations on the coset of N points (dst, N words per column)
// natural (log_b = 0, 2^17 / 2^18 points): the one coset's transform in natural order -- the forward NTT of the column in two passes
// V = 3: Fq3 columns -- interleaved coefficients in, PLANAR scratch between the passes, interleaved evaluations out (lde2_kernels.h)
static int lde2_run(ms_ntt_plan* fwd, unsigned log_n, unsigned log_b, const void* const* src, void* const* dst, unsigned ncols, bool natural, unsigned V, ms_ntt_plan* inv) {
    ms_ctx* ctx = fwd->ctx;
    hipStream_t st = ctx->stream;
    HIPCHK(hipSetDevice(ctx->device));
    ms_ntt_plan::Lde2* tb = nullptr;
    MSCHK(lde2_tables(fwd, log_n, log_b, &tb));
    const size_t n = (size_t)1 << log_n, N = n << log_b, col_bytes = N * 8 * V;
    if (V == 3 && natural) return fail(MS_ERR_INVALID, "internal: natural-order two-pass transform is an Fp plan");
    const unsigned T = (unsigned)(n >> 16);
    unsigned group = (unsigned)std::max<size_t>(1, std::min<size_t>(MAXC, ctx->group_bytes / col_bytes));
    group = std::min(group, ncols);
    void* scratch = nullptr;
    MSCHK(ctx_scratch(ctx, (size_t)group * col_bytes, &scratch));
    for (unsigned c0 = 0; c0 < ncols; c0 += group) {
        const unsigned nc = std::min(group, ncols - c0);
        mslde2::Params P;
        memset(&P, 0, sizeof P);
        P.wr4 = fwd->d_wr4[0]; P.gpl = tb->gpl; P.aux = tb->aux; P.t2 = tb->t2; P.tin4 = tb->tin4; P.tout4 = tb->tout4; P.c3 = tb->c3;
        if (inv) {                                              // an inverse transform: the column backwards in, n^-1 h^-k on the way out
            P.rev = 1;
            if (inv->d_oscale) { P.oscale = inv->d_oscale; P.oscale_mode = 2; }
            else { P.oscale_c = inv->scale_const; P.oscale_mode = 1; }
        }
        const bool uni = tb->tin4 != nullptr;
        const bool stream_hint = 2 * (size_t)nc * col_bytes > ((size_t)256 << 20);       // as for the transforms above
        P.tw_lo = fwd->d_tw_lo; P.tw_hi = fwd->d_tw_hi; P.lo_bits = fwd->lo_bits; P.log_n = log_n; P.log_b = log_b;
        for (unsigned c = 0; c < nc; c++) { P.src[c] = (const uint64_t*)src[c0 + c]; P.dst[c] = (uint64_t*)((char*)scratch + (size_t)c * col_bytes); }
        {
            ProfScope ps(ctx, "lde2_pass_a", (double)(n * 8 * V + col_bytes) * nc);
            const dim3 ga((unsigned)(n >> 14), V << log_b, nc);
            if (V == 3) {
                if (!uni) hipLaunchKernelGGL((mslde2::lde2_strided_pass<true, false, 3>), ga, dim3(msntt2::NT), 0, st, P);       // T = 2
                else if (stream_hint) hipLaunchKernelGGL((mslde2::lde2_strided_pass<true, true, 3>), ga, dim3(msntt2::NT), 0, st, P);
                else hipLaunchKernelGGL((mslde2::lde2_strided_pass<false, true, 3>), ga, dim3(msntt2::NT), 0, st, P);
            } else if (inv) {                                   // (2^18 points: the uniform split)
                if (!uni) return fail(MS_ERR_INVALID, "internal: the backwards pass A is the 2^18-point plan's");
                if (stream_hint) hipLaunchKernelGGL((mslde2::lde2_strided_pass<true, true, 1, true>), ga, dim3(msntt2::NT), 0, st, P);
                else hipLaunchKernelGGL((mslde2::lde2_strided_pass<false, true, 1, true>), ga, dim3(msntt2::NT), 0, st, P);
            } else
            if (stream_hint) { if (uni) hipLaunchKernelGGL((mslde2::lde2_strided_pass<true, true>), ga, dim3(msntt2::NT), 0, st, P);
                               else hipLaunchKernelGGL((mslde2::lde2_strided_pass<true, false>), ga, dim3(msntt2::NT), 0, st, P); }
            else { if (uni) hipLaunchKernelGGL((mslde2::lde2_strided_pass<false, true>), ga, dim3(msntt2::NT), 0, st, P);
                   else hipLaunchKernelGGL((mslde2::lde2_strided_pass<false, false>), ga, dim3(msntt2::NT), 0, st, P); }
        }
        for (unsigned c = 0; c < nc; c++) { P.src[c] = (const uint64_t*)((char*)scratch + (size_t)c * col_bytes); P.dst[c] = (uint64_t*)dst[c0 + c]; }
        {
            ProfScope ps(ctx, "lde2_pass_b", 2.0 * col_bytes * nc);
            const dim3 g(V == 3 ? 16 * T : 4 * T, nc, 1u << log_b), b(msntt2::NT);      // 64 / T rows per workgroup (Fq3: 16 / T rows x 3 planes)
#define MS_RB(T_, UNI_) do { if (stream_hint) hipLaunchKernelGGL((mslde2::lde2_rows_pass<true, T_, UNI_>), g, b, 0, st, P); \
                             else hipLaunchKernelGGL((mslde2::lde2_rows_pass<false, T_, UNI_>), g, b, 0, st, P); } while (0)
#define MS_RB3(T_, UNI_) do { if (stream_hint) hipLaunchKernelGGL((mslde2::lde2_rows_pass<true, T_, UNI_, false, 3>), g, b, 0, st, P); \
                              else hipLaunchKernelGGL((mslde2::lde2_rows_pass<false, T_, UNI_, false, 3>), g, b, 0, st, P); } while (0)
            if (V == 3) {
                switch (T) {
                case 16: MS_RB3(16, true); break;
                case 8: MS_RB3(8, true); break;
                case 4: MS_RB3(4, true); break;
                default: MS_RB3(2, false); break;
                }
            } else
            if (natural) {
                if (T != 4) return fail(MS_ERR_INVALID, "internal: natural-order two-pass transform is the 2^18-point plan");
                if (inv && P.oscale_mode == 2) {
                    if (stream_hint) hipLaunchKernelGGL((mslde2::lde2_rows_pass<true, 4, true, true, 1, 2>), g, b, 0, st, P);
                    else hipLaunchKernelGGL((mslde2::lde2_rows_pass<false, 4, true, true, 1, 2>), g, b, 0, st, P);
                } else if (inv) {
                    if (stream_hint) hipLaunchKernelGGL((mslde2::lde2_rows_pass<true, 4, true, true, 1, 1>), g, b, 0, st, P);
                    else hipLaunchKernelGGL((mslde2::lde2_rows_pass<false, 4, true, true, 1, 1>), g, b, 0, st, P);
                } else
                if (stream_hint) hipLaunchKernelGGL((mslde2::lde2_rows_pass<true, 4, true, true>), g, b, 0, st, P);
                else hipLaunchKernelGGL((mslde2::lde2_rows_pass<false, 4, true, true>), g, b, 0, st, P);
            } else
            switch (T) {
            case 64: MS_RB(64, true); break;
            case 32: MS_RB(32, true); break;
            case 16: if (uni) MS_RB(16, true); else MS_RB(16, false); break;
            case 8: if (uni) MS_RB(8, true); else MS_RB(8, false); break;
            case 4: if (uni) MS_RB(4, true); else MS_RB(4, false); break;
            default: MS_RB(2, false); break;
            }
#undef MS_RB
#undef MS_RB3
        }
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}

extern "C" int ms_ntt_encode(ms_ntt_plan* plan, void* d_column) {
    if (!plan || !d_column) return fail(MS_ERR_INVALID, "ms_ntt_encode: null argument");
    if (!user_plan_alive(plan)) return fail(MS_ERR_INVALID, "ms_ntt_encode: the plan's context has been destroyed");
    std::lock_guard<std::mutex> lk(plan->ctx->mu);
    plan->queue.push_back(d_column);
    return MS_OK;
}
extern "C" int ms_ntt_enqueue(ms_ntt_plan* plan, void* const* d_columns, unsigned ncols) {
    if (!plan || (!d_columns && ncols)) return fail(MS_ERR_INVALID, "ms_ntt_enqueue: null argument");
    if (!user_plan_alive(plan)) return fail(MS_ERR_INVALID, "ms_ntt_enqueue: the plan's context has been destroyed");
    if (ncols == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(plan->ctx->mu);
    return plan_run(plan, (const void* const*)d_columns, d_columns, ncols, 256);
}
extern "C" int ms_ntt_enqueue_to(ms_ntt_plan* plan, const void* const* d_src, void* const* d_dst, unsigned ncols) {
    if (!plan || ((!d_src || !d_dst) && ncols)) return fail(MS_ERR_INVALID, "ms_ntt_enqueue_to: null argument");
    if (!user_plan_alive(plan)) return fail(MS_ERR_INVALID, "ms_ntt_enqueue_to: the plan's context has been destroyed");
    for (unsigned c = 0; c < ncols; c++) if (!d_src[c] || !d_dst[c]) return fail(MS_ERR_INVALID, "ms_ntt_enqueue_to: null column %u", c);
    if (ncols == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(plan->ctx->mu);
    return plan_run(plan, d_src, d_dst, ncols, 256);
}
extern "C" int ms_ntt_execute(ms_ntt_plan* plan) {
    if (!plan) return fail(MS_ERR_INVALID, "ms_ntt_execute: null plan");
    if (!user_plan_alive(plan)) return fail(MS_ERR_INVALID, "ms_ntt_execute: the plan's context has been destroyed");
    std::vector<void*> q;
    { std::lock_guard<std::mutex> lk(plan->ctx->mu); q.swap(plan->queue); }
    if (!q.empty()) MSCHK(ms_ntt_enqueue(plan, q.data(), (unsigned)q.size()));
    HIPCHK(hipStreamSynchronize(plan->ctx->stream));
    return MS_OK;
}

// ---------------------------------------------------------------------------------------
// bit reversal
// ---------------------------------------------------------------------------------------
int bit_reverse_run(ms_ctx* ctx, unsigned V, unsigned log_n, const void* const* src, void* const* dst, unsigned ncols) {
    hipStream_t st = ctx->stream;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << log_n;
    if (log_n >= 10 && V != 4) {
        for (unsigned c0 = 0; c0 < ncols; c0 += MAXC) {
            unsigned nc = std::min<unsigned>(MAXC, ncols - c0);
            msntt::BitrevParams B;
            memset(&B, 0, sizeof B);
            for (unsigned c = 0; c < nc; c++) { B.src[c] = (const uint64_t*)src[c0 + c]; B.dst[c] = (uint64_t*)dst[c0 + c]; }
            B.log_n = log_n;
            dim3 grid((unsigned)(n >> 10), nc);
            ProfScope ps(ctx, "bit_reverse", 2.0 * n * V * 8 * nc);
            if (V == 1) hipLaunchKernelGGL(msntt::bit_reverse_tiled<1>, grid, dim3(msntt::NT), 0, st, B);
            else hipLaunchKernelGGL(msntt::bit_reverse_tiled<3>, grid, dim3(msntt::NT), 0, st, B);
        }
    } else {
        // tiny: out of place through scratch when aliased
        const size_t col_bytes = n * V * 8;
        void* scratch = nullptr;
        const unsigned grp = (unsigned)std::max<size_t>(1, std::min<size_t>(MAXC, ((size_t)256 << 20) / col_bytes));
        MSCHK(ctx_scratch(ctx, (size_t)grp * col_bytes, &scratch));
        for (unsigned c0 = 0; c0 < ncols; c0 += grp) {
            unsigned nc = std::min<unsigned>(grp, ncols - c0);
            msntt::BitrevParams B;
            memset(&B, 0, sizeof B);
            for (unsigned c = 0; c < nc; c++) { B.src[c] = (const uint64_t*)src[c0 + c]; B.dst[c] = (uint64_t*)((char*)scratch + c * col_bytes); }
            B.log_n = log_n;
            dim3 grid((unsigned)((n + msntt::NT - 1) / msntt::NT), nc);
            if (V == 1) hipLaunchKernelGGL(msntt::bit_reverse_simple<1>, grid, dim3(msntt::NT), 0, st, B);
            else if (V == 3) hipLaunchKernelGGL(msntt::bit_reverse_simple<3>, grid, dim3(msntt::NT), 0, st, B);
            else hipLaunchKernelGGL(msntt::bit_reverse_simple<4>, grid, dim3(msntt::NT), 0, st, B);
            for (unsigned c = 0; c < nc; c++)
                HIPCHK(hipMemcpyAsync(dst[c0 + c], (char*)scratch + c * col_bytes, col_bytes, hipMemcpyDeviceToDevice, st));
        }
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}
extern "C" int ms_bit_reverse(ms_ctx* ctx, int field, unsigned log_n, void* const* d_columns, unsigned ncols) {
    if (!ctx || (!d_columns && ncols)) return fail(MS_ERR_INVALID, "ms_bit_reverse: null argument");
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    if (log_n > 40) return fail(MS_ERR_INVALID, "log_n too large");
    std::lock_guard<std::mutex> lk(ctx->mu);
    return bit_reverse_run(ctx, V, log_n, (const void* const*)d_columns, d_columns, ncols);
}

// ---------------------------------------------------------------------------------------
// fused LDE
// ---------------------------------------------------------------------------------------
extern "C" int ms_lde(ms_ctx* ctx, int field, unsigned log_n, unsigned log_blowup, const void* h_offset,
                      const void* const* d_in, void* const* d_out, unsigned ncols, int bit_reversed) {
    if (!ctx || !d_in || !d_out) return fail(MS_ERR_INVALID, "ms_lde: null argument");
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    const unsigned log_N = log_n + log_blowup;
    if (V == 4) {
        // compute-bound field: iNTT into the head of the output column, explicit zero padding, coset NTT,
        // bit reversal -- the plain sequence (the fused / pruned passes are Goldilocks kernels)
        if (log_N > 40) return fail(MS_ERR_INVALID, "LDE domain 2^%u too large", log_N);
        f252::E h252 = f252::one();
        if (h_offset) memcpy(h252.l, h_offset, 32);
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (f252::is_zero(h252) || f252::geq_p(h252)) return fail(MS_ERR_INVALID, "coset offset must be a non-zero canonical element");
        ms_ntt_plan *inv = nullptr, *fwd = nullptr;
        MSCHK(plan252_cached(ctx, log_n, true, f252::one(), &inv));
        MSCHK(plan252_cached(ctx, log_N, false, h252, &fwd));
        MSCHK(plan_run252(inv, d_in, d_out, ncols));
        // tiled passes: the coefficients are read straight from the head of the output column (zeros implicit), the
        // bit reversal is part of the last pass
        if (fwd->np252 && log_blowup <= fwd->lr252[0])
            return plan_run252_tiled(fwd, (const void* const*)d_out, d_out, ncols, log_blowup, bit_reversed != 0);
        const size_t n = (size_t)1 << log_n, N = (size_t)1 << log_N;
        if (N > n)
            for (unsigned c = 0; c < ncols; c++) HIPCHK(hipMemsetAsync((char*)d_out[c] + n * 32, 0, (N - n) * 32, ctx->stream));
        MSCHK(plan_run252(fwd, (const void* const*)d_out, d_out, ncols));
        if (bit_reversed) MSCHK(bit_reverse_run(ctx, 4, log_N, (const void* const*)d_out, d_out, ncols));
        return MS_OK;
    }
    if (log_N > 32) return fail(MS_ERR_INVALID, "LDE domain 2^%u exceeds the two-adicity", log_N);
    uint64_t h = 1;
    if (h_offset) { uint64_t h_m; memcpy(&h_m, h_offset, 8); h = gl::from_mont(h_m); }
    if (h == 0) return fail(MS_ERR_INVALID, "coset offset must be non-zero");
    ms_ntt_plan *inv = nullptr, *fwd = nullptr;
    int rc;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        MSCHK(ctx_plan(ctx, V, log_n, true, 1, &inv));
        MSCHK(ctx_plan(ctx, V, log_N, false, h, &fwd));
        // coefficients land in the first 2^log_n elements of the output column
        rc = plan_run(inv, d_in, d_out, ncols, 256);
        const size_t n = (size_t)1 << log_n, N = (size_t)1 << log_N;
        if (rc == MS_OK && lde2_applicable(fwd, V, log_n, log_blowup)) {
            // beta coset transforms of size n in two passes each, blocks land in the bit-reversed order
            rc = lde2_run(fwd, log_n, log_blowup, (const void* const*)d_out, d_out, ncols, false, V);
            if (rc == MS_OK && !bit_reversed) rc = bit_reverse_run(ctx, V, log_N, (const void* const*)d_out, d_out, ncols);
            return rc;
        }
        if (rc == MS_OK) {
            if (!fwd->small && log_blowup <= 4) {
                // zero padding is implicit in pass 1, the bit reversal is fused into the last pass
                rc = plan_run(fwd, (const void* const*)d_out, d_out, ncols, 256u >> log_blowup, bit_reversed != 0);
                bit_reversed = 0;
            } else {
                for (unsigned c = 0; c < ncols && rc == MS_OK; c++)
                    if (hipMemsetAsync((char*)d_out[c] + n * V * 8, 0, (N - n) * V * 8, ctx->stream) != hipSuccess)
                        rc = fail(MS_ERR_HIP, "hipMemsetAsync failed");
                if (rc == MS_OK) rc = plan_run(fwd, (const void* const*)d_out, d_out, ncols, 256);
            }
        }
        if (rc == MS_OK && bit_reversed) rc = bit_reverse_run(ctx, V, log_N, (const void* const*)d_out, d_out, ncols);
    }
    return rc;
}

// Matrix::into_evaluations / bit_reversed_evaluate on columns shorter than the domain (src/matrix.rs:193-251:
// "resize the column to the domain size", i.e. zero-extend the coefficient vector): the second half of ms_lde.
extern "C" int ms_evaluate(ms_ctx* ctx, int field, unsigned log_n, unsigned log_domain, const void* h_offset,
                           const void* const* d_in, void* const* d_out, unsigned ncols, int bit_reversed) {
    if (!ctx || !d_in || !d_out) return fail(MS_ERR_INVALID, "ms_evaluate: null argument");
    unsigned V = 0;
    MSCHK(field_words(field, &V));
    if (log_n > log_domain) return fail(MS_ERR_INVALID, "more coefficients (2^%u) than domain points (2^%u)", log_n, log_domain);
    const unsigned log_blowup = log_domain - log_n;
    const size_t n = (size_t)1 << log_n, N = (size_t)1 << log_domain;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    ms_ntt_plan* fwd = nullptr;
    if (V == 4) {
        if (log_domain > 40) return fail(MS_ERR_INVALID, "domain 2^%u too large", log_domain);
        f252::E h252 = f252::one();
        if (h_offset) memcpy(h252.l, h_offset, 32);
        if (f252::is_zero(h252) || f252::geq_p(h252)) return fail(MS_ERR_INVALID, "coset offset must be a non-zero canonical element");
        MSCHK(plan252_cached(ctx, log_domain, false, h252, &fwd));
    } else {
        if (log_domain > 32) return fail(MS_ERR_INVALID, "domain 2^%u exceeds the two-adicity", log_domain);
        uint64_t h = 1;
        if (h_offset) { uint64_t h_m; memcpy(&h_m, h_offset, 8); h = gl::from_mont(h_m); }
        if (h == 0) return fail(MS_ERR_INVALID, "coset offset must be non-zero");
        MSCHK(ctx_plan(ctx, V, log_domain, false, h, &fwd));
    }
    if (V != 4 && lde2_applicable(fwd, V, log_n, log_blowup)) {
        MSCHK(lde2_run(fwd, log_n, log_blowup, d_in, d_out, ncols, false, V));
        if (!bit_reversed) MSCHK(bit_reverse_run(ctx, V, log_domain, (const void* const*)d_out, d_out, ncols));
        return MS_OK;
    }
    if (V == 4 && fwd->np252 && log_blowup <= fwd->lr252[0])
        return plan_run252_tiled(fwd, d_in, d_out, ncols, log_blowup, bit_reversed != 0);
    if (V != 4 && !fwd->small && log_blowup >= 2 && log_blowup <= 4) {
        // pass 1 reads only the rows that hold coefficients (straight from d_in), zero padding is implicit,
        // the bit reversal is fused into the last pass
        return plan_run(fwd, d_in, d_out, ncols, 256u >> log_blowup, bit_reversed != 0);
    }
    for (unsigned c = 0; c < ncols; c++) {
        if (d_in[c] != d_out[c]) HIPCHK(hipMemcpyAsync(d_out[c], d_in[c], n * V * 8, hipMemcpyDeviceToDevice, ctx->stream));
        if (N > n) HIPCHK(hipMemsetAsync((char*)d_out[c] + n * V * 8, 0, (N - n) * V * 8, ctx->stream));
    }
    MSCHK(plan_run(fwd, (const void* const*)d_out, d_out, ncols, 256));
    if (bit_reversed) MSCHK(bit_reverse_run(ctx, V, log_domain, (const void* const*)d_out, d_out, ncols));
    return MS_OK;
}

// composition_poly.chunks(k) -> k columns (src/prover.rs:113-121): out[c][j] = in[j*k + c]
extern "C" int ms_deinterleave(ms_ctx* ctx, int field, size_t n_out, unsigned k, const void* d_in, void* const* d_out) {
    if (!ctx || !d_in || !d_out) return fail(MS_ERR_INVALID, "ms_deinterleave: null argument");
    const size_t fb = ms_field_bytes(field);
    if (!fb) return fail(MS_ERR_UNSUPPORTED, "unknown field %d", field);
    if (k == 0 || k > (unsigned)msstage::MAXCOLS) return fail(MS_ERR_UNSUPPORTED, "1..%d columns", msstage::MAXCOLS);
    if (n_out == 0) return MS_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    msscan::DeinterleaveParams P;
    memset(&P, 0, sizeof P);
    for (unsigned c = 0; c < k; c++) { if (!d_out[c]) return fail(MS_ERR_INVALID, "null column %u", c); P.out[c] = (uint64_t*)d_out[c]; }
    P.in = (const uint64_t*)d_in; P.n_out = n_out; P.k = k; P.V = (unsigned)(fb / 8);
    const size_t total = n_out * k * P.V;
    ProfScope ps(ctx, "deinterleave", 16.0 * total);
    hipLaunchKernelGGL(msscan::deinterleave, dim3(stream_grid(total)), dim3(msscan::NT), 0, ctx->stream, P);
    HIPCHK(hipGetLastError());
    return MS_OK;
}
