// DEEP composition on the device: out-of-domain evaluations and the composition polynomial (src/composer.rs:43-188).
#include "ms_internal.h"
#include "stage_kernels.h"
#include "fp252_kernels.h"
#include "deep_kernels.h"

// ---------------------------------------------------------------------------------------
// DEEP composition
// ---------------------------------------------------------------------------------------
static int point_words(int point_field, unsigned* PW) {
    if (point_field == MS_GOLDILOCKS_FP) { *PW = 1; return MS_OK; }
    if (point_field == MS_GOLDILOCKS_FQ3) { *PW = 3; return MS_OK; }
    return fail(MS_ERR_UNSUPPORTED, "point field must be Goldilocks Fp or Fq3");
}
static gl::Fq3 q3_load(const uint64_t* p, unsigned PW) { return PW == 3 ? gl::Fq3{p[0], p[1], p[2]} : gl::Fq3{p[0], 0, 0}; }

// ---- the 252-bit instantiations (Fq = Fp = Fp252)
static int horner_eval252(ms_ctx* ctx, size_t n, const void* const* d_cols, unsigned ncols, const unsigned* h_qcol, const uint64_t* h_qpoints,
                          unsigned nq, uint64_t* h_out) {
    if (ncols > (unsigned)msdeep::MAXCOLS) return fail(MS_ERR_UNSUPPORTED, "at most %d columns per call", msdeep::MAXCOLS);
    if (nq == 0) return MS_OK;
    for (unsigned q = 0; q < nq; q++) if (h_qcol[q] >= ncols) return fail(MS_ERR_INVALID, "query %u names column %u of %u", q, h_qcol[q], ncols);
    const unsigned nblocks = (unsigned)std::max<size_t>(1, (n + 4095) / 4096);
    void *d_qcol = nullptr, *d_pts = nullptr, *d_part = nullptr;
    PoolGuard pooled(ctx);                                 // temporaries go back to the pool on every exit path
    MSCHK(pooled.alloc((size_t)nq * 4, &d_qcol));
    MSCHK(pooled.alloc((size_t)nq * 32, &d_pts));
    MSCHK(pooled.alloc((size_t)nq * nblocks * 32, &d_part));
    std::vector<uint64_t> part((size_t)nq * nblocks * 4);
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        HIPCHK(hipSetDevice(ctx->device));
        HIPCHK(hipMemcpyAsync(d_qcol, h_qcol, (size_t)nq * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(d_pts, h_qpoints, (size_t)nq * 32, hipMemcpyHostToDevice, ctx->stream));
        msdeep252::HornerParams H;
        memset(&H, 0, sizeof H);
        for (unsigned c = 0; c < ncols; c++) H.cols[c] = (const uint64_t*)d_cols[c];
        H.qcol = (const uint32_t*)d_qcol; H.qpoint = (const uint64_t*)d_pts; H.partial = (uint64_t*)d_part; H.n = n; H.nblocks = nblocks;
        {
            ProfScope ps(ctx, "horner_blocks252", 32.0 * n * nq);
            hipLaunchKernelGGL(msdeep252::horner_blocks, dim3(nblocks, nq), dim3(msdeep252::NT), 0, ctx->stream, H);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(part.data(), d_part, part.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    for (unsigned q = 0; q < nq; q++) {                       // sum_b E_b * (x^4096)^b
        f252::E xb;
        memcpy(xb.l, h_qpoints + 4 * (size_t)q, 32);
        for (int sq = 0; sq < 12; sq++) xb = f252::mul(xb, xb);
        f252::E acc = f252::zero();
        for (unsigned b = nblocks; b-- > 0;) {
            f252::E e;
            memcpy(e.l, &part[((size_t)q * nblocks + b) * 4], 32);
            acc = f252::add(f252::mul(acc, xb), e);
        }
        memcpy(h_out + 4 * (size_t)q, acc.l, 32);
    }
    return MS_OK;
}
static int deep_compose252(ms_ctx* ctx, unsigned log_n, const void* h_offset, const void* const* d_polys, unsigned ncols,
                           const uint64_t* h_points, unsigned npoints, const unsigned* h_term_col, const unsigned* h_term_point,
                           const uint64_t* h_term_alpha, const uint64_t* h_term_ood, unsigned nterms,
                           const uint64_t* h_degree_alpha, const uint64_t* h_degree_beta, void* d_out) {
    if (ncols > (unsigned)msdeep::MAXCOLS) return fail(MS_ERR_UNSUPPORTED, "at most %d columns", msdeep::MAXCOLS);
    if (npoints == 0 || npoints > (unsigned)msdeep::MAXPOINTS) return fail(MS_ERR_UNSUPPORTED, "1..%d distinct out-of-domain points", msdeep::MAXPOINTS);
    if (log_n > 40) return fail(MS_ERR_INVALID, "log_n too large");
    for (unsigned t = 0; t < nterms; t++)
        if (h_term_col[t] >= ncols || h_term_point[t] >= npoints) return fail(MS_ERR_INVALID, "term %u out of range", t);
    f252::E h = f252::to_mont(f252::E{{3, 0, 0, 0}});          // the field's generator (gpu/src/fields.rs:241)
    if (h_offset) memcpy(h.l, h_offset, 32);
    if (f252::is_zero(h) || f252::geq_p(h)) return fail(MS_ERR_INVALID, "coset offset must be a non-zero canonical element");
    const size_t n = (size_t)1 << log_n;
    std::vector<void*> ev(ncols, nullptr);
    void *d_terms = nullptr, *d_q = nullptr;
    PoolGuard pooled(ctx);                                 // temporaries go back to the pool on every exit path
    for (unsigned c = 0; c < ncols; c++) MSCHK(pooled.alloc(n * 32, &ev[c]));
    MSCHK(pooled.alloc(std::max<size_t>(1, nterms) * sizeof(msdeep252::Term), &d_terms));
    MSCHK(pooled.alloc(n * 32, &d_q));
    std::vector<msdeep252::Term> terms(nterms);
    for (unsigned t = 0; t < nterms; t++) {
        terms[t].col = h_term_col[t]; terms[t].point = h_term_point[t];
        memcpy(terms[t].alpha, h_term_alpha + 4 * (size_t)t, 32);
        memcpy(terms[t].ood, h_term_ood + 4 * (size_t)t, 32);
    }
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        HIPCHK(hipSetDevice(ctx->device));
        if (nterms) HIPCHK(hipMemcpyAsync(d_terms, terms.data(), nterms * sizeof(msdeep252::Term), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        ms_ntt_plan *fwd = nullptr, *inv = nullptr, *sub = nullptr;
        MSCHK(plan252_cached(ctx, log_n, false, h, &fwd));
        MSCHK(plan252_cached(ctx, log_n, true, h, &inv));
        MSCHK(plan252_cached(ctx, log_n, false, f252::one(), &sub));           // its tables hold w_n^i
        if (ncols) MSCHK(plan_run(fwd, d_polys, ev.data(), ncols, 256));
        msdeep252::DeepParams D;
        memset(&D, 0, sizeof D);
        for (unsigned c = 0; c < ncols; c++) D.cols[c] = (const uint64_t*)ev[c];
        D.terms = (const msdeep252::Term*)d_terms; D.tw_lo = sub->d252_tw_lo; D.tw_hi = sub->d252_tw_hi; D.lo_bits = sub->lo_bits;
        memcpy(D.points, h_points, (size_t)npoints * 32);
        memcpy(D.h, h.l, 32);
        D.out = (uint64_t*)d_q; D.n = n; D.nterms = nterms; D.npoints = npoints;
        const dim3 g((unsigned)((n + msdeep252::NT - 1) / msdeep252::NT));
        { ProfScope ps(ctx, "deep_points252", 32.0 * n * (ncols + 1));
          hipLaunchKernelGGL(msdeep252::deep_points, g, dim3(msdeep252::NT), 0, ctx->stream, D); }
        HIPCHK(hipGetLastError());
        const void* qsrc[1] = {d_q};
        void* qdst[1] = {d_q};
        MSCHK(plan_run(inv, qsrc, qdst, 1, 256));
        f252::E da, db;
        memcpy(da.l, h_degree_alpha, 32);
        memcpy(db.l, h_degree_beta, 32);
        { ProfScope ps(ctx, "deep_degree_adjust252", 64.0 * n);
          hipLaunchKernelGGL(msdeep252::deep_degree_adjust, g, dim3(msdeep252::NT), 0, ctx->stream, (uint64_t*)d_out, (const uint64_t*)d_q, n, da, db); }
        HIPCHK(hipGetLastError());
    }
    return MS_OK;
}

extern "C" int ms_horner_eval(ms_ctx* ctx, int coeff_field, int point_field, size_t n, const void* const* d_cols, unsigned ncols,
                              const unsigned* h_qcol, const void* h_qpoints, unsigned nq, void* h_out) {
    if (!ctx || !d_cols || !h_qcol || !h_qpoints || !h_out) return fail(MS_ERR_INVALID, "ms_horner_eval: null argument");
    if (coeff_field == MS_STARK252_FP || point_field == MS_STARK252_FP) {
        if (coeff_field != point_field) return fail(MS_ERR_UNSUPPORTED, "the 252-bit field has no extension: coefficients and points must both be Fp252");
        return horner_eval252(ctx, n, d_cols, ncols, h_qcol, (const uint64_t*)h_qpoints, nq, (uint64_t*)h_out);
    }
    unsigned CW = 0, PW = 0;
    MSCHK(point_words(coeff_field, &CW));
    MSCHK(point_words(point_field, &PW));
    if (CW > PW) return fail(MS_ERR_UNSUPPORTED, "coefficients must embed into the point field");
    if (ncols > (unsigned)msdeep::MAXCOLS) return fail(MS_ERR_UNSUPPORTED, "at most %d columns per call", msdeep::MAXCOLS);
    if (nq == 0) return MS_OK;
    for (unsigned q = 0; q < nq; q++) if (h_qcol[q] >= ncols) return fail(MS_ERR_INVALID, "query %u names column %u of %u", q, h_qcol[q], ncols);
    if (n == 0) { memset(h_out, 0, (size_t)nq * PW * 8); return MS_OK; }           // the zero polynomial
    // queries on the same column that follow one another (the callers list them per column) share one pass over the coefficients
    constexpr unsigned GQ = 2;
    std::vector<uint32_t> groups, singles;
    for (unsigned q = 0; q < nq; q++) {
        if (!groups.empty() && groups[groups.size() - 1] < GQ && h_qcol[q] == h_qcol[q - 1]) groups[groups.size() - 1]++;
        else { groups.push_back(q); groups.push_back(1); }
        singles.push_back(q); singles.push_back(1);
    }
    // Levels: blocks of 4096 coefficients -> one value each (deep_kernels.h); those values are the coefficients of a polynomial in
    // x^4096, evaluated the same way by the next level (one row of block values per query) until one value per query is left --
    // nothing is combined on the host (17 queries x 1024 blocks of dependent products there cost more than the kernel).
    std::vector<size_t> level_n;
    for (size_t m = n;; m = (m + 4095) >> 12) { level_n.push_back(m); if (m <= 4096) break; }
    const unsigned nlevels = (unsigned)level_n.size();
    // per level and query, for the level's point p = x^(4096^level): p^i and p^(16 i), i < 16 (a lane's weight p^t), and (p^256)^k,
    // k < 16 (the factor of a lane's k-th coefficient), the latter also as 22 / 22 / 20-bit limbs for Fp coefficient columns.
    // Fp points stay scalars here: these are dependent products on the host.
    auto mulq = [PW](const gl::Fq3& a, const gl::Fq3& b) { return PW == 1 ? gl::Fq3{gl::mont_mul(a.c0, b.c0), 0, 0} : gl::mont_mul(a, b); };
    const size_t per_level = (size_t)nq * 48;
    std::vector<uint64_t> xlo(per_level * nlevels), xhi(per_level * nlevels), ypow(per_level * nlevels);
    std::vector<uint32_t> ylimb((size_t)nq * 16 * 3 * 4, 0);                 // level 0 only
    for (unsigned q = 0; q < nq; q++) {
        gl::Fq3 pnt = q3_load((const uint64_t*)h_qpoints + (size_t)q * PW, PW);
        for (unsigned l = 0; l < nlevels; l++) {
            uint64_t* lo = &xlo[per_level * l + (size_t)q * 48];
            uint64_t* hi = &xhi[per_level * l + (size_t)q * 48];
            uint64_t* yp = &ypow[per_level * l + (size_t)q * 48];
            auto fill = [&](uint64_t* out, const gl::Fq3& base) {           // out[i] = base^i, i < 16; returns base^16
                gl::Fq3 cur = {gl::ONE_MONT, 0, 0};
                for (unsigned i = 0; i < 16; i++) { out[3 * i] = cur.c0; out[3 * i + 1] = cur.c1; out[3 * i + 2] = cur.c2; cur = mulq(cur, base); }
                return cur;
            };
            const gl::Fq3 p16 = fill(lo, pnt), p256 = fill(hi, p16);
            pnt = fill(yp, p256);                                           // (p^256)^16 = p^4096: the next level's point
            if (l == 0 && CW == 1)
                for (unsigned i = 0; i < 48; i++) {
                    uint32_t* o = &ylimb[((size_t)q * 48 + i) * 4];
                    o[0] = (uint32_t)(yp[i] & 0x3FFFFF); o[1] = (uint32_t)((yp[i] >> 22) & 0x3FFFFF); o[2] = (uint32_t)(yp[i] >> 44);
                }
        }
    }
    std::vector<unsigned> level_blocks(nlevels);
    size_t part_words = 0;
    for (unsigned l = 0; l < nlevels; l++) { level_blocks[l] = (unsigned)std::max<size_t>(1, (level_n[l] + 4095) >> 12); part_words += (size_t)nq * level_blocks[l] * 3; }
    const unsigned ngroups = (unsigned)(groups.size() / 2);
    if ((uint64_t)level_blocks[0] * ngroups > 0x7FFFFFFFull) return fail(MS_ERR_UNSUPPORTED, "too many (block, query group) pairs for one launch");
    // the seven host-built tables travel as ONE block and one copy command (each copy is a blit launch of its own on the stream: seven of
    // them were 40 us in front of a 60 us kernel, twice per proof)
    std::vector<char> image;
    auto put = [&image](const void* src, size_t bytes) { const size_t at = (image.size() + 63) & ~(size_t)63; image.resize(at + bytes); if (bytes) memcpy(image.data() + at, src, bytes); return at; };
    const size_t o_qcol = put(h_qcol, (size_t)nq * 4), o_groups = put(groups.data(), groups.size() * 4), o_singles = put(singles.data(), singles.size() * 4),
                 o_xlo = put(xlo.data(), xlo.size() * 8), o_xhi = put(xhi.data(), xhi.size() * 8), o_ypow = put(ypow.data(), ypow.size() * 8),
                 o_ylimb = put(ylimb.data(), ylimb.size() * 4);
    void *d_tabs = nullptr, *d_part = nullptr;
    PoolGuard pooled(ctx);                                 // temporaries go back to the pool on every exit path
    MSCHK(pooled.alloc(std::max<size_t>(image.size(), 64), &d_tabs));
    MSCHK(pooled.alloc(part_words * 8, &d_part));
    const char* const tb = (const char*)d_tabs;
    const void *d_qcol = tb + o_qcol, *d_groups = tb + o_groups, *d_singles = tb + o_singles, *d_xlo = tb + o_xlo, *d_xhi = tb + o_xhi, *d_ypow = tb + o_ypow,
               *d_ylimb = tb + o_ylimb;
    std::vector<uint64_t> res((size_t)nq * 3);
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        HIPCHK(hipSetDevice(ctx->device));
        MSCHK(stage_upload(ctx, d_tabs, image.data(), image.size()));
        uint64_t* part = (uint64_t*)d_part;
        const uint64_t* below = nullptr;
        for (unsigned l = 0; l < nlevels; l++) {
            msdeep::HornerParams H;
            memset(&H, 0, sizeof H);
            if (l == 0) for (unsigned c = 0; c < ncols; c++) H.cols[c] = (const uint64_t*)d_cols[c];
            H.qcol = (const uint32_t*)d_qcol; H.partial = part; H.n = level_n[l]; H.nblocks = level_blocks[l];
            H.ngroups = l == 0 ? ngroups : nq; H.group = (const uint32_t*)(l == 0 ? d_groups : d_singles);
            H.xlo = (const uint64_t*)d_xlo + per_level * l; H.xhi = (const uint64_t*)d_xhi + per_level * l; H.ypow = (const uint64_t*)d_ypow + per_level * l;
            H.ylimb = (const uint32_t*)d_ylimb;
            H.self_src = below; H.self_nblocks = l ? level_blocks[l - 1] : 0;
            const dim3 g(H.nblocks * H.ngroups);
            {
                ProfScope ps(ctx, "horner_blocks", l == 0 ? 8.0 * CW * n * nq : 24.0 * level_n[l] * nq);
                if (l > 0) hipLaunchKernelGGL((msdeep::horner_blocks<3, 3, GQ>), g, dim3(msdeep::NT), 0, ctx->stream, H);       // block values: 3 words each
                else if (CW == 1 && PW == 1) hipLaunchKernelGGL((msdeep::horner_blocks<1, 1, GQ>), g, dim3(msdeep::NT), 0, ctx->stream, H);
                else if (CW == 1) hipLaunchKernelGGL((msdeep::horner_blocks<1, 3, GQ>), g, dim3(msdeep::NT), 0, ctx->stream, H);
                else hipLaunchKernelGGL((msdeep::horner_blocks<3, 3, GQ>), g, dim3(msdeep::NT), 0, ctx->stream, H);
            }
            HIPCHK(hipGetLastError());
            below = part;
            part += (size_t)nq * level_blocks[l] * 3;
        }
        HIPCHK(hipMemcpyAsync(res.data(), below, res.size() * 8, hipMemcpyDeviceToHost, ctx->stream));   // the last level: one block per query
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    for (unsigned q = 0; q < nq; q++) memcpy((uint64_t*)h_out + (size_t)q * PW, &res[3 * q], PW * 8);
    return MS_OK;
}

extern "C" int ms_deep_compose(ms_ctx* ctx, int point_field, unsigned log_n, const void* h_offset,
                               const void* const* d_base_polys, unsigned nbase, const void* const* d_ext_polys, unsigned next,
                               const void* h_points, unsigned npoints, const unsigned* h_term_col, const unsigned* h_term_point,
                               const void* h_term_alpha, const void* h_term_ood, unsigned nterms,
                               const void* h_degree_alpha, const void* h_degree_beta, void* d_out) {
    if (!ctx || !h_points || !h_term_col || !h_term_point || !h_term_alpha || !h_term_ood || !h_degree_alpha || !h_degree_beta || !d_out)
        return fail(MS_ERR_INVALID, "ms_deep_compose: null argument");
    if (point_field == MS_STARK252_FP) {
        if (next) return fail(MS_ERR_INVALID, "the 252-bit field has no extension columns: pass every polynomial as a base column");
        if (nbase && !d_base_polys) return fail(MS_ERR_INVALID, "ms_deep_compose: null column table");
        return deep_compose252(ctx, log_n, h_offset, d_base_polys, nbase, (const uint64_t*)h_points, npoints, h_term_col, h_term_point,
                               (const uint64_t*)h_term_alpha, (const uint64_t*)h_term_ood, nterms, (const uint64_t*)h_degree_alpha,
                               (const uint64_t*)h_degree_beta, d_out);
    }
    unsigned PW = 0;
    MSCHK(point_words(point_field, &PW));
    if (PW == 1 && next) return fail(MS_ERR_INVALID, "extension columns need point_field = Fq3");
    if (nbase > (unsigned)msdeep::MAXCOLS || next > (unsigned)msdeep::MAXCOLS) return fail(MS_ERR_UNSUPPORTED, "at most %d columns of each kind", msdeep::MAXCOLS);
    if (npoints == 0 || npoints > (unsigned)msdeep::MAXPOINTS) return fail(MS_ERR_UNSUPPORTED, "1..%d distinct out-of-domain points", msdeep::MAXPOINTS);
    if ((nbase && !d_base_polys) || (next && !d_ext_polys)) return fail(MS_ERR_INVALID, "ms_deep_compose: null column table");
    if (log_n > 32) return fail(MS_ERR_INVALID, "log_n too large");
    for (unsigned t = 0; t < nterms; t++)
        if (h_term_col[t] >= nbase + next || h_term_point[t] >= npoints) return fail(MS_ERR_INVALID, "term %u out of range", t);
    uint64_t h = gl::GENERATOR;
    if (h_offset) { uint64_t h_m; memcpy(&h_m, h_offset, 8); h = gl::from_mont(h_m); }
    if (h == 0) return fail(MS_ERR_INVALID, "coset offset must be non-zero");
    const size_t n = (size_t)1 << log_n;
    // Q is evaluated on the coset h<w_n> and a lane shares ONE inversion between the denominators x - z_k of its points: a point
    // z_k ON that coset (possible only when it lies in the base field: probability about n / p per drawn point) has no quotient
    // there and would zero its neighbours' factors as well.  The reference's synthetic division has no such restriction, so this
    // is reported instead of computed wrongly; the caller re-draws z, as it must when z falls into the trace domain.
    for (unsigned k = 0; k < npoints; k++) {
        const uint64_t* zk = (const uint64_t*)h_points + (size_t)k * PW;
        if (PW == 3 && (zk[1] != 0 || zk[2] != 0)) continue;
        const uint64_t z = gl::from_mont(zk[0]);
        if (z != 0 && gl::pow(gl::mul(z, gl::inv(h)), n) == 1)
            return fail(MS_ERR_INVALID, "ms_deep_compose: out-of-domain point %u lies on the evaluation coset h<w_n> (x - z vanishes there)", k);
    }
    // scratch: coset evaluations of every polynomial + the evaluation/coefficient column of Q
    std::vector<void*> ev(nbase + next, nullptr);
    void *d_terms = nullptr, *d_q = nullptr;
    PoolGuard pooled(ctx);                                 // temporaries go back to the pool on every exit path
    for (unsigned c = 0; c < nbase; c++) MSCHK(pooled.alloc(n * 8, &ev[c]));
    for (unsigned c = 0; c < next; c++) MSCHK(pooled.alloc(n * 24, &ev[nbase + c]));
    MSCHK(pooled.alloc(std::max<size_t>(1, nterms) * sizeof(msdeep::Term), &d_terms));
    MSCHK(pooled.alloc(n * PW * 8, &d_q));
    // terms sorted by point (the kernel multiplies a point's quotient factor into the SUM of its terms; the order of an exact sum is free)
    std::vector<msdeep::Term> terms;
    terms.reserve(nterms);
    gl::Fq3 csum[msdeep::MAXPOINTS];
    for (auto& c : csum) c = {0, 0, 0};
    unsigned term_start[msdeep::MAXPOINTS + 1];
    for (unsigned k = 0; k < npoints; k++) {
        term_start[k] = (unsigned)terms.size();
        for (unsigned t = 0; t < nterms; t++) {
            if (h_term_point[t] != k) continue;
            msdeep::Term T;
            memset(&T, 0, sizeof T);
            T.col = h_term_col[t]; T.point = k;
            memcpy(T.alpha, (const uint64_t*)h_term_alpha + (size_t)t * PW, PW * 8);
            memcpy(T.ood, (const uint64_t*)h_term_ood + (size_t)t * PW, PW * 8);
            for (unsigned w = 0; w < PW; w++) {              // the limbs limb_mac multiplies an Fp column's value with
                T.alimb[w][0] = (uint32_t)(T.alpha[w] & 0x3FFFFF); T.alimb[w][1] = (uint32_t)((T.alpha[w] >> 22) & 0x3FFFFF); T.alimb[w][2] = (uint32_t)(T.alpha[w] >> 44);
            }
            {                                                  // csum_k += alpha_t ood_t (Montgomery products, as the kernel's)
                const gl::Fq3 a = {T.alpha[0], T.alpha[1], T.alpha[2]}, o = {T.ood[0], T.ood[1], T.ood[2]};
                const gl::Fq3 pr = PW == 1 ? gl::Fq3{gl::mont_mul(a.c0, o.c0), 0, 0} : gl::mont_mul(a, o);
                csum[k] = gl::add(csum[k], pr);
            }
            terms.push_back(T);
        }
    }
    for (unsigned k = npoints; k <= (unsigned)msdeep::MAXPOINTS; k++) term_start[k] = (unsigned)terms.size();
    int rc = MS_OK;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        HIPCHK(hipSetDevice(ctx->device));
        if (nterms) MSCHK(stage_upload(ctx, d_terms, terms.data(), nterms * sizeof(msdeep::Term)));     // through the pinned ring: no stream drain
        ms_ntt_plan *f1 = nullptr, *f3 = nullptr, *inv = nullptr, *tw = nullptr;
        if (nbase) { MSCHK(ctx_plan(ctx, 1, log_n, false, h, &f1)); rc = plan_run(f1, d_base_polys, ev.data(), nbase, 256); }
        if (rc == MS_OK && next) { MSCHK(ctx_plan(ctx, 3, log_n, false, h, &f3)); rc = plan_run(f3, d_ext_polys, ev.data() + nbase, next, 256); }
        if (rc != MS_OK) return rc;
        const unsigned tl = std::max(log_n, 12u);
        MSCHK(ctx_plan(ctx, 1, tl, false, 1, &tw));
        msdeep::DeepParams D;
        memset(&D, 0, sizeof D);
        for (unsigned c = 0; c < nbase; c++) D.base[c] = (const uint64_t*)ev[c];
        for (unsigned c = 0; c < next; c++) D.ext[c] = (const uint64_t*)ev[nbase + c];
        D.terms = (const msdeep::Term*)d_terms; D.tw_lo = tw->d_tw_lo; D.tw_hi = tw->d_tw_hi; D.lo_bits = tw->lo_bits; D.xshift = tl - log_n;
        for (unsigned k = 0; k < npoints; k++) memcpy(D.points[k], (const uint64_t*)h_points + (size_t)k * PW, PW * 8);
        D.out = (uint64_t*)d_q; D.h_mont = gl::to_mont(h); D.n = n; D.nbase = nbase; D.nterms = nterms; D.npoints = npoints;
        memcpy(D.term_start, term_start, sizeof term_start);
        for (unsigned k = 0; k < npoints; k++) { D.csum[k][0] = csum[k].c0; D.csum[k][1] = csum[k].c1; D.csum[k][2] = csum[k].c2; }
        dim3 g((unsigned)((n + msdeep::NT - 1) / msdeep::NT));
        {
            // points per lane: as many as keep the shared inversion's operands in registers (4 x <= 3 points over Fp, 2 x <= 4 over Fq3)
            ProfScope ps(ctx, "deep_points", 8.0 * n * (nbase + 3.0 * next + PW));
            auto blocks = [&](unsigned pts) { return dim3((unsigned)((n + (size_t)msdeep::NT * pts - 1) / ((size_t)msdeep::NT * pts))); };
            if (PW == 1) {
                if (npoints <= 3 && n >= 4096) hipLaunchKernelGGL((msdeep::deep_points<1, 4, 3>), blocks(4), dim3(msdeep::NT), 0, ctx->stream, D);
                else hipLaunchKernelGGL((msdeep::deep_points<1, 1, msdeep::MAXPOINTS>), blocks(1), dim3(msdeep::NT), 0, ctx->stream, D);
            } else {
                if (npoints <= 4 && n >= 4096) hipLaunchKernelGGL((msdeep::deep_points<3, 2, 4>), blocks(2), dim3(msdeep::NT), 0, ctx->stream, D);
                else hipLaunchKernelGGL((msdeep::deep_points<3, 1, msdeep::MAXPOINTS>), blocks(1), dim3(msdeep::NT), 0, ctx->stream, D);
            }
        }
        HIPCHK(hipGetLastError());
        MSCHK(ctx_plan(ctx, PW, log_n, true, h, &inv));
        const void* qsrc[1] = {d_q};
        void* qdst[1] = {d_q};
        MSCHK(plan_run(inv, qsrc, qdst, 1, 256));
        msdeep::Q da = {{0, 0, 0}}, db = {{0, 0, 0}};
        memcpy(da.w, h_degree_alpha, PW * 8);
        memcpy(db.w, h_degree_beta, PW * 8);
        {
            ProfScope ps(ctx, "deep_degree_adjust", 16.0 * n * PW);
            if (PW == 1) hipLaunchKernelGGL((msdeep::deep_degree_adjust<1>), g, dim3(msdeep::NT), 0, ctx->stream, (uint64_t*)d_out, (const uint64_t*)d_q, n, da, db);
            else hipLaunchKernelGGL((msdeep::deep_degree_adjust<3>), g, dim3(msdeep::NT), 0, ctx->stream, (uint64_t*)d_out, (const uint64_t*)d_q, n, da, db);
        }
        HIPCHK(hipGetLastError());
    }
    return MS_OK;
}

// The DEEP composition polynomial evaluated where the committed LDEs already are: rows [first, first + count) of the bit-reversed LDE
// domain (2^log_domain points, offset h).  Same terms and points as ms_deep_compose; the columns are the LDE columns' rows (a row
// shard of a multi-GPU run, or the whole domain with first = 0).  The value at x is
//      (alpha + beta x) sum_k 1/(x - z_k) sum_{t: pt = k} alpha_t (P_ct(x) - ood_t),
// the polynomial ms_deep_compose returns in coefficient form, at that point: its LDE (src/prover.rs:149-152) without the transforms.
extern "C" int ms_deep_rows(ms_ctx* ctx, int point_field, unsigned log_domain, const void* h_offset, size_t first, size_t count,
                            const void* const* d_base_rows, unsigned nbase, const void* const* d_ext_rows, unsigned next,
                            const void* h_points, unsigned npoints, const unsigned* h_term_col, const unsigned* h_term_point,
                            const void* h_term_alpha, const void* h_term_ood, unsigned nterms,
                            const void* h_degree_alpha, const void* h_degree_beta, void* d_out) {
    if (!ctx || !h_points || !h_term_col || !h_term_point || !h_term_alpha || !h_term_ood || !h_degree_alpha || !h_degree_beta || !d_out)
        return fail(MS_ERR_INVALID, "ms_deep_rows: null argument");
    if (point_field == MS_STARK252_FP) return fail(MS_ERR_UNSUPPORTED, "ms_deep_rows: Goldilocks fields only (the 252-bit composer goes through ms_deep_compose)");
    unsigned PW = 0;
    MSCHK(point_words(point_field, &PW));
    if (PW == 1 && next) return fail(MS_ERR_INVALID, "extension columns need point_field = Fq3");
    if (nbase > (unsigned)msdeep::MAXCOLS || next > (unsigned)msdeep::MAXCOLS) return fail(MS_ERR_UNSUPPORTED, "at most %d columns of each kind", msdeep::MAXCOLS);
    if (npoints == 0 || npoints > (unsigned)msdeep::MAXPOINTS) return fail(MS_ERR_UNSUPPORTED, "1..%d distinct out-of-domain points", msdeep::MAXPOINTS);
    if ((nbase && !d_base_rows) || (next && !d_ext_rows)) return fail(MS_ERR_INVALID, "ms_deep_rows: null column table");
    if (log_domain == 0 || log_domain > 32) return fail(MS_ERR_INVALID, "ms_deep_rows: domain of 2^%u points", log_domain);
    const size_t N = (size_t)1 << log_domain;
    if (first > N || count > N - first) return fail(MS_ERR_INVALID, "ms_deep_rows: rows [%zu, %zu) outside the domain", first, first + count);
    for (unsigned t = 0; t < nterms; t++)
        if (h_term_col[t] >= nbase + next || h_term_point[t] >= npoints) return fail(MS_ERR_INVALID, "term %u out of range", t);
    uint64_t h = gl::GENERATOR;
    if (h_offset) { uint64_t h_m; memcpy(&h_m, h_offset, 8); h = gl::from_mont(h_m); }
    if (h == 0) return fail(MS_ERR_INVALID, "coset offset must be non-zero");
    for (unsigned k = 0; k < npoints; k++) {           // a point ON the LDE coset has no quotient there (cf. ms_deep_compose)
        const uint64_t* zk = (const uint64_t*)h_points + (size_t)k * PW;
        if (PW == 3 && (zk[1] != 0 || zk[2] != 0)) continue;
        const uint64_t z = gl::from_mont(zk[0]);
        if (z != 0 && gl::pow(gl::mul(z, gl::inv(h)), N) == 1)
            return fail(MS_ERR_INVALID, "ms_deep_rows: out-of-domain point %u lies on the LDE coset h<w_N> (x - z vanishes there)", k);
    }
    if (count == 0) return MS_OK;
    std::vector<msdeep::Term> terms;
    terms.reserve(nterms);
    gl::Fq3 csum[msdeep::MAXPOINTS];
    for (auto& c : csum) c = {0, 0, 0};
    unsigned term_start[msdeep::MAXPOINTS + 1];
    for (unsigned k = 0; k < npoints; k++) {
        term_start[k] = (unsigned)terms.size();
        for (unsigned t = 0; t < nterms; t++) {
            if (h_term_point[t] != k) continue;
            msdeep::Term T;
            memset(&T, 0, sizeof T);
            T.col = h_term_col[t]; T.point = k;
            memcpy(T.alpha, (const uint64_t*)h_term_alpha + (size_t)t * PW, PW * 8);
            memcpy(T.ood, (const uint64_t*)h_term_ood + (size_t)t * PW, PW * 8);
            for (unsigned w = 0; w < PW; w++) {              // the limbs limb_mac multiplies an Fp column's value with
                T.alimb[w][0] = (uint32_t)(T.alpha[w] & 0x3FFFFF); T.alimb[w][1] = (uint32_t)((T.alpha[w] >> 22) & 0x3FFFFF); T.alimb[w][2] = (uint32_t)(T.alpha[w] >> 44);
            }
            {                                                  // csum_k += alpha_t ood_t (Montgomery products, as the kernel's)
                const gl::Fq3 a = {T.alpha[0], T.alpha[1], T.alpha[2]}, o = {T.ood[0], T.ood[1], T.ood[2]};
                const gl::Fq3 pr = PW == 1 ? gl::Fq3{gl::mont_mul(a.c0, o.c0), 0, 0} : gl::mont_mul(a, o);
                csum[k] = gl::add(csum[k], pr);
            }
            terms.push_back(T);
        }
    }
    for (unsigned k = npoints; k <= (unsigned)msdeep::MAXPOINTS; k++) term_start[k] = (unsigned)terms.size();
    void* d_terms = nullptr;
    PoolGuard pooled(ctx);
    MSCHK(pooled.alloc(std::max<size_t>(1, nterms) * sizeof(msdeep::Term), &d_terms));
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    if (nterms) MSCHK(stage_upload(ctx, d_terms, terms.data(), nterms * sizeof(msdeep::Term)));
    ms_ntt_plan* tw = nullptr;
    const unsigned tl = std::max(log_domain, 12u);
    MSCHK(ctx_plan(ctx, 1, tl, false, 1, &tw));
    msdeep::DeepParams D;
    memset(&D, 0, sizeof D);
    for (unsigned c = 0; c < nbase; c++) D.base[c] = (const uint64_t*)d_base_rows[c];
    for (unsigned c = 0; c < next; c++) D.ext[c] = (const uint64_t*)d_ext_rows[c];
    D.terms = (const msdeep::Term*)d_terms; D.tw_lo = tw->d_tw_lo; D.tw_hi = tw->d_tw_hi; D.lo_bits = tw->lo_bits; D.xshift = tl - log_domain;
    for (unsigned k = 0; k < npoints; k++) memcpy(D.points[k], (const uint64_t*)h_points + (size_t)k * PW, PW * 8);
    D.out = (uint64_t*)d_out; D.h_mont = gl::to_mont(h); D.n = count; D.nbase = nbase; D.nterms = nterms; D.npoints = npoints;
    memcpy(D.term_start, term_start, sizeof term_start);
        for (unsigned k = 0; k < npoints; k++) { D.csum[k][0] = csum[k].c0; D.csum[k][1] = csum[k].c1; D.csum[k][2] = csum[k].c2; }
    D.first = first; D.log_dom = log_domain; D.adjust = 1;
    memcpy(D.adj_alpha, h_degree_alpha, PW * 8);
    memcpy(D.adj_beta, h_degree_beta, PW * 8);
    {
        ProfScope ps(ctx, "deep_rows", 8.0 * count * (nbase + 3.0 * next + PW));
        auto blocks = [&](unsigned pts) { return dim3((unsigned)((count + (size_t)msdeep::NT * pts - 1) / ((size_t)msdeep::NT * pts))); };
        if (PW == 1) {
            // two rows per lane: with the inversion pooled over the workgroup (deep_kernels.h) more rows per lane no longer pay for
            // themselves -- 2^24 rows x 9 columns: 466 us against 494 with four and 622 with one (scripts/deep_rows_probe.py)
            // (the inversion pooled over EIGHT waves -- deep_points<1, 2, 3, 8>, 512 threads -- is slower: 536 against 463 us for the same rows; 76
            // registers instead of 60 and a longer wait at the pool's barriers cost more than the halved inversions save.  Round 6, measured, not used.)
            if (npoints <= 3 && count >= 4096) hipLaunchKernelGGL((msdeep::deep_points<1, 2, 3>), blocks(2), dim3(msdeep::NT), 0, ctx->stream, D);
            else hipLaunchKernelGGL((msdeep::deep_points<1, 1, msdeep::MAXPOINTS>), blocks(1), dim3(msdeep::NT), 0, ctx->stream, D);
        } else {
            if (npoints <= 4 && count >= 4096) hipLaunchKernelGGL((msdeep::deep_points<3, 2, 4>), blocks(2), dim3(msdeep::NT), 0, ctx->stream, D);
            else hipLaunchKernelGGL((msdeep::deep_points<3, 1, msdeep::MAXPOINTS>), blocks(1), dim3(msdeep::NT), 0, ctx->stream, D);
        }
    }
    HIPCHK(hipGetLastError());
    return MS_OK;
}
