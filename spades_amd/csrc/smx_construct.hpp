// spades_amd/csrc/smx_construct.hpp — host side of the construction path: rank indexes, link records / vertices on the device,
// run_graph (masks, early clippers, successors, walks, loops) and run_coverage (included by smx_api.hip after smx_pipeline.hpp).
#pragma once

void clear_graph(smx_ctx *ctx) {
    if (ctx->g_kpo) arena_put(ctx, ctx->g_kpo);
    if (ctx->g_kmers) {
        if (ctx->d_result == ctx->g_kmers) ctx->d_result = nullptr;
        arena_put(ctx, ctx->g_kmers);
    }
    if (ctx->g_mask) arena_put(ctx, ctx->g_mask);
    if (ctx->g_ix_kmers.off) arena_put(ctx, (void *)ctx->g_ix_kmers.off);
    if (ctx->g_ix_kpo.off) arena_put(ctx, (void *)ctx->g_ix_kpo.off);
    ctx->g_ix_kmers = smx::RankIndex{};
    ctx->g_ix_kpo = smx::RankIndex{};
    ctx->g_kpo = ctx->g_kmers = nullptr;
    ctx->g_mask = nullptr;
    ctx->g_nkpo = ctx->g_nkmers = 0;
    ctx->g_ready = false;
    ctx->gh = smxh::GraphHost();
}

template <typename T>
int d2h(smx_ctx *ctx, std::vector<T> &dst, const void *src, size_t n) {
    dst.resize(n);
    if (n) HIPCHK(hipMemcpy(dst.data(), src, n * sizeof(T), hipMemcpyDeviceToHost));
    return 0;
}


// Sort distinct 64-bit keys with the counting pipeline itself: one bucket (B = 1), K = 32 so that the whole word is the key;
// keys are left-aligned first so that the MSD digits see a spread-out fraction. Used for the link records of the graph.
int device_sort_u64(smx_ctx *ctx, std::vector<uint64_t> &keys) {
    const size_t n = keys.size();
    if (n < (1u << 16)) {  // not worth a launch sequence
        smxh::radix_sort_u64(keys);
        return 0;
    }
    uint64_t mx = 0;
    for (uint64_t v : keys) mx |= v;
    const int sh = mx ? __builtin_clzll(mx) : 0;
    if (sh)
        for (auto &v : keys) v <<= sh;
    // save the count-result view (the k-mer file) that run_count overwrites
    void *sv_res = ctx->d_result;
    const uint64_t sv_n = ctx->n_records, sv_inst = ctx->n_instances;
    const unsigned sv_nw = ctx->nw, sv_K = ctx->K, sv_B = ctx->num_buckets;
    std::vector<uint64_t> sv_boff = ctx->bucket_off;
    ctx->d_result = nullptr;  // non-owning view; d_result_buf is null here
    Rec<1> *d;
    int rc = dalloc(ctx, &d, n);
    if (!rc && hipMemcpy(d, keys.data(), n * 8, hipMemcpyHostToDevice) != hipSuccess) rc = fail(ctx, SMX_DEVICE_ERROR, "key upload failed");
    if (!rc) rc = run_count<1>(ctx, 32, SMX_MODE_ALL, 1, d, n, nullptr, /*recs_reusable=*/true);
    if (!rc && ctx->n_records != n) rc = fail(ctx, SMX_DEVICE_ERROR, "link keys are not distinct");
    if (!rc && hipMemcpy(keys.data(), ctx->d_result, n * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(ctx, SMX_DEVICE_ERROR, "key download failed");
    ctx->d_result_buf = nullptr;  // the result block is one of the temps
    free_temps(ctx);
    ctx->d_result = sv_res;
    ctx->n_records = sv_n;
    ctx->n_instances = sv_inst;
    ctx->nw = sv_nw;
    ctx->K = sv_K;
    ctx->num_buckets = sv_B;
    ctx->bucket_off = sv_boff;
    if (!rc && sh)
        for (auto &v : keys) v >>= sh;
    return rc;
}

// Sort + unique 64-bit keys that are already in HBM (left-aligned) with the counting pipeline (one bucket, K = 32 so that the whole
// word is the key). *out points into the temp list (valid until free_temps); the count-result view of the context is preserved.
int device_sort_keys_dev(smx_ctx *ctx, void *d_keys, uint64_t n, unsigned long long **out, uint64_t *n_out) {
    void *sv_res = ctx->d_result;
    const uint64_t sv_n = ctx->n_records, sv_inst = ctx->n_instances;
    const unsigned sv_nw = ctx->nw, sv_K = ctx->K, sv_B = ctx->num_buckets;
    std::vector<uint64_t> sv_boff = ctx->bucket_off;
    const bool sv_want = ctx->want_index;
    ctx->want_index = false;
    ctx->d_result = nullptr;  // non-owning view; d_result_buf is null here
    int rc = run_count<1>(ctx, 32, SMX_MODE_ALL, 1, d_keys, n, nullptr, /*recs_reusable=*/true);
    *out = (unsigned long long *)ctx->d_result_buf;
    *n_out = ctx->n_records;
    ctx->d_result_buf = nullptr;  // the result block stays in the temp list
    ctx->d_result = sv_res;
    ctx->n_records = sv_n;
    ctx->n_instances = sv_inst;
    ctx->nw = sv_nw;
    ctx->K = sv_K;
    ctx->num_buckets = sv_B;
    ctx->bucket_off = sv_boff;
    ctx->want_index = sv_want;
    return rc;
}

// Link records and vertices of the graph on the device; fills g.recs / g.vstart / g.n_vertices exactly like smxh::build_links.
// Returns 1 when the sizes do not fit the packed keys (the caller then takes the host path).
int device_build_links(smx_ctx *ctx, smxh::GraphHost &g, uint64_t n_ranks) {
    const uint64_t ne = g.n_edges();
    if (ne == 0 || (ne < (1u << 16) && ctx->opt_device_links < 2) || ne >= (1ull << 29) || n_ranks >= (1ull << 31)) return 1;
    uint32_t *estart, *eend;
    uint8_t *eself;
    unsigned long long *keys, *sorted = nullptr, *one, *vidx;
    if (int rc = dalloc(ctx, &estart, ne)) return rc;
    if (int rc = dalloc(ctx, &eend, ne)) return rc;
    if (int rc = dalloc(ctx, &eself, ne)) return rc;
    if (int rc = dalloc(ctx, &keys, 2 * ne)) return rc;
    HIPCHK(hipMemcpyAsync(estart, g.estart.data(), ne * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(eend, g.eend.data(), ne * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(eself, g.eself.data(), ne, hipMemcpyHostToDevice, ctx->stream));
    const uint64_t maxkey = ((n_ranks ? n_ranks - 1 : 0) << 33) | ((1ull << 33) - 1);
    const unsigned sh = (unsigned)__builtin_clzll(maxkey | 1);
    const unsigned g1 = (unsigned)std::min<uint64_t>((ne + BLK - 1) / BLK, 1u << 16);
    hipLaunchKernelGGL(k_link_keys, dim3(g1), dim3(BLK), 0, ctx->stream, (const uint32_t *)estart, (const uint32_t *)eend, (const uint8_t *)eself, ne, sh,
                       keys);
    HIPCHK(hipGetLastError());
    uint64_t nrec = 0;
    if (int rc = device_sort_keys_dev(ctx, keys, 2 * ne, &sorted, &nrec)) return rc;
    uint64_t nself = 0;
    for (uint8_t f : g.eself) nself += f;
    if (nrec != 2 * ne - nself) return fail(ctx, SMX_DEVICE_ERROR, "link records: %llu after sort, expected %llu", (unsigned long long)nrec,
                                            (unsigned long long)(2 * ne - nself));
    if (int rc = dalloc(ctx, &one, nrec)) return rc;
    if (int rc = dalloc(ctx, &vidx, nrec + 1)) return rc;
    const unsigned g2 = (unsigned)std::min<uint64_t>((nrec + BLK - 1) / BLK, 1u << 16);
    hipLaunchKernelGGL(k_vertex_flags, dim3(g2), dim3(BLK), 0, ctx->stream, sorted, nrec, sh, one);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_unshift, dim3(g2), dim3(BLK), 0, ctx->stream, sorted, nrec, sh);
    HIPCHK(hipGetLastError());
    if (int rc = scan_u64(ctx, one, vidx, nrec)) return rc;
    unsigned long long nv = 0;
    HIPCHK(hipMemcpyAsync(&nv, vidx + nrec, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (nv >= (1ull << 31)) return 1;
    unsigned long long *vpos, *vkeys, *vsorted = nullptr, *vstart;
    if (int rc = dalloc(ctx, &vpos, nv + 1)) return rc;
    if (int rc = dalloc(ctx, &vkeys, nv + 1)) return rc;
    if (int rc = dalloc(ctx, &vstart, nv + 1)) return rc;
    const uint64_t maxv = (((((3 + 2 * ne) << 2) | 3ull) << 31) | ((1ull << 31) - 1));
    const unsigned sh2 = (unsigned)__builtin_clzll(maxv | 1);
    hipLaunchKernelGGL(k_vertex_collect, dim3(g2), dim3(BLK), 0, ctx->stream, (const unsigned long long *)sorted, (const unsigned long long *)one,
                       (const unsigned long long *)vidx, nrec, sh2, vpos, vkeys);
    HIPCHK(hipGetLastError());
    // the keys of the records are needed after the second sort: copy them out first (the pipeline reuses the arena)
    std::vector<uint64_t> hkeys(nrec);
    HIPCHK(hipMemcpyAsync(hkeys.data(), sorted, nrec * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    uint64_t nv2 = 0;
    if (nv >= (1u << 16) || ctx->opt_device_links >= 2) {
        if (int rc = device_sort_keys_dev(ctx, vkeys, nv, &vsorted, &nv2)) return rc;
        if (nv2 != nv) return fail(ctx, SMX_DEVICE_ERROR, "vertex keys are not distinct");
    } else {
        std::vector<uint64_t> hv(nv);
        HIPCHK(hipMemcpy(hv.data(), vkeys, nv * 8, hipMemcpyDeviceToHost));
        smxh::radix_sort_u64(hv);
        HIPCHK(hipMemcpy(vkeys, hv.data(), nv * 8, hipMemcpyHostToDevice));
        vsorted = vkeys;
    }
    hipLaunchKernelGGL(k_vertex_permute, dim3((unsigned)std::min<uint64_t>((nv + BLK - 1) / BLK, 1u << 16)), dim3(BLK), 0, ctx->stream,
                       (const unsigned long long *)vsorted, (const unsigned long long *)vpos, (uint64_t)nv, sh2, vstart);
    HIPCHK(hipGetLastError());
    std::vector<unsigned long long> hvs(nv);
    HIPCHK(hipMemcpyAsync(hvs.data(), vstart, nv * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    g.recs.resize(2 * ne);
    for (size_t i = 0; i < nrec; ++i) {
        const uint64_t e = hkeys[i] & ((1ull << 33) - 1);
        g.recs[i] = {((hkeys[i] >> 33) << 2) | (e & 3), e >> 2};
    }
    for (size_t i = nrec; i < 2 * ne; ++i) g.recs[i] = {~0ull, 0};
    g.vstart.assign(hvs.begin(), hvs.end());
    g.n_vertices = nv;
    return 0;
}

// Adopt the fine-bin offsets of the pipeline run that just produced a sorted file as its lookup index (bucket offsets if the run
// kept none).
int take_rank_index(smx_ctx *ctx, smx::RankIndex &ix, unsigned K, uint32_t B) {
    ix = smx::RankIndex{};
    ix.B = B;
    ix.K = K;
    ix.S1 = 1;
    if (ctx->last_idx_off && ctx->last_idx_bins) {
        ix.off = ctx->last_idx_off;
        ctx->last_idx_off = nullptr;
        ix.S1 = ctx->last_idx_S1;
        ix.nf = (uint32_t)std::min<size_t>(ctx->last_idx_f.size(), 6);
        for (uint32_t i = 0; i < ix.nf; ++i) ix.f[i] = ctx->last_idx_f[i];
        return 0;
    }
    unsigned long long *d;
    if (int rc = dalloc(ctx, &d, B + 1, false)) return rc;
    std::vector<unsigned long long> hb(ctx->bucket_off.begin(), ctx->bucket_off.end());
    hb.resize(B + 1, hb.empty() ? 0 : hb.back());
    HIPCHK(hipMemcpy(d, hb.data(), (size_t)(B + 1) * 8, hipMemcpyHostToDevice));
    ix.off = d;
    return 0;
}

template <int NW>
int run_graph(smx_ctx *ctx, unsigned k, unsigned B, const void *kpo_recs = nullptr, uint64_t n_kpo_recs = 0) {
    clear_graph(ctx);
    struct IndexScope {  // the pipeline keeps its fine-bin offsets only while a graph is being built
        smx_ctx *c;
        explicit IndexScope(smx_ctx *c_) : c(c_) { c->want_index = true; }
        ~IndexScope() {
            c->want_index = false;
            if (c->last_idx_off) arena_put(c, c->last_idx_off);
            c->last_idx_off = nullptr;
        }
    } index_scope(ctx);
    WallTrace gwt;
    ctx->g_k = k;
    ctx->g_nw = NW;
    ctx->g_B = B;
    ctx->gh.k = k;
    ctx->gh.eoff.assign(1, 0);
    // ---- 1. canonical (k+1)-mers -------------------------------------------------------------
    if (kpo_recs) {  // multi-GPU: the (k+1)-mer file gathered from its owner ranks (any order; re-sorted here)
        if (int rc = run_count<NW>(ctx, k + 1, SMX_MODE_ALL, B, kpo_recs, n_kpo_recs)) return rc;
    } else {
        if (int rc = count_reads<NW>(ctx, k + 1, SMX_MODE_CANONICAL, B)) return rc;
    }
    ctx->g_kpo = ctx->d_result_buf;
    ctx->g_nkpo = ctx->n_records;
    ctx->g_kpoboff = ctx->bucket_off;
    if (ctx->n_records)
        if (int rc = take_rank_index(ctx, ctx->g_ix_kpo, k + 1, B)) return rc;
    ctx->d_result_buf = ctx->d_result = nullptr;
    free_temps(ctx, ctx->g_kpo);
    const uint64_t nkpo = ctx->g_nkpo;
    ctx->g_kboff.assign(B + 1, 0);
    if (nkpo == 0) {
        smxh::build_links(ctx->gh);
        ctx->g_ready = true;
        ctx->n_records = 0;
        ctx->K = k;
        ctx->bucket_off.assign(B + 1, 0);
        return 0;
    }
    // ---- 2. canonical k-mers in k-mer-file order ----------------------------------------------
    {
        Rec<NW> *derived;
        if (int rc = dalloc(ctx, &derived, 2 * nkpo)) return rc;
        tbegin(ctx, "derive_kmers");
        hipLaunchKernelGGL((k_derive_kmers<NW>), dim3((unsigned)std::min<uint64_t>((nkpo + BLK - 1) / BLK, 1u << 16)), dim3(BLK), 0,
                           ctx->stream, (const void *)ctx->g_kpo, nkpo, k, (void *)derived);
        HIPCHK(hipGetLastError());
        tend(ctx);
        if (int rc = run_count<NW>(ctx, k, SMX_MODE_ALL, B, derived, 2 * nkpo, nullptr, /*recs_reusable=*/true)) return rc;
        ctx->g_kmers = ctx->d_result_buf;
        ctx->g_nkmers = ctx->n_records;
        ctx->g_kboff = ctx->bucket_off;
        if (int rc = take_rank_index(ctx, ctx->g_ix_kmers, k, B)) return rc;
        ctx->d_result_buf = nullptr;
        ctx->d_result = ctx->g_kmers;  // smx_copy_final_kmers() now yields the k-mer file
        free_temps(ctx, ctx->g_kmers);
    }
    gwt.mark(ctx, "g:counts");
    const uint64_t D0 = ctx->g_nkmers;
    if (D0 >= (1ull << 31)) return fail(ctx, SMX_MEMORY_LIMIT_EXCEEDED, "%llu k-mers exceed the 2^31 node-id limit", (unsigned long long)D0);
    const unsigned grid = (unsigned)std::min<uint64_t>((2 * D0 + BLK - 1) / BLK, 1u << 16);
    // ---- 3. extension masks ---------------------------------------------------------------------
    uint32_t *d_err;
    if (int rc = dalloc(ctx, &d_err, 1)) return rc;
    HIPCHK(hipMemsetAsync(d_err, 0, 4, ctx->stream));
    const smx::RankIndex ixk = ctx->g_ix_kmers;
    if (int rc = dalloc(ctx, &ctx->g_mask, (size_t)((D0 + 3) / 4 * 4 + 4), false)) return rc;
    HIPCHK(hipMemsetAsync(ctx->g_mask, 0, (size_t)((D0 + 3) / 4 * 4 + 4), ctx->stream));
    tbegin(ctx, "fill_masks");
    hipLaunchKernelGGL((k_fill_masks<NW>), dim3((unsigned)std::min<uint64_t>((nkpo + BLK - 1) / BLK, 1u << 16)), dim3(BLK), 0, ctx->stream,
                       (const void *)ctx->g_kpo, nkpo, k, (const void *)ctx->g_kmers, ixk, (uint32_t *)ctx->g_mask, d_err);
    HIPCHK(hipGetLastError());
    tend(ctx);
    uint32_t *succ;
    if (int rc = dalloc(ctx, &succ, 2 * D0)) return rc;
    // ---- 3a. early A/T remover (RNA pipelines: EarlyATClipper::run, stages/construction.cpp:317-326) --------------
    ctx->g_at_edges = ctx->g_at_tip_kmers = 0;
    if (ctx->opt_early_at) {
        const double ratio = 0.8;
        const uint32_t min_len = 10, max_len = 200;
        // math::ls(a, b) = !AlmostEquals(a, b) && a < b (4 ULPs, math/xmath.h:284-312): thresholds as the smallest count that is NOT ls
        auto almost_eq = [](double a, double b) {
            int64_t x, y;
            memcpy(&x, &a, 8);
            memcpy(&y, &b, 8);
            if (x < 0) x = (int64_t)0x8000000000000000ull - x;
            if (y < 0) y = (int64_t)0x8000000000000000ull - y;
            const int64_t d = x > y ? x - y : y - x;
            return d <= 4;
        };
        auto not_less = [&](double thr) {
            uint32_t c = 0;
            while (!almost_eq((double)c, thr) && (double)c < thr) ++c;
            return c;
        };
        const uint32_t thr_edge = not_less((double)k * ratio);
        std::vector<uint16_t> h_thr(max_len + 2);
        for (uint32_t n = 0; n <= max_len + 1; ++n) h_thr[n] = (uint16_t)not_less((double)std::max(n, min_len) * ratio);
        uint8_t *atflag, *isolate, *tipped;
        uint16_t *d_thr;
        unsigned long long *astats;
        if (int rc = dalloc(ctx, &atflag, 2 * D0 + 1)) return rc;
        if (int rc = dalloc(ctx, &isolate, D0 + 1)) return rc;
        if (int rc = dalloc(ctx, &tipped, 2 * D0 + 1)) return rc;
        if (int rc = dalloc(ctx, &d_thr, h_thr.size())) return rc;
        if (int rc = dalloc(ctx, &astats, 4)) return rc;
        HIPCHK(hipMemsetAsync(atflag, 0, 2 * D0 + 1, ctx->stream));
        HIPCHK(hipMemsetAsync(isolate, 0, D0 + 1, ctx->stream));
        HIPCHK(hipMemsetAsync(tipped, 0, 2 * D0 + 1, ctx->stream));
        HIPCHK(hipMemsetAsync(astats, 0, 32, ctx->stream));
        HIPCHK(hipMemcpyAsync(d_thr, h_thr.data(), h_thr.size() * 2, hipMemcpyHostToDevice, ctx->stream));
        tbegin(ctx, "early_at");
        hipLaunchKernelGGL((k_at_edges_mark<NW>), dim3(grid), dim3(BLK), 0, ctx->stream, (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask, D0, k, ixk,
                           thr_edge, atflag, d_err);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL((k_at_edges_apply<NW>), dim3(grid), dim3(BLK), 0, ctx->stream, (const void *)ctx->g_kmers, (uint32_t *)ctx->g_mask, D0, k, ixk,
                           (const uint8_t *)atflag, astats, d_err);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL((k_succ<NW>), dim3(grid), dim3(BLK), 0, ctx->stream, (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask, D0, k, ixk, succ, d_err);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL((k_at_tips_mark<NW>), dim3(grid), dim3(BLK), 0, ctx->stream, (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask,
                           (const uint32_t *)succ, D0, k, ixk, min_len, max_len, (const uint16_t *)d_thr, isolate, tipped, astats, d_err);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_tip_apply, dim3(grid), dim3(BLK), 0, ctx->stream, ctx->g_mask, (const uint8_t *)isolate, D0);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL((k_tip_fix<NW>), dim3(grid), dim3(BLK), 0, ctx->stream, (const void *)ctx->g_kmers, (uint32_t *)ctx->g_mask,
                           (const uint8_t *)tipped, D0, k, ixk, d_err);
        HIPCHK(hipGetLastError());
        tend(ctx);
        unsigned long long hs[4] = {0, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(hs, astats, 32, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        ctx->g_at_tip_kmers = hs[0];
        ctx->g_at_edges = hs[2];
    }
    // ---- 3b. early tip clipper (spades-core variant, off for spades-gbuilder) ---------------------
    ctx->g_tip_kmers = ctx->g_tips = 0;
    if (ctx->opt_early_tip_bound > 0) {
        uint8_t *isolate, *tipped;
        unsigned long long *tstats;
        if (int rc = dalloc(ctx, &isolate, D0 + 1)) return rc;
        if (int rc = dalloc(ctx, &tipped, 2 * D0 + 1)) return rc;
        if (int rc = dalloc(ctx, &tstats, 2)) return rc;
        HIPCHK(hipMemsetAsync(isolate, 0, D0 + 1, ctx->stream));
        HIPCHK(hipMemsetAsync(tipped, 0, 2 * D0 + 1, ctx->stream));
        HIPCHK(hipMemsetAsync(tstats, 0, 16, ctx->stream));
        tbegin(ctx, "early_tips");
        hipLaunchKernelGGL((k_succ<NW>), dim3(grid), dim3(BLK), 0, ctx->stream, (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask, D0, k, ixk, succ, d_err);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL((k_tip_mark<NW>), dim3(grid), dim3(BLK), 0, ctx->stream, (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask,
                           (const uint32_t *)succ, D0, k, ixk, (uint32_t)std::min<int64_t>(ctx->opt_early_tip_bound, 0x7FFFFFFF), isolate, tipped, tstats,
                           d_err);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_tip_apply, dim3(grid), dim3(BLK), 0, ctx->stream, ctx->g_mask, (const uint8_t *)isolate, D0);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL((k_tip_fix<NW>), dim3(grid), dim3(BLK), 0, ctx->stream, (const void *)ctx->g_kmers, (uint32_t *)ctx->g_mask,
                           (const uint8_t *)tipped, D0, k, ixk, d_err);
        HIPCHK(hipGetLastError());
        tend(ctx);
        unsigned long long hs[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(hs, tstats, 16, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        ctx->g_tip_kmers = hs[0];
        ctx->g_tips = hs[1];
    }
    // ---- 4. successors + start de-edges -------------------------------------------------------
    unsigned long long *ccnt, *coff;
    if (int rc = dalloc(ctx, &ccnt, D0)) return rc;
    if (int rc = dalloc(ctx, &coff, D0 + 1)) return rc;
    tbegin(ctx, "succ");
    hipLaunchKernelGGL((k_succ<NW>), dim3(grid), dim3(BLK), 0, ctx->stream, (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask, D0, k, ixk, succ, d_err);
    HIPCHK(hipGetLastError());
    tend(ctx);
    tbegin(ctx, "candidates");
    hipLaunchKernelGGL(k_cand_count, dim3(grid), dim3(BLK), 0, ctx->stream, (const uint8_t *)ctx->g_mask, D0, ccnt);
    HIPCHK(hipGetLastError());
    if (int rc = scan_u64(ctx, ccnt, coff, D0)) return rc;
    unsigned long long C = 0;
    HIPCHK(hipMemcpyAsync(&C, coff + D0, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    tend(ctx);
    gwt.mark(ctx, "g:masks+succ");
    std::vector<unsigned long long> h_eoff;
    uint64_t n_paths = 0;
    uint8_t *visited;
    if (int rc = dalloc(ctx, &visited, D0 + 1)) return rc;
    HIPCHK(hipMemsetAsync(visited, 0, D0 + 1, ctx->stream));
    if (C > 0) {
        unsigned long long *cand, *len, *soff, *keeplen, *koff, *one, *eidx;
        uint32_t *first, *last;
        uint8_t *flags;
        if (int rc = dalloc(ctx, &cand, C)) return rc;
        if (int rc = dalloc(ctx, &len, C)) return rc;
        if (int rc = dalloc(ctx, &soff, C + 1)) return rc;
        if (int rc = dalloc(ctx, &keeplen, C)) return rc;
        if (int rc = dalloc(ctx, &koff, C + 1)) return rc;
        if (int rc = dalloc(ctx, &one, C)) return rc;
        if (int rc = dalloc(ctx, &eidx, C + 1)) return rc;
        if (int rc = dalloc(ctx, &first, C)) return rc;
        if (int rc = dalloc(ctx, &last, C)) return rc;
        if (int rc = dalloc(ctx, &flags, C)) return rc;
        const unsigned cgrid = (unsigned)std::min<uint64_t>((C + BLK - 1) / BLK, 1u << 16);
        hipLaunchKernelGGL(k_cand_expand, dim3(grid), dim3(BLK), 0, ctx->stream, (const uint8_t *)ctx->g_mask,
                           (const unsigned long long *)coff, D0, cand);
        HIPCHK(hipGetLastError());
        tbegin(ctx, "walk_len");
        hipLaunchKernelGGL((k_walk_len<NW>), dim3(cgrid), dim3(BLK), 0, ctx->stream, (const unsigned long long *)cand, (uint64_t)C,
                           (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask, (const uint32_t *)succ, k, ixk,
                           (uint64_t)(2 * D0 + 2), len, first, last, d_err);
        HIPCHK(hipGetLastError());
        if (int rc = scan_u64(ctx, len, soff, C)) return rc;
        unsigned long long total = 0;
        HIPCHK(hipMemcpyAsync(&total, soff + C, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        tend(ctx);
        char *seq;
        if (int rc = dalloc(ctx, &seq, total + 1)) return rc;
        tbegin(ctx, "walk_write");
        hipLaunchKernelGGL((k_walk_write<NW>), dim3(cgrid), dim3(BLK), 0, ctx->stream, (const unsigned long long *)cand, (uint64_t)C,
                           (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask, (const uint32_t *)succ, k, (const uint32_t *)first,
                           (const unsigned long long *)soff, seq, visited);
        HIPCHK(hipGetLastError());
        tend(ctx);
        tbegin(ctx, "keep_gather");
        hipLaunchKernelGGL(k_keep, dim3(cgrid), dim3(BLK), 0, ctx->stream, (const char *)seq, (const unsigned long long *)soff, (uint64_t)C,
                           keeplen, flags);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_keep_flag, dim3(cgrid), dim3(BLK), 0, ctx->stream, (const uint8_t *)flags, (uint64_t)C, one);
        HIPCHK(hipGetLastError());
        if (int rc = scan_u64(ctx, keeplen, koff, C)) return rc;
        if (int rc = scan_u64(ctx, one, eidx, C)) return rc;
        unsigned long long ktotal = 0, nkept = 0;
        HIPCHK(hipMemcpyAsync(&ktotal, koff + C, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(&nkept, eidx + C, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        char *kseq;
        unsigned long long *eoff;
        uint32_t *estart, *eend;
        uint8_t *eself;
        if (int rc = dalloc(ctx, &kseq, ktotal + 1)) return rc;
        if (int rc = dalloc(ctx, &eoff, nkept + 1)) return rc;
        if (int rc = dalloc(ctx, &estart, nkept + 1)) return rc;
        if (int rc = dalloc(ctx, &eend, nkept + 1)) return rc;
        if (int rc = dalloc(ctx, &eself, nkept + 1)) return rc;
        hipLaunchKernelGGL(k_gather, dim3(cgrid), dim3(BLK), 0, ctx->stream, (const char *)seq, (const unsigned long long *)soff,
                           (const unsigned long long *)koff, (const uint8_t *)flags, (const unsigned long long *)eidx,
                           (const unsigned long long *)cand, (const uint32_t *)last, (uint64_t)C, kseq, eoff, estart, eend, eself);
        HIPCHK(hipGetLastError());
        tend(ctx);
        HIPCHK(hipStreamSynchronize(ctx->stream));
        gwt.mark(ctx, "g:walks");
        // ---- to host ----
        n_paths = nkept;
        if (int rc = d2h(ctx, h_eoff, eoff, nkept)) return rc;
        ctx->gh.seq.resize(ktotal);
        if (ktotal) HIPCHK(hipMemcpy(&ctx->gh.seq[0], kseq, ktotal, hipMemcpyDeviceToHost));
        if (int rc = d2h(ctx, ctx->gh.estart, estart, nkept)) return rc;
        if (int rc = d2h(ctx, ctx->gh.eend, eend, nkept)) return rc;
        if (int rc = d2h(ctx, ctx->gh.eself, eself, nkept)) return rc;
        ctx->gh.eoff.assign(h_eoff.begin(), h_eoff.end());
        ctx->gh.eoff.push_back(ktotal);
    }
    gwt.mark(ctx, "g:d2h");
    {
        // ---- perfect loops: non-junction k-mers on no path ----
        uint32_t *lcount, *llist;
        const uint32_t lcap = (uint32_t)std::min<uint64_t>(D0, 1u << 26);
        if (int rc = dalloc(ctx, &lcount, 1)) return rc;
        if (int rc = dalloc(ctx, &llist, lcap)) return rc;
        HIPCHK(hipMemsetAsync(lcount, 0, 4, ctx->stream));
        hipLaunchKernelGGL(k_loop_nodes, dim3(grid), dim3(BLK), 0, ctx->stream, (const uint8_t *)ctx->g_mask, (const uint8_t *)visited, D0,
                           lcount, llist, lcap);
        HIPCHK(hipGetLastError());
        uint32_t nloopk = 0;
        HIPCHK(hipMemcpyAsync(&nloopk, lcount, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (nloopk > lcap) return fail(ctx, SMX_MEMORY_LIMIT_EXCEEDED, "%u k-mers on perfect loops exceed the host-side limit", nloopk);
        if (nloopk && ctx->opt_keep_loops) {
            std::vector<uint32_t> ranks;
            if (int rc = d2h(ctx, ranks, llist, nloopk)) return rc;
            std::sort(ranks.begin(), ranks.end());  // k-mer-file order
            HIPCHK(hipMemcpy(llist, ranks.data(), (size_t)nloopk * 4, hipMemcpyHostToDevice));
            Rec<NW> *lk;
            if (int rc = dalloc(ctx, &lk, nloopk)) return rc;
            hipLaunchKernelGGL((k_gather_kmers<NW>), dim3((nloopk + BLK - 1) / BLK), dim3(BLK), 0, ctx->stream, (const void *)ctx->g_kmers,
                               (const uint32_t *)llist, nloopk, (void *)lk);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(ctx->stream));
            std::vector<uint64_t> hk;
            if (int rc = d2h(ctx, hk, lk, (size_t)nloopk * NW)) return rc;
            std::vector<uint8_t> hmask;
            if (int rc = d2h(ctx, hmask, ctx->g_mask, (size_t)D0)) return rc;
            std::vector<smxh::LoopNode> nodes(nloopk);
            for (uint32_t i = 0; i < nloopk; ++i) {
                nodes[i].rank = ranks[i];
                nodes[i].kmer.resize(k);
                for (unsigned j = 0; j < k; ++j) nodes[i].kmer[j] = "ACGT"[(hk[(size_t)i * NW + (j >> 5)] >> ((j & 31) << 1)) & 3];
                nodes[i].mask = hmask[ranks[i]];
            }
            smxh::LoopCollector lc(nodes, k);
            std::vector<std::string> loops;
            lc.collect(loops);
            // node ids must be taken from the untouched masks' k-mers: rebuild a lookup (masks were zeroed by collect)
            for (auto &s : loops) {
                const std::string fk = s.substr(0, k), lk2 = s.substr(s.size() - k);
                ctx->gh.estart.push_back(lc.node_of(fk));
                ctx->gh.eend.push_back(lc.node_of(lk2));
                ctx->gh.eself.push_back(s == smxh::revcomp(s) ? 1 : 0);
                ctx->gh.seq += s;
                ctx->gh.eoff.push_back(ctx->gh.seq.size());
            }
            ctx->gh.n_loops = loops.size();
        }
    }
    ctx->gh.n_paths = n_paths;
    unsigned herr = 0;
    HIPCHK(hipMemcpy(&herr, d_err, 4, hipMemcpyDeviceToHost));
    if (herr) return fail(ctx, SMX_DEVICE_ERROR, "inconsistent k-mer index: %u failed lookups/walks", herr);
    gwt.mark(ctx, "g:loops");
    if (ctx->opt_sort_edges) smxh::sort_edges_raw(ctx->gh);
    free_temps(ctx);  // walk buffers are no longer needed; the link sort reuses the arena
    int lrc = ctx->opt_device_links ? device_build_links(ctx, ctx->gh, D0) : 1;
    free_temps(ctx);
    if (lrc > 1) return lrc;
    if (lrc == 1) {  // small graphs, or sizes beyond the packed keys: host link records with the device (or host) key sort
        int sort_rc = 0;
        smxh::build_links(ctx->gh, [&](std::vector<uint64_t> &keys) {
            if (!sort_rc) sort_rc = device_sort_u64(ctx, keys);
            if (sort_rc) smxh::radix_sort_u64(keys);
        });
        if (sort_rc) return sort_rc;
    }
    gwt.mark(ctx, "g:links");
    ctx->g_ready = true;
    return 0;
}


template <int NW>
int run_coverage(smx_ctx *ctx) {
    const unsigned K1 = ctx->g_k + 1, B = ctx->g_B;
    const uint64_t D1 = ctx->g_nkpo, ne = ctx->gh.n_edges();
    ctx->gh.ecov.assign(ne, 0);
    if (D1 == 0 || ne == 0) return 0;
    std::vector<uint64_t *> masks;
    uint64_t nwin = 0;
    if (int rc = mark_windows(ctx, K1, masks, &nwin)) return rc;
    uint32_t *cnt, *ecov, *fls, *fle;
    unsigned long long *d_eoff;
    char *d_seq;
    if (int rc = dalloc(ctx, &cnt, D1)) return rc;
    if (int rc = dalloc(ctx, &ecov, ne)) return rc;
    if (int rc = dalloc(ctx, &fls, ne)) return rc;
    if (int rc = dalloc(ctx, &fle, ne)) return rc;
    if (int rc = dalloc(ctx, &d_eoff, ne + 1)) return rc;
    if (int rc = dalloc(ctx, &d_seq, ctx->gh.seq.size() + 1)) return rc;
    HIPCHK(hipMemsetAsync(cnt, 0, D1 * 4, ctx->stream));
    HIPCHK(hipMemsetAsync(ecov, 0, ne * 4, ctx->stream));
    HIPCHK(hipMemsetAsync(fls, 0, ne * 4, ctx->stream));
    HIPCHK(hipMemsetAsync(fle, 0, ne * 4, ctx->stream));
    std::vector<unsigned long long> he(ctx->gh.eoff.begin(), ctx->gh.eoff.end());
    const smx::RankIndex ixp = ctx->g_ix_kpo;
    HIPCHK(hipMemcpyAsync(d_eoff, he.data(), (ne + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_seq, ctx->gh.seq.data(), ctx->gh.seq.size(), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    tbegin(ctx, "kpo_coverage");
    for (size_t ci = 0; ci < ctx->chunks.size(); ++ci) {
        const ReadChunk &ch = ctx->chunks[ci];
        if (ch.n_bases == 0 || !masks[ci] || ch.contigs) continue;  // contigs: "separate stream for not counting it in coverage"
        hipLaunchKernelGGL((k_kpo_coverage<NW>), dim3((unsigned)std::min<uint64_t>((ch.n_bases + BLK - 1) / BLK, 1u << 16)), dim3(BLK), 0,
                           ctx->stream, (const uint64_t *)ch.d_words, (const uint64_t *)masks[ci], ch.n_bases, K1, (const void *)ctx->g_kpo, ixp, cnt);
        HIPCHK(hipGetLastError());
    }
    tend(ctx);
    tbegin(ctx, "edge_coverage");
    const uint64_t total = ctx->gh.seq.size();
    hipLaunchKernelGGL((k_edge_coverage<NW>), dim3((unsigned)std::min<uint64_t>((total + BLK - 1) / BLK, 1u << 16)), dim3(BLK), 0, ctx->stream,
                       (const char *)d_seq, (const unsigned long long *)d_eoff, ne, total, K1, (const void *)ctx->g_kpo, ixp,
                       (const uint32_t *)cnt, ecov, (uint32_t)std::max<int64_t>(ctx->opt_flank_range, 1), fls, fle);
    HIPCHK(hipGetLastError());
    tend(ctx);
    ctx->gh.eflank_s.assign(ne, 0);
    ctx->gh.eflank_e.assign(ne, 0);
    HIPCHK(hipMemcpyAsync(ctx->gh.ecov.data(), ecov, ne * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->gh.eflank_s.data(), fls, ne * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->gh.eflank_e.data(), fle, ne * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
