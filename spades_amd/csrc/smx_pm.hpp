// spades_amd/csrc/smx_pm.hpp — host side of the partition-major construction route (kernels: smx_pm.hip; included by smx_api.hip
// after smx_construct.hpp).
//
// Route: mark the k-mer windows of the reads that hold a (k+1)-mer -> super-k-mer pre-dedupe with extension bytes, winners in
// partition-major order with their side arrays (run_prededupe, MODE 2) -> node table + jump words per chunk (k_pm_tab) ->
// graph_from_masks on that numbering (smx_construct.hpp; the junction k-mers alone are sorted into the reference's k-mer-file order to
// number the start de-edges). The sorted k-mer file itself (smx_copy_final_kmers & co. after smx_build_graph) is made on demand.
#pragma once

void pm_release(smx_ctx *ctx) {
    PmState &P = ctx->pm;
    arena_put(ctx, P.pinfo);
    arena_put(ctx, P.cinfo);
    arena_put(ctx, P.meta);
    arena_put(ctx, P.overflow);
    arena_put(ctx, P.mask);
    arena_put(ctx, P.llink);
    arena_put(ctx, P.pals);
    arena_put(ctx, P.tab);
    arena_put(ctx, P.jmp);
    arena_put(ctx, P.rbits);
    arena_put(ctx, P.tab_stats);
    arena_put(ctx, P.dk);
    arena_put(ctx, (void *)P.ddir.dir);
    arena_put(ctx, (void *)P.ddir.boff);
    P = PmState();
}

// XXH3-64 of a k-mer record on the host (xxh3_rec of smx_device.hpp, for the handful of k-mers the host has to put into k-mer-file order)
inline uint64_t pm_host_xxh3(const uint64_t *w, int nw) {
    auto fold = [](uint64_t a, uint64_t b) {
        const unsigned __int128 p = (unsigned __int128)a * b;
        return (uint64_t)p ^ (uint64_t)(p >> 64);
    };
    auto rotl = [](uint64_t v, int r) { return (v << r) | (v >> (64 - r)); };
    auto aval = [](uint64_t h) {
        h ^= h >> 37;
        h *= SMX_XXH_MX1;
        h ^= h >> 32;
        return h;
    };
    if (nw == 1) {
        uint64_t h = rotl(w[0], 32) ^ SMX_XXH_BITFLIP8;
        h ^= rotl(h, 49) ^ rotl(h, 24);
        h *= SMX_XXH_MX2;
        h ^= (h >> 35) + 8;
        h *= SMX_XXH_MX2;
        return h ^ (h >> 28);
    }
    if (nw == 2) {
        const uint64_t lo = w[0] ^ SMX_XXH_BITFLIP16A, hi = w[1] ^ SMX_XXH_BITFLIP16B;
        return aval(16 + __builtin_bswap64(lo) + hi + fold(lo, hi));
    }
    uint64_t acc = (uint64_t)(8 * nw) * SMX_XXH_P64_1;
    acc += fold(w[0] ^ SMX_XXH_SEC0, w[1] ^ SMX_XXH_SEC8);
    acc += fold(w[nw - 2] ^ SMX_XXH_SEC16, w[nw - 1] ^ SMX_XXH_SEC24);
    return aval(acc);
}
inline uint32_t pm_host_bucket(const uint64_t *w, int nw, uint32_t B) {
    return B == 1 ? 0u : (uint32_t)(((unsigned __int128)pm_host_xxh3(w, nw) * B) >> 64);
}

// 0: the graph is built; SMX_ROUTE_NA: the route does not apply to this input or did not fit (nothing left behind, the caller takes
// another route); anything else: an error.
template <int NW>
int pm_route(smx_ctx *ctx, unsigned k, unsigned B, WallTrace &gwt) {
    if (ctx->opt_pm_route == 0 || ctx->opt_ext_route == 0 || ctx->chunks.empty() || ctx->opt_derive_batches != 0 ||
        ctx->opt_ext_presort == 0 || ctx->opt_batch_records > 0)
        return SMX_ROUTE_NA;
    // spades-core's early clippers (construction.info: early_tip_clipper { enable true } — its default) edit the extension masks before the
    // condensation. Until round 5 they sent the build to the sorted routes ("their lookups may miss"); pm_find answers exactly for absent
    // k-mers too, so they run here (round 6): on the partition-major records, walking the route's own node table, which is made again from the
    // clipped masks (graph_from_masks: early_clippers<PmFind>, retab below).
    const bool clip = ctx->opt_early_at || ctx->opt_early_tip_bound > 0;
    // k = 29, 31, 61, 63, 93, 95, 125, 127 leave the last record word no room for the InOutMask byte (EXT layout): the route then runs on PLAIN k-mer
    // records ("nx") — the byte of every k-mer lives in the mask array the route keeps anyway, the junction k-mers are sorted without it and get it
    // back by their rank lookups, the survivors of cut partitions gather theirs by a search in their own sorted set (round 5; option nx_route = 0:
    // such k take the (k+1)-mer route as before)
    const bool nx = !ext_layout_fits(k, NW);
    if (nx && ctx->opt_nx_route == 0) return SMX_ROUTE_NA;
    const size_t W = sizeof(Rec<NW>);
    auto bail = [&](int rc) {  // leave nothing behind
        if (getenv("SMX_DEBUG") || getenv("SMX_DEBUG_BAIL")) {
            fprintf(stderr, "[smx] partition-major route gives up (code %d: %s); %.1f GB obtainable; last stage begun: %s; live blocks of 1 GB and more:", rc,
                    rc == SMX_ROUTE_NA ? "does not apply" : ctx->err.c_str(), (double)arena_avail(ctx) / 1e9, ctx->timings.empty() ? "-" : ctx->timings.back().name.c_str());
            std::vector<size_t> big;
            for (auto &b : ctx->arena.live)
                if (b.second >= ((size_t)1 << 30)) big.push_back(b.second);
            std::sort(big.rbegin(), big.rend());
            for (size_t v : big) fprintf(stderr, " %.1f", (double)v / 1e9);
            fprintf(stderr, "\n");
        }
        ctx->ext_mode = false;
        ctx->pm.active = false;
        ctx->pm.nx = false;
        if (ctx->side_stream) (void)hipStreamSynchronize(ctx->side_stream);
        (void)hipStreamSynchronize(ctx->stream);
        for (auto &t : ctx->timings) {
            (void)hipEventDestroy(t.e0);
            (void)hipEventDestroy(t.e1);
        }
        ctx->timings.clear();
        ctx->tprefix.clear();
        free_temps(ctx);
        clear_graph(ctx);
        clear_result(ctx);
        return (rc == SMX_MEMORY_LIMIT_EXCEEDED || rc == SMX_RETRY_SMALLER) ? SMX_ROUTE_NA : rc;
    };
    ctx->tprefix = "kmers:";
    std::vector<uint64_t *> masks;
    uint64_t nwin = 0;
    tbegin(ctx, "mark_windows");
    int rc = mark_windows(ctx, k, masks, &nwin, /*temp_masks=*/true, /*min_len=*/k + 1, /*wait_words=*/false);  // (the first scan follows the upload piece by piece)
    tend(ctx);
    if (rc) return bail(rc);
    if (!prededupe_applies<NW>(ctx, k, nwin)) return bail(SMX_ROUTE_NA);
    // HBM plan. The stage: super-k-mer slots + staging (< 5 B per window, run_prededupe) next to the output buffer and its side arrays;
    // then per distinct k-mer: record + byte + 2 node-table entries + 2 jump words + ~8 B of walk arrays and group words.
    uint64_t out_cap = nwin;
    {
        const double avail = (double)arena_avail(ctx);
        if (10.0 * (double)nwin > avail) return bail(SMX_ROUTE_NA);  // the stage alone would need batches
        // record + mask byte + share of the group words + local links, or (fuse_tab) the node table and jump words the stage writes itself
        const bool fuse = ctx->opt_pm_fuse_tab != 0 && !(clip && ctx->opt_pm_full_retab != 0);
        const double fit1 = (avail - 5.0 * (double)nwin) / ((double)W + (fuse ? 3.0 + 16.0 + 8.0 + (clip ? 4.0 : 0.0) : 7.0));
        const double fit2 = avail / ((double)W + 1.0 + 16.0 + 8.0 + 8.0);
        const double fit = std::max(std::min(fit1, fit2), 1.0);
        if (fit < (double)nwin) out_cap = (uint64_t)fit;
    }
    Rec<NW> *recs = nullptr;
    uint64_t n = 0;
    {
        ReadSel sel;
        sel.masks = &masks;
        sel.nrec = nwin;
        ctx->ext_mode = !nx;
        ctx->pm.active = true;
        ctx->pm.nx = nx;
        // (the whole table again after an early clipper — option pm_full_retab — is k_pm_tab on the edited masks: it needs the link array)
        ctx->pm.fuse_tab = ctx->opt_pm_fuse_tab != 0 && !(clip && ctx->opt_pm_full_retab != 0);
        ctx->pm.keep_links = clip;
        rc = run_prededupe<NW>(ctx, k, sel, nwin, &recs, &n, out_cap);
        ctx->ext_mode = false;
        ctx->pm.active = false;
    }
    if (rc) return bail(rc);
    ctx->tprefix.clear();
    PmState &P = ctx->pm;
    if (P.nclean + P.ndirty != n || n >= (1ull << 39)) return bail(fail(ctx, SMX_DEVICE_ERROR, "partition-major dedupe: %llu clean + %llu dirty k-mers, %llu in all",
                                                                      (unsigned long long)P.nclean, (unsigned long long)P.ndirty, (unsigned long long)n));
    // the stage's temporaries go; its output becomes the k-mer array of the graph (EXT records, partition-major)
    free_temps(ctx, recs);
    ctx->g_kmers = recs;
    ctx->g_pm = true;
    ctx->g_pm_nx = nx;
    ctx->g_nkmers = n;
    ctx->n_instances = nwin;
    arena_shrink(ctx, recs, (size_t)std::max<uint64_t>(n, 1) * W);
    arena_shrink(ctx, P.meta, (size_t)std::max<uint32_t>(P.nchunks, 1) * (P.T >> 4) * 4);
    arena_shrink(ctx, P.cinfo, (size_t)std::max<uint32_t>(P.nchunks, 1) * 8);
    arena_shrink(ctx, P.mask, (size_t)n + 16);
    if (P.llink) arena_shrink(ctx, P.llink, ((size_t)n + 16) * 4);
    const bool fused = P.tab != nullptr;  // the dedupe stage wrote the node table of the clean chunks
    if (fused) {
        arena_shrink(ctx, P.tab, (2 * (size_t)n + 2) * 8);
        arena_shrink(ctx, P.jmp, (2 * (size_t)n + 2) * 4);
        arena_shrink(ctx, P.rbits, (size_t)std::max<uint32_t>(P.nchunks, 1) * (P.T >> 5) * 4);
    }
    ctx->g_mask = P.mask;
    P.mask = nullptr;
    gwt.mark(ctx, "g:kmers+masks");
    const uint64_t D0 = n;
    if (D0 == 0) {
        ctx->g_nkpo = 0;
        ctx->g_host_valid = true;  // the empty graph
        ctx->g_ready = true;
        ctx->n_records = 0;
        ctx->K = k;
        ctx->nw = NW;
        ctx->num_buckets = B;
        ctx->bucket_off.assign(B + 1, 0);
        return 0;
    }
    if (hipMemsetAsync(ctx->g_mask + D0, 0, 16, ctx->stream) != hipSuccess) return bail(fail(ctx, SMX_DEVICE_ERROR, "mask tail reset failed"));
    // dirty region: k-mers without their bytes + rank directory (one bucket), bytes into the mask array
    if (P.ndirty) {
        Rec<NW> *dk;
        if ((rc = dalloc(ctx, &dk, P.ndirty, false))) return bail(rc);
        P.dk = dk;
        if (nx) {  // plain k-mers already, their bytes in the mask array already (run_prededupe): the directory's copy is a copy
            if (hipMemcpyAsync(dk, recs + P.nclean, (size_t)P.ndirty * W, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
                return bail(fail(ctx, SMX_DEVICE_ERROR, "copy of the sorted tail failed"));
        } else {
            hipLaunchKernelGGL((k_pm_dirty_split<NW>), dim3(grid_for(P.ndirty)), dim3(BLK), 0, ctx->stream, (const void *)recs, (uint64_t)P.nclean, (uint64_t)P.ndirty,
                               (void *)dk, ctx->g_mask);
            if (hipGetLastError() != hipSuccess) return bail(fail(ctx, SMX_DEVICE_ERROR, "k_pm_dirty_split launch failed"));
        }
        if (P.dirty_boff.size() != (size_t)P.dirty_B + 1 || P.dirty_boff.back() != P.ndirty)
            return bail(fail(ctx, SMX_DEVICE_ERROR, "bucket offsets of the sorted tail do not add up (%llu of %llu)",
                             (unsigned long long)(P.dirty_boff.empty() ? 0 : P.dirty_boff.back()), (unsigned long long)P.ndirty));
        tbegin(ctx, "rank_dir");
        rc = build_rank_dir<NW>(ctx, dk, P.ndirty, P.dirty_boff, P.dirty_B, k, P.ddir);
        tend(ctx);
        if (rc) return bail(rc);
    }
    PmWalk pw{};
    pw.ix.recs = recs;
    pw.ix.pinfo = P.pinfo;
    pw.ix.meta = P.meta;
    pw.ix.T = P.T;
    pw.ix.ngroups = P.T >> 4;
    pw.ix.K = k;
    pw.ix.m = P.m;
    pw.ix.w = P.w;
    pw.ix.pshift = P.pshift;
    pw.ix.nclean = P.nclean;
    pw.ix.dk = P.dk;
    pw.ix.ddir = P.ddir;
    pw.ix.xs = nx ? 0u : EXT_BITS;
    pw.ix.mask = ctx->g_mask;
    pw.ix.bym = (nx || clip) ? 1u : 0u;  // (the mask array holds every k-mer's byte on this route either way; a clipper edits IT, not the records)
    uint32_t *d_err, *jmp;
    node_t *tab;
    unsigned long long *stats;
    if ((rc = dalloc(ctx, &d_err, 1))) return bail(rc);
    if ((rc = dalloc(ctx, &stats, 2))) return bail(rc);
    if (fused) {  // (temporaries of this build from here on, like the ones asked for below)
        tab = P.tab;
        jmp = P.jmp;
        P.tab = nullptr;
        P.jmp = nullptr;
        ctx->temps.push_back(tab);
        ctx->temps.push_back(jmp);
    } else {
        if ((rc = dalloc(ctx, &tab, 2 * D0 + 2))) return bail(rc);
        if ((rc = dalloc(ctx, &jmp, 2 * D0 + 2))) return bail(rc);
    }
    pw.jmp = jmp;
    uint32_t *cob;
    if ((rc = dalloc(ctx, &cob, (size_t)(D0 >> 8) + 2))) return bail(rc);
    if (hipMemsetAsync(cob, 0, ((size_t)(D0 >> 8) + 2) * 4, ctx->stream) != hipSuccess) return bail(fail(ctx, SMX_DEVICE_ERROR, "chunk map reset failed"));
    if (P.nchunks) hipLaunchKernelGGL(k_pm_cob, dim3(std::min<uint32_t>((P.nchunks + BLK - 1) / BLK, 1u << 16)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)P.cinfo, P.nchunks, cob);
    pw.cinfo = P.cinfo;
    pw.cob = cob;
    pw.nchunks = P.nchunks;
    if (hipMemsetAsync(d_err, 0, 4, ctx->stream) != hipSuccess || hipMemsetAsync(stats, 0, 16, ctx->stream) != hipSuccess)
        return bail(fail(ctx, SMX_DEVICE_ERROR, "counter reset failed"));
    const uint32_t maxn = P.T / 2;  // winners of a chunk <= its instance capacity
    const size_t lds = (size_t)maxn * 8 + (size_t)maxn * 4 + 2 * (size_t)(maxn >> 4) * 4 + 16;
    if ((rc = set_lds(ctx, k_pm_tab, lds))) return bail(rc);
    // per chunk maxn / 16 words: the nodes whose successor lies in another chunk (k_pm_tab marks, k_pm_remote looks up)
    const uint32_t wpc = maxn >> 4;
    uint32_t *rbits = nullptr;
    if (fused) {
        rbits = P.rbits;
        P.rbits = nullptr;
        ctx->temps.push_back(rbits);
    } else {
        if ((rc = dalloc(ctx, &rbits, (size_t)std::max<uint32_t>(P.nchunks, 1) * wpc))) return bail(rc);
        if (hipMemsetAsync(rbits, 0, (size_t)std::max<uint32_t>(P.nchunks, 1) * wpc * 4, ctx->stream) != hipSuccess) return bail(fail(ctx, SMX_DEVICE_ERROR, "counter reset failed"));
    }
    unsigned long long *prof = nullptr;
    if (getenv("SMX_DEBUG")) {
        if ((rc = dalloc(ctx, &prof, 8))) return bail(rc);
        if (hipMemsetAsync(prof, 0, 64, ctx->stream) != hipSuccess) return bail(fail(ctx, SMX_DEVICE_ERROR, "counter reset failed"));
    }
    // The successor table (k_pm_tab: HBM writes; k_pm_remote: random sectors, latency) and the reference's order of the start de-edges (the
    // junction k-mers through the sort pipeline: streaming) need nothing from each other — both read the records and the mask bytes only. Round 6:
    // the table is filled on a side stream while graph_from_masks sorts the junction k-mers on the main one; the host joins them before the
    // first walk (tab_ready below). Their stage times then overlap: the sum of the stage times exceeds the step's wall clock by what was hidden.
    const bool overlap = ctx->opt_pm_overlap != 0 && !getenv("SMX_DEBUG") && !clip;
    hipStream_t main_stream = ctx->stream;
    if (overlap) {
        if (!ctx->side_stream && hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(ctx, SMX_DEVICE_ERROR, "side stream"));
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, main_stream) != hipSuccess ||
            hipStreamWaitEvent(ctx->side_stream, ev, 0) != hipSuccess)
            return bail(fail(ctx, SMX_DEVICE_ERROR, "side stream: %s", hipGetErrorString(hipGetLastError())));
        (void)hipEventDestroy(ev);  // (released once it has completed)
        ctx->stream = ctx->side_stream;  // tbegin / tend and the launches below take the context's stream
    }
    tbegin(ctx, "pm_tab");
    if (fused) {  // the clean chunks' entries are there; their extension bits were counted by the stage
        if (hipMemcpyAsync(stats, P.tab_stats, 8, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) return bail(fail(ctx, SMX_DEVICE_ERROR, "counter copy failed"));
    } else if (P.nchunks)
        hipLaunchKernelGGL(k_pm_tab, dim3(std::min<uint32_t>(P.nchunks, 256 * 16)), dim3(BLK), lds, ctx->stream, (const unsigned long long *)P.cinfo, P.nchunks,
                           maxn, (const uint8_t *)ctx->g_mask, (const uint32_t *)P.llink, tab, jmp, stats, d_err, prof, rbits, (const uint8_t *)nullptr);
    if (P.ndirty)
        hipLaunchKernelGGL((k_pm_tab_dirty<NW>), dim3(grid_for(P.ndirty)), dim3(BLK), 0, ctx->stream, pw.ix, (uint64_t)P.ndirty, k, tab, jmp, stats, d_err);
    tend(ctx);
    tbegin(ctx, "pm_remote");
    if (P.nchunks) {
        // (an edge that leaves its chunk is answered once for both of its ends, then what is still pending: see k_pm_remote; option pm_remote_mirror = 0: every end asks)
        const bool mirror = ctx->opt_pm_remote_mirror != 0;
        for (unsigned mode = mirror ? 1u : 0u; mode <= (mirror ? 2u : 0u); ++mode)
            hipLaunchKernelGGL((k_pm_remote<NW>), dim3((unsigned)std::min<uint64_t>((P.nchunks + PMR_CH - 1) / PMR_CH, 256 * 32)), dim3(BLK), 0, ctx->stream, pw.ix,
                               (const unsigned long long *)P.cinfo, P.nchunks, wpc, rbits, k, tab, d_err, mode);
    }
    tend(ctx);
    ctx->stream = main_stream;
    if (hipGetLastError() != hipSuccess) {
        if (overlap) (void)hipStreamSynchronize(ctx->side_stream);
        return bail(fail(ctx, SMX_DEVICE_ERROR, "k_pm_tab launch failed"));
    }
    bool joined = false;
    const std::function<int()> tab_ready = [&]() -> int {
        if (joined) return 0;
        joined = true;
        unsigned long long hs[2] = {0, 0}, hpal = 0;
        if ((overlap && hipStreamSynchronize(ctx->side_stream) != hipSuccess) || hipMemcpyAsync(hs, stats, 16, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(&hpal, P.pals, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
            return fail(ctx, SMX_DEVICE_ERROR, "node table pass failed: %s", hipGetErrorString(hipGetLastError()));
        hs[1] += hpal;  // palindromic (k+1)-mers: the clean winners' by the dedupe stage, the dirty region's by k_pm_tab_dirty
        if (!clip) {  // (k_pm_tab reads the local links once; with an early clipper they stay until the last renewal of the table: k_pm_isolate_chains follows them, and
                      // option pm_full_retab makes the whole table again from them)
            arena_put(ctx, P.llink);
            P.llink = nullptr;
        }
        if (prof) {
            unsigned long long hp[8];
            if (hipMemcpy(hp, prof, 64, hipMemcpyDeviceToHost) == hipSuccess)
                fprintf(stderr, "[smx] pm_tab: %u chunks (%llu k-mers clean, %llu dirty); 100MHz ticks per chunk: stage %.1f, node table %.1f, chain heads %.1f, jumps %.1f; "
                                "successors outside their chunk %llu, chain heads per chunk %.1f\n", P.nchunks, (unsigned long long)P.nclean, (unsigned long long)P.ndirty,
                        hp[4] ? (double)hp[0] / hp[4] : 0.0, hp[4] ? (double)hp[1] / hp[4] : 0.0, hp[4] ? (double)hp[2] / hp[4] : 0.0, hp[4] ? (double)hp[3] / hp[4] : 0.0,
                        hp[5], hp[4] ? (double)hp[6] / hp[4] : 0.0);
        }
        if ((hs[0] + hs[1]) & 1) return fail(ctx, SMX_DEVICE_ERROR, "odd number of extension bits (%llu + %llu palindromes)", hs[0], hs[1]);
        ctx->g_ext_bits = hs[0];
        ctx->g_ext_pals = hs[1];
        ctx->g_nkpo = (hs[0] + hs[1]) / 2;
        return 0;
    };
    ctx->g_route_stats[0] = 0;
    ctx->g_route_stats[1] = P.nclean;
    ctx->g_route_stats[2] = P.ndirty;
    ctx->g_route_stats[3] = P.nchunks;
    ctx->g_route_stats[6] = P.nslots;
    ctx->g_route_stats[7] = P.nfolded;
    if (!overlap)
        if ((rc = tab_ready())) return bail(rc);
    // with early clippers: the masks as the reads gave them stay beside the array the clippers edit (a local link is a successor only where the
    // k-mer had ONE extension when the links were made), and the node table can be made again from the edited masks
    uint8_t *mask_orig = nullptr;
    unsigned long long *stats2 = nullptr;
    if (clip) {
        if ((rc = dalloc(ctx, &mask_orig, (size_t)D0 + 16))) return bail(rc);
        if ((rc = dalloc(ctx, &stats2, 2))) return bail(rc);
        if (hipMemcpyAsync(mask_orig, ctx->g_mask, (size_t)D0 + 16, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
            return bail(fail(ctx, SMX_DEVICE_ERROR, "copy of the masks failed"));
    }
    const std::function<int(bool)> retab = [&](bool last) -> int {
        if (!ctx->opt_pm_full_retab) {  // the entries of the edited k-mers only (k_pm_retab_changed)
            const uint64_t subcap = std::min<uint64_t>(std::max<uint64_t>(D0 / 8 / PM_RL_SUB, 64), (uint64_t)1 << 20);
            unsigned long long *rlist = nullptr, *rn = nullptr;
            if (int rc2 = dalloc(ctx, &rlist, subcap * PM_RL_SUB)) return rc2;
            if (int rc2 = dalloc(ctx, &rn, (size_t)PM_RL_SUB * 8)) return rc2;
            if (hipMemsetAsync(rn, 0, (size_t)PM_RL_SUB * 64, ctx->stream) != hipSuccess) return fail(ctx, SMX_DEVICE_ERROR, "counter reset failed");
            tbegin(ctx, "pm_retab");
            hipLaunchKernelGGL((k_pm_retab_changed<NW>), dim3(grid_for(D0)), dim3(BLK), 0, ctx->stream, pw.ix, mask_orig, (uint64_t)D0, k, tab, d_err, rlist, rn, subcap);
            hipLaunchKernelGGL((k_pm_retab_lookups<NW>), dim3(grid_for(subcap * PM_RL_SUB)), dim3(BLK), 0, ctx->stream, pw.ix, (const unsigned long long *)rlist,
                               (const unsigned long long *)rn, subcap, k, tab, d_err);
            tend(ctx);
            if (hipGetLastError() != hipSuccess) return fail(ctx, SMX_DEVICE_ERROR, "k_pm_retab_changed launch failed");
            (void)hipStreamSynchronize(ctx->stream);
            for (void *p : {(void *)rlist, (void *)rn}) {
                detach_temp(ctx, p);
                arena_put(ctx, p);
            }
            if (last) {
                if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(ctx, SMX_DEVICE_ERROR, "node table pass failed: %s", hipGetErrorString(hipGetLastError()));
                for (void *p : {(void *)mask_orig, (void *)rbits}) {
                    detach_temp(ctx, p);
                    arena_put(ctx, p);
                }
                mask_orig = nullptr;
                rbits = nullptr;
                arena_put(ctx, P.llink);
                P.llink = nullptr;
            }
            return 0;
        }
        if (hipMemsetAsync(rbits, 0, (size_t)std::max<uint32_t>(P.nchunks, 1) * wpc * 4, ctx->stream) != hipSuccess || hipMemsetAsync(stats2, 0, 16, ctx->stream) != hipSuccess)
            return fail(ctx, SMX_DEVICE_ERROR, "counter reset failed");
        tbegin(ctx, "pm_tab");
        if (P.nchunks)
            hipLaunchKernelGGL(k_pm_tab, dim3(std::min<uint32_t>(P.nchunks, 256 * 16)), dim3(BLK), lds, ctx->stream, (const unsigned long long *)P.cinfo, P.nchunks,
                               maxn, (const uint8_t *)ctx->g_mask, (const uint32_t *)P.llink, tab, jmp, stats2, d_err, (unsigned long long *)nullptr, rbits,
                               (const uint8_t *)mask_orig);
        if (P.ndirty)
            hipLaunchKernelGGL((k_pm_tab_dirty<NW>), dim3(grid_for(P.ndirty)), dim3(BLK), 0, ctx->stream, pw.ix, (uint64_t)P.ndirty, k, tab, jmp, stats2, d_err);
        tend(ctx);
        tbegin(ctx, "pm_remote");
        if (P.nchunks)
            hipLaunchKernelGGL((k_pm_remote<NW>), dim3((unsigned)std::min<uint64_t>((P.nchunks + PMR_CH - 1) / PMR_CH, 256 * 32)), dim3(BLK), 0, ctx->stream, pw.ix,
                               (const unsigned long long *)P.cinfo, P.nchunks, wpc, rbits, k, tab, d_err, 0u);
        tend(ctx);
        if (hipGetLastError() != hipSuccess) return fail(ctx, SMX_DEVICE_ERROR, "k_pm_tab launch failed");
        if (last) {  // 5 B per k-mer of local links and unclipped masks + the remote bits: back to the arena before the walks' arrays are asked for
            if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(ctx, SMX_DEVICE_ERROR, "node table pass failed: %s", hipGetErrorString(hipGetLastError()));
            arena_put(ctx, P.llink);
            P.llink = nullptr;
            for (void *p : {(void *)mask_orig, (void *)rbits}) {
                detach_temp(ctx, p);
                arena_put(ctx, p);
            }
            mask_orig = nullptr;
            rbits = nullptr;
        }
        return 0;
    };
    rc = graph_from_masks<NW>(ctx, k, tab, /*tab_valid=*/true, d_err, gwt, /*present=*/true, &pw, &tab_ready, clip ? &retab : nullptr);
    if (P.llink) {
        (void)hipStreamSynchronize(ctx->stream);
        arena_put(ctx, P.llink);
        P.llink = nullptr;
    }
    if (overlap) (void)hipStreamSynchronize(ctx->side_stream);  // (an early way out of graph_from_masks: nothing of the table may still be in flight when its blocks go)
    if (rc) return bail(rc);
    if ((rc = tab_ready())) return bail(rc);
    ctx->g_pm_clipped = clip && !nx;
    // the count-result view: the k-mer file is made when somebody asks for it (pm_materialize_file)
    ctx->d_result = ctx->d_result_buf = nullptr;
    ctx->n_records = D0;
    ctx->K = k;
    ctx->nw = NW;
    ctx->num_buckets = B;
    ctx->bucket_off.clear();
    ctx->pm_view_pending = true;
    return 0;
}

// The sorted k-mer file + InOutMask bytes of a graph built by pm_route, on demand (smx_copy_final_kmers, smx_bucket_sizes,
// smx_graph_copy_kmers, ... after smx_build_graph): the partition-major records go through the sort pipeline (they are consumed) and
// the split pass of the sorted route. Node ids of the graph keep referring to the partition-major numbering (they are opaque).
template <int NW>
int pm_materialize_file(smx_ctx *ctx, bool for_view) {
    if (!ctx->g_pm) return 0;
    // for_view: the count-result view is waiting for this file (nothing was counted since the graph was built). Otherwise the file is
    // asked for through the graph (smx_graph_copy_kmers) while the view shows a later count: that count's result is kept.
    void *sv_buf = ctx->d_result_buf, *sv_res = ctx->d_result;
    const uint64_t sv_n = ctx->n_records, sv_inst = ctx->n_instances;
    const unsigned sv_nw = ctx->nw, sv_K = ctx->K, sv_B = ctx->num_buckets;
    const std::vector<uint64_t> sv_boff = ctx->bucket_off;
    const bool sv_host = ctx->result_on_host;
    std::vector<smx_ctx::HostChunk> sv_h;
    sv_h.swap(ctx->h_result);
    ctx->result_on_host = false;
    auto restore_view = [&]() {
        ctx->d_result_buf = sv_buf;
        ctx->d_result = sv_res;
        ctx->n_records = sv_n;
        ctx->n_instances = sv_inst;
        ctx->nw = sv_nw;
        ctx->K = sv_K;
        ctx->num_buckets = sv_B;
        ctx->bucket_off = sv_boff;
        ctx->result_on_host = sv_host;
        ctx->h_result.swap(sv_h);
    };
    const unsigned k = ctx->g_k, B = ctx->g_B;
    const uint64_t D0 = ctx->g_nkmers;
    void *src = ctx->g_kmers;
    uint8_t *old_mask = ctx->g_mask;
    const uint64_t nkpo = ctx->g_nkpo;
    const bool nx = ctx->g_pm_nx;
    const bool clipped = ctx->g_pm_clipped;
    if (clipped && !nx) {
        hipLaunchKernelGGL((k_pm_sync_bytes<NW>), dim3(grid_for(D0)), dim3(BLK), 0, ctx->stream, src, (const uint8_t *)old_mask, D0);
        ctx->g_pm_clipped = false;
    }
    ctx->g_kmers = nullptr;
    ctx->g_mask = nullptr;
    ctx->g_pm = false;
    ctx->g_pm_nx = false;
    // (EXT records are consumed by the sort: no lookups on the old numbering from here on. Plain records — nx — are only read: their side arrays
    // stay until the file stands, so that a file that does not fit leaves the graph as it was)
    if (!nx) pm_release(ctx);
    if (!nx) arena_put(ctx, old_mask);
    ctx->d_result = ctx->d_result_buf = nullptr;
    int rc;
    if (!nx) {
        ctx->ext_mode = true;
        rc = run_count<NW>(ctx, k, SMX_MODE_ALL, B, src, D0, nullptr, /*recs_reusable=*/true, false, /*distinct_hint=*/true);
        ctx->ext_mode = false;
        if (rc == 0) {
            if (ctx->d_result_buf != src) arena_put(ctx, src);
            src = nullptr;
            rc = ext_result_to_file<NW>(ctx, k, B, /*whole=*/!clipped);  // (clipped masks need not pair up bit for bit: the (k+1)-mer count is the one the reads gave)
        }
    } else {
        // Plain records, the bytes beside them: a COPY goes through the sort (the old array stays), the file's rank directory is built, and every
        // k-mer of the old array takes its byte to its place in the file (k_nx_file_masks). Costs one more copy of the records than the EXT
        // layout needs; an input for which that does not fit gets the memory-limit code here.
        rc = run_count<NW>(ctx, k, SMX_MODE_ALL, B, src, D0, nullptr, /*recs_reusable=*/false, false, /*distinct_hint=*/true);
        uint8_t *new_mask = nullptr;
        Rec<NW> *file = nullptr;
        smx::RankDir dir{};
        if (rc == 0 && ctx->n_records != D0) rc = fail(ctx, SMX_DEVICE_ERROR, "k-mer file: %llu records after the sort, %llu k-mers in the graph", (unsigned long long)ctx->n_records, (unsigned long long)D0);
        const size_t mask_bytes = (size_t)((D0 + 7) / 8 * 8 + 8);
        void *sorted = rc == 0 ? ctx->d_result_buf : nullptr;
        if (rc == 0) {  // the sort's other buffers go before the file's own block is asked for: old array + sorted copy + file at the peak, not four
            (void)hipStreamSynchronize(ctx->stream);
            free_temps(ctx, sorted);
        }
        if (rc == 0) rc = dalloc(ctx, &file, D0 + 1, false);
        if (rc == 0) rc = dalloc(ctx, &new_mask, mask_bytes, false);
        uint32_t *d_e = nullptr;
        if (rc == 0) rc = dalloc(ctx, &d_e, 1);
        if (rc == 0) {
            const std::vector<uint64_t> boff = ctx->bucket_off;
            hipError_t e = hipMemcpyAsync(file, ctx->d_result_buf, (size_t)D0 * sizeof(Rec<NW>), hipMemcpyDeviceToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemsetAsync(new_mask, 0, mask_bytes, ctx->stream);
            if (e == hipSuccess) e = hipMemsetAsync(d_e, 0, 4, ctx->stream);
            if (e != hipSuccess) rc = fail(ctx, SMX_DEVICE_ERROR, "k-mer file of a plain-record graph: %s", hipGetErrorString(e));
            if (rc == 0) rc = build_rank_dir<NW>(ctx, file, D0, boff, B, k, dir);
            if (rc == 0) {
                hipLaunchKernelGGL((k_nx_file_masks<NW>), dim3(grid_for(D0)), dim3(BLK), 0, ctx->stream, (const void *)src, (const uint8_t *)old_mask, D0, (const void *)file, dir,
                                   new_mask, d_e);
                uint32_t he = 0;
                e = hipGetLastError();
                if (e == hipSuccess) e = hipMemcpyAsync(&he, d_e, 4, hipMemcpyDeviceToHost, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess) rc = fail(ctx, SMX_DEVICE_ERROR, "k-mer file of a plain-record graph: %s", hipGetErrorString(e));
                else if (he) rc = fail(ctx, SMX_DEVICE_ERROR, "k-mer file of a plain-record graph: %u k-mers are missing from their own sorted file", he);
            }
            drop_rank_dir(ctx, dir);
            if (rc == 0) {
                ctx->g_kmers = file;
                ctx->g_mask = new_mask;
                file = nullptr;
                new_mask = nullptr;
                ctx->g_kboff = boff;
                ctx->g_nkmers = D0;
                ctx->n_records = D0;
                ctx->bucket_off = boff;
                ctx->K = k;
                ctx->nw = NW;
                ctx->num_buckets = B;
            }
        }
        arena_put(ctx, file);
        arena_put(ctx, new_mask);
        if (sorted) arena_put(ctx, sorted);  // the sorted copy (kept out of the temp list above)
        ctx->d_result_buf = ctx->d_result = nullptr;
        if (rc) {
            // nothing of the graph was touched: the partition-major records, their bytes and the side arrays of the route are all there — the
            // caller hears the error (normally the memory limit) and the graph stays what it was
            (void)hipStreamSynchronize(ctx->stream);
            for (auto &t : ctx->timings) {
                (void)hipEventDestroy(t.e0);
                (void)hipEventDestroy(t.e1);
            }
            ctx->timings.clear();
            free_temps(ctx);
            ctx->g_kmers = src;
            ctx->g_mask = old_mask;
            ctx->g_pm = true;
            ctx->g_pm_nx = true;
            ctx->g_nkpo = nkpo;
            restore_view();
            return rc;
        }
        arena_put(ctx, old_mask);
        pm_release(ctx);
    }
    if (src) arena_put(ctx, src);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &t : ctx->timings) {
        (void)hipEventDestroy(t.e0);
        (void)hipEventDestroy(t.e1);
    }
    ctx->timings.clear();
    free_temps(ctx);
    ctx->pm_view_pending = false;
    if (rc) {
        if (ctx->g_kmers) arena_put(ctx, ctx->g_kmers);
        if (ctx->g_mask) arena_put(ctx, ctx->g_mask);
        ctx->g_kmers = nullptr;
        ctx->g_mask = nullptr;
        ctx->d_result = ctx->d_result_buf = nullptr;
        if (!for_view) restore_view();
        else {
            ctx->n_records = 0;
            ctx->bucket_off.assign((size_t)ctx->num_buckets + 1, 0);
        }
        // the partition-major records were consumed and no file took their place: the graph has no k-mers to look up or copy any
        // more — it is not a graph any longer (callers see "no graph" instead of a copy from a null block)
        clear_graph(ctx);
        return rc;
    }
    ctx->g_nkpo = nkpo;
    if (for_view) {
        ctx->d_result = ctx->g_kmers;
        ctx->d_result_buf = nullptr;
    } else {
        restore_view();
    }
    return 0;
}
// view: the caller reads the count-result view (smx_copy_final_kmers, smx_bucket_sizes, ...): only a view that still stands for the
// graph's file needs it made; else (smx_graph_copy_kmers & co.) whenever the graph has none yet
int ensure_kmer_file(smx_ctx *ctx, bool view = true) {
    if (!ctx || !ctx->g_pm) return 0;
    if (view && !ctx->pm_view_pending) return 0;
    const bool for_view = ctx->pm_view_pending;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, SMX_DEVICE_ERROR, "hipSetDevice failed");
    switch (ctx->g_nw) {
        case 1: return pm_materialize_file<1>(ctx, for_view);
        case 2: return pm_materialize_file<2>(ctx, for_view);
        case 3: return pm_materialize_file<3>(ctx, for_view);
        default: return pm_materialize_file<4>(ctx, for_view);
    }
}
