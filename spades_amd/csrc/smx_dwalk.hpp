// spades_amd/csrc/smx_dwalk.hpp — host side of the distributed unitig walks (kernels in smx_dwalk.hip).
// Included by smx_api.hip after smx_construct.hpp. The exchanges themselves are the caller's (spades_amd/dist.py over RCCL, or any
// MPI_Alltoallv): the library never touches a communicator.
#pragma once

// The shard must be what smx_graph_shard_from_ext / smx_graph_shard_build left: sorted k-mers of this rank's bucket range + masks.
// Builds the rank directory of the shard (the lookups of the other ranks go through it) and lists the start de-edges.
template <int NW>
int dw_prepare(smx_ctx *ctx) {
    if (ctx->dw_ready) return 0;
    if (ctx->g_ready || ctx->g_pm) return fail(ctx, SMX_INVALID_PARAMETER, "the context holds a whole graph, not a shard of the k-mer file");
    const uint64_t D0 = ctx->g_nkmers;
    ctx->dw_ncand = ctx->dw_nchain = 0;
    if (D0 == 0) {
        ctx->dw_ready = true;
        return 0;
    }
    if (!ctx->g_kmers || !ctx->g_mask) return fail(ctx, SMX_INVALID_PARAMETER, "no shard of the k-mer file in this context");
    if (!ctx->g_dir_kmers.dir)
        if (int rc = build_rank_dir<NW>(ctx, ctx->g_kmers, D0, ctx->g_kboff, ctx->g_B, ctx->g_k, ctx->g_dir_kmers)) return rc;
    const uint64_t ntiles = (D0 + CAND_TILE - 1) / CAND_TILE;
    unsigned long long *tcnt, *toff, *nj;
    if (int rc = dalloc(ctx, &tcnt, ntiles)) return rc;
    if (int rc = dalloc(ctx, &toff, ntiles + 1)) return rc;
    if (int rc = dalloc(ctx, &nj, CAND_NJ)) return rc;
    HIPCHK(hipMemsetAsync(nj, 0, CAND_NJ * 8, ctx->stream));
    hipLaunchKernelGGL(k_cand_tiles, dim3((unsigned)ntiles), dim3(BLK), 0, ctx->stream, (const uint8_t *)ctx->g_mask, D0, tcnt, nj, (unsigned long long *)nullptr);
    HIPCHK(hipGetLastError());
    if (int rc = scan_u64(ctx, tcnt, toff, ntiles)) return rc;
    unsigned long long C = 0, h_nj[CAND_NJ];
    HIPCHK(hipMemcpyAsync(&C, toff + ntiles, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(h_nj, nj, CAND_NJ * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    uint64_t n_junction = 0;
    for (int i = 0; i < CAND_NJ; ++i) n_junction += h_nj[i];
    if (int rc = dalloc(ctx, &ctx->dw_cand, (size_t)C + 1, false)) return rc;
    if (C) {
        hipLaunchKernelGGL(k_cand_expand, dim3((unsigned)ntiles), dim3(BLK), 0, ctx->stream, (const uint8_t *)ctx->g_mask, (const unsigned long long *)toff, D0,
                           ctx->dw_cand);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->dw_ncand = C;
    ctx->dw_nchain = 2 * (D0 - n_junction);
    ctx->dw_ready = true;
    return 0;
}

// requests of the chain k-mers (cand = false) or of the start de-edges (cand = true), grouped by owner rank
template <int NW>
int dw_requests(smx_ctx *ctx, bool cand, unsigned world, void *d_recs, unsigned long long *d_tags, uint64_t *counts, uint64_t item0 = 0,
                uint64_t item_n = ~0ull) {
    if (int rc = dw_prepare<NW>(ctx)) return rc;
    for (unsigned i = 0; i < world; ++i) counts[i] = 0;
    const uint64_t all_items = cand ? ctx->dw_ncand : 2 * ctx->g_nkmers;
    const bool whole = item0 == 0 && item_n >= all_items;  // (a range: the caller sized its buffers for one request per item)
    item0 = std::min(item0, all_items);
    const uint64_t n_items = std::min(item_n, all_items - item0);
    const uint64_t expect = cand ? ctx->dw_ncand : ctx->dw_nchain;
    if (expect == 0 || n_items == 0) return 0;
    if (!d_recs || !d_tags) return fail(ctx, SMX_INVALID_PARAMETER, "null request buffers");
    unsigned long long *hist, *off, *cur;
    if (int rc = dalloc(ctx, &hist, world)) return rc;
    if (int rc = dalloc(ctx, &off, world + 1)) return rc;
    if (int rc = dalloc(ctx, &cur, world)) return rc;
    HIPCHK(hipMemsetAsync(hist, 0, (size_t)world * 8, ctx->stream));
    const size_t lds = (size_t)world * 16;
    const unsigned grid = grid_for(n_items, 4096);
    const unsigned k = ctx->g_k, B = ctx->g_B;
    if (cand)
        hipLaunchKernelGGL((k_dw_requests<NW, true, 0>), dim3(grid), dim3(BLK), lds, ctx->stream, (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask,
                           (const unsigned long long *)ctx->dw_cand, item0, n_items, k, B, world, hist, (void *)nullptr, (unsigned long long *)nullptr);
    else
        hipLaunchKernelGGL((k_dw_requests<NW, false, 0>), dim3(grid), dim3(BLK), lds, ctx->stream, (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask,
                           (const unsigned long long *)nullptr, item0, n_items, k, B, world, hist, (void *)nullptr, (unsigned long long *)nullptr);
    HIPCHK(hipGetLastError());
    if (int rc = scan_u64(ctx, hist, off, world)) return rc;
    HIPCHK(hipMemcpyAsync(cur, off, (size_t)world * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (cand)
        hipLaunchKernelGGL((k_dw_requests<NW, true, 1>), dim3(grid), dim3(BLK), lds, ctx->stream, (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask,
                           (const unsigned long long *)ctx->dw_cand, item0, n_items, k, B, world, cur, d_recs, d_tags);
    else
        hipLaunchKernelGGL((k_dw_requests<NW, false, 1>), dim3(grid), dim3(BLK), lds, ctx->stream, (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask,
                           (const unsigned long long *)nullptr, item0, n_items, k, B, world, cur, d_recs, d_tags);
    HIPCHK(hipGetLastError());
    std::vector<unsigned long long> h(world);
    HIPCHK(hipMemcpyAsync(h.data(), hist, (size_t)world * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    uint64_t tot = 0;
    for (unsigned i = 0; i < world; ++i) tot += (counts[i] = h[i]);
    if (whole && tot != expect) return fail(ctx, SMX_DEVICE_ERROR, "%llu requests placed, %llu expected", (unsigned long long)tot, (unsigned long long)expect);
    return 0;
}

template <int NW>
int dw_lookup(smx_ctx *ctx, const void *d_recs, uint64_t n, unsigned long long *d_reply) {
    if (int rc = dw_prepare<NW>(ctx)) return rc;
    if (n == 0) return 0;
    if (ctx->g_nkmers == 0) {  // an empty shard owns nothing: whoever asks it has an inconsistent ownership map
        HIPCHK(hipMemsetAsync(d_reply, 0xFF, n * 8, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return 0;
    }
    hipLaunchKernelGGL((k_dw_lookup<NW>), dim3(grid_for(n)), dim3(BLK), 0, ctx->stream, d_recs, n, (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask,
                       ctx->g_dir_kmers, d_reply);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

// the unitigs of this rank's start de-edges: steps[i] chain k-mers behind de-edge i, their outgoing nucleotides at
// bases[boff[i] .. boff[i] + steps[i]), last[i] = global node the walk ends at. Kept ones stay in the context (edge arrays of a graph
// that is not ready) until smx_shard_unitigs_copy / the next shard.
template <int NW>
int dw_unitigs(smx_ctx *ctx, uint64_t first_rank, const unsigned long long *d_steps, const unsigned long long *d_last, const unsigned long long *d_boff,
               const uint8_t *d_bases, uint64_t *n_kept, uint64_t *n_words) {
    if (int rc = dw_prepare<NW>(ctx)) return rc;
    drop_device_graph(ctx);
    const uint64_t C = ctx->dw_ncand;
    const unsigned k = ctx->g_k;
    uint64_t nkept = 0, tw = 0;
    if (C) {
        if (!d_steps || !d_last || !d_boff) return fail(ctx, SMX_INVALID_PARAMETER, "null chain arrays");
        unsigned long long *kw, *one;
        uint8_t *flags;
        if (int rc = dalloc(ctx, &kw, C + 1)) return rc;
        if (int rc = dalloc(ctx, &one, C + 1)) return rc;
        if (int rc = dalloc(ctx, &flags, C)) return rc;
        const unsigned cgrid = grid_for(C);
        hipLaunchKernelGGL((k_dw_keep<NW>), dim3(cgrid), dim3(BLK), 0, ctx->stream, (const unsigned long long *)ctx->dw_cand, C, (const void *)ctx->g_kmers, k, d_steps,
                           d_boff, d_bases, flags, kw, one);
        HIPCHK(hipGetLastError());
        if (int rc = scan_u64(ctx, kw, kw, C)) return rc;
        if (int rc = scan_u64(ctx, one, one, C)) return rc;
        unsigned long long htw = 0, hnk = 0;
        HIPCHK(hipMemcpyAsync(&htw, kw + C, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(&hnk, one + C, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        nkept = hnk;
        tw = htw;
        if (int rc = dalloc(ctx, &ctx->g_uwords, tw + 8, false)) return rc;
        if (int rc = dalloc(ctx, &ctx->g_eoffw, nkept + 1, false)) return rc;
        if (int rc = dalloc(ctx, &ctx->g_elen, nkept + 1, false)) return rc;
        if (int rc = dalloc(ctx, &ctx->g_estart, nkept + 1, false)) return rc;
        if (int rc = dalloc(ctx, &ctx->g_eend, nkept + 1, false)) return rc;
        if (int rc = dalloc(ctx, &ctx->g_eself, nkept + 1, false)) return rc;
        HIPCHK(hipMemsetAsync(ctx->g_uwords + tw, 0, 64, ctx->stream));
        hipLaunchKernelGGL((k_dw_write<NW>), dim3(cgrid), dim3(BLK), 0, ctx->stream, (const unsigned long long *)ctx->dw_cand, C, (const void *)ctx->g_kmers, k, d_steps,
                           d_last, d_boff, d_bases, (const uint8_t *)flags, (const unsigned long long *)kw, (const unsigned long long *)one,
                           (unsigned long long)(2 * first_rank), ctx->g_uwords, ctx->g_eoffw, ctx->g_elen, ctx->g_estart, ctx->g_eend, ctx->g_eself);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    ctx->g_ne = ctx->g_npaths = nkept;
    ctx->g_nuwords = tw;
    *n_kept = nkept;
    *n_words = tw;
    return 0;
}

// The graph from unitigs that were walked elsewhere (every rank's kept unitigs, concatenated in rank order = k-mer-file order of their
// start k-mers) + the k-mers left on perfect loops (k-mer-file order, global ranks): link records and vertices as in
// graph_from_masks step 5. No k-mer file and no masks on this rank afterwards.
template <int NW>
int graph_from_unitigs(smx_ctx *ctx, unsigned k, unsigned B, uint64_t n_kmers_total, uint64_t n_kpomers, const uint64_t *d_words, uint64_t n_words,
                       const unsigned long long *d_elen, const unsigned long long *d_estart, const unsigned long long *d_eend, const uint8_t *d_eself, uint64_t ne,
                       const uint64_t *loop_ranks, const uint64_t *loop_kmers, const uint8_t *loop_masks, uint64_t n_loop_kmers) {
    clear_graph(ctx);
    WallTrace gwt;
    ctx->g_k = k;
    ctx->g_nw = NW;
    ctx->g_B = B;
    ctx->gh.k = k;
    ctx->gh.eoff.assign(1, 0);
    ctx->g_kboff.assign(B + 1, 0);
    ctx->g_sharded_file = true;
    if (int rc = dalloc(ctx, &ctx->g_uwords, n_words + 8, false)) return rc;
    if (int rc = dalloc(ctx, &ctx->g_eoffw, ne + 1, false)) return rc;
    if (int rc = dalloc(ctx, &ctx->g_elen, ne + 1, false)) return rc;
    if (int rc = dalloc(ctx, &ctx->g_estart, ne + 1, false)) return rc;
    if (int rc = dalloc(ctx, &ctx->g_eend, ne + 1, false)) return rc;
    if (int rc = dalloc(ctx, &ctx->g_eself, ne + 1, false)) return rc;
    HIPCHK(hipMemsetAsync(ctx->g_uwords + n_words, 0, 64, ctx->stream));
    if (ne) {
        HIPCHK(hipMemcpyAsync(ctx->g_uwords, d_words, n_words * 8, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->g_elen, d_elen, ne * 8, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->g_estart, d_estart, ne * 8, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->g_eend, d_eend, ne * 8, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(ctx->g_eself, d_eself, ne, hipMemcpyDeviceToDevice, ctx->stream));
        unsigned long long *w;
        if (int rc = dalloc(ctx, &w, ne + 1)) return rc;
        hipLaunchKernelGGL(k_dw_words_of, dim3(grid_for(ne)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)ctx->g_elen, ne, w);
        HIPCHK(hipGetLastError());
        if (int rc = scan_u64(ctx, w, w, ne)) return rc;
        HIPCHK(hipMemcpyAsync(ctx->g_eoffw, w, ne * 8, hipMemcpyDeviceToDevice, ctx->stream));
        unsigned long long tw = 0;
        HIPCHK(hipMemcpyAsync(&tw, w + ne, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (tw != n_words) return fail(ctx, SMX_INVALID_PARAMETER, "the unitig lengths need %llu words, %llu given", tw, (unsigned long long)n_words);
    }
    ctx->g_ne = ctx->g_npaths = ne;
    ctx->g_nuwords = n_words;
    ctx->g_nkmers = n_kmers_total;
    ctx->g_nkpo = n_kpomers;
    if (n_loop_kmers && ctx->opt_keep_loops) {
        for (uint64_t t = 1; t < n_loop_kmers; ++t)
            if (loop_ranks[t] <= loop_ranks[t - 1]) return fail(ctx, SMX_INVALID_PARAMETER, "loop k-mers must come in k-mer-file order");
        if (int rc = append_loops(ctx, k, loop_kmers, loop_ranks, loop_masks, n_loop_kmers, ne, n_words)) return rc;
    }
    free_temps(ctx);
    ctx->g_dev_valid = true;
    {
        unsigned long long *one1;
        const uint64_t n2 = ctx->g_ne;
        if (int rc = dalloc(ctx, &one1, n2 + 1)) return rc;
        ctx->g_nbases = 0;
        if (n2) {
            if (int rc = scan_u64(ctx, ctx->g_elen, one1, n2)) return rc;
            unsigned long long nb = 0;
            HIPCHK(hipMemcpyAsync(&nb, one1 + n2, 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            ctx->g_nbases = nb;
        }
        free_temps(ctx);
    }
    const bool host_links = ctx->opt_sort_edges || ctx->g_ne == 0 || ctx->opt_device_links == 0 || (ctx->g_ne < (1u << 16) && ctx->opt_device_links < 2);
    if (host_links) {
        if (int rc = materialize_host(ctx)) return rc;
        if (ctx->opt_sort_edges) {
            smxh::sort_edges_raw(ctx->gh);
            if (int rc = upload_graph(ctx)) return rc;
        }
        int sort_rc = 0;
        smxh::build_links(ctx->gh, [&](std::vector<uint64_t> &keys) {
            if (!sort_rc) sort_rc = device_sort_u64(ctx, keys);
            if (sort_rc) smxh::radix_sort_u64(keys);
        });
        if (sort_rc) return sort_rc;
    } else {
        ctx->tprefix = "links:";
        const int rc = device_build_links(ctx, n_kmers_total);
        ctx->tprefix.clear();
        if (rc) return rc;
        free_temps(ctx);
    }
    ctx->g_ready = true;
    return 0;
}
