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

// Host side of the two-pass grouping of smx_dwalk.hip: launch(0, bc) counts per (owner, workgroup), the scan of bc (owner-major) is the place of every
// (owner, workgroup) in the send order, launch(1, off) places. counts[p] = what goes to rank p. `grid` must be the grid of both launches.
template <class Launch>
int dw_two_pass(smx_ctx *ctx, unsigned world, unsigned grid, Launch &&launch, uint64_t *counts, unsigned long long **d_off) {
    unsigned long long *bc, *off;
    const size_t n = (size_t)world * grid;
    if (int rc = dalloc(ctx, &bc, n)) return rc;
    if (int rc = dalloc(ctx, &off, n + 1)) return rc;
    launch(0, bc);
    HIPCHK(hipGetLastError());
    if (int rc = scan_u64(ctx, bc, off, n)) return rc;
    std::vector<unsigned long long> h(world + 1);
    for (unsigned p = 0; p <= world; ++p) HIPCHK(hipMemcpyAsync(&h[p], off + (size_t)p * grid, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (unsigned p = 0; p < world; ++p) counts[p] = h[p + 1] - h[p];
    *d_off = off;
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
    const size_t lds = (size_t)world * 8;
    const unsigned grid = grid_for(n_items, 4096);
    const unsigned k = ctx->g_k, B = ctx->g_B;
    auto launch = [&](int pass, unsigned long long *bc_or_off) {
        void *o = pass ? d_recs : nullptr;
        unsigned long long *t = pass ? d_tags : nullptr;
        const unsigned long long *cd = cand ? (const unsigned long long *)ctx->dw_cand : (const unsigned long long *)nullptr;
        if (cand && pass == 0) hipLaunchKernelGGL((k_dw_requests<NW, true, 0>), dim3(grid), dim3(BLK), lds, ctx->stream, (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask, cd, item0, n_items, k, B, world, bc_or_off, o, t);
        else if (cand) hipLaunchKernelGGL((k_dw_requests<NW, true, 1>), dim3(grid), dim3(BLK), lds, ctx->stream, (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask, cd, item0, n_items, k, B, world, bc_or_off, o, t);
        else if (pass == 0) hipLaunchKernelGGL((k_dw_requests<NW, false, 0>), dim3(grid), dim3(BLK), lds, ctx->stream, (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask, cd, item0, n_items, k, B, world, bc_or_off, o, t);
        else hipLaunchKernelGGL((k_dw_requests<NW, false, 1>), dim3(grid), dim3(BLK), lds, ctx->stream, (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask, cd, item0, n_items, k, B, world, bc_or_off, o, t);
    };
    unsigned long long *d_off = nullptr;
    if (int rc = dw_two_pass(ctx, world, grid, launch, counts, &d_off)) return rc;
    launch(1, d_off);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    uint64_t tot = 0;
    for (unsigned i = 0; i < world; ++i) tot += counts[i];
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

// ---- the whole of the distributed walks behind ONE entry point (smx_shard_walks, round 6) ------------------------------------------------
// Everything between "the owner-side shard is built" and "this rank's kept unitigs are in the context": lookups by exchange, pointer
// doubling, chain nucleotides to the heads, the chains of the start de-edges, assembly — the steps of the header comment of smx_dwalk.hip,
// all on the device (kernels there), with the CALLER's collectives between them (include/smx.h: smx_collectives). The library still never
// touches a communicator: the C++ host hands in grouped ncclSend / ncclRecv (tools/gbuilder_mgpu.hpp), dist.py torch.distributed.
// Failure protocol: a rank whose local step fails goes on to the NEXT collective and poisons it (counts of all ones; 2^62 added to a sum),
// so every rank leaves at the same collective — nobody waits for a rank that is gone (dist.py's _guarded did this with one extra
// all-reduce per step).
constexpr uint64_t DW_POISON = ~0ull;
struct DwKeep {  // long-lived blocks of one run of the walks: returned on every way out
    smx_ctx *ctx;
    std::vector<void *> v;
    ~DwKeep() {
        for (void *p : v) arena_put(ctx, p);
    }
    template <typename T>
    int get(T **p, size_t n) {
        if (int rc = dalloc(ctx, p, n, false)) return rc;
        v.push_back(*p);
        return 0;
    }
    void drop(void *p) {
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i] == p) {
                v.erase(v.begin() + i);
                arena_put(ctx, p);
                return;
            }
    }
};

template <int NW>
int dw_walks(smx_ctx *ctx, const uint64_t *kmers_per_rank, const smx_collectives *co, uint64_t *info) {
    const unsigned world = co->world, rank = co->rank;
    if (world < 1 || world > DW_MAX_WORLD || rank >= world) return fail(ctx, SMX_INVALID_PARAMETER, "distributed walks: world %u (1..%u), rank %u", world, DW_MAX_WORLD, rank);
    if (!co->exchange_counts || !co->alltoallv || !co->allreduce_u64) return fail(ctx, SMX_INVALID_PARAMETER, "distributed walks: a collective is missing");
    const bool dbg = getenv("SMX_DEBUG") != nullptr && rank == 0;
    double t_last = wall_now();
    auto mark = [&](const char *what) {
        if (!dbg) return;
        (void)hipStreamSynchronize(ctx->stream);
        const double now = wall_now();
        fprintf(stderr, "[smx] walks: %-34s %8.1f ms\n", what, (now - t_last) * 1e3);
        t_last = now;
    };
    int bad = 0;  // the first local failure; carried into the next collective
#define DW_LOCAL(expr)            \
    do {                          \
        if (!bad) bad = (expr);   \
    } while (0)
#define DW_HIP(call)                                                                                              \
    do {                                                                                                          \
        if (!bad) {                                                                                               \
            hipError_t e_ = (call);                                                                               \
            if (e_ != hipSuccess)                                                                                 \
                bad = fail(ctx, e_ == hipErrorOutOfMemory ? SMX_MEMORY_LIMIT_EXCEEDED : SMX_DEVICE_ERROR, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
        }                                                                                                         \
    } while (0)
    // test hook (option "walk_fail_at" = phase, set on ONE rank): that rank's local step fails there with the memory-limit code — what the
    // failure protocol is for; the tests check that every rank comes back and none waits
#define DW_HOOK(phase) DW_LOCAL(ctx->opt_walk_fail_at == (phase) ? fail(ctx, SMX_MEMORY_LIMIT_EXCEEDED, "test hook: this rank fails in phase %d of the distributed walks", (phase)) : 0)
    auto peer_failed = [&]() { return fail(ctx, SMX_DEVICE_ERROR, "distributed walks: another rank failed (its own message says why); all ranks leave together"); };
    // collectives with the failure protocol. `bad` set: poison goes out and the local code comes back.
    auto xcounts = [&](const std::vector<uint64_t> &send, std::vector<uint64_t> &recv) -> int {
        std::vector<uint64_t> s(world);
        for (unsigned p = 0; p < world; ++p) s[p] = bad ? DW_POISON : send[p];
        recv.assign(world, 0);
        (void)hipStreamSynchronize(ctx->stream);
        if (co->exchange_counts(co->user, s.data(), recv.data())) return bad ? bad : fail(ctx, SMX_DEVICE_ERROR, "distributed walks: the exchange of counts failed");
        if (bad) return bad;
        for (unsigned p = 0; p < world; ++p)
            if (recv[p] == DW_POISON) return peer_failed();
        return 0;
    };
    auto a2a = [&](const void *d_send, const std::vector<uint64_t> &sc, void *d_recv, const std::vector<uint64_t> &rc_, unsigned unit) -> int {
        (void)hipStreamSynchronize(ctx->stream);  // (the caller's streams know nothing of the library's)
        if (co->alltoallv(co->user, d_send, sc.data(), d_recv, rc_.data(), unit)) return fail(ctx, SMX_DEVICE_ERROR, "distributed walks: an all-to-all failed");
        return 0;
    };
    auto allred = [&](uint64_t *vals, unsigned n, int op) -> int {  // op 0 sum (values < 2^62), 1 max
        std::vector<uint64_t> v(vals, vals + n);
        v.push_back(bad ? 1 : 0);  // (an extra slot: any rank's failure reaches everybody under either operation)
        if (co->allreduce_u64(co->user, v.data(), n + 1, op)) return bad ? bad : fail(ctx, SMX_DEVICE_ERROR, "distributed walks: an all-reduce failed");
        if (bad) return bad;
        if (v[n]) return peer_failed();
        for (unsigned i = 0; i < n; ++i) vals[i] = v[i];
        return 0;
    };

    // ---- geometry -----------------------------------------------------------------------------------------------------------------
    std::vector<uint64_t> first(world + 1, 0);
    for (unsigned p = 0; p < world; ++p) first[p + 1] = first[p] + kmers_per_rank[p];
    const uint64_t n_mine = kmers_per_rank[rank], n2 = 2 * n_mine;
    const unsigned long long my_base = 2 * first[rank];
    DwBits B{};
    B.hb = (unsigned)std::min<int64_t>(std::max<int64_t>(ctx->opt_walk_hop_bits, 2), 30);
    B.idm = (1ull << (62 - B.hb)) - 1;
    B.hm = (1ull << B.hb) - 1;
    DwOwners OW{};
    OW.world = world;
    for (unsigned p = 0; p < world; ++p) OW.bound[p] = 2 * first[p + 1];
    DW_LOCAL(2 * first[world] > B.idm ? fail(ctx, SMX_INVALID_PARAMETER, "%llu k-mers: node ids beyond %u bits", (unsigned long long)first[world], 62 - B.hb) : 0);
    DW_LOCAL(n_mine != ctx->g_nkmers ? fail(ctx, SMX_INVALID_PARAMETER, "this rank's shard holds %llu k-mers, kmers_per_rank says %llu", (unsigned long long)ctx->g_nkmers,
                                            (unsigned long long)n_mine) : 0);
    DW_LOCAL(dw_prepare<NW>(ctx));
    const uint64_t n_cand = bad ? 0 : ctx->dw_ncand;
    const uint64_t CH = (uint64_t)std::max<int64_t>(ctx->opt_walk_chunk > 0 ? ctx->opt_walk_chunk : ((int64_t)1 << 27), 2) & ~1ull;  // nodes per round (whole k-mers)
    const uint64_t SCH = (uint64_t)std::max<int64_t>(ctx->opt_walk_start_chunk > 0 ? ctx->opt_walk_start_chunk : ((int64_t)1 << 22), 1);
    uint64_t rounds3[4] = {(n2 + CH - 1) / CH, (n_cand + CH - 1) / CH, (n_cand + SCH - 1) / SCH, n2};
    if (int rc = allred(rounds3, 4, 1)) return rc;
    const uint64_t node_rounds = rounds3[0], cand_rounds = rounds3[1], start_rounds = rounds3[2], max_n2 = rounds3[3];
    const size_t lds = (size_t)world * 8;

    DwKeep keep{ctx, {}};
    unsigned long long *word = nullptr, *c_first = nullptr;
    uint8_t *flag = nullptr, *c_fj = nullptr;
    uint32_t *d_err = nullptr;          // [0] failed lookups / bad indices, [1] chains too long
    unsigned long long *d_stats = nullptr;  // [0] open nodes, [1] placed, [2] bad heads, [3] unset ends
    DW_LOCAL(keep.get(&word, std::max<uint64_t>(n2, 1)));
    DW_LOCAL(keep.get(&flag, std::max<uint64_t>(n2, 1) + 8));
    DW_LOCAL(keep.get(&c_first, std::max<uint64_t>(n_cand, 1)));
    DW_LOCAL(keep.get(&c_fj, std::max<uint64_t>(n_cand, 1) + 8));
    DW_LOCAL(keep.get(&d_err, 4));
    DW_LOCAL(keep.get(&d_stats, 8));
    DW_HIP(hipMemsetAsync(word, 0, std::max<uint64_t>(n2, 1) * 8, ctx->stream));
    DW_HIP(hipMemsetAsync(flag, 0, std::max<uint64_t>(n2, 1) + 8, ctx->stream));
    DW_HIP(hipMemsetAsync(c_fj, 0, std::max<uint64_t>(n_cand, 1) + 8, ctx->stream));
    DW_HIP(hipMemsetAsync(d_err, 0, 16, ctx->stream));
    DW_HIP(hipMemsetAsync(d_stats, 0, 64, ctx->stream));
    auto read_err = [&](unsigned which, uint32_t *out) -> int {
        uint32_t h[4] = {0, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(h, d_err, 16, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        *out = h[which];
        return 0;
    };
    auto segs_of = [&](const std::vector<uint64_t> &counts, bool add_first) {
        DwSegs sg{};
        sg.world = world;
        sg.off[0] = 0;
        for (unsigned p = 0; p < world; ++p) {
            sg.off[p + 1] = sg.off[p] + counts[p];
            sg.add[p] = add_first ? first[p] : 0;
        }
        return sg;
    };
    auto sum = [](const std::vector<uint64_t> &v) {
        uint64_t t = 0;
        for (uint64_t x : v) t += x;
        return t;
    };

    // ---- 1. successors of the chain k-mers, first nodes of the start de-edges: one lookup exchange per range ------------------------
    DW_HOOK(1);
    for (int cand = 0; cand < 2; ++cand) {
        const uint64_t n_items_all = cand ? n_cand : n2, n_rounds = cand ? cand_rounds : node_rounds;
        for (uint64_t c = 0; c < n_rounds; ++c) {
            const uint64_t a = std::min(c * CH, n_items_all), n_it = std::min(CH, n_items_all - a);
            std::vector<uint64_t> counts(world, 0), rcounts;
            Rec<NW> *recs = nullptr, *rrecs = nullptr;
            unsigned long long *tags = nullptr, *reply = nullptr, *back = nullptr;
            DW_LOCAL(dalloc(ctx, &recs, std::max<uint64_t>(n_it, 1)));
            DW_LOCAL(dalloc(ctx, &tags, std::max<uint64_t>(n_it, 1)));
            DW_LOCAL(dw_requests<NW>(ctx, cand != 0, world, recs, tags, counts.data(), a, n_it));
            if (int rc = xcounts(counts, rcounts)) return rc;
            const uint64_t n_send = sum(counts), n_recv = sum(rcounts);
            DW_LOCAL(dalloc(ctx, &rrecs, std::max<uint64_t>(n_recv, 1)));
            DW_LOCAL(dalloc(ctx, &reply, std::max<uint64_t>(n_recv, 1)));
            DW_LOCAL(dalloc(ctx, &back, std::max<uint64_t>(n_send, 1)));
            // (a rank that failed its allocations cannot take part in the all-to-all: it says so in a second round of counts)
            std::vector<uint64_t> ok_s(world, 0), ok_r;
            if (int rc = xcounts(ok_s, ok_r)) return rc;
            if (int rc = a2a(recs, counts, rrecs, rcounts, (unsigned)sizeof(Rec<NW>))) return rc;
            DW_LOCAL(dw_lookup<NW>(ctx, rrecs, n_recv, reply));
            if (int rc = xcounts(ok_s, ok_r)) return rc;
            if (int rc = a2a(reply, rcounts, back, counts, 8)) return rc;
            if (n_send) {
                const DwSegs sg = segs_of(counts, true);
                if (cand)
                    hipLaunchKernelGGL((k_dw_apply_succ<true>), dim3(grid_for(n_send)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)tags,
                                       (const unsigned long long *)back, n_send, sg, B, word, flag, c_first, c_fj, d_err);
                else
                    hipLaunchKernelGGL((k_dw_apply_succ<false>), dim3(grid_for(n_send)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)tags,
                                       (const unsigned long long *)back, n_send, sg, B, word, flag, c_first, c_fj, d_err);
                DW_HIP(hipGetLastError());
            }
            DW_HIP(hipStreamSynchronize(ctx->stream));
            free_temps(ctx);
        }
    }
    {
        uint32_t he = 0;
        DW_LOCAL(read_err(0, &he));
        DW_LOCAL(he ? fail(ctx, SMX_DEVICE_ERROR, "%u successor k-mers are in no shard: the k-mer file and the masks disagree", he) : 0);
    }
    mark("successor lookups");

    // ---- 2. pointer doubling over the chains ----------------------------------------------------------------------------------------
    uint64_t prev_open = ~0ull, rounds = 0;
    for (;;) {
        unsigned long long h_open = 0;
        DW_HIP(hipMemsetAsync(d_stats, 0, 8, ctx->stream));
        if (!bad && n2) hipLaunchKernelGGL(k_dw_count_open, dim3(grid_for(n2, 4096)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)word, (const uint8_t *)flag, n2, d_stats);
        DW_HIP(hipMemcpyAsync(&h_open, d_stats, 8, hipMemcpyDeviceToHost, ctx->stream));
        DW_HIP(hipStreamSynchronize(ctx->stream));
        uint64_t tot = h_open;
        if (int rc = allred(&tot, 1, 0)) return rc;
        if (tot == 0 || tot == prev_open) break;  // every round ends at least one k-mer of every open chain: what is left runs in circles
        prev_open = tot;
        ++rounds;
        if (rounds == 2) DW_HOOK(2);
        if (dbg) fprintf(stderr, "[smx] walks: round %llu: %llu open\n", (unsigned long long)rounds, (unsigned long long)tot);
        // A round's exchanges carry the OPEN nodes only, and they thin out round by round (at 30x: 92 %, 86, 74, 50, 15, 0.5 % of the nodes): the
        // node range of one exchange grows as they do, so that every exchange carries about the same load — a chunk of the last rounds is the whole
        // shard, not 1 / 80 of it with a few thousand requests and a dozen host round trips each. Every rank derives the same ranges (tot and the
        // largest shard are global figures).
        uint64_t CHr = CH;
        {
            const long double frac = (long double)tot / (long double)std::max<uint64_t>(2 * first[world], 1);
            const long double grow = frac > 0 ? 1.0L / frac : 1.0L;
            CHr = (uint64_t)std::min<long double>((long double)CH * std::min<long double>(grow, 4096.0L), (long double)std::max<uint64_t>(max_n2, 2));
            CHr = std::max<uint64_t>(CHr & ~1ull, 2);
        }
        const uint64_t rounds_r = std::max<uint64_t>((max_n2 + CHr - 1) / CHr, 1);
        for (uint64_t c = 0; c < rounds_r; ++c) {
            const uint64_t a = std::min(c * CHr, n2), n_it = std::min(CHr, n2 - a);
            unsigned long long *d_off = nullptr, *q = nullptr, *tag = nullptr, *qin = nullptr, *rows = nullptr, *wp = nullptr;
            std::vector<uint64_t> counts(world, 0), rcounts;
            const unsigned grid = grid_for(std::max<uint64_t>(n_it, 1), 4096);
            auto launch = [&](int pass, unsigned long long *bc_or_off) {
                if (pass == 0) hipLaunchKernelGGL((k_dw_open_req<0>), dim3(grid), dim3(BLK), lds, ctx->stream, (const unsigned long long *)word, (const uint8_t *)flag, a, n_it, B, OW,
                                                  bc_or_off, (unsigned long long *)nullptr, (unsigned long long *)nullptr);
                else hipLaunchKernelGGL((k_dw_open_req<1>), dim3(grid), dim3(BLK), lds, ctx->stream, (const unsigned long long *)word, (const uint8_t *)flag, a, n_it, B, OW, bc_or_off, q, tag);
            };
            DW_LOCAL(dw_two_pass(ctx, world, grid, launch, counts.data(), &d_off));
            const uint64_t n_send = bad ? 0 : sum(counts);
            DW_LOCAL(dalloc(ctx, &q, std::max<uint64_t>(n_send, 1)));
            DW_LOCAL(dalloc(ctx, &tag, std::max<uint64_t>(n_send, 1)));
            DW_LOCAL(dalloc(ctx, &wp, std::max<uint64_t>(n_send, 1)));
            if (!bad) launch(1, d_off);
            DW_HIP(hipGetLastError());
            if (int rc = xcounts(counts, rcounts)) return rc;
            const uint64_t n_recv = sum(rcounts);
            DW_LOCAL(dalloc(ctx, &qin, std::max<uint64_t>(n_recv, 1)));
            DW_LOCAL(dalloc(ctx, &rows, std::max<uint64_t>(n_recv, 1)));
            std::vector<uint64_t> ok_s(world, 0), ok_r;
            if (int rc = xcounts(ok_s, ok_r)) return rc;
            if (int rc = a2a(q, counts, qin, rcounts, 8)) return rc;
            if (n_recv) hipLaunchKernelGGL(k_dw_gather_words, dim3(grid_for(n_recv)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)qin, n_recv, my_base, n2,
                                           (const unsigned long long *)word, rows, d_err);
            DW_HIP(hipGetLastError());
            if (int rc = a2a(rows, rcounts, wp, counts, 8)) return rc;
            if (n_send) hipLaunchKernelGGL(k_dw_double_apply, dim3(grid_for(n_send)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)tag, (const unsigned long long *)wp,
                                           n_send, B, word, d_err + 1);
            DW_HIP(hipGetLastError());
            DW_HIP(hipStreamSynchronize(ctx->stream));
            free_temps(ctx);
        }
    }
    {
        uint32_t he = 0, hl = 0;
        DW_LOCAL(read_err(0, &he));
        DW_LOCAL(read_err(1, &hl));
        DW_LOCAL(he ? fail(ctx, SMX_DEVICE_ERROR, "%u pointers of the doubling lead outside their owner's shard", he) : 0);
        DW_LOCAL(hl ? fail(ctx, SMX_INVALID_PARAMETER, "a chain of 2^%u k-mers or more: beyond the packed hop count of the distributed walks", B.hb) : 0);
    }
    // what never finished lies on perfect loops: the local ranks of those k-mers, ascending
    uint64_t n_loop = 0;
    arena_put(ctx, ctx->dw_loops);
    ctx->dw_loops = nullptr;
    ctx->dw_nloops = 0;
    if (!bad && prev_open != ~0ull && n_mine) {
        const uint64_t ntiles = (n_mine + CAND_TILE - 1) / CAND_TILE;
        unsigned long long *tcnt = nullptr, *toff = nullptr;
        DW_LOCAL(dalloc(ctx, &tcnt, ntiles));
        DW_LOCAL(dalloc(ctx, &toff, ntiles + 1));
        if (!bad) hipLaunchKernelGGL((k_dw_loops<0>), dim3((unsigned)ntiles), dim3(BLK), 0, ctx->stream, (const unsigned long long *)word, (const uint8_t *)flag, n_mine, tcnt,
                                     (unsigned long long *)nullptr);
        DW_HIP(hipGetLastError());
        DW_LOCAL(scan_u64(ctx, tcnt, toff, ntiles));
        unsigned long long hn = 0;
        DW_HIP(hipMemcpyAsync(&hn, toff + ntiles, 8, hipMemcpyDeviceToHost, ctx->stream));
        DW_HIP(hipStreamSynchronize(ctx->stream));
        n_loop = bad ? 0 : hn;
        if (n_loop) {
            DW_LOCAL(dalloc(ctx, &ctx->dw_loops, n_loop, false));
            if (!bad) hipLaunchKernelGGL((k_dw_loops<1>), dim3((unsigned)ntiles), dim3(BLK), 0, ctx->stream, (const unsigned long long *)word, (const uint8_t *)flag, n_mine, toff,
                                         ctx->dw_loops);
            DW_HIP(hipGetLastError());
            DW_HIP(hipStreamSynchronize(ctx->stream));
            if (!bad) ctx->dw_nloops = n_loop;
        }
        free_temps(ctx);
    }
    mark("doubling");

    // ---- 3. every chain k-mer to the head of its chain ----------------------------------------------------------------------------------
    DW_HOOK(3);
    const uint64_t nwords = (n2 + 63) / 64;
    unsigned long long *hbits = nullptr, *hpre = nullptr, *hoff = nullptr, *hend = nullptr;
    uint8_t *bases = nullptr;
    uint64_t n_heads = 0, total = 0;
    DW_LOCAL(keep.get(&hbits, std::max<uint64_t>(nwords, 1)));
    DW_LOCAL(keep.get(&hpre, nwords + 1));
    {
        unsigned long long *hcnt = nullptr;
        DW_LOCAL(dalloc(ctx, &hcnt, std::max<uint64_t>(nwords, 1)));
        if (!bad && nwords) hipLaunchKernelGGL(k_dw_head_bits, dim3(grid_for(nwords * 64, 4096)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)word, (const uint8_t *)flag, n2,
                                               nwords, hbits, hcnt);
        DW_HIP(hipGetLastError());
        DW_LOCAL(scan_u64(ctx, hcnt, hpre, nwords));
        unsigned long long hn = 0;
        DW_HIP(hipMemcpyAsync(&hn, hpre + nwords, 8, hipMemcpyDeviceToHost, ctx->stream));
        DW_HIP(hipStreamSynchronize(ctx->stream));
        n_heads = bad ? 0 : hn;
        free_temps(ctx);
    }
    DW_LOCAL(keep.get(&hoff, n_heads + 1));
    DW_LOCAL(keep.get(&hend, std::max<uint64_t>(n_heads, 1)));
    {
        unsigned long long *hlen = nullptr;
        DW_LOCAL(dalloc(ctx, &hlen, std::max<uint64_t>(n_heads, 1)));
        if (!bad && n2) hipLaunchKernelGGL(k_dw_head_len, dim3(grid_for(n2, 4096)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)word, (const unsigned long long *)hbits,
                                           (const unsigned long long *)hpre, n2, B, hlen);
        DW_HIP(hipGetLastError());
        DW_LOCAL(scan_u64(ctx, hlen, hoff, n_heads));
        unsigned long long ht = 0;
        DW_HIP(hipMemcpyAsync(&ht, hoff + n_heads, 8, hipMemcpyDeviceToHost, ctx->stream));
        DW_HIP(hipMemsetAsync(hend, 0xFF, std::max<uint64_t>(n_heads, 1) * 8, ctx->stream));
        DW_HIP(hipStreamSynchronize(ctx->stream));
        total = bad ? 0 : ht;
        free_temps(ctx);
    }
    DW_LOCAL(keep.get(&bases, std::max<uint64_t>(total, 1) + 8));
    DW_HIP(hipMemsetAsync(d_stats, 0, 64, ctx->stream));
    for (uint64_t c = 0; c < node_rounds; ++c) {
        const uint64_t a = std::min(c * CH, n2), n_it = std::min(CH, n2 - a);
        unsigned long long *d_off = nullptr;
        ulonglong2 *msg = nullptr, *got = nullptr;
        std::vector<uint64_t> counts(world, 0), rcounts;
        const unsigned grid = grid_for(std::max<uint64_t>(n_it, 1), 4096);
        auto launch = [&](int pass, unsigned long long *bc_or_off) {
            if (pass == 0) hipLaunchKernelGGL((k_dw_head_msgs<0>), dim3(grid), dim3(BLK), lds, ctx->stream, (const unsigned long long *)word, (const uint8_t *)flag, a, n_it, my_base, B, OW,
                                              bc_or_off, (ulonglong2 *)nullptr);
            else hipLaunchKernelGGL((k_dw_head_msgs<1>), dim3(grid), dim3(BLK), lds, ctx->stream, (const unsigned long long *)word, (const uint8_t *)flag, a, n_it, my_base, B, OW, bc_or_off, msg);
        };
        DW_LOCAL(dw_two_pass(ctx, world, grid, launch, counts.data(), &d_off));
        const uint64_t n_send = bad ? 0 : sum(counts);
        DW_LOCAL(dalloc(ctx, &msg, std::max<uint64_t>(n_send, 1)));
        if (!bad) launch(1, d_off);
        DW_HIP(hipGetLastError());
        if (int rc = xcounts(counts, rcounts)) return rc;
        const uint64_t n_recv = sum(rcounts);
        DW_LOCAL(dalloc(ctx, &got, std::max<uint64_t>(n_recv, 1)));
        std::vector<uint64_t> ok_s(world, 0), ok_r;
        if (int rc = xcounts(ok_s, ok_r)) return rc;
        if (int rc = a2a(msg, counts, got, rcounts, 16)) return rc;
        if (n_recv) hipLaunchKernelGGL(k_dw_place, dim3(grid_for(n_recv)), dim3(BLK), 0, ctx->stream, (const ulonglong2 *)got, n_recv, my_base, n2, (const unsigned long long *)hbits,
                                       (const unsigned long long *)hpre, (const unsigned long long *)hoff, hend, bases, d_stats + 1);
        DW_HIP(hipGetLastError());
        DW_HIP(hipStreamSynchronize(ctx->stream));
        free_temps(ctx);
    }
    if (!bad && n_heads) hipLaunchKernelGGL(k_dw_count_unset, dim3(grid_for(n_heads, 4096)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)hend, n_heads, d_stats + 3);
    {
        unsigned long long hs[4] = {0, 0, 0, 0};
        DW_HIP(hipMemcpyAsync(hs, d_stats, 32, hipMemcpyDeviceToHost, ctx->stream));
        DW_HIP(hipStreamSynchronize(ctx->stream));
        DW_LOCAL(hs[2] ? fail(ctx, SMX_DEVICE_ERROR, "%llu chain nucleotides arrived at a k-mer that heads no chain", hs[2]) : 0);
        DW_LOCAL(hs[1] != total ? fail(ctx, SMX_DEVICE_ERROR, "%llu chain nucleotides arrived for chains of %llu k-mers", hs[1], (unsigned long long)total) : 0);
        DW_LOCAL(hs[3] ? fail(ctx, SMX_DEVICE_ERROR, "%llu chains whose end node never reached their head", hs[3]) : 0);
    }
    keep.drop(word);
    keep.drop(flag);
    word = nullptr;
    flag = nullptr;
    mark("chain nucleotides to the heads");

    // ---- 4. the chains behind this rank's start de-edges: lengths + end nodes first (so that every chain has its place), then the nucleotides ----
    unsigned long long *steps = nullptr, *last = nullptr, *boff = nullptr;
    uint8_t *my_bases = nullptr;
    DW_LOCAL(keep.get(&steps, std::max<uint64_t>(n_cand, 1)));
    DW_LOCAL(keep.get(&last, std::max<uint64_t>(n_cand, 1)));
    DW_LOCAL(keep.get(&boff, n_cand + 1));
    DW_HIP(hipMemsetAsync(steps, 0, std::max<uint64_t>(n_cand, 1) * 8, ctx->stream));
    if (n_cand) DW_HIP(hipMemcpyAsync(last, c_first, n_cand * 8, hipMemcpyDeviceToDevice, ctx->stream));  // (a de-edge that leads straight to a junction k-mer ends there)
    DW_HIP(hipMemsetAsync(d_err, 0, 16, ctx->stream));
    uint64_t have = 0;
    for (int sweep = 0; sweep < 2; ++sweep) {
        if (sweep == 1) {
            DW_HOOK(4);
            DW_LOCAL(scan_u64(ctx, steps, boff, n_cand));
            unsigned long long ht = 0;
            DW_HIP(hipMemcpyAsync(&ht, boff + n_cand, 8, hipMemcpyDeviceToHost, ctx->stream));
            DW_HIP(hipStreamSynchronize(ctx->stream));
            have = bad ? 0 : ht;
            free_temps(ctx);
            DW_LOCAL(keep.get(&my_bases, std::max<uint64_t>(have, 1) + 8));
        }
        for (uint64_t c = 0; c < start_rounds; ++c) {
            const uint64_t a = std::min(c * SCH, n_cand), n_it = std::min(SCH, n_cand - a);
            unsigned long long *d_off = nullptr, *q = nullptr, *tag = nullptr, *asks = nullptr;
            std::vector<uint64_t> counts(world, 0), rcounts;
            const unsigned grid = grid_for(std::max<uint64_t>(n_it, 1), 4096);
            auto launch = [&](int pass, unsigned long long *bc_or_off) {
                if (pass == 0) hipLaunchKernelGGL((k_dw_start_asks<0>), dim3(grid), dim3(BLK), lds, ctx->stream, (const unsigned long long *)c_first, (const uint8_t *)c_fj, a, n_it, OW, bc_or_off,
                                                  (unsigned long long *)nullptr, (unsigned long long *)nullptr);
                else hipLaunchKernelGGL((k_dw_start_asks<1>), dim3(grid), dim3(BLK), lds, ctx->stream, (const unsigned long long *)c_first, (const uint8_t *)c_fj, a, n_it, OW, bc_or_off, q, tag);
            };
            DW_LOCAL(dw_two_pass(ctx, world, grid, launch, counts.data(), &d_off));
            const uint64_t n_send = bad ? 0 : sum(counts);
            DW_LOCAL(dalloc(ctx, &q, std::max<uint64_t>(n_send, 1)));
            DW_LOCAL(dalloc(ctx, &tag, std::max<uint64_t>(n_send, 1)));
            if (!bad) launch(1, d_off);
            DW_HIP(hipGetLastError());
            if (int rc = xcounts(counts, rcounts)) return rc;
            const uint64_t n_recv = sum(rcounts);
            DW_LOCAL(dalloc(ctx, &asks, std::max<uint64_t>(n_recv, 1)));
            std::vector<uint64_t> ok_s(world, 0), ok_r;
            if (int rc = xcounts(ok_s, ok_r)) return rc;
            if (int rc = a2a(q, counts, asks, rcounts, 8)) return rc;
            if (sweep == 0) {
                ulonglong2 *rows = nullptr, *back = nullptr;
                DW_LOCAL(dalloc(ctx, &rows, std::max<uint64_t>(n_recv, 1)));
                DW_LOCAL(dalloc(ctx, &back, std::max<uint64_t>(n_send, 1)));
                if (!bad && n_recv) hipLaunchKernelGGL(k_dw_answer, dim3(grid_for(n_recv)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)asks, n_recv, my_base, n2,
                                                       (const unsigned long long *)hbits, (const unsigned long long *)hpre, (const unsigned long long *)hoff,
                                                       (const unsigned long long *)hend, rows, (unsigned long long *)nullptr, (unsigned long long *)nullptr, d_err);
                DW_HIP(hipGetLastError());
                if (int rc = xcounts(ok_s, ok_r)) return rc;
                if (int rc = a2a(rows, rcounts, back, counts, 16)) return rc;
                if (n_send) hipLaunchKernelGGL(k_dw_place_rows, dim3(grid_for(n_send)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)tag, (const ulonglong2 *)back, n_send, steps,
                                               last, d_err);
                DW_HIP(hipGetLastError());
            } else {
                unsigned long long *slots = nullptr, *lens = nullptr, *loff = nullptr, *mlens = nullptr, *moff = nullptr;
                uint8_t *flat = nullptr, *mine = nullptr;
                DW_LOCAL(dalloc(ctx, &slots, std::max<uint64_t>(n_recv, 1)));
                DW_LOCAL(dalloc(ctx, &lens, std::max<uint64_t>(n_recv, 1)));
                DW_LOCAL(dalloc(ctx, &loff, n_recv + 1));
                DW_LOCAL(dalloc(ctx, &mlens, std::max<uint64_t>(n_send, 1)));
                DW_LOCAL(dalloc(ctx, &moff, n_send + 1));
                if (!bad && n_recv) hipLaunchKernelGGL(k_dw_answer, dim3(grid_for(n_recv)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)asks, n_recv, my_base, n2,
                                                       (const unsigned long long *)hbits, (const unsigned long long *)hpre, (const unsigned long long *)hoff,
                                                       (const unsigned long long *)hend, (ulonglong2 *)nullptr, slots, lens, d_err);
                DW_HIP(hipGetLastError());
                DW_LOCAL(scan_u64(ctx, lens, loff, n_recv));
                if (!bad && n_send) hipLaunchKernelGGL(k_dw_lens_of, dim3(grid_for(n_send)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)tag, n_send,
                                                       (const unsigned long long *)steps, mlens);
                DW_HIP(hipGetLastError());
                DW_LOCAL(scan_u64(ctx, mlens, moff, n_send));
                // bytes per pair: what the asks of rank p add up to on either side (both sides know the lengths: no exchange of counts)
                std::vector<uint64_t> bsend(world, 0), brecv(world, 0);
                {
                    uint64_t ro = 0, so = 0;
                    unsigned long long v0 = 0, v1 = 0;
                    for (unsigned p = 0; p < world && !bad; ++p) {
                        DW_HIP(hipMemcpyAsync(&v0, loff + ro, 8, hipMemcpyDeviceToHost, ctx->stream));
                        DW_HIP(hipMemcpyAsync(&v1, loff + ro + rcounts[p], 8, hipMemcpyDeviceToHost, ctx->stream));
                        DW_HIP(hipStreamSynchronize(ctx->stream));
                        bsend[p] = v1 - v0;
                        ro += rcounts[p];
                        DW_HIP(hipMemcpyAsync(&v0, moff + so, 8, hipMemcpyDeviceToHost, ctx->stream));
                        DW_HIP(hipMemcpyAsync(&v1, moff + so + counts[p], 8, hipMemcpyDeviceToHost, ctx->stream));
                        DW_HIP(hipStreamSynchronize(ctx->stream));
                        brecv[p] = v1 - v0;
                        so += counts[p];
                    }
                }
                const uint64_t nb_send = bad ? 0 : sum(bsend), nb_recv = bad ? 0 : sum(brecv);
                DW_LOCAL(dalloc(ctx, &flat, std::max<uint64_t>(nb_send, 1) + 8));
                DW_LOCAL(dalloc(ctx, &mine, std::max<uint64_t>(nb_recv, 1) + 8));
                if (!bad && n_recv) hipLaunchKernelGGL(k_dw_ragged_copy, dim3(grid_for(n_recv)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)slots, (const unsigned long long *)loff,
                                                       n_recv, (const unsigned long long *)hoff, (const uint8_t *)bases, flat);
                DW_HIP(hipGetLastError());
                if (int rc = xcounts(ok_s, ok_r)) return rc;
                if (int rc = a2a(flat, bsend, mine, brecv, 1)) return rc;
                if (n_send) hipLaunchKernelGGL(k_dw_scatter_bases, dim3(grid_for(n_send)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)tag, (const unsigned long long *)moff,
                                               n_send, (const unsigned long long *)boff, (const uint8_t *)mine, my_bases);
                DW_HIP(hipGetLastError());
            }
            DW_HIP(hipStreamSynchronize(ctx->stream));
            free_temps(ctx);
        }
        if (sweep == 0) {
            uint32_t he = 0;
            DW_LOCAL(read_err(0, &he));
            DW_LOCAL(he ? fail(ctx, SMX_DEVICE_ERROR, "%u start de-edges lead to a k-mer that heads no chain", he) : 0);
        }
    }
    keep.drop(bases);
    keep.drop(hbits);
    keep.drop(hpre);
    keep.drop(hoff);
    keep.drop(hend);
    mark("chains of the start de-edges");

    // ---- 5. assembly: keep / drop as the single-GPU route, kept unitigs in k-mer-file order of their start k-mers --------------------
    uint64_t n_kept = 0, n_words = 0;
    DW_HOOK(5);
    DW_LOCAL(dw_unitigs<NW>(ctx, first[rank], steps, last, boff, my_bases, &n_kept, &n_words));
    free_temps(ctx);
    mark("unitigs");
    {   // every rank hears how it went everywhere: the gathers that follow are collectives of the caller's
        uint64_t z = 0;
        if (int rc = allred(&z, 1, 1)) return rc;
    }
    info[0] = n_kept;
    info[1] = n_words;
    info[2] = ctx->dw_nloops;
    info[3] = rounds;
#undef DW_HOOK
#undef DW_LOCAL
#undef DW_HIP
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
