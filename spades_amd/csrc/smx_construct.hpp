// spades_amd/csrc/smx_construct.hpp — host side of the construction path: rank directories, the k-mer file from the (k+1)-mer file
// (whole or in bucket ranges), run_graph (masks + successors, early clippers, walks, loops, link records / vertices on the device),
// the lazily filled host mirror of the graph, and run_coverage (included by smx_api.hip after smx_pipeline.hpp).
//
// The graph stays in HBM: unitigs 2-bit packed (the reference's Sequence word layout) with word-aligned starts, edge arrays in
// the reference's enumeration order, sorted link records and the vertex order. The host mirror (smxh::GraphHost, what the
// writers format) is filled the first time something asks for it. Ranks and node ids are 64-bit (size_t in the reference,
// debruijn_graph_constructor.hpp:399-406,506-567).
#pragma once
#include <functional>

void pm_release(smx_ctx *ctx);  // smx_pm.hpp
template <int NW>
int pm_route(smx_ctx *ctx, unsigned k, unsigned B, WallTrace &gwt);
inline uint32_t pm_host_bucket(const uint64_t *w, int nw, uint32_t B);

void drop_device_graph(smx_ctx *ctx) {
    arena_put(ctx, ctx->g_uwords);
    arena_put(ctx, ctx->g_eoffw);
    arena_put(ctx, ctx->g_elen);
    arena_put(ctx, ctx->g_estart);
    arena_put(ctx, ctx->g_eend);
    arena_put(ctx, ctx->g_eself);
    arena_put(ctx, ctx->g_lrecs);
    arena_put(ctx, ctx->g_vstart);
    ctx->g_uwords = nullptr;
    ctx->g_eoffw = ctx->g_elen = nullptr;
    ctx->g_estart = ctx->g_eend = nullptr;
    ctx->g_eself = nullptr;
    ctx->g_lrecs = nullptr;
    ctx->g_vstart = nullptr;
    ctx->g_nlrec = ctx->g_nv = 0;
    ctx->g_links_dev = false;
    ctx->g_dev_valid = false;
}

void drop_rank_dir(smx_ctx *ctx, smx::RankDir &ix) {
    arena_put(ctx, (void *)ix.dir);
    arena_put(ctx, (void *)ix.boff);
    ix = smx::RankDir{};
}

void drop_kpo(smx_ctx *ctx) {
    arena_put(ctx, ctx->g_kpo);
    ctx->g_kpo = nullptr;
    drop_rank_dir(ctx, ctx->g_dir_kpo);
}

void clear_graph(smx_ctx *ctx) {
    drop_kpo(ctx);
    // the count-result view may BE this graph's k-mer file — pending (pm route: to be materialised on request) or shared (d_result ==
    // g_kmers): with the graph gone there is nothing behind it any more, so it becomes an empty result (smx_bucket_sizes and
    // smx_copy_final_kmers then report 0 records instead of indexing an empty offset table / copying from a freed block)
    if (ctx->pm_view_pending || (ctx->g_kmers && ctx->d_result == ctx->g_kmers)) {
        ctx->n_records = 0;
        ctx->bucket_off.assign((size_t)ctx->num_buckets + 1, 0);
        ctx->d_result = nullptr;
    }
    if (ctx->g_kmers) {
        if (ctx->d_result == ctx->g_kmers) ctx->d_result = nullptr;
        arena_put(ctx, ctx->g_kmers);
    }
    arena_put(ctx, ctx->g_mask);
    drop_rank_dir(ctx, ctx->g_dir_kmers);
    drop_device_graph(ctx);
    pm_release(ctx);
    arena_put(ctx, ctx->dw_cand);
    ctx->dw_cand = nullptr;
    arena_put(ctx, ctx->dw_loops);
    ctx->dw_loops = nullptr;
    ctx->dw_nloops = 0;
    ctx->dw_ncand = ctx->dw_nchain = 0;
    ctx->dw_ready = false;
    ctx->g_sharded_file = false;
    ctx->g_pm = false;
    ctx->g_pm_nx = false;
    ctx->g_pm_clipped = false;
    ctx->pm_view_pending = false;
    ctx->g_kmers = nullptr;
    ctx->g_mask = nullptr;
    ctx->g_nkpo = ctx->g_nkmers = 0;
    ctx->g_nkpo_total = 0;
    ctx->g_ne = ctx->g_nuwords = ctx->g_nbases = ctx->g_npaths = ctx->g_nloops = 0;
    ctx->g_ready = false;
    ctx->g_host_valid = false;
    ctx->g_cov_hist.clear();
    ctx->gh = smxh::GraphHost();
}

// A pipeline result that becomes part of the graph state leaves the temporaries' end of the arena for the long-lived end (one
// device copy at the r+w rate); the other temporaries are released first. Without the VMM arena the block just changes owner.
int adopt_result(smx_ctx *ctx, void **p, size_t bytes) {
    free_temps(ctx, *p);
    detach_temp(ctx, *p);
    if (!ctx->arena.vmm || bytes == 0) return 0;
    void *q = arena_get(ctx, std::max<size_t>(bytes, 256), /*top=*/true);
    if (!q) return 0;  // no room for a second copy: it stays where it is
    hipError_t e = hipMemcpyAsync(q, *p, bytes, hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        arena_put(ctx, q);
        return fail(ctx, SMX_DEVICE_ERROR, "moving a result failed: %s", hipGetErrorString(e));
    }
    arena_put(ctx, *p);
    *p = q;
    return 0;
}

template <typename T>
int d2h(smx_ctx *ctx, std::vector<T> &dst, const void *src, size_t n) {
    dst.resize(n);
    if (n) HIPCHK(hipMemcpy(dst.data(), src, n * sizeof(T), hipMemcpyDeviceToHost));
    return 0;
}

// Sort distinct 64-bit keys that the host holds with the counting pipeline itself: one bucket (B = 1), K = 32 so that the whole word
// is the key; keys are left-aligned first so that the MSD digits see a spread-out fraction (host link path of mid-sized graphs).
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

// Sort + unique two-word records (w0, w1) that are already in HBM with the counting pipeline (one bucket, K = 64: the record is
// its own key, word 0 most significant). *out points into the temp list (valid until free_temps); the count-result view of the
// context is preserved.
int device_sort_recs2(smx_ctx *ctx, void *d_keys, uint64_t n, Rec<2> **out, uint64_t *n_out) {
    void *sv_res = ctx->d_result;
    const uint64_t sv_n = ctx->n_records, sv_inst = ctx->n_instances;
    const unsigned sv_nw = ctx->nw, sv_K = ctx->K, sv_B = ctx->num_buckets;
    std::vector<uint64_t> sv_boff = ctx->bucket_off;
    ctx->d_result = nullptr;  // non-owning view; d_result_buf is null here
    int rc = run_count<2>(ctx, 64, SMX_MODE_ALL, 1, d_keys, n, nullptr, /*recs_reusable=*/true);
    *out = (Rec<2> *)ctx->d_result_buf;
    *n_out = ctx->n_records;
    ctx->d_result_buf = nullptr;  // the result block stays in the temp list
    ctx->d_result = sv_res;
    ctx->n_records = sv_n;
    ctx->n_instances = sv_inst;
    ctx->nw = sv_nw;
    ctx->K = sv_K;
    ctx->num_buckets = sv_B;
    ctx->bucket_off = sv_boff;
    return rc;
}

unsigned grid_for(uint64_t n, unsigned cap = 1u << 16) { return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n + BLK - 1) / BLK, cap)); }

// Link records and vertices of the graph on the device, from the edge arrays in HBM: g_lrecs (sorted records) and g_vstart
// (first record of every vertex, in vertex-id order).
int device_build_links(smx_ctx *ctx, uint64_t n_ranks) {
    const uint64_t ne = ctx->g_ne;
    Rec<2> *keys, *sorted = nullptr;
    if (int rc = dalloc(ctx, &keys, 2 * ne)) return rc;
    const unsigned sh = (unsigned)__builtin_clzll((n_ranks ? n_ranks - 1 : 0) | 1);
    hipLaunchKernelGGL(k_link_keys, dim3(grid_for(ne)), dim3(BLK), 0, ctx->stream, (const node_t *)ctx->g_estart, (const node_t *)ctx->g_eend,
                       (const uint8_t *)ctx->g_eself, ne, sh, keys);
    HIPCHK(hipGetLastError());
    uint64_t nrec = 0;
    if (int rc = device_sort_recs2(ctx, keys, 2 * ne, &sorted, &nrec)) return rc;
    if (nrec == 0 || nrec > 2 * ne) return fail(ctx, SMX_DEVICE_ERROR, "link records: %llu after sort from %llu edges", (unsigned long long)nrec, (unsigned long long)ne);
    unsigned long long *one, *vidx;
    if (int rc = dalloc(ctx, &one, nrec)) return rc;
    if (int rc = dalloc(ctx, &vidx, nrec + 1)) return rc;
    hipLaunchKernelGGL(k_vertex_flags, dim3(grid_for(nrec)), dim3(BLK), 0, ctx->stream, (const Rec<2> *)sorted, nrec, one);
    HIPCHK(hipGetLastError());
    if (int rc = scan_u64(ctx, one, vidx, nrec)) return rc;
    unsigned long long nv = 0;
    HIPCHK(hipMemcpyAsync(&nv, vidx + nrec, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    // the sorted records become part of the graph before the second sort reuses the arena
    Rec<2> *lrecs;
    if (int rc = dalloc(ctx, &lrecs, nrec, false)) return rc;
    ctx->g_lrecs = lrecs;
    HIPCHK(hipMemcpyAsync(lrecs, sorted, nrec * sizeof(Rec<2>), hipMemcpyDeviceToDevice, ctx->stream));
    unsigned long long *vpos, *vstart;
    Rec<2> *vkeys, *vsorted = nullptr;
    if (int rc = dalloc(ctx, &vpos, nv + 1)) return rc;
    if (int rc = dalloc(ctx, &vkeys, nv + 1)) return rc;
    if (int rc = dalloc(ctx, &vstart, nv + 1, false)) return rc;
    ctx->g_vstart = vstart;
    const unsigned sh2 = (unsigned)__builtin_clzll((((3 + 2 * ne) << 2) | 3ull) | 1);
    hipLaunchKernelGGL(k_vertex_collect, dim3(grid_for(nrec)), dim3(BLK), 0, ctx->stream, (const Rec<2> *)lrecs, (const unsigned long long *)one,
                       (const unsigned long long *)vidx, nrec, sh2, vpos, vkeys);
    HIPCHK(hipGetLastError());
    uint64_t nv2 = 0;
    if (int rc = device_sort_recs2(ctx, vkeys, nv, &vsorted, &nv2)) return rc;
    if (nv2 != nv) return fail(ctx, SMX_DEVICE_ERROR, "vertex keys are not distinct");
    hipLaunchKernelGGL(k_vertex_permute, dim3(grid_for(nv)), dim3(BLK), 0, ctx->stream, (const Rec<2> *)vsorted, (const unsigned long long *)vpos,
                       (uint64_t)nv, vstart);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->g_nlrec = nrec;
    ctx->g_nv = nv;
    ctx->g_lsh = sh;
    ctx->g_links_dev = true;
    return 0;
}

// Rank directory of a sorted file whose bucket offsets the host knows.
template <int NW>
int build_rank_dir(smx_ctx *ctx, const void *recs, uint64_t n, const std::vector<uint64_t> &boff, uint32_t B, unsigned K, smx::RankDir &ix) {
    drop_rank_dir(ctx, ix);
    uint64_t mx = 0;
    for (uint32_t b = 0; b < B; ++b) mx = std::max(mx, boff[b + 1] - boff[b]);
    if (mx >= (1ull << 32)) return fail(ctx, SMX_MEMORY_LIMIT_EXCEEDED, "a bucket of %llu records exceeds the directory's 32-bit offsets (use more buckets)", (unsigned long long)mx);
    // slots per bucket: one per record on average, 8 B each (option "dir_slots": slots per record). Measured at 4.3 G k-mers with
    // 4-byte entries without fingerprints: 1 -> 2 slots per record took the successor lookups from 340 to 284 ms for 9 ms more
    // here, 4 slots cost more than they saved; the fingerprints settle more lookups in the same 8 B per record (smx_graph.hip).
    // (next to a resident (k+1)-mer file — the second construction route at config 3 — HBM is short: one slot per two records there)
    const uint64_t per2 = ctx->opt_dir_slots > 0 ? 2 * (uint64_t)std::min<int64_t>(ctx->opt_dir_slots, 8) : (ctx->g_kpo ? 1 : 2);  // half slots per record
    const uint64_t sb = std::min<uint64_t>(std::max<uint64_t>(1, n / B * per2 / 2), 1ull << 30);
    unsigned long long *d_boff;
    uint64_t *dir;
    if (int rc = dalloc(ctx, &d_boff, (size_t)B + 1, false)) return rc;
    ix.boff = d_boff;
    if (int rc = dalloc(ctx, &dir, (size_t)B * (sb + 1), false)) return rc;
    ix.dir = dir;
    ix.B = B;
    ix.SB = (uint32_t)sb;
    ix.K = K;
    ix.verify = ctx->opt_verify_lookups ? 1u : 0u;
    std::vector<unsigned long long> hb(boff.begin(), boff.begin() + B + 1);
    HIPCHK(hipMemcpyAsync(d_boff, hb.data(), ((size_t)B + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    // (k_dir_fill writes every entry of a bucket that has records; only the entries of empty buckets have to be cleared)
    for (uint32_t b = 0; b < B; ++b)
        if (boff[b + 1] == boff[b]) HIPCHK(hipMemsetAsync(dir + (size_t)b * (sb + 1), 0, (size_t)(sb + 1) * 8, ctx->stream));
    if (n) {
        hipLaunchKernelGGL((k_dir_fill<NW>), dim3(grid_for(n)), dim3(BLK), 0, ctx->stream, recs, n, ix, dir);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));  // hb goes out of scope
    return 0;
}

// ---- 2. canonical k-mers in k-mer-file order (DeBruijnKMerKMerSplitter + KMerDiskCounter, kmer_splitters.hpp:138-207) ----
// Whole: 2 derived records per (k+1)-mer through the pipeline at once. In bucket ranges when that does not fit HBM (or on request):
// the buckets of the k-mer file are disjoint, so the sorted-unique output of a range is final and only has to be appended.
template <int NW>
int derive_kmer_file(smx_ctx *ctx, unsigned k, unsigned B, bool from_reads) {
    const uint64_t nkpo = ctx->g_nkpo;
    const size_t W = (size_t)NW * 8;
    // From the reads when they are resident: the canonical k-mers of every valid run that holds a (k+1)-mer ARE the prefixes and
    // suffixes of the (k+1)-mers, and behind the pre-dedupe stage counting them costs less than sorting 2 derived records per
    // (k+1)-mer. (Not when the (k+1)-mer file came from other ranks: there are no reads for it here.)
    uint64_t n_bases = 0;
    for (auto &c : ctx->chunks) n_bases += c.n_bases;
    // (only when one batch should fit next to the (k+1)-mer file — first the super-k-mer slots, < 5 B per window, next to one buffer of
    // about as many k-mers as there are (k+1)-mers, then two such buffers; batches, folds or a host spill would cost more than the
    // derivation, and count_reads gives up instead of taking them: single_batch_only)
    const double need_reads_route = 1.08 * std::max(5.0 * (double)n_bases + 1.1 * (double)nkpo * (double)W, 2.25 * (double)nkpo * (double)W);
    if (from_reads && ctx->opt_kmers_from_reads != 0 && ctx->opt_derive_batches == 0 &&
        (ctx->opt_kmers_from_reads > 1 || need_reads_route <= (double)arena_avail(ctx))) {
        ctx->single_batch_only = true;
        int rc = count_reads<NW>(ctx, k, SMX_MODE_CANONICAL, B, k + 1);
        ctx->single_batch_only = false;
        if (rc == 0 && !ctx->result_on_host) {
            ctx->g_kmers = ctx->d_result_buf;
            ctx->g_nkmers = ctx->n_records;
            ctx->g_kboff = ctx->bucket_off;
            ctx->d_result_buf = ctx->d_result = nullptr;
            return adopt_result(ctx, &ctx->g_kmers, (size_t)ctx->g_nkmers * W);
        }
        if (rc && rc != SMX_MEMORY_LIMIT_EXCEEDED) return rc;
        free_temps(ctx);  // no room for that route: derive from the (k+1)-mer file in bucket ranges
        clear_result(ctx);
    }
    const size_t avail = arena_avail(ctx);
    const size_t need_whole = (size_t)(4.25 * (double)nkpo * (double)W) + ((size_t)64 << 20);
    const bool whole = ctx->opt_derive_batches > 1 ? false : (ctx->opt_derive_batches == 1 || need_whole <= avail);
    if (whole) {
        Rec<NW> *derived;
        if (int rc = dalloc(ctx, &derived, 2 * nkpo)) return rc;
        tbegin(ctx, "derive_kmers");
        hipLaunchKernelGGL((k_derive_kmers<NW>), dim3(grid_for(nkpo)), dim3(BLK), 0, ctx->stream, (const void *)ctx->g_kpo, nkpo, k, (void *)derived);
        HIPCHK(hipGetLastError());
        tend(ctx);
        if (int rc = run_count<NW>(ctx, k, SMX_MODE_ALL, B, derived, 2 * nkpo, nullptr, /*recs_reusable=*/true)) return rc;
        ctx->g_kmers = ctx->d_result_buf;
        ctx->g_nkmers = ctx->n_records;
        ctx->g_kboff = ctx->bucket_off;
        ctx->d_result_buf = nullptr;
        return adopt_result(ctx, &ctx->g_kmers, (size_t)ctx->g_nkmers * W);
    }
    if (B > 12 * 1024) return fail(ctx, SMX_INVALID_PARAMETER, "num_buckets=%u too large for the bucket histogram", B);
    unsigned long long *d_hist, *d_count;
    uint32_t *d_tags = nullptr;  // buckets of the two k-mers of every (k+1)-mer, found once by the histogram pass (4 B per (k+1)-mer)
    if (int rc = dalloc(ctx, &d_hist, B)) return rc;
    if (int rc = dalloc(ctx, &d_count, 1)) return rc;
    if (B <= 0xFFFF)
        if (int rc = dalloc(ctx, &d_tags, nkpo)) return rc;
    HIPCHK(hipMemsetAsync(d_hist, 0, (size_t)B * 8, ctx->stream));
    tbegin(ctx, "derive_hist");
    hipLaunchKernelGGL((k_derive_hist<NW>), dim3(grid_for(nkpo, 4096)), dim3(BLK), (size_t)B * 4, ctx->stream, (const void *)ctx->g_kpo, nkpo, k, B, d_hist,
                       d_tags);
    HIPCHK(hipGetLastError());
    tend(ctx);
    std::vector<unsigned long long> h(B);
    HIPCHK(hipMemcpyAsync(h.data(), d_hist, (size_t)B * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    // the k-mer file: a graph has about as many k-mers as (k+1)-mers; grown (copy) in the rare case it has more
    uint64_t cap = std::min<uint64_t>(2 * nkpo, nkpo + nkpo / 8 + (1u << 20));
    Rec<NW> *file;
    if (int rc = dalloc(ctx, &file, cap, false)) return rc;
    ctx->g_kmers = file;
    uint64_t max_batch = (uint64_t)((double)arena_avail(ctx) / (2.6 * (double)W));  // two buffers + bin bookkeeping + slack for placement
    if (ctx->opt_derive_batches > 1) max_batch = (2 * nkpo + ctx->opt_derive_batches - 1) / ctx->opt_derive_batches;
    uint64_t used = 0, remaining = 2 * nkpo;
    ctx->g_kboff.assign(B + 1, 0);
    WallTrace dwt;
    for (uint32_t b0 = 0; b0 < B;) {
        uint32_t b1 = b0;
        uint64_t sum = 0;
        while (b1 < B && (b1 == b0 || sum + h[b1] <= max_batch)) sum += h[b1++];
        if (sum > max_batch && ctx->opt_derive_batches <= 1)
            return fail(ctx, SMX_MEMORY_LIMIT_EXCEEDED, "bucket %u alone holds %llu derived k-mers: more than HBM has room to sort (use more buckets)", b0, (unsigned long long)sum);
        if (sum) {
            Rec<NW> *derived;
            if (int rc = dalloc(ctx, &derived, sum)) return rc;
            HIPCHK(hipMemsetAsync(d_count, 0, 8, ctx->stream));
            tbegin(ctx, "derive_kmers");
            hipLaunchKernelGGL((k_derive_range<NW>), dim3(grid_for(nkpo, 256 * 32)), dim3(BLK), 0, ctx->stream, (const void *)ctx->g_kpo, nkpo, k, B, b0, b1,
                               (const uint32_t *)d_tags, (void *)derived, d_count);
            HIPCHK(hipGetLastError());
            tend(ctx);
            dwt.mark(ctx, "d:derive");
            if (int rc = run_count<NW>(ctx, k, SMX_MODE_ALL, B, derived, sum, nullptr, /*recs_reusable=*/true, false, false, b0, b1 - b0)) return rc;
            dwt.mark(ctx, "d:sort");
            const uint64_t nres = ctx->n_records;
            remaining -= sum;
            if (used + nres > cap) {
                const uint64_t ncap = used + nres + remaining;
                Rec<NW> *bigger;
                if (int rc = dalloc(ctx, &bigger, ncap, false)) return rc;
                HIPCHK(hipMemcpyAsync(bigger, file, used * W, hipMemcpyDeviceToDevice, ctx->stream));
                HIPCHK(hipStreamSynchronize(ctx->stream));
                arena_put(ctx, file);
                ctx->g_kmers = file = bigger;
                cap = ncap;
            }
            HIPCHK(hipMemcpyAsync(file + used, ctx->d_result_buf, nres * W, hipMemcpyDeviceToDevice, ctx->stream));
            for (uint32_t b = b0; b < b1; ++b) ctx->g_kboff[b + 1] = used + ctx->bucket_off[b + 1];
            used += nres;
            ctx->d_result_buf = ctx->d_result = nullptr;  // one of the temps
            HIPCHK(hipStreamSynchronize(ctx->stream));
            void *keep[3] = {d_hist, d_count, d_tags};
            std::vector<void *> rest;
            for (void *p : ctx->temps) {
                if (p == keep[0] || p == keep[1] || p == keep[2]) rest.push_back(p);
                else arena_put(ctx, p);
            }
            ctx->temps = rest;
            dwt.mark(ctx, "d:append");
        } else {
            for (uint32_t b = b0; b < b1; ++b) ctx->g_kboff[b + 1] = used;
        }
        b0 = b1;
    }
    ctx->g_nkmers = used;
    // the count-result view of the context describes the k-mer file (smx_copy_final_kmers, smx_bucket_sizes)
    ctx->n_records = used;
    ctx->bucket_off = ctx->g_kboff;
    ctx->K = k;
    ctx->nw = NW;
    ctx->num_buckets = B;
    free_temps(ctx);
    return 0;
}

// The k-mer file AND the InOutMask bytes from one count of the resident reads (no (k+1)-mer file, no mask fill): the pre-dedupe
// stage knows the bases next to every k-mer instance — the (k+1)-mers around it — and hands the OR of their extension bits on inside
// the records (EXT layout, smx_device.hpp); after the sort one streaming pass splits records into k-mers and bytes.
// Equivalent to the reference's route (count the (k+1)-mers, out[prefix] / in[suffix] per (k+1)-mer,
// kmer_extension_index_builder.hpp:45-60): a (k+1)-mer is in that file iff it is a valid window of some read, and then it is the
// window behind its prefix instance and before its suffix instance. Needs >= 8 spare bits in the last record word, two words
// or more, the pre-dedupe stage (k >= 21, enough windows) and one batch; SMX_ROUTE_NA otherwise (nothing changed).
template <int NW>
int ext_result_to_file(smx_ctx *ctx, unsigned k, unsigned B, bool whole);
template <int NW>
int kmer_file_with_masks(smx_ctx *ctx, unsigned k, unsigned B) {
    if (ctx->opt_ext_route == 0 || !ext_layout_fits(k, NW) || ctx->chunks.empty() || ctx->opt_derive_batches != 0) return SMX_ROUTE_NA;
    ctx->ext_mode = true;
    ctx->single_batch_only = true;
    int rc = count_reads<NW>(ctx, k, SMX_MODE_CANONICAL, B, k + 1);
    ctx->ext_mode = false;
    ctx->single_batch_only = false;
    if (rc == SMX_ROUTE_NA) return rc;
    if (rc == SMX_MEMORY_LIMIT_EXCEEDED || (rc == 0 && ctx->result_on_host)) {
        free_temps(ctx);
        clear_result(ctx);
        return SMX_ROUTE_NA;
    }
    if (rc) return rc;
    if (int rc2 = ext_result_to_file<NW>(ctx, k, B, /*whole=*/true)) return rc2;
    return 0;
}

// The context's count result in the EXT layout (sorted, bucket-major) -> k-mer file, InOutMask bytes, bucket offsets; the raw
// records are released. whole: the result covers all reads of the graph (one GPU) — an odd number of extension bits is then an
// error and g_nkpo is set; a shard only reports its bits and palindromes (g_ext_bits / g_ext_pals), the caller adds the ranks up.
template <int NW>
int ext_result_to_file(smx_ctx *ctx, unsigned k, unsigned B, bool whole) {
    void *raw = ctx->d_result_buf;
    const uint64_t n = ctx->n_records;
    const std::vector<uint64_t> raw_off = ctx->bucket_off;
    ctx->d_result_buf = ctx->d_result = nullptr;
    free_temps(ctx, raw);
    ctx->temps.push_back(raw);
    ctx->g_kboff.assign(B + 1, 0);
    ctx->g_ext_bits = ctx->g_ext_pals = 0;
    if (n == 0) {
        free_temps(ctx);
        ctx->g_nkmers = ctx->g_nkpo = 0;
        return 0;
    }
    // the usual case first: every k-mer occurs once (run_prededupe merged the survivors of cut partitions) — one streaming pass
    unsigned long long *stats;
    if (int rc2 = dalloc(ctx, &stats, 3)) return rc2;
    HIPCHK(hipMemsetAsync(stats, 0, 24, ctx->stream));
    Rec<NW> *file;
    if (int rc2 = dalloc(ctx, &file, n, false)) return rc2;
    ctx->g_kmers = file;
    size_t mask_bytes = (size_t)((n + 7) / 8 * 8 + 8);
    if (int rc2 = dalloc(ctx, &ctx->g_mask, mask_bytes, false)) return rc2;
    tbegin(ctx, "ext_merge");
    hipLaunchKernelGGL((k_ext_split<NW>), dim3(grid_for(n)), dim3(BLK), 0, ctx->stream, (const void *)raw, n, k, (void *)file, ctx->g_mask, stats);
    HIPCHK(hipGetLastError());
    tend(ctx);
    unsigned long long hs[3] = {0, 0, 0};
    HIPCHK(hipMemcpyAsync(hs, stats, 24, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    uint64_t nk = n;
    if (hs[2] == 0) {
        HIPCHK(hipMemsetAsync(ctx->g_mask + nk, 0, mask_bytes - nk, ctx->stream));
        for (unsigned b = 0; b <= B; ++b) ctx->g_kboff[b] = raw_off[b];
    } else {  // copies of a k-mer with different bytes are still there: heads, scan, merge
        arena_put(ctx, file);
        arena_put(ctx, ctx->g_mask);
        ctx->g_kmers = nullptr;
        ctx->g_mask = nullptr;
        const uint64_t ntiles = (n + XM_TILE - 1) / XM_TILE;
        unsigned long long *tcnt, *toff, *d_old, *d_new;
        if (int rc2 = dalloc(ctx, &tcnt, ntiles)) return rc2;
        if (int rc2 = dalloc(ctx, &toff, ntiles + 1)) return rc2;
        if (int rc2 = dalloc(ctx, &d_old, B + 1)) return rc2;
        if (int rc2 = dalloc(ctx, &d_new, B + 1)) return rc2;
        HIPCHK(hipMemsetAsync(stats, 0, 24, ctx->stream));
        {
            std::vector<unsigned long long> h(raw_off.begin(), raw_off.end());
            HIPCHK(hipMemcpyAsync(d_old, h.data(), (size_t)(B + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
        }
        tbegin(ctx, "ext_merge");
        const unsigned grid = (unsigned)std::min<uint64_t>(ntiles, 256 * 16);
        hipLaunchKernelGGL((k_ext_heads<NW>), dim3(grid), dim3(BLK), 0, ctx->stream, (const void *)raw, n, tcnt);
        HIPCHK(hipGetLastError());
        if (int rc2 = scan_u64(ctx, tcnt, toff, ntiles)) return rc2;
        unsigned long long nkd = 0;
        HIPCHK(hipMemcpyAsync(&nkd, toff + ntiles, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        nk = nkd;
        if (int rc2 = dalloc(ctx, &file, nk, false)) return rc2;
        ctx->g_kmers = file;
        mask_bytes = (size_t)((nk + 7) / 8 * 8 + 8);
        if (int rc2 = dalloc(ctx, &ctx->g_mask, mask_bytes, false)) return rc2;
        HIPCHK(hipMemsetAsync(ctx->g_mask + nk, 0, mask_bytes - nk, ctx->stream));
        hipLaunchKernelGGL((k_ext_merge<NW, true>), dim3(grid), dim3(BLK), 0, ctx->stream, (const void *)raw, n, (const unsigned long long *)toff, k, (void *)file,
                           ctx->g_mask, stats);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL((k_ext_boff<NW>), dim3((B + 1 + BLK - 1) / BLK), dim3(BLK), 0, ctx->stream, (const void *)raw, n, (const unsigned long long *)toff,
                           (const unsigned long long *)d_old, B + 1, d_new);
        HIPCHK(hipGetLastError());
        tend(ctx);
        std::vector<unsigned long long> hn(B + 1);
        HIPCHK(hipMemcpyAsync(hn.data(), d_new, (size_t)(B + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(hs, stats, 16, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        for (unsigned b = 0; b <= B; ++b) ctx->g_kboff[b] = hn[b];
    }
    ctx->g_nkmers = nk;
    if (ctx->g_kboff[B] != nk) return fail(ctx, SMX_DEVICE_ERROR, "inconsistent bucket offsets after the extension merge (%llu vs %llu)",
                                           (unsigned long long)ctx->g_kboff[B], (unsigned long long)nk);
    ctx->g_ext_bits = hs[0];
    ctx->g_ext_pals = hs[1];
    if (whole) {
        if ((hs[0] + hs[1]) & 1) return fail(ctx, SMX_DEVICE_ERROR, "odd number of extension bits (%llu + %llu palindromes)", hs[0], hs[1]);
        ctx->g_nkpo = (hs[0] + hs[1]) / 2;
    }
    free_temps(ctx);
    ctx->n_records = nk;
    ctx->bucket_off = ctx->g_kboff;
    ctx->K = k;
    ctx->nw = NW;
    ctx->num_buckets = B;
    return 0;
}

// ---- host mirror of the device graph -----------------------------------------------------------------------------------------
// 2-bit words -> ACGT, several threads
inline void unpack_unitigs(const uint64_t *words, const unsigned long long *eoffw, const uint64_t *eoff, size_t ne, char *seq) {
    smxh::parallel_blocks(ne, (size_t)1 << 12, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            const uint64_t *w = words + eoffw[i];
            char *s = seq + eoff[i];
            const uint64_t len = eoff[i + 1] - eoff[i];
            for (uint64_t t = 0; t < len; ++t) s[t] = "ACGT"[(w[t >> 5] >> ((t & 31) << 1)) & 3];
        }
    });
}

int materialize_host(smx_ctx *ctx) {
    if (ctx->g_host_valid) return 0;
    smxh::GraphHost &g = ctx->gh;
    const uint64_t ne = ctx->g_ne;
    std::vector<unsigned long long> eoffw, elen;
    if (int rc = d2h(ctx, eoffw, ctx->g_eoffw, ne)) return rc;
    if (int rc = d2h(ctx, elen, ctx->g_elen, ne)) return rc;
    g.eoff.assign(ne + 1, 0);
    for (uint64_t i = 0; i < ne; ++i) g.eoff[i + 1] = g.eoff[i] + elen[i];
    {
        std::vector<uint64_t> words;
        if (int rc = d2h(ctx, words, ctx->g_uwords, ctx->g_nuwords)) return rc;
        g.seq.resize(g.eoff[ne]);
        if (ne) unpack_unitigs(words.data(), eoffw.data(), g.eoff.data(), ne, &g.seq[0]);
    }
    if (int rc = d2h(ctx, g.estart, ctx->g_estart, ne)) return rc;
    if (int rc = d2h(ctx, g.eend, ctx->g_eend, ne)) return rc;
    if (int rc = d2h(ctx, g.eself, ctx->g_eself, ne)) return rc;
    g.n_paths = ctx->g_npaths;
    g.n_loops = ctx->g_nloops;
    if (ctx->g_links_dev) {
        std::vector<Rec<2>> lr;
        if (int rc = d2h(ctx, lr, ctx->g_lrecs, ctx->g_nlrec)) return rc;
        std::vector<unsigned long long> vs;
        if (int rc = d2h(ctx, vs, ctx->g_vstart, ctx->g_nv)) return rc;
        g.recs.resize(2 * ne);
        const unsigned sh = ctx->g_lsh;
        const size_t nrec = lr.size();
        smxh::parallel_blocks(nrec, (size_t)1 << 16, [&](size_t b, size_t e) {
            for (size_t i = b; i < e; ++i) g.recs[i] = {((lr[i].w[0] >> sh) << 2) | (lr[i].w[1] & 3), lr[i].w[1] >> 2};
        });
        for (size_t i = nrec; i < 2 * ne; ++i) g.recs[i] = {~0ull, 0};  // the missing end records of self-conjugate edges
        g.vstart.assign(vs.begin(), vs.end());
        g.n_vertices = ctx->g_nv;
    }
    ctx->g_host_valid = true;
    return 0;
}

// host -> device (after a host-side reordering of the edges): packed unitigs + edge arrays
int upload_graph(smx_ctx *ctx) {
    drop_device_graph(ctx);
    const smxh::GraphHost &g = ctx->gh;
    const uint64_t ne = g.n_edges();
    std::vector<unsigned long long> eoffw(ne + 1, 0), elen(ne);
    for (uint64_t i = 0; i < ne; ++i) {
        elen[i] = g.eoff[i + 1] - g.eoff[i];
        eoffw[i + 1] = eoffw[i] + (elen[i] + 31) / 32;
    }
    std::vector<uint64_t> words(eoffw[ne] + 8, 0);
    smxh::parallel_blocks(ne, (size_t)1 << 12, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            const char *s = g.seq.data() + g.eoff[i];
            uint64_t *w = words.data() + eoffw[i];
            for (uint64_t t = 0; t < elen[i]; ++t) {
                const char ch = s[t];
                const uint64_t code = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3;
                w[t >> 5] |= code << ((t & 31) << 1);
            }
        }
    });
    if (int rc = dalloc(ctx, &ctx->g_uwords, words.size(), false)) return rc;
    if (int rc = dalloc(ctx, &ctx->g_eoffw, ne + 1, false)) return rc;
    if (int rc = dalloc(ctx, &ctx->g_elen, ne + 1, false)) return rc;
    if (int rc = dalloc(ctx, &ctx->g_estart, ne + 1, false)) return rc;
    if (int rc = dalloc(ctx, &ctx->g_eend, ne + 1, false)) return rc;
    if (int rc = dalloc(ctx, &ctx->g_eself, ne + 1, false)) return rc;
    HIPCHK(hipMemcpy(ctx->g_uwords, words.data(), words.size() * 8, hipMemcpyHostToDevice));
    if (ne) {
        HIPCHK(hipMemcpy(ctx->g_eoffw, eoffw.data(), ne * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(ctx->g_elen, elen.data(), ne * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(ctx->g_estart, g.estart.data(), ne * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(ctx->g_eend, g.eend.data(), ne * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(ctx->g_eself, g.eself.data(), ne, hipMemcpyHostToDevice));
    }
    ctx->g_ne = ne;
    ctx->g_nuwords = eoffw[ne];
    ctx->g_nbases = g.eoff[ne];
    ctx->g_dev_valid = true;
    return 0;
}

// Perfect loops (CollectLoops, debruijn_graph_constructor.hpp:359-397; serial in the reference) from their k-mers in k-mer-file order
// (packed, nw words each; ranks ascending; InOutMask bytes) — collected on the host by all cores (smx_loops_host.hpp) — and
// appended to the device graph: bigger arrays, old content copied, loop edges uploaded.
inline int append_loops(smx_ctx *ctx, unsigned k, const uint64_t *lkmers, const uint64_t *lranks, const uint8_t *lmasks, uint64_t nloopk, uint64_t nkept,
                        uint64_t ktotalw) {
    std::vector<smxl::PackedLoop> loops;
    if (const int lrc = smxl::collect_loops(lkmers, lranks, lmasks, nloopk, k, loops))
        return fail(ctx, SMX_DEVICE_ERROR, "inconsistent k-mer index: the k-mers left over by the unbranching paths do not form perfect loops (%d)", lrc);
    const uint64_t nl = loops.size();
    if (nl) {
        std::vector<unsigned long long> l_offw(nl), l_len(nl), l_start(nl), l_end(nl);
        std::vector<uint8_t> l_self(nl);
        std::vector<uint64_t> lwords;
        for (uint64_t i = 0; i < nl; ++i) {
            const smxl::PackedLoop &lp = loops[i];
            l_offw[i] = ktotalw + lwords.size();
            l_len[i] = lp.len;
            l_start[i] = lp.start_node;
            l_end[i] = lp.end_node;
            l_self[i] = lp.self_rc;
            lwords.insert(lwords.end(), lp.words.begin(), lp.words.end());
        }
        const uint64_t ne2 = nkept + nl, tw2 = ktotalw + lwords.size();
        uint64_t *uw2;
        unsigned long long *eo2, *el2;
        node_t *es2, *ee2;
        uint8_t *sf2;
        if (int rc = dalloc(ctx, &uw2, tw2 + 8, false)) return rc;
        if (int rc = dalloc(ctx, &eo2, ne2 + 1, false)) return rc;
        if (int rc = dalloc(ctx, &el2, ne2 + 1, false)) return rc;
        if (int rc = dalloc(ctx, &es2, ne2 + 1, false)) return rc;
        if (int rc = dalloc(ctx, &ee2, ne2 + 1, false)) return rc;
        if (int rc = dalloc(ctx, &sf2, ne2 + 1, false)) return rc;
        hipError_t e = hipSuccess;
        auto cp = [&](void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
            if (bytes && e == hipSuccess) e = hipMemcpy(dst, src, bytes, kind);
        };
        cp(uw2, ctx->g_uwords, ktotalw * 8, hipMemcpyDeviceToDevice);
        cp(uw2 + ktotalw, lwords.data(), lwords.size() * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemset(uw2 + tw2, 0, 64);
        cp(eo2, ctx->g_eoffw, nkept * 8, hipMemcpyDeviceToDevice);
        cp(eo2 + nkept, l_offw.data(), nl * 8, hipMemcpyHostToDevice);
        cp(el2, ctx->g_elen, nkept * 8, hipMemcpyDeviceToDevice);
        cp(el2 + nkept, l_len.data(), nl * 8, hipMemcpyHostToDevice);
        cp(es2, ctx->g_estart, nkept * 8, hipMemcpyDeviceToDevice);
        cp(es2 + nkept, l_start.data(), nl * 8, hipMemcpyHostToDevice);
        cp(ee2, ctx->g_eend, nkept * 8, hipMemcpyDeviceToDevice);
        cp(ee2 + nkept, l_end.data(), nl * 8, hipMemcpyHostToDevice);
        cp(sf2, ctx->g_eself, nkept, hipMemcpyDeviceToDevice);
        cp(sf2 + nkept, l_self.data(), nl, hipMemcpyHostToDevice);
        arena_put(ctx, ctx->g_uwords);
        arena_put(ctx, ctx->g_eoffw);
        arena_put(ctx, ctx->g_elen);
        arena_put(ctx, ctx->g_estart);
        arena_put(ctx, ctx->g_eend);
        arena_put(ctx, ctx->g_eself);
        ctx->g_uwords = uw2;
        ctx->g_eoffw = eo2;
        ctx->g_elen = el2;
        ctx->g_estart = es2;
        ctx->g_eend = ee2;
        ctx->g_eself = sf2;
        if (e != hipSuccess) return fail(ctx, SMX_DEVICE_ERROR, "appending the perfect loops failed: %s", hipGetErrorString(e));
        ctx->g_ne = ne2;
        ctx->g_nuwords = tw2;
        ctx->g_nloops = nl;
    }
    return 0;
}

// Perfect loops on the device (smx_loops.hip; option "device_loops"): llist = the left-over k-mers (node ranks, any order), tab = the
// successor table of the walks. O(loop k-mers) work — cycle leaders, rotation, palindrome search, orientation, the 2-bit sequences — runs
// in kernels; the host only orders the leaders (one per loop) and lays out the edges. Same result as append_loops (tests run both).
template <int NW>
int device_loops(smx_ctx *ctx, unsigned k, bool pm, const unsigned long long *llist, uint64_t L, const node_t *tab, uint64_t D0, uint64_t nkept, uint64_t ktotalw) {
    unsigned long long *lead, *llen, *cnt;
    if (int rc = dalloc(ctx, &lead, L)) return rc;
    if (int rc = dalloc(ctx, &llen, L)) return rc;
    if (int rc = dalloc(ctx, &cnt, 2)) return rc;
    HIPCHK(hipMemsetAsync(cnt, 0, 16, ctx->stream));
    const bool ext = pm && !ctx->g_pm_nx;  // the partition-major records carry their mask byte — unless the k-mer leaves it no room (nx: plain k-mers)
    if (pm && !ext)
        hipLaunchKernelGGL((k_lp_leaders<NW, false, true>), dim3(grid_for(L)), dim3(BLK), 0, ctx->stream, llist, L, tab, (const void *)ctx->g_kmers, (uint32_t)ctx->g_B,
                           (unsigned long long)(2 * D0 + 2), lead, llen, cnt);
    else if (pm)
        hipLaunchKernelGGL((k_lp_leaders<NW, true, true>), dim3(grid_for(L)), dim3(BLK), 0, ctx->stream, llist, L, tab, (const void *)ctx->g_kmers, (uint32_t)ctx->g_B,
                           (unsigned long long)(2 * D0 + 2), lead, llen, cnt);
    else
        hipLaunchKernelGGL((k_lp_leaders<NW, false, false>), dim3(grid_for(L)), dim3(BLK), 0, ctx->stream, llist, L, tab, (const void *)ctx->g_kmers, (uint32_t)ctx->g_B,
                           (unsigned long long)(2 * D0 + 2), lead, llen, cnt);
    HIPCHK(hipGetLastError());
    unsigned long long hc[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(hc, cnt, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (hc[1]) return fail(ctx, SMX_DEVICE_ERROR, "inconsistent k-mer index: %llu left-over k-mers lie on no cycle", hc[1]);
    const uint64_t nl = hc[0];
    if (!nl) return 0;
    std::vector<unsigned long long> hlead, hlen;
    if (int rc = d2h(ctx, hlead, lead, nl)) return rc;
    if (int rc = d2h(ctx, hlen, llen, nl)) return rc;
    // One thread walks one loop (rotation, palindrome search, orientation, sequence): fine for plasmids by the thousand, not for ONE loop of
    // millions of k-mers (a dependent load per step: seconds). Such an input goes to the host collector, which cuts a cycle among all cores.
    for (uint64_t i = 0; i < nl; ++i)
        if (hlen[i] > (1ull << 18)) return SMX_ROUTE_NA;
    std::vector<uint64_t> order(nl);
    for (uint64_t i = 0; i < nl; ++i) order[i] = i;
    if (pm) {  // the reference meets the loops in k-mer-file order of their first k-mers: (bucket, record) of the leaders
        Rec<NW> *lk;
        uint8_t *lm;
        if (int rc = dalloc(ctx, &lk, nl)) return rc;
        if (int rc = dalloc(ctx, &lm, nl)) return rc;
        hipLaunchKernelGGL((k_gather_kmers<NW>), dim3(grid_for(nl)), dim3(BLK), 0, ctx->stream, (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask,
                           (const unsigned long long *)lead, (uint64_t)nl, (void *)lk, lm);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));
        std::vector<uint64_t> hk;
        if (int rc = d2h(ctx, hk, lk, (size_t)nl * NW)) return rc;
        if (ext)
            for (uint64_t i = 0; i < nl; ++i) hk[(size_t)i * NW + NW - 1] >>= smx::EXT_BITS;
        std::vector<uint32_t> bk(nl);
        for (uint64_t i = 0; i < nl; ++i) bk[i] = pm_host_bucket(&hk[(size_t)i * NW], NW, ctx->g_B);
        std::sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) {
            if (bk[a] != bk[b]) return bk[a] < bk[b];
            for (int w = 0; w < NW; ++w)
                if (hk[(size_t)a * NW + w] != hk[(size_t)b * NW + w]) return hk[(size_t)a * NW + w] < hk[(size_t)b * NW + w];
            return false;
        });
    } else {
        std::sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return hlead[a] < hlead[b]; });
    }
    {
        std::vector<unsigned long long> sl(nl), sn(nl);
        for (uint64_t i = 0; i < nl; ++i) {
            sl[i] = hlead[order[i]];
            sn[i] = hlen[order[i]];
        }
        hlead.swap(sl);
        hlen.swap(sn);
    }
    HIPCHK(hipMemcpy(lead, hlead.data(), nl * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(llen, hlen.data(), nl * 8, hipMemcpyHostToDevice));
    unsigned long long *meas;
    if (int rc = dalloc(ctx, &meas, 4 * nl)) return rc;
    if (ext)
        hipLaunchKernelGGL((k_lp_measure<NW, true>), dim3(grid_for(nl)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)lead, (const unsigned long long *)llen, nl, tab,
                           (const void *)ctx->g_kmers, meas);
    else
        hipLaunchKernelGGL((k_lp_measure<NW, false>), dim3(grid_for(nl)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)lead, (const unsigned long long *)llen, nl, tab,
                           (const void *)ctx->g_kmers, meas);
    HIPCHK(hipGetLastError());
    std::vector<unsigned long long> hm;
    if (int rc = d2h(ctx, hm, meas, 4 * nl)) return rc;
    // edges: the loop from its smallest k-mer once around, or — split at its first palindromic (k+1)-mer — that (k+1)-mer and the rest
    std::vector<unsigned long long> ea, eb, nnt, offw, elen;
    uint64_t words = 0;
    auto add = [&](unsigned long long a, unsigned long long b, unsigned long long n) {
        ea.push_back(a);
        eb.push_back(b);
        nnt.push_back(n);
        offw.push_back(ktotalw + words);
        elen.push_back(k + n);
        words += (k + n + 31) / 32;
    };
    for (uint64_t i = 0; i < nl; ++i) {
        if (hm[4 * i + 1] == ~0ull) add(hm[4 * i], hm[4 * i], hlen[i]);
        else {
            add(hm[4 * i + 2], hm[4 * i + 3], 1);
            add(hm[4 * i + 3], hm[4 * i + 2], hlen[i] - 1);
        }
    }
    const uint64_t nle = ea.size(), ne2 = nkept + nle, tw2 = ktotalw + words;
    uint64_t *uw2;
    unsigned long long *eo2, *el2, *dea, *deb, *dnn, *dof;
    node_t *es2, *ee2;
    uint8_t *sf2;
    if (int rc = dalloc(ctx, &uw2, tw2 + 8, false)) return rc;
    if (int rc = dalloc(ctx, &eo2, ne2 + 1, false)) return rc;
    if (int rc = dalloc(ctx, &el2, ne2 + 1, false)) return rc;
    if (int rc = dalloc(ctx, &es2, ne2 + 1, false)) return rc;
    if (int rc = dalloc(ctx, &ee2, ne2 + 1, false)) return rc;
    if (int rc = dalloc(ctx, &sf2, ne2 + 1, false)) return rc;
    if (int rc = dalloc(ctx, &dea, nle)) return rc;
    if (int rc = dalloc(ctx, &deb, nle)) return rc;
    if (int rc = dalloc(ctx, &dnn, nle)) return rc;
    if (int rc = dalloc(ctx, &dof, nle)) return rc;
    hipError_t e = hipSuccess;
    auto cp = [&](void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
        if (bytes && e == hipSuccess) e = hipMemcpy(dst, src, bytes, kind);
    };
    cp(uw2, ctx->g_uwords, ktotalw * 8, hipMemcpyDeviceToDevice);
    if (e == hipSuccess) e = hipMemset(uw2 + ktotalw, 0, (words + 8) * 8);
    cp(eo2, ctx->g_eoffw, nkept * 8, hipMemcpyDeviceToDevice);
    cp(eo2 + nkept, offw.data(), nle * 8, hipMemcpyHostToDevice);
    cp(el2, ctx->g_elen, nkept * 8, hipMemcpyDeviceToDevice);
    cp(el2 + nkept, elen.data(), nle * 8, hipMemcpyHostToDevice);
    cp(es2, ctx->g_estart, nkept * 8, hipMemcpyDeviceToDevice);
    cp(ee2, ctx->g_eend, nkept * 8, hipMemcpyDeviceToDevice);
    cp(sf2, ctx->g_eself, nkept, hipMemcpyDeviceToDevice);
    cp(dea, ea.data(), nle * 8, hipMemcpyHostToDevice);
    cp(deb, eb.data(), nle * 8, hipMemcpyHostToDevice);
    cp(dnn, nnt.data(), nle * 8, hipMemcpyHostToDevice);
    cp(dof, offw.data(), nle * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        if (ext)
            hipLaunchKernelGGL((k_lp_write<NW, true>), dim3(grid_for(nle)), dim3(BLK), 0, ctx->stream, (const node_t *)dea, (const node_t *)deb, (const unsigned long long *)dnn,
                               (const unsigned long long *)dof, nle, tab, (const void *)ctx->g_kmers, k, uw2, es2 + nkept, ee2 + nkept, sf2 + nkept);
        else
            hipLaunchKernelGGL((k_lp_write<NW, false>), dim3(grid_for(nle)), dim3(BLK), 0, ctx->stream, (const node_t *)dea, (const node_t *)deb, (const unsigned long long *)dnn,
                               (const unsigned long long *)dof, nle, tab, (const void *)ctx->g_kmers, k, uw2, es2 + nkept, ee2 + nkept, sf2 + nkept);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    arena_put(ctx, ctx->g_uwords);
    arena_put(ctx, ctx->g_eoffw);
    arena_put(ctx, ctx->g_elen);
    arena_put(ctx, ctx->g_estart);
    arena_put(ctx, ctx->g_eend);
    arena_put(ctx, ctx->g_eself);
    ctx->g_uwords = uw2;
    ctx->g_eoffw = eo2;
    ctx->g_elen = el2;
    ctx->g_estart = es2;
    ctx->g_eend = ee2;
    ctx->g_eself = sf2;
    if (e != hipSuccess) return fail(ctx, SMX_DEVICE_ERROR, "the perfect loops on the device failed: %s", hipGetErrorString(e));
    ctx->g_ne = ne2;
    ctx->g_nuwords = tw2;
    ctx->g_nloops = nle;
    return 0;
}

// The early clippers of spades-core's Construction stage on the extension masks (stages/construction.cpp:289-341), on either view of the graph
// (IX: FileFind — sorted k-mer file + rank directory; PmFind — the partition-major records of route 0). resucc(&succ): the successor table of the
// masks as they are NOW, in the format IX::next reads; *edited is set whenever a pass has changed the masks.
template <int NW, class IX>
int early_clippers(smx_ctx *ctx, unsigned k, IX ix, uint64_t D0, uint32_t *d_err, const std::function<int(const node_t **)> &resucc, bool *edited,
                   const std::function<int(const uint8_t *, uint8_t *)> *marked_chains = nullptr /* route 0: isolates the chains whose heads the tip clipper marked */) {
    const unsigned grid = grid_for(2 * D0);
    const node_t *succ = nullptr;
    // ---- 3a. early A/T remover (RNA pipelines: EarlyATClipper::run, stages/construction.cpp:317-326) --------------
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
        hipLaunchKernelGGL((k_at_edges_mark<NW, IX>), dim3(grid), dim3(BLK), 0, ctx->stream, ix, (const uint8_t *)ctx->g_mask, D0, k, thr_edge, atflag, d_err);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL((k_at_edges_apply<NW, IX>), dim3(grid), dim3(BLK), 0, ctx->stream, ix, (uint32_t *)ctx->g_mask, D0, k, (const uint8_t *)atflag, astats, d_err);
        HIPCHK(hipGetLastError());
        *edited = true;
        tend(ctx);  // (the stage is timed in two pieces: making the successor table again opens stages of its own in between)
        if (int rc = resucc(&succ)) return rc;
        tbegin(ctx, "early_at");
        hipLaunchKernelGGL((k_at_tips_mark<NW, IX>), dim3(grid), dim3(BLK), 0, ctx->stream, ix, (const uint8_t *)ctx->g_mask, succ, D0, k, min_len, max_len,
                           (const uint16_t *)d_thr, isolate, tipped, astats, d_err);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL(k_tip_apply, dim3(grid), dim3(BLK), 0, ctx->stream, ctx->g_mask, (const uint8_t *)isolate, D0);
        HIPCHK(hipGetLastError());
        hipLaunchKernelGGL((k_tip_fix<NW, IX>), dim3(grid), dim3(BLK), 0, ctx->stream, ix, (uint32_t *)ctx->g_mask, (const uint8_t *)tipped, D0, k, d_err);
        HIPCHK(hipGetLastError());
        *edited = true;
        tend(ctx);
        unsigned long long hs[4] = {0, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(hs, astats, 32, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        ctx->g_at_tip_kmers = hs[0];
        ctx->g_at_edges = hs[2];
        for (void *p : {(void *)atflag, (void *)isolate, (void *)tipped}) {  // (16 B per k-mer of marks: gone before the walks ask for their arrays)
            detach_temp(ctx, p);
            arena_put(ctx, p);
        }
    }
    // ---- 3b. early tip clipper (spades-core variant, off for spades-gbuilder) ---------------------
    if (ctx->opt_early_tip_bound > 0) {
        uint8_t *isolate, *hmark = nullptr;
        unsigned long long *tstats;
        if (int rc = dalloc(ctx, &isolate, D0 + 1)) return rc;
        if (int rc = dalloc(ctx, &tstats, 2)) return rc;
        HIPCHK(hipMemsetAsync(isolate, 0, D0 + 1, ctx->stream));
        if (marked_chains) {
            if (int rc = dalloc(ctx, &hmark, 2 * D0 + 16)) return rc;
            HIPCHK(hipMemsetAsync(hmark, 0, 2 * D0 + 16, ctx->stream));
        }
        HIPCHK(hipMemsetAsync(tstats, 0, 16, ctx->stream));
        if (int rc = resucc(&succ)) return rc;
        tbegin(ctx, "early_tips");
        {   // the branches of the junction k-mers, listed densely (k_cand_tiles / k_cand_expand on the masks as they are now): one lane per branch
            const uint64_t ntiles = (D0 + CAND_TILE - 1) / CAND_TILE;
            unsigned long long *tcnt, *toff, *nj, *bcand;
            uint32_t *blen;
            node_t *bfirst;
            if (int rc = dalloc(ctx, &tcnt, ntiles)) return rc;
            if (int rc = dalloc(ctx, &toff, ntiles + 1)) return rc;
            if (int rc = dalloc(ctx, &nj, CAND_NJ)) return rc;
            HIPCHK(hipMemsetAsync(nj, 0, CAND_NJ * 8, ctx->stream));
            hipLaunchKernelGGL(k_cand_tiles, dim3((unsigned)ntiles), dim3(BLK), 0, ctx->stream, (const uint8_t *)ctx->g_mask, D0, tcnt, nj, (unsigned long long *)nullptr);
            HIPCHK(hipGetLastError());
            if (int rc = scan_u64(ctx, tcnt, toff, ntiles)) return rc;
            unsigned long long Cb = 0;
            HIPCHK(hipMemcpyAsync(&Cb, toff + ntiles, 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            if (Cb) {
                if (int rc = dalloc(ctx, &bcand, Cb)) return rc;
                if (int rc = dalloc(ctx, &blen, Cb)) return rc;
                if (int rc = dalloc(ctx, &bfirst, Cb)) return rc;
                hipLaunchKernelGGL(k_cand_expand, dim3((unsigned)ntiles), dim3(BLK), 0, ctx->stream, (const uint8_t *)ctx->g_mask, (const unsigned long long *)toff, D0, bcand);
                HIPCHK(hipGetLastError());
                hipLaunchKernelGGL((k_tip_branch<NW, IX>), dim3(grid_for(Cb)), dim3(BLK), 0, ctx->stream, ix, (const uint8_t *)ctx->g_mask, succ, (const unsigned long long *)bcand,
                                   (uint64_t)Cb, k, (uint32_t)std::min<int64_t>(ctx->opt_early_tip_bound, 0x7FFFFFFF), blen, bfirst, d_err);
                HIPCHK(hipGetLastError());
                hipLaunchKernelGGL((k_tip_decide<IX>), dim3(grid_for(Cb)), dim3(BLK), 0, ctx->stream, ix, succ, (const unsigned long long *)bcand, (uint64_t)Cb,
                                   (const uint32_t *)blen, (const node_t *)bfirst, isolate, hmark, (uint32_t *)ctx->g_mask, tstats);
                HIPCHK(hipGetLastError());
                if (marked_chains)
                    if (int rc = (*marked_chains)(hmark, isolate)) return rc;
                HIPCHK(hipStreamSynchronize(ctx->stream));
                for (void *p : {(void *)bcand, (void *)blen, (void *)bfirst}) {
                    detach_temp(ctx, p);
                    arena_put(ctx, p);
                }
            }
        }
        hipLaunchKernelGGL(k_tip_apply, dim3(grid), dim3(BLK), 0, ctx->stream, ctx->g_mask, (const uint8_t *)isolate, D0);
        HIPCHK(hipGetLastError());
        *edited = true;
        tend(ctx);
        unsigned long long hs[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(hs, tstats, 16, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        ctx->g_tip_kmers = hs[0];
        ctx->g_tips = hs[1];
        detach_temp(ctx, isolate);
        arena_put(ctx, isolate);
        if (hmark) {
            detach_temp(ctx, hmark);
            arena_put(ctx, hmark);
        }
    }
    return 0;
}

// Everything after the extension masks: early clippers (options), node table of the final masks, start de-edges, walks, perfect
// loops, link records + vertices. tab: 2 * D0 + 2 entries; tab_valid: k_fill_tab has already filled it for the current masks.
// pm: the partition-major route (smx_pm.hpp) — g_kmers holds EXT records in the dedupe stage's order, tab is filled, walks cross chunks by
// jump words, and the start de-edges are numbered through the sorted junction k-mers.
template <int NW>
int graph_from_masks(smx_ctx *ctx, unsigned k, node_t *tab, bool tab_valid, uint32_t *d_err, WallTrace &gwt, bool present = false, const PmWalk *pm = nullptr,
                     const std::function<int()> *tab_ready = nullptr /* pm: the node table is still being filled on the side stream; called before the first use of it */,
                     const std::function<int(bool)> *retab = nullptr /* pm with early clippers: makes the node table again from the masks as they are now (true: for the last time) */) {
    bool masks_edited = false;
    const uint64_t D0 = ctx->g_nkmers;
    const unsigned grid = grid_for(2 * D0);
    const smx::RankDir ixk = ctx->g_dir_kmers;
    struct Prefix {
        smx_ctx *c;
        Prefix(smx_ctx *c_, const char *p) : c(c_) { c->tprefix = p; }
        ~Prefix() { c->tprefix.clear(); }
    };
    bool clipped = !tab_valid;
    ctx->g_at_edges = ctx->g_at_tip_kmers = 0;
    ctx->g_tip_kmers = ctx->g_tips = 0;
    if (ctx->opt_early_at || ctx->opt_early_tip_bound > 0) {
        clipped = true;
        if (pm) {
            // route 0 (round 6): the clippers look k-mers up through the partition table and walk the route's own node table; whenever they have
            // edited the masks, the table is made again (retab: k_pm_tab with the unclipped masks beside the clipped ones + k_pm_remote)
            if (!retab) return fail(ctx, SMX_DEVICE_ERROR, "the early clippers on the partition-major route need a way to renew the node table");
            PmFind<NW> ixp{};
            ixp.ix = pm->ix;
            ixp.jmp = pm->jmp;
            ixp.cinfo = pm->cinfo;
            ixp.cob = pm->cob;
            ixp.nchunks = pm->nchunks;
            ixp.bytes_ok = ctx->opt_early_at ? 0u : 1u;  // (the tip clipper is then the first to edit the mask array: its branch walks still see the records' own bytes)
            const std::function<int(const node_t **)> resucc = [&](const node_t **s) -> int {
                if (masks_edited)
                    if (int rc = (*retab)(false)) return rc;
                masks_edited = false;
                *s = tab;
                return 0;
            };
            const uint32_t maxn = pm->ix.T / 2;
            const size_t ilds = (size_t)maxn * 12 + 16;
            if (int rc = set_lds(ctx, k_pm_isolate_chains, ilds)) return rc;
            const std::function<int(const uint8_t *, uint8_t *)> marked_chains = [&](const uint8_t *hmark, uint8_t *isolate) -> int {
                if (pm->nchunks)
                    hipLaunchKernelGGL(k_pm_isolate_chains, dim3(std::min<uint32_t>(pm->nchunks, 256 * 16)), dim3(BLK), ilds, ctx->stream, pm->cinfo, pm->nchunks, maxn,
                                       (const node_t *)tab, pm->jmp, hmark, isolate, (const uint32_t *)ctx->pm.llink);
                HIPCHK(hipGetLastError());
                return 0;
            };
            if (int rc = early_clippers<NW>(ctx, k, ixp, D0, d_err, resucc, &masks_edited, &marked_chains)) return rc;
            if (int rc = (*retab)(true)) return rc;  // (the last time: what only this needs — local links, unclipped masks — goes before the walks ask for their arrays)
            masks_edited = false;
        } else {
            node_t *succ = nullptr;  // successor table of the early clippers (their own format, by lookup)
            if (int rc = dalloc(ctx, &succ, 2 * D0)) return rc;
            FileFind<NW> ixf{(const Rec<NW> *)ctx->g_kmers, ixk};
            const std::function<int(const node_t **)> resucc = [&](const node_t **s) -> int {
                hipLaunchKernelGGL((k_succ<NW, FileFind<NW>>), dim3(grid), dim3(BLK), 0, ctx->stream, ixf, (const uint8_t *)ctx->g_mask, D0, k, succ, d_err);
                HIPCHK(hipGetLastError());
                *s = succ;
                return 0;
            };
            if (int rc = early_clippers<NW>(ctx, k, ixf, D0, d_err, resucc, &masks_edited)) return rc;
        }
    }
    if (clipped && !(pm && (ctx->opt_early_at || ctx->opt_early_tip_bound > 0))) {  // the node table has to describe the clipped masks (route 0 has renewed its own)
        tbegin(ctx, "succ");
        // (present: masks that came from the reads and were not clipped — every extension leads to a k-mer of the file)
        if (present && !ctx->opt_early_at && ctx->opt_early_tip_bound <= 0)
            hipLaunchKernelGGL((k_tab_from_masks<NW, true>), dim3(grid), dim3(BLK), 0, ctx->stream, (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask, D0, k,
                               ixk, tab, d_err);
        else
            hipLaunchKernelGGL((k_tab_from_masks<NW, false>), dim3(grid), dim3(BLK), 0, ctx->stream, (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask, D0, k,
                               ixk, tab, d_err);
        HIPCHK(hipGetLastError());
        tend(ctx);
    }
    // ---- 4. start de-edges -----------------------------------------------------------------------
    const uint64_t ntiles = (D0 + CAND_TILE - 1) / CAND_TILE;
    unsigned long long *tcnt, *toff, *counters;  // counters: [0] spare, [1] non-junction k-mers on kept paths, [2..3] loop k-mers, [4..] junction k-mers
    unsigned long long *tjcnt = nullptr, *tjoff = nullptr;  // pm: junction k-mers per tile
    if (int rc = dalloc(ctx, &tcnt, ntiles)) return rc;
    if (int rc = dalloc(ctx, &toff, ntiles + 1)) return rc;
    if (int rc = dalloc(ctx, &counters, 4 + CAND_NJ)) return rc;
    if (pm) {
        if (int rc = dalloc(ctx, &tjcnt, ntiles)) return rc;
        if (int rc = dalloc(ctx, &tjoff, ntiles + 1)) return rc;
    }
    HIPCHK(hipMemsetAsync(counters, 0, (4 + CAND_NJ) * 8, ctx->stream));
    tbegin(ctx, "candidates");
    hipLaunchKernelGGL(k_cand_tiles, dim3((unsigned)ntiles), dim3(BLK), 0, ctx->stream, (const uint8_t *)ctx->g_mask, D0, tcnt, counters + 4, tjcnt);
    HIPCHK(hipGetLastError());
    if (pm)
        if (int rc = scan_u64(ctx, tjcnt, tjoff, ntiles)) return rc;
    if (int rc = scan_u64(ctx, tcnt, toff, ntiles)) return rc;
    unsigned long long C = 0, n_junction = 0, h_nj[CAND_NJ];
    HIPCHK(hipMemcpyAsync(&C, toff + ntiles, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(h_nj, counters + 4, CAND_NJ * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < CAND_NJ; ++i) n_junction += h_nj[i];
    tend(ctx);
    ctx->g_route_stats[4] = n_junction;
    ctx->g_route_stats[5] = C;
    gwt.mark(ctx, "g:masks+succ");
    uint64_t nkept = 0, ktotalw = 0;
    unsigned long long interior = 0;
    unsigned long long *cand = nullptr, *len = nullptr;
    node_t *first = nullptr;
    uint8_t *flags = nullptr;
    if (C > 0) {
        unsigned long long *kw = nullptr, *one = nullptr;
        node_t *last;
        // route 0: word offset (35 bits: 2^35 words of unitigs are 275 GB) and edge index (29 bits) of a kept path come out of ONE scan side by side in one word
        // where the start de-edges number less than 2^29 (option walk_pack = 0: two arrays, two scans, as on the sorted routes)
        const unsigned pack_shift = (pm && ctx->opt_walk_pack != 0 && C < (1ull << 29)) ? 35u : 0u;
        if (int rc = dalloc(ctx, &cand, C)) return rc;
        if (int rc = dalloc(ctx, &len, C)) return rc;
        if (!pack_shift) {
            if (int rc = dalloc(ctx, &kw, C + 1)) return rc;
            if (int rc = dalloc(ctx, &one, C + 1)) return rc;
        }
        if (int rc = dalloc(ctx, &first, C)) return rc;
        if (int rc = dalloc(ctx, &last, C)) return rc;
        if (int rc = dalloc(ctx, &flags, C)) return rc;
        const unsigned cgrid = grid_for(C);
        unsigned long long *qidx = nullptr, *vq = nullptr;  // pm: number of every start de-edge in the reference's order; words << 1 | keep there
        if (!pm) {
            hipLaunchKernelGGL(k_cand_expand, dim3((unsigned)ntiles), dim3(BLK), 0, ctx->stream, (const uint8_t *)ctx->g_mask,
                               (const unsigned long long *)toff, D0, cand);
            HIPCHK(hipGetLastError());
        } else {
            // The reference visits the junction k-mers in k-mer-file order (AddStartDeEdges, debruijn_graph_constructor.hpp:203-226): they
            // alone — n_junction of D0 k-mers — go through the sort pipeline (EXT records: the byte gives every k-mer its number of start
            // de-edges), and the de-edges of a k-mer are numbered from the prefix sum at its place in that file.
            // (the junction k-mers that have start de-edges: after an early tip clipper most k-mers of a 30x data set with 1 % errors are ISOLATED — junction
            // k-mers by the mask rule, 2.6 G of 4.3 G at config 3 — and sorting those for nothing was 41 + 21 GB and most of the stage's time)
            unsigned long long nj_ = 0;
            HIPCHK(hipMemcpyAsync(&nj_, tjoff + ntiles, 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            const uint64_t nj = nj_;
            if (int rc = dalloc(ctx, &qidx, C)) return rc;
            if (int rc = dalloc(ctx, &vq, C + 1)) return rc;
            Rec<NW> *jrecs, *jk;
            uint8_t *jm;
            unsigned long long *jstats, *jcnt, *candoff, *qbase, *jrank_of, *coff;
            if (int rc = dalloc(ctx, &jrecs, nj + 1)) return rc;
            if (int rc = dalloc(ctx, &qbase, nj + 1)) return rc;
            if (int rc = dalloc(ctx, &jrank_of, nj + 1)) return rc;
            if (int rc = dalloc(ctx, &coff, nj + 2)) return rc;
            tbegin(ctx, "junctions");
            hipLaunchKernelGGL((k_pm_junc_write<NW>), dim3((unsigned)ntiles), dim3(BLK), 0, ctx->stream, (const uint8_t *)ctx->g_mask, (const void *)ctx->g_kmers,
                               (const unsigned long long *)tjoff, D0, (void *)jrecs, jrank_of);
            HIPCHK(hipGetLastError());
            const bool nx = pm->ix.xs == 0;  // plain k-mer records: the bytes of the junction k-mers come from the graph's mask array (smx_pm.hip)
            const bool bym = pm->ix.bym != 0;  // ... and so they do where an early clipper has edited the masks since the records were written (EXT records, stale bytes)
            hipLaunchKernelGGL((k_pm_cand_counts_node<NW>), dim3(grid_for(nj)), dim3(BLK), 0, ctx->stream, (const void *)jrecs, nj, coff,
                               bym ? (const uint8_t *)ctx->g_mask : (const uint8_t *)nullptr, (const unsigned long long *)jrank_of);
            HIPCHK(hipGetLastError());
            if (int rc = scan_u64(ctx, coff, coff, nj)) return rc;
            tend(ctx);
            {
                Prefix pf(ctx, "jsort:");
                ctx->ext_mode = !nx;
                const int rc = run_count<NW>(ctx, k, SMX_MODE_ALL, ctx->g_B, jrecs, nj, nullptr, /*recs_reusable=*/false, false, /*distinct_hint=*/true);
                ctx->ext_mode = false;
                if (rc) return rc;
            }
            const Rec<NW> *jsorted = (const Rec<NW> *)ctx->d_result_buf;
            const std::vector<uint64_t> jboff = ctx->bucket_off;
            const uint64_t nj2 = ctx->n_records;
            ctx->d_result_buf = ctx->d_result = nullptr;  // stays in the temp list
            ctx->n_records = 0;
            if (nj2 != nj) return fail(ctx, SMX_DEVICE_ERROR, "junction k-mers: %llu after the sort, %llu before", (unsigned long long)nj2, (unsigned long long)nj);
            if (nx) jk = const_cast<Rec<NW> *>(jsorted);  // (the sorted records ARE the k-mers)
            else if (int rc = dalloc(ctx, &jk, nj + 1)) return rc;
            if (int rc = dalloc(ctx, &jm, nj + 16)) return rc;
            if (int rc = dalloc(ctx, &jstats, 3)) return rc;
            if (int rc = dalloc(ctx, &jcnt, nj + 1)) return rc;
            if (int rc = dalloc(ctx, &candoff, nj + 2)) return rc;
            HIPCHK(hipMemsetAsync(jstats, 0, 24, ctx->stream));
            tbegin(ctx, "junction_order");
            if (!nx) {
                hipLaunchKernelGGL((k_ext_split<NW>), dim3(grid_for(nj)), dim3(BLK), 0, ctx->stream, (const void *)jsorted, nj, k, (void *)jk, jm, jstats);
                HIPCHK(hipGetLastError());
            }
            smx::RankDir jix{};
            if (int rc = build_rank_dir<NW>(ctx, jk, nj, jboff, ctx->g_B, k, jix)) return rc;
            if (bym) {  // the bytes reach the sorted order through the rank lookups themselves (k_pm_jrank_nx1), then the counts, then the numbers
                hipLaunchKernelGGL((k_pm_jrank_nx1<NW>), dim3(grid_for(nj)), dim3(BLK), 0, ctx->stream, (const void *)jrecs, nj, (const void *)jk, jix,
                                   (const uint8_t *)ctx->g_mask, (const unsigned long long *)jrank_of, qbase, jm, d_err, pm->ix.xs);
            }
            hipLaunchKernelGGL(k_pm_cand_counts, dim3(grid_for(nj)), dim3(BLK), 0, ctx->stream, (const uint8_t *)jm, nj, jcnt);
            if (int rc = scan_u64(ctx, jcnt, candoff, nj)) {
                drop_rank_dir(ctx, jix);
                return rc;
            }
            if (bym)
                hipLaunchKernelGGL(k_pm_jrank_nx2, dim3(grid_for(nj)), dim3(BLK), 0, ctx->stream, nj, (const unsigned long long *)candoff, qbase);
            else
                hipLaunchKernelGGL((k_pm_jrank<NW>), dim3(grid_for(nj)), dim3(BLK), 0, ctx->stream, (const void *)jrecs, nj, (const void *)jk, jix,
                                   (const unsigned long long *)candoff, qbase, d_err);
            hipLaunchKernelGGL((k_pm_cand_expand<NW>), dim3(grid_for(nj)), dim3(BLK), 0, ctx->stream, (const void *)jrecs, (const unsigned long long *)jrank_of, nj,
                               (const unsigned long long *)coff, (const unsigned long long *)qbase, cand, qidx,
                               bym ? (const uint8_t *)ctx->g_mask : (const uint8_t *)nullptr);
            unsigned long long ctot = 0;
            hipError_t e1 = hipGetLastError();
            if (e1 == hipSuccess) e1 = hipMemcpyAsync(&ctot, candoff + nj, 8, hipMemcpyDeviceToHost, ctx->stream);
            if (e1 == hipSuccess) e1 = hipStreamSynchronize(ctx->stream);
            tend(ctx);
            drop_rank_dir(ctx, jix);
            if (e1 != hipSuccess) return fail(ctx, SMX_DEVICE_ERROR, "numbering the start de-edges failed: %s", hipGetErrorString(e1));
            if (ctot != C) return fail(ctx, SMX_DEVICE_ERROR, "start de-edges: %llu by the sorted junction k-mers, %llu by the masks", ctot, C);
        }
        if (tab_ready)
            if (int rc = (*tab_ready)()) return rc;
        tbegin(ctx, "walk_len");
        if (pm)
            hipLaunchKernelGGL((k_pm_walk_len<NW>), dim3(cgrid), dim3(BLK), 0, ctx->stream, (const unsigned long long *)cand, (uint64_t)C, pm->ix, pm->cinfo, pm->cob,
                               pm->nchunks, (const node_t *)tab, pm->jmp, k, (uint64_t)(2 * D0), len, first, last, flags, d_err);
        else
            hipLaunchKernelGGL((k_walk_len<NW>), dim3(cgrid), dim3(BLK), 0, ctx->stream, (const unsigned long long *)cand, (uint64_t)C,
                               (const void *)ctx->g_kmers, (const node_t *)tab, k, ixk, (uint64_t)(2 * D0), len, first, last, d_err);
        HIPCHK(hipGetLastError());
        tend(ctx);
        tbegin(ctx, "keep");
        if (pm) {
            HIPCHK(hipMemsetAsync(vq, 0, (size_t)(C + 1) * 8, ctx->stream));
            hipLaunchKernelGGL((k_pm_keep<NW>), dim3(cgrid), dim3(BLK), 0, ctx->stream, (const unsigned long long *)cand, (const unsigned long long *)qidx, (uint64_t)C,
                               (const void *)ctx->g_kmers, (const node_t *)tab, k, (const unsigned long long *)len, (const node_t *)first, flags, vq, counters + 1,
                               pm->ix.xs, pack_shift);
            HIPCHK(hipGetLastError());
            if (!pack_shift) hipLaunchKernelGGL(k_pm_unpack, dim3(cgrid), dim3(BLK), 0, ctx->stream, (const unsigned long long *)vq, (uint64_t)C, kw, one);  // kw / one: indexed by q
        } else {
            hipLaunchKernelGGL((k_keep<NW>), dim3(cgrid), dim3(BLK), 0, ctx->stream, (const unsigned long long *)cand, (uint64_t)C,
                               (const void *)ctx->g_kmers, (const node_t *)tab, k, (const unsigned long long *)len,
                               (const node_t *)first, (const node_t *)last, flags, kw, one, counters + 1);
        }
        HIPCHK(hipGetLastError());
        // word offsets and edge indices of the kept paths (scans in place: kw -> woff, one -> eidx)
        unsigned long long tw = 0, nk = 0;
        if (pack_shift) {  // one scan: word offset and edge index of a kept path side by side in one word
            if (int rc = scan_u64(ctx, vq, vq, C)) return rc;
            HIPCHK(hipMemcpyAsync(&tw, vq + C, 8, hipMemcpyDeviceToHost, ctx->stream));
        } else {
            if (int rc = scan_u64(ctx, kw, kw, C)) return rc;
            if (int rc = scan_u64(ctx, one, one, C)) return rc;
            HIPCHK(hipMemcpyAsync(&tw, kw + C, 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipMemcpyAsync(&nk, one + C, 8, hipMemcpyDeviceToHost, ctx->stream));
        }
        HIPCHK(hipMemcpyAsync(&interior, counters + 1, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        tend(ctx);
        if (pack_shift) {
            nk = tw >> pack_shift;
            tw &= (1ull << pack_shift) - 1;
        }
        nkept = nk;
        ktotalw = tw;
        if (int rc = dalloc(ctx, &ctx->g_uwords, ktotalw + 8, false)) return rc;
        if (int rc = dalloc(ctx, &ctx->g_eoffw, nkept + 1, false)) return rc;
        if (int rc = dalloc(ctx, &ctx->g_elen, nkept + 1, false)) return rc;
        if (int rc = dalloc(ctx, &ctx->g_estart, nkept + 1, false)) return rc;
        if (int rc = dalloc(ctx, &ctx->g_eend, nkept + 1, false)) return rc;
        if (int rc = dalloc(ctx, &ctx->g_eself, nkept + 1, false)) return rc;
        HIPCHK(hipMemsetAsync(ctx->g_uwords + ktotalw, 0, 64, ctx->stream));
        tbegin(ctx, "walk_write");
        if (pm) {
            ulonglong4 *erec;
            if (int rc = dalloc(ctx, &erec, nkept + 1)) return rc;
            hipLaunchKernelGGL((k_pm_walk_write<NW>), dim3(cgrid), dim3(BLK), 0, ctx->stream, (const unsigned long long *)cand, (const unsigned long long *)qidx,
                               (uint64_t)C, (const void *)ctx->g_kmers, (const node_t *)tab, pm->jmp, k, (const unsigned long long *)len, (const node_t *)first,
                               (const node_t *)last, (const uint8_t *)flags, (const unsigned long long *)(pack_shift ? vq : kw), (const unsigned long long *)one, ctx->g_uwords, erec,
                               pm->ix.xs, pack_shift);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(k_pm_edges, dim3(grid_for(nkept)), dim3(BLK), 0, ctx->stream, (const ulonglong4 *)erec, (uint64_t)nkept, ctx->g_eoffw, ctx->g_elen,
                               ctx->g_estart, ctx->g_eend, ctx->g_eself);
        } else
            hipLaunchKernelGGL((k_walk_write<NW>), dim3(cgrid), dim3(BLK), 0, ctx->stream, (const unsigned long long *)cand, (uint64_t)C,
                               (const void *)ctx->g_kmers, (const node_t *)tab, k, (const unsigned long long *)len,
                               (const node_t *)first, (const node_t *)last, (const uint8_t *)flags, (const unsigned long long *)kw,
                               (const unsigned long long *)one, ctx->g_uwords, ctx->g_eoffw, ctx->g_elen, ctx->g_estart, ctx->g_eend, ctx->g_eself);
        HIPCHK(hipGetLastError());
        tend(ctx);
    } else {
        if (tab_ready)
            if (int rc = (*tab_ready)()) return rc;
        if (int rc = dalloc(ctx, &ctx->g_uwords, 8, false)) return rc;
        if (int rc = dalloc(ctx, &ctx->g_eoffw, 1, false)) return rc;
        if (int rc = dalloc(ctx, &ctx->g_elen, 1, false)) return rc;
        if (int rc = dalloc(ctx, &ctx->g_estart, 1, false)) return rc;
        if (int rc = dalloc(ctx, &ctx->g_eend, 1, false)) return rc;
        if (int rc = dalloc(ctx, &ctx->g_eself, 1, false)) return rc;
        HIPCHK(hipMemsetAsync(ctx->g_uwords, 0, 64, ctx->stream));
    }
    ctx->g_ne = ctx->g_npaths = nkept;
    ctx->g_nuwords = ktotalw;
    gwt.mark(ctx, "g:walks");
    // ---- perfect loops: non-junction k-mers on no path (CollectLoops, :359-397; serial in the reference too) ----
    // Every non-junction k-mer lies on exactly one kept path or on a perfect loop: the count of k_keep tells whether there is any.
    if (D0 - n_junction != interior && ctx->opt_keep_loops) {
        uint8_t *visited;
        unsigned long long *lcount = counters + 2;
        if (int rc = dalloc(ctx, &visited, D0 + 8)) return rc;
        HIPCHK(hipMemsetAsync(visited, 0, D0 + 8, ctx->stream));
        if (C > 0) {
            hipLaunchKernelGGL(k_walk_mark, dim3(grid_for(C)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)cand, (uint64_t)C, (const node_t *)tab, k,
                               (const unsigned long long *)len, (const node_t *)first, (const uint8_t *)flags, visited);
            HIPCHK(hipGetLastError());
        }
        hipLaunchKernelGGL(k_loop_count, dim3(grid_for(D0, 4096)), dim3(BLK), 0, ctx->stream, (const uint8_t *)ctx->g_mask, (const uint8_t *)visited, D0, lcount);
        HIPCHK(hipGetLastError());
        unsigned long long nloopk = 0;
        HIPCHK(hipMemcpyAsync(&nloopk, lcount, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (nloopk) {
            unsigned long long *llist;
            if (int rc = dalloc(ctx, &llist, nloopk)) return rc;
            hipLaunchKernelGGL(k_loop_list, dim3(grid_for(D0)), dim3(BLK), 0, ctx->stream, (const uint8_t *)ctx->g_mask, (const uint8_t *)visited, D0,
                               lcount + 1, llist, nloopk);
            HIPCHK(hipGetLastError());
            int lrc = SMX_ROUTE_NA;  // (an even k has k-mers that are their own reverse complement: one node for two strands — host)
            if (ctx->opt_device_loops > 0 && (k & 1)) lrc = device_loops<NW>(ctx, k, pm, llist, nloopk, (const node_t *)tab, D0, nkept, ktotalw);
            if (lrc != 0 && lrc != SMX_ROUTE_NA) return lrc;
            if (lrc == SMX_ROUTE_NA) {
            std::vector<unsigned long long> ranks;
            if (int rc = d2h(ctx, ranks, llist, nloopk)) return rc;
            std::sort(ranks.begin(), ranks.end());  // k-mer-file order
            HIPCHK(hipMemcpy(llist, ranks.data(), (size_t)nloopk * 8, hipMemcpyHostToDevice));
            Rec<NW> *lk;
            uint8_t *lm;
            if (int rc = dalloc(ctx, &lk, nloopk)) return rc;
            if (int rc = dalloc(ctx, &lm, nloopk)) return rc;
            hipLaunchKernelGGL((k_gather_kmers<NW>), dim3(grid_for(nloopk)), dim3(BLK), 0, ctx->stream, (const void *)ctx->g_kmers,
                               (const uint8_t *)ctx->g_mask, (const unsigned long long *)llist, (uint64_t)nloopk, (void *)lk, lm);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(ctx->stream));
            std::vector<uint64_t> hk;
            if (int rc = d2h(ctx, hk, lk, (size_t)nloopk * NW)) return rc;
            std::vector<uint8_t> hmask;
            if (int rc = d2h(ctx, hmask, lm, (size_t)nloopk)) return rc;
            std::vector<uint64_t> order(nloopk);
            for (uint64_t i = 0; i < nloopk; ++i) order[i] = i;
            if (pm) {  // EXT records in partition-major order: drop the byte, then k-mer-file order = (bucket, words) on the host
                if (!ctx->g_pm_nx)
                    for (uint64_t i = 0; i < nloopk; ++i) hk[(size_t)i * NW + NW - 1] >>= smx::EXT_BITS;
                std::vector<uint32_t> bk(nloopk);
                for (uint64_t i = 0; i < nloopk; ++i) bk[i] = pm_host_bucket(&hk[(size_t)i * NW], NW, ctx->g_B);
                std::sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) {
                    if (bk[a] != bk[b]) return bk[a] < bk[b];
                    for (int w = 0; w < NW; ++w)
                        if (hk[(size_t)a * NW + w] != hk[(size_t)b * NW + w]) return hk[(size_t)a * NW + w] < hk[(size_t)b * NW + w];
                    return false;
                });
            }
            std::vector<uint64_t> fk((size_t)nloopk * NW), fr(nloopk);  // in k-mer-file order
            std::vector<uint8_t> fm(nloopk);
            for (uint64_t t = 0; t < nloopk; ++t) {
                const uint64_t i = order[t];
                fr[t] = ranks[i];
                for (int w = 0; w < NW; ++w) fk[(size_t)t * NW + w] = hk[(size_t)i * NW + w];
                fm[t] = hmask[i];
            }
            if (int rc = append_loops(ctx, k, fk.data(), fr.data(), fm.data(), nloopk, nkept, ktotalw)) return rc;
            }
        }
    }
    unsigned herr = 0;
    HIPCHK(hipMemcpy(&herr, d_err, 4, hipMemcpyDeviceToHost));
    if (herr) return fail(ctx, SMX_DEVICE_ERROR, "inconsistent k-mer index: %u failed lookups/walks", herr);
    gwt.mark(ctx, "g:loops");
    free_temps(ctx);  // walk buffers are no longer needed; the link sort reuses the arena
    ctx->g_dev_valid = true;
    {
        // total nucleotides (graph info) without a device pass: elen is summed where the host mirror is built; keep a device sum here
        unsigned long long *one1, *tot;
        const uint64_t ne = ctx->g_ne;
        if (int rc = dalloc(ctx, &one1, ne + 1)) return rc;
        (void)tot;
        if (ne) {
            if (int rc = scan_u64(ctx, ctx->g_elen, one1, ne)) return rc;
            unsigned long long nb = 0;
            HIPCHK(hipMemcpyAsync(&nb, one1 + ne, 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            ctx->g_nbases = nb;
        } else {
            ctx->g_nbases = 0;
        }
        free_temps(ctx);
    }
    // ---- 5. link records + vertices ---------------------------------------------------------------
    const bool host_links = ctx->opt_sort_edges || ctx->g_ne == 0 || ctx->opt_device_links == 0 || (ctx->g_ne < (1u << 16) && ctx->opt_device_links < 2);
    if (host_links) {
        if (int rc = materialize_host(ctx)) return rc;
        if (ctx->opt_sort_edges) {
            smxh::sort_edges_raw(ctx->gh);
            if (int rc = upload_graph(ctx)) return rc;  // the device copy follows the new edge order (coverage walks it)
        }
        int sort_rc = 0;
        smxh::build_links(ctx->gh, [&](std::vector<uint64_t> &keys) {
            if (!sort_rc) sort_rc = device_sort_u64(ctx, keys);
            if (sort_rc) smxh::radix_sort_u64(keys);
        });
        if (sort_rc) return sort_rc;
    } else {
        Prefix pf(ctx, "links:");
        if (int rc = device_build_links(ctx, D0)) return rc;
        free_temps(ctx);
    }
    gwt.mark(ctx, "g:links");
    ctx->g_ready = true;
    return 0;
}

template <int NW>
int run_graph(smx_ctx *ctx, unsigned k, unsigned B, const void *kpo_recs = nullptr, uint64_t n_kpo_recs = 0) {
    clear_graph(ctx);
    WallTrace gwt;
    ctx->g_k = k;
    ctx->g_nw = NW;
    ctx->g_B = B;
    ctx->gh.k = k;
    ctx->gh.eoff.assign(1, 0);
    struct Prefix {  // stage names of the pipeline runs below tell which part of the construction they belong to
        smx_ctx *c;
        Prefix(smx_ctx *c_, const char *p) : c(c_) { c->tprefix = p; }
        ~Prefix() { c->tprefix.clear(); }
    };
    // ---- 0. k-mers and masks from one count of the reads, when that applies ---------------------
    for (auto &v : ctx->g_route_stats) v = 0;
    if (!kpo_recs) {
        int rc = pm_route<NW>(ctx, k, B, gwt);  // no sorted k-mer file at all where that applies (smx_pm.hpp)
        if (rc != SMX_ROUTE_NA) return rc;
        for (auto &v : ctx->g_route_stats) v = 0;
        ctx->g_k = k;  // (a route that gave up cleared the graph state)
        ctx->g_nw = NW;
        ctx->g_B = B;
        ctx->gh.k = k;
        ctx->gh.eoff.assign(1, 0);
        {
            Prefix pf(ctx, "kmers:");
            rc = kmer_file_with_masks<NW>(ctx, k, B);
        }
        if (rc == 0) {
            gwt.mark(ctx, "g:kmers+masks");
            ctx->g_route_stats[0] = 1;
            const uint64_t D0 = ctx->g_nkmers;
            if (D0 == 0) {
                ctx->g_host_valid = true;  // the empty graph
                ctx->g_ready = true;
                ctx->n_records = 0;
                ctx->K = k;
                ctx->bucket_off.assign(B + 1, 0);
                return 0;
            }
            ctx->d_result = ctx->g_kmers;
            if (D0 >= (1ull << 60)) return fail(ctx, SMX_MEMORY_LIMIT_EXCEEDED, "%llu k-mers exceed the node-id range", (unsigned long long)D0);
            tbegin(ctx, "rank_dir");
            if (int rc2 = build_rank_dir<NW>(ctx, ctx->g_kmers, D0, ctx->g_kboff, B, k, ctx->g_dir_kmers)) return rc2;
            tend(ctx);
            uint32_t *d_err;
            if (int rc2 = dalloc(ctx, &d_err, 1)) return rc2;
            HIPCHK(hipMemsetAsync(d_err, 0, 4, ctx->stream));
            node_t *tab;
            if (int rc2 = dalloc(ctx, &tab, 2 * D0 + 2)) return rc2;
            return graph_from_masks<NW>(ctx, k, tab, /*tab_valid=*/false, d_err, gwt, /*present=*/true);
        }
        if (rc != SMX_ROUTE_NA) return rc;
    }
    // ---- 1. canonical (k+1)-mers -------------------------------------------------------------
    ctx->g_route_stats[0] = 2;
    if (kpo_recs) {  // multi-GPU: the (k+1)-mer file gathered from its owner ranks (any order; re-sorted here)
        if (int rc = run_count<NW>(ctx, k + 1, SMX_MODE_ALL, B, kpo_recs, n_kpo_recs)) return rc;
    } else {
        if (int rc = count_reads<NW>(ctx, k + 1, SMX_MODE_CANONICAL, B)) return rc;
    }
    if (ctx->result_on_host) return fail(ctx, SMX_MEMORY_LIMIT_EXCEEDED, "the (k+1)-mer file does not fit the HBM budget (it was spilled to the host): the construction needs it resident");
    ctx->g_kpo = ctx->d_result_buf;
    ctx->d_result_buf = ctx->d_result = nullptr;
    if (int rc = adopt_result(ctx, &ctx->g_kpo, (size_t)ctx->n_records * NW * 8)) return rc;
    ctx->g_nkpo = ctx->n_records;
    ctx->g_kpoboff = ctx->bucket_off;
    const uint64_t nkpo = ctx->g_nkpo;
    ctx->g_kboff.assign(B + 1, 0);
    gwt.mark(ctx, "g:kpo count");
    if (nkpo == 0) {
        ctx->g_host_valid = true;  // the empty graph
        ctx->g_ready = true;
        ctx->n_records = 0;
        ctx->K = k;
        ctx->bucket_off.assign(B + 1, 0);
        return 0;
    }
    // ---- 2. canonical k-mers in k-mer-file order ----------------------------------------------
    {
        Prefix pf(ctx, "kmers:");
        if (int rc = derive_kmer_file<NW>(ctx, k, B, /*from_reads=*/kpo_recs == nullptr)) return rc;
    }
    ctx->d_result = ctx->g_kmers;  // smx_copy_final_kmers() now yields the k-mer file
    gwt.mark(ctx, "g:kmer file");
    const uint64_t D0 = ctx->g_nkmers;
    if (D0 >= (1ull << 60)) return fail(ctx, SMX_MEMORY_LIMIT_EXCEEDED, "%llu k-mers exceed the node-id range", (unsigned long long)D0);
    const unsigned grid = grid_for(2 * D0);
    tbegin(ctx, "rank_dir");
    if (int rc = build_rank_dir<NW>(ctx, ctx->g_kmers, D0, ctx->g_kboff, B, k, ctx->g_dir_kmers)) return rc;
    tend(ctx);
    // ---- 3. node table (extensions + successors) and the InOutMask bytes ------------------------
    uint32_t *d_err;
    if (int rc = dalloc(ctx, &d_err, 1)) return rc;
    HIPCHK(hipMemsetAsync(d_err, 0, 4, ctx->stream));
    const smx::RankDir ixk = ctx->g_dir_kmers;
    const size_t mask_bytes = (size_t)((D0 + 7) / 8 * 8 + 8);
    if (int rc = dalloc(ctx, &ctx->g_mask, mask_bytes, false)) return rc;
    node_t *tab;
    if (int rc = dalloc(ctx, &tab, 2 * D0 + 2)) return rc;
    HIPCHK(hipMemsetAsync(tab, 0, (size_t)(2 * D0 + 2) * 8, ctx->stream));
    tbegin(ctx, "fill_masks");
    hipLaunchKernelGGL((k_fill_tab<NW>), dim3(grid_for(nkpo)), dim3(BLK), 0, ctx->stream, (const void *)ctx->g_kpo, nkpo, k, (const void *)ctx->g_kmers,
                       ixk, tab, d_err);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_tab_masks, dim3(grid_for(D0)), dim3(BLK), 0, ctx->stream, (const node_t *)tab, D0, ctx->g_mask);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(ctx->g_mask + D0, 0, mask_bytes - D0, ctx->stream));
    tend(ctx);
    {
        // the (k+1)-mer file is only needed again by -c; when HBM is short it goes now and the coverage pass recounts it
        const size_t later = (size_t)D0 * 40;  // candidate and edge arrays of the walks, generously
        const bool keep = ctx->opt_keep_kpo > 0 || (ctx->opt_keep_kpo < 0 && arena_avail(ctx) > later);
        if (!keep) {
            HIPCHK(hipStreamSynchronize(ctx->stream));
            drop_kpo(ctx);
        }
    }
    return graph_from_masks<NW>(ctx, k, tab, /*tab_valid=*/true, d_err, gwt);
}

// ---- sharded construction: owner-side mask fill, replicated compact structure --------------------------------
// 1. extension updates of this rank's (k+1)-mer shard (= the context's current count result), grouped by the owner rank of the k-mer
template <int NW>
int shard_updates(smx_ctx *ctx, unsigned k, unsigned B, unsigned world, void *d_out, uint64_t capacity, uint64_t *counts) {
    const uint64_t n = ctx->n_records;
    if (2 * n > capacity) return fail(ctx, SMX_MEMORY_LIMIT_EXCEEDED, "update buffer too small: need %llu records", (unsigned long long)(2 * n));
    unsigned long long *hist, *off, *cur;
    if (int rc = dalloc(ctx, &hist, world)) return rc;
    if (int rc = dalloc(ctx, &off, world + 1)) return rc;
    if (int rc = dalloc(ctx, &cur, world)) return rc;
    HIPCHK(hipMemsetAsync(hist, 0, (size_t)world * 8, ctx->stream));
    const size_t lds = (size_t)world * 16;
    if (n) {
        hipLaunchKernelGGL((k_upd_partition<NW, 0>), dim3(grid_for(n, 4096)), dim3(BLK), lds, ctx->stream, (const void *)ctx->d_result, n, k, B, world, hist, (void *)nullptr);
        HIPCHK(hipGetLastError());
    }
    if (int rc = scan_u64(ctx, hist, off, world)) return rc;
    HIPCHK(hipMemcpyAsync(cur, off, (size_t)world * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (n) {
        hipLaunchKernelGGL((k_upd_partition<NW, 1>), dim3(grid_for(n, 4096)), dim3(BLK), lds, ctx->stream, (const void *)ctx->d_result, n, k, B, world, cur, d_out);
        HIPCHK(hipGetLastError());
    }
    std::vector<unsigned long long> h(world);
    HIPCHK(hipMemcpyAsync(h.data(), hist, (size_t)world * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (unsigned i = 0; i < world; ++i) counts[i] = h[i];
    return 0;
}
// 2. owner side: the k-mers this rank owns (sorted-unique, its bucket range of the k-mer file) and their InOutMask bytes
template <int NW>
int shard_build(smx_ctx *ctx, unsigned k, unsigned B, unsigned world, unsigned rank, const void *d_upd, uint64_t n) {
    clear_graph(ctx);
    ctx->g_k = k;
    ctx->g_nw = NW;
    ctx->g_B = B;
    ctx->g_kboff.assign(B + 1, 0);
    const unsigned b0 = (unsigned)(((uint64_t)rank * B + world - 1) / world), b1 = (unsigned)(((uint64_t)(rank + 1) * B + world - 1) / world);
    if (n == 0) return 0;
    Rec<NW> *recs;
    if (int rc = dalloc(ctx, &recs, n)) return rc;
    hipLaunchKernelGGL((k_upd_strip<NW>), dim3(grid_for(n)), dim3(BLK), 0, ctx->stream, d_upd, n, (void *)recs);
    HIPCHK(hipGetLastError());
    if (int rc = run_count<NW>(ctx, k, SMX_MODE_ALL, B, recs, n, nullptr, /*recs_reusable=*/true, false, false, b0, std::max(1u, b1 - b0))) return rc;
    ctx->g_kmers = ctx->d_result_buf;
    ctx->g_nkmers = ctx->n_records;
    ctx->g_kboff = ctx->bucket_off;
    ctx->d_result_buf = ctx->d_result = nullptr;
    if (int rc = adopt_result(ctx, &ctx->g_kmers, (size_t)ctx->g_nkmers * NW * 8)) return rc;
    const uint64_t D = ctx->g_nkmers;
    if (int rc = build_rank_dir<NW>(ctx, ctx->g_kmers, D, ctx->g_kboff, B, k, ctx->g_dir_kmers)) return rc;
    const size_t mask_bytes = (size_t)((D + 7) / 8 * 8 + 8);
    if (int rc = dalloc(ctx, &ctx->g_mask, mask_bytes, false)) return rc;
    HIPCHK(hipMemsetAsync(ctx->g_mask, 0, mask_bytes, ctx->stream));
    uint32_t *d_err;
    if (int rc = dalloc(ctx, &d_err, 1)) return rc;
    HIPCHK(hipMemsetAsync(d_err, 0, 4, ctx->stream));
    hipLaunchKernelGGL((k_upd_apply<NW>), dim3(grid_for(n)), dim3(BLK), 0, ctx->stream, d_upd, n, (const void *)ctx->g_kmers, ctx->g_dir_kmers,
                       (uint32_t *)ctx->g_mask, d_err);
    HIPCHK(hipGetLastError());
    unsigned herr = 0;
    HIPCHK(hipMemcpy(&herr, d_err, 4, hipMemcpyDeviceToHost));
    if (herr) return fail(ctx, SMX_DEVICE_ERROR, "%u extension updates did not find their k-mer in this rank's shard", herr);
    free_temps(ctx);
    return 0;
}
// 3. the graph from the gathered compact structure {k-mer file, InOutMask bytes} (replicated on every rank)
// 2'. owner side of the one-exchange route: the records received are canonical k-mers in the EXT layout, every rank's copy with the
// extension byte ITS reads gave the k-mer; sort (exact copies drop out), OR the bytes of the copies, split into file + bytes
template <int NW>
int shard_from_ext(smx_ctx *ctx, unsigned k, unsigned B, unsigned world, unsigned rank, const void *d_recs, uint64_t n, bool recs_reusable) {
    clear_graph(ctx);
    ctx->g_k = k;
    ctx->g_nw = NW;
    ctx->g_B = B;
    ctx->g_kboff.assign(B + 1, 0);
    ctx->g_ext_bits = ctx->g_ext_pals = 0;
    if (!ext_layout_fits(k, NW)) return fail(ctx, SMX_INVALID_PARAMETER, "k=%u leaves no room for the extension byte", k);
    const unsigned b0 = (unsigned)(((uint64_t)rank * B + world - 1) / world), b1 = (unsigned)(((uint64_t)(rank + 1) * B + world - 1) / world);
    if (n == 0) return 0;
    ctx->ext_mode = true;
    int rc = run_count<NW>(ctx, k, SMX_MODE_ALL, B, d_recs, n, nullptr, recs_reusable, false, false, b0, std::max(1u, b1 - b0));
    ctx->ext_mode = false;
    if (rc) return rc;
    return ext_result_to_file<NW>(ctx, k, B, /*whole=*/false);
}

template <int NW>
int run_graph_from_kmers(smx_ctx *ctx, unsigned k, unsigned B, const void *d_kmers, const void *d_masks, uint64_t n, const uint64_t *bucket_sizes) {
    clear_graph(ctx);
    WallTrace gwt;
    ctx->g_k = k;
    ctx->g_nw = NW;
    ctx->g_B = B;
    ctx->gh.k = k;
    ctx->gh.eoff.assign(1, 0);
    ctx->g_kboff.assign(B + 1, 0);
    for (unsigned b = 0; b < B; ++b) ctx->g_kboff[b + 1] = ctx->g_kboff[b] + bucket_sizes[b];
    if (ctx->g_kboff[B] != n) return fail(ctx, SMX_INVALID_PARAMETER, "bucket sizes add up to %llu, %llu k-mers given", (unsigned long long)ctx->g_kboff[B], (unsigned long long)n);
    ctx->n_records = n;
    ctx->K = k;
    ctx->nw = NW;
    ctx->num_buckets = B;
    ctx->bucket_off = ctx->g_kboff;
    if (n == 0) {
        ctx->g_host_valid = true;
        ctx->g_ready = true;
        return 0;
    }
    Rec<NW> *file;
    if (int rc = dalloc(ctx, &file, n, false)) return rc;
    ctx->g_kmers = file;
    ctx->g_nkmers = n;
    ctx->d_result = file;
    const size_t mask_bytes = (size_t)((n + 7) / 8 * 8 + 8);
    if (int rc = dalloc(ctx, &ctx->g_mask, mask_bytes, false)) return rc;
    HIPCHK(hipMemsetAsync(ctx->g_mask, 0, mask_bytes, ctx->stream));
    HIPCHK(hipMemcpyAsync(file, d_kmers, (size_t)n * NW * 8, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->g_mask, d_masks, (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
    if (int rc = build_rank_dir<NW>(ctx, ctx->g_kmers, n, ctx->g_kboff, B, k, ctx->g_dir_kmers)) return rc;
    uint32_t *d_err;
    if (int rc = dalloc(ctx, &d_err, 1)) return rc;
    HIPCHK(hipMemsetAsync(d_err, 0, 4, ctx->stream));
    node_t *tab;
    if (int rc = dalloc(ctx, &tab, 2 * n + 2)) return rc;
    return graph_from_masks<NW>(ctx, k, tab, /*tab_valid=*/false, d_err, gwt);
}

// -c: per-(k+1)-mer multiplicities over the resident reads, summed per edge (and over the edge flanks)
template <int NW>
int run_coverage(smx_ctx *ctx) {
    const unsigned K1 = ctx->g_k + 1, B = ctx->g_B;
    const uint64_t D1 = ctx->g_nkpo, ne = ctx->g_ne;
    ctx->gh.ecov.assign(ne, 0);
    ctx->gh.eflank_s.assign(ne, 0);
    ctx->gh.eflank_e.assign(ne, 0);
    if (D1 == 0 || ne == 0) return 0;
    if (!ctx->g_dev_valid)
        if (int rc = upload_graph(ctx)) return rc;
    if (!ctx->g_kpo) {  // dropped to make room for the walks: count the canonical (k+1)-mers again
        void *sv_res = ctx->d_result;
        const uint64_t sv_n = ctx->n_records;
        const unsigned sv_K = ctx->K;
        std::vector<uint64_t> sv_boff = ctx->bucket_off;
        ctx->d_result = nullptr;
        auto restore = [&]() {  // the count-result view of the context describes the graph's k-mer file again (every way out of here)
            ctx->d_result = sv_res;
            ctx->n_records = sv_n;
            ctx->K = sv_K;
            ctx->bucket_off = sv_boff;
        };
        if (int rc = count_reads<NW>(ctx, K1, SMX_MODE_CANONICAL, B)) {
            free_temps(ctx);
            clear_result(ctx);
            restore();
            return rc;
        }
        if (ctx->result_on_host) {
            clear_result(ctx);
            restore();
            return fail(ctx, SMX_MEMORY_LIMIT_EXCEEDED, "the coverage pass needs the (k+1)-mer file resident next to the graph: not enough HBM");
        }
        if (ctx->n_records != D1) {
            const unsigned long long got = ctx->n_records;
            free_temps(ctx);
            clear_result(ctx);
            restore();
            return fail(ctx, SMX_DEVICE_ERROR, "recount of the (k+1)-mers gave %llu records, the graph was built from %llu", got, (unsigned long long)D1);
        }
        ctx->g_kpo = ctx->d_result_buf;
        ctx->g_kpoboff = ctx->bucket_off;
        ctx->d_result_buf = nullptr;
        if (int rc = adopt_result(ctx, &ctx->g_kpo, (size_t)D1 * NW * 8)) {
            restore();
            return rc;
        }
        restore();
        ctx->n_instances = 0;
        free_temps(ctx);
    }
    if (!ctx->g_dir_kpo.dir)
        if (int rc = build_rank_dir<NW>(ctx, ctx->g_kpo, D1, ctx->g_kpoboff, B, K1, ctx->g_dir_kpo)) return rc;
    std::vector<uint64_t *> masks;
    uint64_t nwin = 0;
    if (int rc = mark_windows(ctx, K1, masks, &nwin)) return rc;
    uint32_t *cnt, *ecov, *fls, *fle;
    if (int rc = dalloc(ctx, &cnt, D1)) return rc;
    if (int rc = dalloc(ctx, &ecov, ne)) return rc;
    if (int rc = dalloc(ctx, &fls, ne)) return rc;
    if (int rc = dalloc(ctx, &fle, ne)) return rc;
    HIPCHK(hipMemsetAsync(cnt, 0, D1 * 4, ctx->stream));
    HIPCHK(hipMemsetAsync(ecov, 0, ne * 4, ctx->stream));
    HIPCHK(hipMemsetAsync(fls, 0, ne * 4, ctx->stream));
    HIPCHK(hipMemsetAsync(fle, 0, ne * 4, ctx->stream));
    const smx::RankDir ixp = ctx->g_dir_kpo;
    tbegin(ctx, "kpo_coverage");
    for (size_t ci = 0; ci < ctx->chunks.size(); ++ci) {
        const ReadChunk &ch = ctx->chunks[ci];
        if (ch.n_bases == 0 || !masks[ci] || ch.contigs) continue;  // contigs: "separate stream for not counting it in coverage"
        hipLaunchKernelGGL((k_kpo_coverage<NW>), dim3(grid_for(ch.n_bases)), dim3(BLK), 0,
                           ctx->stream, (const uint64_t *)ch.d_words, (const uint64_t *)masks[ci], ch.n_bases, K1, (const void *)ctx->g_kpo, ixp, cnt);
        HIPCHK(hipGetLastError());
    }
    tend(ctx);
    {   // multiplicity histogram of the (k+1)-mers
        unsigned long long *d_hist, *d_nbig;
        uint32_t *d_big;
        const uint32_t bigcap = 1u << 20;
        if (int rc = dalloc(ctx, &d_hist, COVH_N)) return rc;
        if (int rc = dalloc(ctx, &d_nbig, 1)) return rc;
        if (int rc = dalloc(ctx, &d_big, bigcap)) return rc;
        HIPCHK(hipMemsetAsync(d_hist, 0, (size_t)COVH_N * 8, ctx->stream));
        HIPCHK(hipMemsetAsync(d_nbig, 0, 8, ctx->stream));
        hipLaunchKernelGGL(k_cov_hist, dim3(grid_for(D1, 4096)), dim3(BLK), 0, ctx->stream, (const uint32_t *)cnt, D1, d_hist, d_nbig, d_big, bigcap);
        HIPCHK(hipGetLastError());
        std::vector<unsigned long long> hh(COVH_N);
        unsigned long long nbig = 0;
        HIPCHK(hipMemcpyAsync(hh.data(), d_hist, (size_t)COVH_N * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(&nbig, d_nbig, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (nbig > bigcap) return fail(ctx, SMX_MEMORY_LIMIT_EXCEEDED, "%llu (k+1)-mers occur 65536 times or more: beyond the histogram's overflow list", nbig);
        std::vector<uint32_t> hb;
        if (int rc = d2h(ctx, hb, d_big, (size_t)nbig)) return rc;
        uint64_t mx = 0;
        for (uint32_t c = 0; c < COVH_N; ++c)
            if (hh[c]) mx = c;
        for (uint32_t c : hb) mx = std::max<uint64_t>(mx, c);
        ctx->g_cov_hist.assign(mx + 1, 0);
        for (uint32_t c = 0; c <= std::min<uint64_t>(mx, COVH_N - 1); ++c) ctx->g_cov_hist[c] = hh[c];
        for (uint32_t c : hb) ctx->g_cov_hist[c]++;
    }
    // the unitigs as a read batch: (k+1)-mer windows marked the same way
    tbegin(ctx, "edge_coverage");
    const uint64_t G = ctx->g_nuwords * 32;
    uint64_t *ustart, *wmask;
    uint32_t *ulen;
    unsigned long long *d_total;
    if (int rc = dalloc(ctx, &ustart, ne)) return rc;
    if (int rc = dalloc(ctx, &ulen, ne)) return rc;
    if (int rc = dalloc(ctx, &wmask, G / 64 + 2)) return rc;
    if (int rc = dalloc(ctx, &d_total, 1)) return rc;
    HIPCHK(hipMemsetAsync(wmask, 0, (G / 64 + 2) * 8, ctx->stream));
    HIPCHK(hipMemsetAsync(d_total, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_edge_as_reads, dim3(grid_for(ne)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)ctx->g_eoffw,
                       (const unsigned long long *)ctx->g_elen, ne, ustart, ulen);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_mark_windows, dim3((unsigned)((ne + BLK - 1) / BLK)), dim3(BLK), 0, ctx->stream, (const uint64_t *)ustart, (const uint32_t *)ulen, ne, K1,
                       (unsigned long long *)wmask, d_total);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL((k_edge_coverage<NW>), dim3(grid_for(G)), dim3(BLK), 0, ctx->stream, (const uint64_t *)ctx->g_uwords, (const uint64_t *)wmask,
                       (const unsigned long long *)ctx->g_eoffw, (const unsigned long long *)ctx->g_elen, ne, G, K1, (const void *)ctx->g_kpo, ixp,
                       (const uint32_t *)cnt, ecov, (uint32_t)std::max<int64_t>(ctx->opt_flank_range, 1), fls, fle);
    HIPCHK(hipGetLastError());
    tend(ctx);
    HIPCHK(hipMemcpyAsync(ctx->gh.ecov.data(), ecov, ne * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->gh.eflank_s.data(), fls, ne * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->gh.eflank_e.data(), fle, ne * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
