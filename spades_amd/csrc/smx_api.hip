// spades_amd/csrc/smx_api.hip — C ABI (include/smx.h) + host-side pipeline driving the gfx950 kernels.
// Built by hipcc into libspades_mi355x.so; no torch / no reference headers involved.
#include "../../include/smx.h"
#include "smx_kernels.hip"
#include "smx_superkmer.hip"
#include "smx_ingest.hip"
#include "smx_graph.hip"
#include "smx_graph_host.hpp"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <ctime>
#include <cstdlib>
#include <string>
#include <unordered_map>
#include <vector>

using namespace smx;

namespace {

struct ReadChunk {
    uint64_t *d_words = nullptr;
    uint64_t *d_start = nullptr;
    uint32_t *d_len = nullptr;
    uint64_t n_words = 0, n_reads = 0, n_bases = 0;
    bool owned = true;
    bool contigs = false;  // takes part in the construction, not in the coverage (trusted / previous-k contigs)
};

struct Timing {
    std::string name;
    hipEvent_t e0, e1;
};

}  // namespace

struct smx_ctx {
    int device = 0;
    size_t budget = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::vector<ReadChunk> chunks;
    // result of the last count
    void *d_result_buf = nullptr;  // allocation holding the result
    void *d_result = nullptr;
    uint64_t n_records = 0, n_instances = 0;
    unsigned nw = 0, K = 0, num_buckets = 0;
    std::vector<uint64_t> bucket_off;
    // tuning / test hooks
    int64_t opt_leaf_cap = 0, opt_leaf_target = 0, opt_s1 = -1, opt_s2 = -1, opt_batch_records = 0;
    int64_t opt_sort_edges = 0, opt_keep_loops = 1;
    int64_t opt_flank_range = 50;     // FlankingCoverage averaging range ((k+1)-mers at either end of an edge)
    int64_t opt_submit_contigs = 0;   // reads submitted while this is 1 are contigs: construction yes, coverage no
    int64_t opt_early_at = 0;         // 1: the early A/T remover of the RNA pipelines before the tip clipper
    int64_t opt_early_tip_bound = 0;  // > 0: spades-core's early tip clipper with this length bound (RL - K there) before condensation
    int64_t opt_skm_stage = 1;     // pass 0 of the super-k-mer scan stages its output so that placing it needs no second scan
    int64_t opt_device_links = 1;  // link records + vertices of the graph on the device (0: host, 2: also for tiny graphs)
    int64_t opt_joint_hist = 1;  // fuse the level-2 histogram into the level-1 histogram pass (records source)
    int64_t opt_prededupe = -1;  // super-k-mer pre-deduplication: -1 auto, 0 off, 1 on whenever K allows it
    int64_t opt_skm_cap = 0;     // instances per LDS dedupe chunk (0 = default)
    int64_t opt_skm_scap = 0;    // slots staged per chunk (0 = default)
    int64_t opt_leaf_grid = 0, opt_leaf_tab = 0;  // tuning experiments (tools/sweep.py)  // spades-core construction variant (debruijn_graph_constructor.hpp:590-604)
    // timings
    std::vector<Timing> timings;
    std::vector<std::string> tnames, xnames;  // last count stages / last extract_partition stages
    std::vector<float> tms, xms;
    std::vector<void *> temps;  // allocations of the pipeline in flight
    // grow-only device arena: blocks are recycled across calls (hipMalloc/hipFree of tens of GB stalls for seconds)
    std::vector<std::pair<void *, size_t>> arena_free;  // cached blocks
    std::unordered_map<void *, size_t> arena_size;      // every live or cached block -> bytes
    // construction state (smx_build_graph)
    void *g_kpo = nullptr, *g_kmers = nullptr;
    uint8_t *g_mask = nullptr;
    uint64_t g_nkpo = 0, g_nkmers = 0;
    unsigned g_k = 0, g_nw = 0, g_B = 0;
    std::vector<uint64_t> g_kboff, g_kpoboff;
    // fine-bin offsets of the last pipeline run (rank lookups of the construction stage), kept when want_index is set
    bool want_index = false;
    unsigned long long *last_idx_off = nullptr;
    uint64_t last_idx_bins = 0;
    uint32_t last_idx_S1 = 1;
    std::vector<uint32_t> last_idx_f;
    smx::RankIndex g_ix_kmers{}, g_ix_kpo{};  // .off owned by the graph state
    bool g_ready = false;
    uint64_t g_tip_kmers = 0, g_tips = 0;  // early tip clipper: k-mers isolated, tips removed
    uint64_t g_at_edges = 0, g_at_tip_kmers = 0;  // early A/T remover: length-1 edges marked, tip k-mers isolated
    smxh::GraphHost gh;
};

namespace {

int fail(smx_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define HIPCHK(call)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            return fail(ctx, e_ == hipErrorOutOfMemory ? SMX_MEMORY_LIMIT_EXCEEDED : SMX_DEVICE_ERROR,        \
                        "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);           \
    } while (0)

void *arena_get(smx_ctx *ctx, size_t bytes) {
    // best fit among cached blocks that waste at most 2x
    size_t best = (size_t)-1, bi = 0;
    for (size_t i = 0; i < ctx->arena_free.size(); ++i) {
        size_t sz = ctx->arena_free[i].second;
        if (sz >= bytes && sz <= 2 * bytes + (1u << 20) && sz < best) {
            best = sz;
            bi = i;
        }
    }
    if (best != (size_t)-1) {
        void *p = ctx->arena_free[bi].first;
        ctx->arena_free.erase(ctx->arena_free.begin() + bi);
        return p;
    }
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) {  // release the cache and retry once
        (void)hipGetLastError();
        for (auto &b : ctx->arena_free) {
            ctx->arena_size.erase(b.first);
            (void)hipFree(b.first);
        }
        ctx->arena_free.clear();
        e = hipMalloc(&q, bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
    }
    ctx->arena_size[q] = bytes;
    return q;
}
void arena_put(smx_ctx *ctx, void *p) {
    if (!p) return;
    auto it = ctx->arena_size.find(p);
    if (it == ctx->arena_size.end()) {
        (void)hipFree(p);
        return;
    }
    ctx->arena_free.emplace_back(p, it->second);
}
void arena_release(smx_ctx *ctx) {
    for (auto &b : ctx->arena_free) {
        ctx->arena_size.erase(b.first);
        (void)hipFree(b.first);
    }
    ctx->arena_free.clear();
}

template <typename T>
int dalloc(smx_ctx *ctx, T **p, size_t count, bool temp = true) {
    size_t bytes = std::max<size_t>(count * sizeof(T), 256);
    void *q = arena_get(ctx, bytes);
    if (!q) return fail(ctx, SMX_MEMORY_LIMIT_EXCEEDED, "device allocation of %zu bytes failed", bytes);
    if (temp) ctx->temps.push_back(q);
    *p = (T *)q;
    return 0;
}

void free_temps(smx_ctx *ctx, void *keep = nullptr) {
    for (void *p : ctx->temps)
        if (p != keep) arena_put(ctx, p);
    ctx->temps.clear();
}

double wall_now() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}
struct WallTrace {  // SMX_DEBUG=1: host wall-clock per pipeline section (includes allocation / implicit syncs)
    bool on;
    double t;
    WallTrace() : on(getenv("SMX_DEBUG") != nullptr), t(wall_now()) {}
    void mark(smx_ctx *ctx, const char *what) {
        if (!on) return;
        (void)hipStreamSynchronize(ctx->stream);
        double n = wall_now();
        fprintf(stderr, "[smx] %-14s %8.2f ms\n", what, (n - t) * 1e3);
        t = n;
    }
};

void tbegin(smx_ctx *ctx, const char *name) {
    Timing t;
    t.name = name;
    (void)hipEventCreate(&t.e0);
    (void)hipEventCreate(&t.e1);
    (void)hipEventRecord(t.e0, ctx->stream);
    ctx->timings.push_back(t);
}
void tend(smx_ctx *ctx) { (void)hipEventRecord(ctx->timings.back().e1, ctx->stream); }
void tcollect(smx_ctx *ctx) {
    ctx->tnames.clear();
    ctx->tms.clear();
    for (auto &t : ctx->timings) {
        float ms = 0;
        (void)hipEventSynchronize(t.e1);
        (void)hipEventElapsedTime(&ms, t.e0, t.e1);
        ctx->tnames.push_back(t.name);
        ctx->tms.push_back(ms);
        (void)hipEventDestroy(t.e0);
        (void)hipEventDestroy(t.e1);
    }
    ctx->timings.clear();
}

unsigned ceil_log2(uint64_t v) {
    unsigned r = 0;
    while ((1ull << r) < v) ++r;
    return r;
}

template <typename KernelT>
int set_lds(smx_ctx *ctx, KernelT k, size_t bytes) {
    if (bytes > 160 * 1024) return fail(ctx, SMX_INVALID_PARAMETER, "LDS request %zu exceeds 160 KiB", bytes);
    if (bytes > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

// exclusive scan of n u64 values, out has n+1 entries (out[n] = total)
int scan_u64(smx_ctx *ctx, const unsigned long long *in, unsigned long long *out, uint64_t n) {
    if (n <= 8192) {
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(BLK), 0, ctx->stream, in, out, (uint32_t)n);
        HIPCHK(hipGetLastError());
        return 0;
    }
    uint64_t nt = (n + SCAN_TILE - 1) / SCAN_TILE;
    unsigned long long *partial, *poff;
    if (int rc = dalloc(ctx, &partial, nt)) return rc;
    if (int rc = dalloc(ctx, &poff, nt + 1)) return rc;
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nt), dim3(BLK), 0, ctx->stream, in, n, partial);
    HIPCHK(hipGetLastError());
    if (int rc = scan_u64(ctx, partial, poff, nt)) return rc;
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nt), dim3(BLK), 0, ctx->stream, in, n, poff, out);
    HIPCHK(hipGetLastError());
    return 0;
}

template <int NW>
struct Tune {
    static constexpr int RPT = (NW <= 2) ? 16 : 8;                        // records per thread in a scatter tile (RPT 8 at NW=2: L1 scatter 32.5 vs 20.6 ms)
    // LDS leaf classes: CAP1 = the common class (small LDS footprint -> 3-4 workgroups per CU), CAP = 4*CAP1 for skewed bins.
    // Sweep at 10 M reads (tools/sweep.py): k=55 cap 2048/avg 915 -> 40.6 ms, cap 1024/avg 915 -> 20.2 ms; k=21 cap 4096 -> 72.7, 2048 -> 27.9 ms
    static constexpr uint32_t CAP1 = (NW == 1) ? 2048 : (NW == 2 ? 1024 : 512);
    static constexpr uint32_t CAP = (NW == 1 ? 2 : 4) * CAP1;  // second class must still fit 160 KiB of LDS
    static constexpr int LPT1 = CAP1 / BLK;
    static constexpr int LPT = CAP / BLK;
    // fan-out per MSD level: runs of >= 16 records (>= 256 B) per bin and tile on average keep the scattered
    // writes at streaming speed and the reservation atomics at <= 1/16 per record (tools/ubench.hip);
    // 512 bins (128-B average runs) measured 1.6x slower on the level-1 scatter
    static constexpr uint32_t FMAX = RPT * BLK / 16;
    // level 1 (extraction from reads). Measured at NW=2, 10 M reads: tiles of 2048 records 32.3 ms (64 or 256 bins alike),
    // 4096 records 20.6 ms, 8192 records 29.2 ms -> 4096 records, 256 bins.
    static constexpr int RPT1 = RPT;  // RPT1 = 32 with tag staging: scatter 16.4 (-0.8) but hist 7.7 ms (+1.6)
    static constexpr uint32_t FMAX1 = FMAX;
};

template <int NW, int RPT>
size_t scatter_lds(uint32_t F) {
    return (size_t)RPT * BLK * NW * 8 + (size_t)F * 8 + (size_t)F * 4 + (size_t)RPT * BLK * 2;
}

// ---- level-1 passes over the resident read chunks (hist or scatter) ----
template <int NW, int BINF>
int pass_reads(smx_ctx *ctx, int mode, bool scatter, PassArgs a, const std::vector<uint64_t *> &masks,
               const std::vector<std::pair<uint64_t, uint64_t>> *ranges = nullptr) {
    constexpr int RPT = Tune<NW>::RPT1;
    for (size_t ci = 0; ci < ctx->chunks.size(); ++ci) {
        const ReadChunk &ch = ctx->chunks[ci];
        if (ch.n_bases == 0 || !masks[ci]) continue;
        a.seq = ch.d_words;
        a.mask = masks[ci];
        a.g0 = ranges ? (*ranges)[ci].first : 0;
        a.G = ranges ? (*ranges)[ci].second : ch.n_bases;
        if (a.G <= a.g0) continue;
        const int rpp = mode == SMX_MODE_ALL ? 2 : 1;
        const uint64_t tp = (uint64_t)(RPT / rpp) * BLK;
        const uint64_t ntiles = (a.G - a.g0 + tp - 1) / tp;
        if (!scatter) {
            unsigned grid = (unsigned)std::min<uint64_t>(ntiles, 4096);
            size_t lds = (size_t)a.F * 4;
            if (mode == SMX_MODE_ALL) {
                if (int rc = set_lds(ctx, k_hist<NW, SRC_READS_ALL, BINF, RPT>, lds)) return rc;
                hipLaunchKernelGGL((k_hist<NW, SRC_READS_ALL, BINF, RPT>), dim3(grid), dim3(BLK), lds, ctx->stream, a);
            } else {
                if (int rc = set_lds(ctx, k_hist<NW, SRC_READS_CANON, BINF, RPT>, lds)) return rc;
                hipLaunchKernelGGL((k_hist<NW, SRC_READS_CANON, BINF, RPT>), dim3(grid), dim3(BLK), lds, ctx->stream, a);
            }
        } else {
            size_t lds = (size_t)a.F * 12 + (size_t)RPT * BLK * 4;
            if (mode == SMX_MODE_ALL) {
                if (int rc = set_lds(ctx, k_scatter_reads<NW, SRC_READS_ALL, BINF, RPT>, lds)) return rc;
                hipLaunchKernelGGL((k_scatter_reads<NW, SRC_READS_ALL, BINF, RPT>), dim3((unsigned)ntiles), dim3(BLK), lds, ctx->stream, a);
            } else {
                if (int rc = set_lds(ctx, k_scatter_reads<NW, SRC_READS_CANON, BINF, RPT>, lds)) return rc;
                hipLaunchKernelGGL((k_scatter_reads<NW, SRC_READS_CANON, BINF, RPT>), dim3((unsigned)ntiles), dim3(BLK), lds, ctx->stream, a);
            }
        }
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// ---- passes over records already in HBM, segmented by a.seg_off ----
// d_tcnt / d_tstart: scratch arrays of nseg+1 u64
template <int NW, int BINF>
int pass_recs(smx_ctx *ctx, bool scatter, PassArgs a, uint64_t nrec, unsigned long long *d_tcnt, unsigned long long *d_tstart) {
    constexpr int RPT = Tune<NW>::RPT;
    const uint32_t tile = scatter ? RPT * BLK : 16 * RPT * BLK;
    hipLaunchKernelGGL(k_tile_counts, dim3((a.nseg + BLK - 1) / BLK), dim3(BLK), 0, ctx->stream, a.seg_off, a.nseg, tile, d_tcnt);
    HIPCHK(hipGetLastError());
    if (int rc = scan_u64(ctx, d_tcnt, d_tstart, a.nseg)) return rc;
    a.tile_start = d_tstart;
    a.tile_recs = tile;
    const uint64_t grid = nrec / tile + a.nseg + 1;
    if (grid > 0x7FFFFFFFull) return fail(ctx, SMX_INVALID_PARAMETER, "batch too large for one launch");
    if (!scatter) {
        size_t lds = (size_t)a.F * 4;
        if (int rc = set_lds(ctx, k_hist<NW, SRC_RECS, BINF, RPT>, lds)) return rc;
        hipLaunchKernelGGL((k_hist<NW, SRC_RECS, BINF, RPT>), dim3((unsigned)grid), dim3(BLK), lds, ctx->stream, a);
    } else {
        size_t lds = scatter_lds<NW, RPT>(a.F);
        if (int rc = set_lds(ctx, k_scatter<NW, SRC_RECS, BINF, RPT>, lds)) return rc;
        hipLaunchKernelGGL((k_scatter<NW, SRC_RECS, BINF, RPT>), dim3((unsigned)grid), dim3(BLK), lds, ctx->stream, a);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// mark valid windows of every chunk; returns total windows
int mark_windows(smx_ctx *ctx, unsigned K, std::vector<uint64_t *> &masks, uint64_t *total, bool temp_masks = true) {
    unsigned long long *d_total;
    if (int rc = dalloc(ctx, &d_total, 1)) return rc;
    HIPCHK(hipMemsetAsync(d_total, 0, 8, ctx->stream));
    masks.assign(ctx->chunks.size(), nullptr);
    for (size_t ci = 0; ci < ctx->chunks.size(); ++ci) {
        const ReadChunk &ch = ctx->chunks[ci];
        if (ch.n_reads == 0) continue;
        size_t mw = (size_t)(ch.n_bases / 64 + 2);
        if (int rc = dalloc(ctx, &masks[ci], mw, temp_masks)) return rc;
        HIPCHK(hipMemsetAsync(masks[ci], 0, mw * 8, ctx->stream));
        unsigned grid = (unsigned)((ch.n_reads + BLK - 1) / BLK);
        hipLaunchKernelGGL(k_mark_windows, dim3(grid), dim3(BLK), 0, ctx->stream, ch.d_start, ch.d_len, ch.n_reads, K,
                           (unsigned long long *)masks[ci], d_total);
        HIPCHK(hipGetLastError());
    }
    unsigned long long t = 0;
    HIPCHK(hipMemcpyAsync(&t, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *total = t;
    return 0;
}

void clear_result(smx_ctx *ctx) {
    if (ctx->d_result_buf) arena_put(ctx, ctx->d_result_buf);
    ctx->d_result_buf = ctx->d_result = nullptr;
    ctx->n_records = 0;
    ctx->bucket_off.clear();
}

// The whole count: from reads (d_recs == nullptr) or from records already in HBM.
struct ReadSel {  // which part of the resident reads one pipeline run covers
    const std::vector<uint64_t *> *masks = nullptr;                         // precomputed window masks (else computed here)
    const std::vector<std::pair<uint64_t, uint64_t>> *ranges = nullptr;     // per chunk position range
    uint64_t nrec = 0;                                                      // records in the selection (when masks given)
};

template <int NW>
int run_count(smx_ctx *ctx, unsigned K, int mode, unsigned B, const void *d_recs, uint64_t n_in, const ReadSel *sel = nullptr,
              bool recs_reusable = false, bool expand_rc = false, bool distinct_hint = false) {
    uint32_t cap = Tune<NW>::CAP;
    if (ctx->opt_leaf_cap > 0) cap = (uint32_t)std::min<int64_t>(ctx->opt_leaf_cap, cap);
    const uint32_t cap1 = std::min<uint32_t>(Tune<NW>::CAP1, cap);
    const bool from_reads = d_recs == nullptr;
    ctx->last_idx_bins = 0;
    clear_result(ctx);
    ctx->K = K;
    ctx->nw = NW;
    ctx->num_buckets = B;
    ctx->bucket_off.assign(B + 1, 0);

    WallTrace wt;
    std::vector<uint64_t *> masks;
    uint64_t nrec = expand_rc ? 2 * n_in : n_in;  // expand_rc: every input record also stands for its reverse complement
    if (expand_rc) recs_reusable = false;
    if (from_reads && sel && sel->masks) {
        masks = *sel->masks;
        nrec = sel->nrec;
    } else if (from_reads) {
        tbegin(ctx, "mark_windows");
        uint64_t nwin = 0;
        int rc = mark_windows(ctx, K, masks, &nwin);
        tend(ctx);
        if (rc) return rc;
        nrec = mode == SMX_MODE_ALL ? 2 * nwin : nwin;
    }
    const std::vector<std::pair<uint64_t, uint64_t>> *ranges = (from_reads && sel) ? sel->ranges : nullptr;
    ctx->n_instances = nrec;
    if (nrec == 0) return 0;
    if (B > 4096) return fail(ctx, SMX_INVALID_PARAMETER, "num_buckets=%u too large (level-1 fan-out limit 4096)", B);

    // ---- choose the MSD split: level 1 = bucket + s1 key bits, then levels of <= log2(FMAX) bits ----
    const unsigned avail = std::min(64u, 2 * K);  // key bits visible in key_top64
    // average leaf = 0.7 * cap1, hit exactly thanks to the mixed-radix fan-outs. Leaf sizes are compound-Poisson (every genomic
    // k-mer arrives ~coverage times), sigma ~ sqrt(coverage * mean) ~ 110 at mean 716: ~3 sigma below cap1, the tail goes to the
    // 4x class. (At mean 915 one leaf in five overflowed: sort_unique2 9.6 ms.)
    uint64_t leaf = std::max<uint32_t>(cap1 * 7 / 10, 1);
    if (ctx->opt_leaf_target > 0) leaf = (uint64_t)ctx->opt_leaf_target;
    // per-bucket key fan-out needed, realised as a mixed-radix product S1 * F2 * F3 ... (every factor <= FMAX)
    uint64_t R = ((nrec + leaf - 1) / leaf + B - 1) / B;
    const uint64_t rmax = 1ull << std::min(avail, 40u);  // no more key bins than key values
    R = std::max<uint64_t>(1, std::min(R, rmax));
    const uint32_t fmax1 = from_reads ? Tune<NW>::FMAX1 : Tune<NW>::FMAX;
    uint32_t S1 = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(R, B <= fmax1 ? fmax1 / B : 1));
    std::vector<uint32_t> lv;  // fan-outs of levels 2..
    if (ctx->opt_s1 >= 0 || ctx->opt_s2 >= 0) {  // test hook: explicit split (powers of two)
        unsigned s1 = (unsigned)std::min<int64_t>(std::max<int64_t>(ctx->opt_s1, 0), std::min(avail, 12u));
        while (s1 > 0 && ((uint64_t)B << s1) > 4096) --s1;
        S1 = 1u << s1;
        unsigned s2 = (unsigned)std::min<int64_t>(std::max<int64_t>(ctx->opt_s2, 0), std::min(avail - std::min(avail, s1), 11u));
        if (s2) lv.push_back(1u << s2);
    } else {
        uint64_t rem = (R + S1 - 1) / S1;
        while (rem > 1) {
            unsigned nl = 1;
            for (uint64_t capf = Tune<NW>::FMAX; capf < rem; capf *= Tune<NW>::FMAX) ++nl;
            uint32_t F = (uint32_t)std::ceil(std::pow((double)rem, 1.0 / nl));
            F = std::min<uint32_t>(std::max<uint32_t>(F, 2), Tune<NW>::FMAX);
            lv.push_back(F);
            rem = (rem + F - 1) / F;
            if (lv.size() >= 5) break;
        }
    }
    const uint32_t F1 = B * S1;
    uint64_t nb = F1;  // fine bins after all levels
    uint64_t nb_parent_max = F1;
    for (uint32_t t : lv) {
        nb_parent_max = nb;
        nb *= t;
    }
    if (nb > (1ull << 31)) return fail(ctx, SMX_INVALID_PARAMETER, "batch too large: %llu fine bins", (unsigned long long)nb);
    FracArgs fa{};
    fa.n = 0;
    if (S1 > 1) fa.f[fa.n++] = S1;
    for (uint32_t t : lv) fa.f[fa.n++] = t;

    // ---- allocations ------------------------------------------------------------------------
    Rec<NW> *bufA, *bufB;
    if (int rc = dalloc(ctx, &bufA, nrec)) return rc;
    if (!from_reads && recs_reusable) bufB = (Rec<NW> *)const_cast<void *>(d_recs);  // the source is dead after the level-1 scatter
    else if (int rc = dalloc(ctx, &bufB, nrec)) return rc;
    unsigned long long *histA, *offA, *offB, *cur, *tcnt, *tstart, *ucount, *uoff, *bucket_off;
    uint32_t *biglist, *bigcount, *runlen, *medlist, *medcount, *smalllist, *smallcount, *med2list, *med2count, *fblist, *fbcount;
    if (int rc = dalloc(ctx, &histA, nb)) return rc;
    if (int rc = dalloc(ctx, &offA, nb + 1)) return rc;
    if (int rc = dalloc(ctx, &offB, nb_parent_max + 1)) return rc;
    if (int rc = dalloc(ctx, &cur, nb)) return rc;
    if (int rc = dalloc(ctx, &tcnt, nb_parent_max + 1)) return rc;
    if (int rc = dalloc(ctx, &tstart, nb_parent_max + 2)) return rc;
    if (int rc = dalloc(ctx, &ucount, nb)) return rc;
    if (int rc = dalloc(ctx, &uoff, nb + 1)) return rc;
    if (int rc = dalloc(ctx, &biglist, nb)) return rc;
    if (int rc = dalloc(ctx, &bigcount, 1)) return rc;
    if (int rc = dalloc(ctx, &medlist, nb)) return rc;
    if (int rc = dalloc(ctx, &medcount, 1)) return rc;
    HIPCHK(hipMemsetAsync(medcount, 0, 4, ctx->stream));
    if (int rc = dalloc(ctx, &med2list, nb)) return rc;
    if (int rc = dalloc(ctx, &med2count, 1)) return rc;
    HIPCHK(hipMemsetAsync(med2count, 0, 4, ctx->stream));
    if (int rc = dalloc(ctx, &fblist, nb)) return rc;
    if (int rc = dalloc(ctx, &fbcount, 1)) return rc;
    HIPCHK(hipMemsetAsync(fbcount, 0, 4, ctx->stream));
    if (int rc = dalloc(ctx, &smalllist, nb)) return rc;
    if (int rc = dalloc(ctx, &smallcount, 1)) return rc;
    HIPCHK(hipMemsetAsync(smallcount, 0, 4, ctx->stream));
    if (int rc = dalloc(ctx, &runlen, nrec / cap + nb + 2)) return rc;
    if (int rc = dalloc(ctx, &bucket_off, B + 1)) return rc;
    HIPCHK(hipMemsetAsync(bigcount, 0, 4, ctx->stream));
    wt.mark(ctx, "mark+alloc");

    PassArgs a{};
    a.K = K;
    a.num_buckets = B;
    a.S1 = S1;
    a.world = 1;

    // ---- level 1 ----------------------------------------------------------------------------
    unsigned long long *seg1 = nullptr;  // records source: single segment [0, nrec)
    if (!from_reads) {
        if (int rc = dalloc(ctx, &seg1, 2)) return rc;
        unsigned long long h[2] = {0, nrec};
        HIPCHK(hipMemcpyAsync(seg1, h, 16, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    a.F = F1;
    a.hist = histA;
    HIPCHK(hipMemsetAsync(histA, 0, (size_t)F1 * 8, ctx->stream));
    // records source: level-1 histogram fused with the level-2 one (LDS table of F1*F2 counters)
    const bool joint = !from_reads && !lv.empty() && (uint64_t)F1 * lv[0] <= 24 * 1024 && F1 <= 256 && ctx->opt_joint_hist != 0;
    unsigned long long *histJ = nullptr;
    tbegin(ctx, "l1_hist");
    if (from_reads) {
        if (int rc = pass_reads<NW, BIN_L1>(ctx, mode, false, a, masks, ranges)) return rc;
    } else {
        a.recs = d_recs;
        a.seg_off = seg1;
        a.nseg = 1;
        a.expand = expand_rc ? 1u : 0u;
        if (joint) {
            const uint32_t F2 = lv[0];
            if (int rc = dalloc(ctx, &histJ, (size_t)F1 * F2)) return rc;
            HIPCHK(hipMemsetAsync(histJ, 0, (size_t)F1 * F2 * 8, ctx->stream));
            PassArgs aj = a;
            aj.hist = histJ;
            const size_t lds = (size_t)F1 * F2 * 4;
            if (int rc = set_lds(ctx, k_hist_l1_joint<NW>, lds)) return rc;
            hipLaunchKernelGGL((k_hist_l1_joint<NW>), dim3(256 * 2), dim3(1024), lds, ctx->stream, aj, F2, n_in);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(k_rowsum, dim3((F1 + BLK - 1) / BLK), dim3(BLK), 0, ctx->stream, (const unsigned long long *)histJ, F1, F2, histA);
            HIPCHK(hipGetLastError());
        } else {
            if (int rc = pass_recs<NW, BIN_L1>(ctx, false, a, nrec, tcnt, tstart)) return rc;
        }
    }
    tend(ctx);
    tbegin(ctx, "l1_scan");
    unsigned long long *off_cur = offA, *off_other = offB;
    if (lv.size() % 2 == 1) std::swap(off_cur, off_other);  // so that the final offsets land in offA (nb+1 entries)
    if (int rc = scan_u64(ctx, histA, off_cur, F1)) return rc;
    HIPCHK(hipMemcpyAsync(cur, off_cur, (size_t)F1 * 8, hipMemcpyDeviceToDevice, ctx->stream));
    tend(ctx);
    a.cursor = cur;
    a.out = bufA;
    tbegin(ctx, "l1_scatter");
    if (from_reads) {
        if (int rc = pass_reads<NW, BIN_L1>(ctx, mode, true, a, masks, ranges)) return rc;
    } else {
        if (int rc = pass_recs<NW, BIN_L1>(ctx, true, a, nrec, tcnt, tstart)) return rc;
    }
    tend(ctx);

    a.expand = 0;
    wt.mark(ctx, "level1");
    // ---- levels 2.. -------------------------------------------------------------------------
    Rec<NW> *sortbuf = bufA, *other = bufB;
    uint64_t nseg = F1;
    a.nprev = 0;
    if (S1 > 1) a.fprev[a.nprev++] = S1;
    static const char *lname[3][3] = {{"l2_hist", "l2_scan", "l2_scatter"}, {"l3_hist", "l3_scan", "l3_scatter"}, {"lN_hist", "lN_scan", "lN_scatter"}};
    for (size_t li = 0; li < lv.size(); ++li) {
        const uint32_t t = lv[li];
        const uint64_t nchild = nseg * t;
        const char **nm = lname[std::min<size_t>(li, 2)];
        a.recs = sortbuf;
        a.seg_off = off_cur;
        a.nseg = (uint32_t)nseg;
        a.F = t;
        a.hist = histA;
        const unsigned long long *hsrc = histA;
        tbegin(ctx, nm[0]);
        if (li == 0 && joint) {
            hsrc = histJ;  // counted together with level 1
        } else {
            HIPCHK(hipMemsetAsync(histA, 0, (size_t)nchild * 8, ctx->stream));
            if (int rc = pass_recs<NW, BIN_LK>(ctx, false, a, nrec, tcnt, tstart)) return rc;
        }
        tend(ctx);
        tbegin(ctx, nm[1]);
        if (int rc = scan_u64(ctx, hsrc, off_other, nchild)) return rc;
        HIPCHK(hipMemcpyAsync(cur, off_other, (size_t)nchild * 8, hipMemcpyDeviceToDevice, ctx->stream));
        tend(ctx);
        a.cursor = cur;
        a.out = other;
        tbegin(ctx, nm[2]);
        if (int rc = pass_recs<NW, BIN_LK>(ctx, true, a, nrec, tcnt, tstart)) return rc;
        tend(ctx);
        std::swap(sortbuf, other);
        std::swap(off_cur, off_other);
        nseg = nchild;
        a.fprev[a.nprev++] = t;
    }
    const unsigned long long *fine_off = off_cur;
    wt.mark(ctx, "levels2+");

    // ---- leaf sort + unique -----------------------------------------------------------------
    {
        auto leaf_geom = [&](uint32_t c, unsigned &sub_bits, uint32_t &T, size_t &lds, bool notab = false) {
            sub_bits = 10;  // 1024 in-LDS digits (512 measured the same)
            while (sub_bits > 0 && (1u << sub_bits) > c) --sub_bits;
            T = 64;
            if (!notab)
                while (T < (ctx->opt_leaf_tab > 0 ? (uint32_t)ctx->opt_leaf_tab : 2u) * c) T <<= 1;
            lds = (size_t)c * NW * 8 + ((size_t)T + 2 * ((size_t)1 << sub_bits) + 1 + c + 4) * 4;
        };
        unsigned sb1, sb2;
        uint32_t T1, T2, T1n, T2n;
        size_t lds1, lds2, lds1n, lds2n;  // ..n: the distinct-input kernels have no hash set; their LDS goes to more workgroups per CU
        leaf_geom(cap1, sb1, T1, lds1);
        leaf_geom(cap, sb2, T2, lds2);
        leaf_geom(cap1, sb1, T1n, lds1n, true);
        leaf_geom(cap, sb2, T2n, lds2n, true);
        if (int rc = set_lds(ctx, k_sort_small<NW, Tune<NW>::LPT1>, lds1)) return rc;
        if (int rc = set_lds(ctx, k_sort_small<NW, Tune<NW>::LPT>, lds2)) return rc;
        if (int rc = set_lds(ctx, k_sort_big<NW>, (size_t)cap * NW * 8)) return rc;
        tbegin(ctx, "classify");
        hipLaunchKernelGGL(k_classify, dim3((unsigned)((nb + BLK - 1) / BLK)), dim3(BLK), 0, ctx->stream, fine_off, (uint32_t)nb, cap1, cap, ucount,
                           smalllist, smallcount, medlist, medcount, med2list, med2count, biglist, bigcount);
        HIPCHK(hipGetLastError());
        tend(ctx);
        tbegin(ctx, "sort_wave");
        hipLaunchKernelGGL((k_sort_wave<NW>), dim3((unsigned)std::min<uint64_t>((nb + 3) / 4, 256 * 16)), dim3(BLK), 0, ctx->stream,
                           (void *)sortbuf, fine_off, ucount, (const uint32_t *)smalllist, (const uint32_t *)smallcount);
        HIPCHK(hipGetLastError());
        tend(ctx);
        if (int rc = set_lds(ctx, k_sort_hash<NW, Tune<NW>::LPT1, false>, lds1)) return rc;
        if (int rc = set_lds(ctx, k_sort_hash<NW, Tune<NW>::LPT, false>, lds2)) return rc;
        if (int rc = set_lds(ctx, k_sort_hash<NW, Tune<NW>::LPT1, true>, lds1n)) return rc;
        if (int rc = set_lds(ctx, k_sort_hash<NW, Tune<NW>::LPT, true>, lds2n)) return rc;
        // distinct_hint: the records are expected to be distinct already (pre-dedupe stage): the leaves skip the hash set and
        // only watch for equal records while ranking; a leaf that has some goes to the general kernel below
        tbegin(ctx, "sort_unique");
        if (sb1 > 0) {
            const dim3 grid1(ctx->opt_leaf_grid > 0 ? (unsigned)ctx->opt_leaf_grid : 256 * 16);
            if (distinct_hint)
                hipLaunchKernelGGL((k_sort_hash<NW, Tune<NW>::LPT1, true>), grid1, dim3(BLK), lds1n, ctx->stream, (void *)sortbuf, fine_off, cap1, K, fa,
                                   sb1, T1n, ucount, (const uint32_t *)medlist, (const uint32_t *)medcount, fblist, fbcount);
            else
                hipLaunchKernelGGL((k_sort_hash<NW, Tune<NW>::LPT1, false>), grid1, dim3(BLK), lds1, ctx->stream, (void *)sortbuf, fine_off, cap1, K, fa,
                                   sb1, T1, ucount, (const uint32_t *)medlist, (const uint32_t *)medcount, fblist, fbcount);
        } else {  // leaves too small for the digit table (test-sized caps): the general kernel takes the list directly
            hipLaunchKernelGGL((k_sort_small<NW, Tune<NW>::LPT1>), dim3(256 * 8), dim3(BLK), lds1, ctx->stream, (void *)sortbuf,
                               fine_off, (uint32_t)nb, cap1, K, fa, sb1, T1, ucount, biglist, bigcount,
                               (const uint32_t *)medlist, (const uint32_t *)medcount);
        }
        HIPCHK(hipGetLastError());
        tend(ctx);
        tbegin(ctx, "sort_unique2");
        if (sb2 > 0) {
            if (distinct_hint)
                hipLaunchKernelGGL((k_sort_hash<NW, Tune<NW>::LPT, true>), dim3(256 * 2), dim3(BLK), lds2n, ctx->stream, (void *)sortbuf, fine_off, cap,
                                   K, fa, sb2, T2n, ucount, (const uint32_t *)med2list, (const uint32_t *)med2count, fblist, fbcount);
            else
                hipLaunchKernelGGL((k_sort_hash<NW, Tune<NW>::LPT, false>), dim3(256 * 2), dim3(BLK), lds2, ctx->stream, (void *)sortbuf, fine_off, cap,
                                   K, fa, sb2, T2, ucount, (const uint32_t *)med2list, (const uint32_t *)med2count, fblist, fbcount);
        } else {
            hipLaunchKernelGGL((k_sort_small<NW, Tune<NW>::LPT>), dim3(256 * 2), dim3(BLK), lds2, ctx->stream, (void *)sortbuf,
                               fine_off, (uint32_t)nb, cap, K, fa, sb2, T2, ucount, biglist, bigcount,
                               (const uint32_t *)med2list, (const uint32_t *)med2count);
        }
        HIPCHK(hipGetLastError());
        // skewed leaves left over by the fast kernel (any size <= cap): general kernel with the bitonic fallback
        hipLaunchKernelGGL((k_sort_small<NW, Tune<NW>::LPT>), dim3(256 * 2), dim3(BLK), lds2, ctx->stream, (void *)sortbuf,
                           fine_off, (uint32_t)nb, cap, K, fa, sb2, T2, ucount, biglist, bigcount,
                           (const uint32_t *)fblist, (const uint32_t *)fbcount);
        HIPCHK(hipGetLastError());
        tend(ctx);
        tbegin(ctx, "sort_big");
        hipLaunchKernelGGL((k_sort_big<NW>), dim3(1024), dim3(BLK), (size_t)cap * NW * 8, ctx->stream, (void *)sortbuf, (void *)other,
                           fine_off, cap, ucount, (const uint32_t *)biglist, (const uint32_t *)bigcount, runlen);
        HIPCHK(hipGetLastError());
        tend(ctx);
    }
    if (wt.on) {
        uint32_t c[5] = {0, 0, 0, 0, 0};
        (void)hipMemcpy(&c[0], smallcount, 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&c[1], medcount, 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&c[2], med2count, 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&c[3], fbcount, 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&c[4], bigcount, 4, hipMemcpyDeviceToHost);
        fprintf(stderr, "[smx] leaves: %llu bins (F1=%u, levels=%zu) wave=%u med=%u med2=%u fallback=%u big=%u\n", (unsigned long long)nb, F1, lv.size(),
                c[0], c[1], c[2], c[3], c[4]);
        for (uint32_t i = 0; i < std::min<uint32_t>(c[4], 20); ++i) {
            uint32_t bi = 0;
            unsigned long long o[2] = {0, 0};
            (void)hipMemcpy(&bi, biglist + i, 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(o, fine_off + bi, 16, hipMemcpyDeviceToHost);
            fprintf(stderr, "[smx]   big leaf: bin %u size %llu\n", bi, o[1] - o[0]);
        }
    }
    wt.mark(ctx, "leaf sort");
    // ---- compact ----------------------------------------------------------------------------
    tbegin(ctx, "compact");
    if (int rc = scan_u64(ctx, ucount, uoff, nb)) return rc;
    unsigned long long n_unique = 0;
    HIPCHK(hipMemcpyAsync(&n_unique, uoff + nb, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (n_unique == nrec) {
        // nothing was removed (the usual case behind the pre-dedupe stage): the leaves, sorted in place and contiguous, already
        // are the bucket-major output
        std::swap(sortbuf, other);
    } else if (nrec / nb >= 128) {
        hipLaunchKernelGGL((k_compact<NW>), dim3((unsigned)std::min<uint64_t>(nb, 1u << 20)), dim3(BLK), 0, ctx->stream,
                           (const void *)sortbuf, fine_off, (const unsigned long long *)ucount, (const unsigned long long *)uoff,
                           (uint32_t)nb, (void *)other);
    } else {
        hipLaunchKernelGGL((k_compact_wave<NW>), dim3((unsigned)std::min<uint64_t>((nb + 3) / 4, 256 * 32)), dim3(BLK), 0, ctx->stream,
                           (const void *)sortbuf, fine_off, (const unsigned long long *)ucount, (const unsigned long long *)uoff,
                           (uint32_t)nb, (void *)other);
    }
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_bucket_offsets, dim3((B + 1 + BLK - 1) / BLK), dim3(BLK), 0, ctx->stream,
                       (const unsigned long long *)uoff, B, (uint32_t)(nb / B), bucket_off);
    HIPCHK(hipGetLastError());
    tend(ctx);
    if (ctx->want_index) {
        if (ctx->last_idx_off) arena_put(ctx, ctx->last_idx_off);
        ctx->last_idx_off = nullptr;
        if (int rc = dalloc(ctx, &ctx->last_idx_off, nb + 1, false)) return rc;
        HIPCHK(hipMemcpyAsync(ctx->last_idx_off, uoff, (size_t)(nb + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream));
        ctx->last_idx_bins = nb;
        ctx->last_idx_S1 = S1;
        ctx->last_idx_f = lv;
    }
    std::vector<unsigned long long> h(B + 1);
    HIPCHK(hipMemcpyAsync(h.data(), bucket_off, (size_t)(B + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (unsigned i = 0; i <= B; ++i) ctx->bucket_off[i] = h[i];
    ctx->n_records = h[B];
    ctx->d_result_buf = other;
    ctx->d_result = other;
    wt.mark(ctx, "compact");
    return 0;
}


// Super-k-mer pre-deduplication of the selected windows (smx_superkmer.hip). *out: canonical K-mers, every k-mer of the
// selection at least once and most of them exactly once (temp buffer of nwin records), *n_out: how many.
template <int NW>
int run_prededupe(smx_ctx *ctx, unsigned K, const ReadSel &sel, uint64_t nwin, Rec<NW> **out, uint64_t *n_out) {
    constexpr int SW = 2 * NW;
    const std::vector<uint64_t *> &masks = *sel.masks;
    unsigned long long *cnt, *soff, *ocount, *cursor;  // ocount[0] clean, ocount[1] dirty survivors
    if (int rc = dalloc(ctx, &cnt, SKM_NKEY)) return rc;
    if (int rc = dalloc(ctx, &soff, SKM_NKEY + 1)) return rc;
    if (int rc = dalloc(ctx, &cursor, SKM_NKEY)) return rc;
    if (int rc = dalloc(ctx, &ocount, 2)) return rc;
    HIPCHK(hipMemsetAsync(cnt, 0, (size_t)SKM_NKEY * 8, ctx->stream));
    HIPCHK(hipMemsetAsync(ocount, 0, 16, ctx->stream));
    SkmArgs a{};
    a.K = K;
    a.m = skm_m(K);
    a.w = K - a.m + 1;
    a.cnt = cnt;
    if (getenv("SMX_DEBUG")) {
        if (int rc = dalloc(ctx, &a.prof, 16)) return rc;
        HIPCHK(hipMemsetAsync(a.prof, 0, 128, ctx->stream));
    }
    auto pass = [&](int phase) -> int {
        for (size_t ci = 0; ci < ctx->chunks.size(); ++ci) {
            const ReadChunk &ch = ctx->chunks[ci];
            if (ch.n_bases == 0 || !masks[ci]) continue;
            a.seq = ch.d_words;
            a.nwords = ch.n_words;
            a.mask = masks[ci];
            a.g0 = sel.ranges ? (*sel.ranges)[ci].first : 0;
            a.G = sel.ranges ? (*sel.ranges)[ci].second : ch.n_bases;
            if (a.G <= a.g0) continue;
            const uint64_t ntiles = (a.G - a.g0 + SKM_TP - 1) / SKM_TP;
            const unsigned grid = (unsigned)std::min<uint64_t>(ntiles, 256 * 16);
            if (phase == 0) hipLaunchKernelGGL((k_skm_scan<0, NW>), dim3(grid), dim3(BLK), 0, ctx->stream, a);
            else hipLaunchKernelGGL((k_skm_scan<1, NW>), dim3(grid), dim3(BLK), 0, ctx->stream, a);
            HIPCHK(hipGetLastError());
        }
        return 0;
    };
    // staging area for the super-k-mers of pass 0 (scan order): expected 2/(w+1) starts per window, 1.5x + slack; an overflow (flag)
    // only costs the second scan of the reads
    unsigned long long *st_alloc = nullptr;
    // (short runs — K < 35, w < 16 windows — make placing atomics-bound either way and the staging round trip a loss: K=21 17.9 vs 14.4 ms)
    if (ctx->opt_skm_stage >= 2 || (ctx->opt_skm_stage == 1 && a.w >= 16)) {
        const uint64_t blocks = 256 * 16;
        a.stage_cap = (uint64_t)((double)nwin * 3.0 / (double)(a.w + 1)) + blocks * 4096 + 4096;
        if (ctx->opt_skm_stage == 2) a.stage_cap = 4096;  // tests: force the overflow fallback
        if (int rc = dalloc(ctx, &a.stage_slots, (size_t)a.stage_cap * SW)) return rc;
        if (int rc = dalloc(ctx, &a.stage_part, (size_t)a.stage_cap)) return rc;
        if (int rc = dalloc(ctx, &st_alloc, 2)) return rc;
        HIPCHK(hipMemsetAsync(a.stage_part, 0xFF, (size_t)a.stage_cap * 4, ctx->stream));
        HIPCHK(hipMemsetAsync(st_alloc, 0, 16, ctx->stream));
        a.stage_alloc = st_alloc;
    }
    tbegin(ctx, "skm_count");
    if (int rc = pass(0)) return rc;
    tend(ctx);
    tbegin(ctx, "skm_scan");
    if (int rc = scan_u64(ctx, cnt, soff, SKM_NKEY)) return rc;
    unsigned long long nslots = 0, st[2] = {0, 1};
    HIPCHK(hipMemcpyAsync(&nslots, soff + SKM_NKEY, 8, hipMemcpyDeviceToHost, ctx->stream));
    if (st_alloc) HIPCHK(hipMemcpyAsync(st, st_alloc, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    tend(ctx);
    uint64_t *slots;
    if (int rc = dalloc(ctx, &slots, (size_t)nslots * SW + SW)) return rc;
    HIPCHK(hipMemcpyAsync(cursor, soff, (size_t)SKM_NKEY * 8, hipMemcpyDeviceToDevice, ctx->stream));
    a.cursor = cursor;
    a.slots = slots;
    tbegin(ctx, "skm_scatter");
    if (st_alloc && st[1] == 0) {  // place the staged super-k-mers
        const uint64_t n_stage = std::min<uint64_t>(st[0], a.stage_cap);
        if (n_stage) {
            hipLaunchKernelGGL((k_skm_permute<NW>), dim3((unsigned)std::min<uint64_t>((n_stage + BLK - 1) / BLK, 1u << 16)), dim3(BLK), 0, ctx->stream,
                               (const uint64_t *)a.stage_slots, (const uint32_t *)a.stage_part, n_stage, cursor, slots);
            HIPCHK(hipGetLastError());
        }
    } else {
        a.stage_slots = nullptr;
        if (int rc = pass(1)) return rc;
    }
    tend(ctx);
    if (a.prof) {
        unsigned long long hp[16];
        HIPCHK(hipMemcpy(hp, a.prof, 128, hipMemcpyDeviceToHost));
        for (int ph = 0; ph < 2; ++ph)
            fprintf(stderr, "[smx] skm_scan phase %d: tiles=%llu; ticks per tile: stage+keys %.1f, minpos+flags %.1f, emit %.1f\n", ph, hp[8 * ph + 3],
                    hp[8 * ph + 3] ? (double)hp[8 * ph] / hp[8 * ph + 3] : 0.0, hp[8 * ph + 3] ? (double)hp[8 * ph + 1] / hp[8 * ph + 3] : 0.0,
                    hp[8 * ph + 3] ? (double)hp[8 * ph + 2] / hp[8 * ph + 3] : 0.0);
    }
    if (int rc = dalloc(ctx, out, nwin + 1)) return rc;
    // Chunk capacity of the LDS hash set: every copy of a k-mer sits in ONE partition, and a partition that does not fit a chunk is
    // cut (its survivors need a unique pass of their own). The partition a typical super-k-mer lives in holds sum(c^2)/sum(c) slots —
    // one genomic locus at coverage 30 is ~46 slots = ~860 instances; deeper coverage grows it linearly. 2048 instances is the fastest
    // geometry (4 workgroups per CU; 1024: 17.7 ms, 4096: 14.0 ms, 2048: 9.6 ms at bench scale); larger only when the data need it.
    uint32_t cap = 2048;
    if (ctx->opt_skm_cap > 0) {
        cap = 512;
        while (cap < (uint32_t)std::min<int64_t>(ctx->opt_skm_cap, 8192)) cap <<= 1;
    } else if (nslots) {
        unsigned long long *sums;
        if (int rc = dalloc(ctx, &sums, 2)) return rc;
        HIPCHK(hipMemsetAsync(sums, 0, 16, ctx->stream));
        hipLaunchKernelGGL(k_skm_moments, dim3(1024), dim3(BLK), 0, ctx->stream, (const unsigned long long *)cnt, SKM_NKEY, sums);
        HIPCHK(hipGetLastError());
        unsigned long long hs[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(hs, sums, 16, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        const double typical_slots = hs[0] ? (double)hs[1] / (double)hs[0] : 0.0;
        const double typical_inst = typical_slots * (double)nwin / (double)nslots;
        while (cap < 8192 && typical_inst * 1.6 > cap) cap <<= 1;
        if (getenv("SMX_DEBUG")) fprintf(stderr, "[smx] prededupe: typical partition %.0f slots = %.0f instances -> chunk capacity %u\n", typical_slots, typical_inst, cap);
    }
    const uint32_t T = 2 * cap;
    const uint32_t scap = std::min<uint32_t>(SKM_SCAP, ctx->opt_skm_scap > 0 ? (uint32_t)ctx->opt_skm_scap : cap / 8);  // ~12+ windows per slot on average
    const size_t lds = (size_t)scap * SW * 8 + (size_t)T * 4 + 514 * 4 + (size_t)cap * 4 + 512 + 16;
    if (int rc = set_lds(ctx, k_skm_dedupe<NW>, lds)) return rc;
    const uint32_t nitems = SKM_NKEY / SKM_KEYS_PER_ITEM;
    unsigned long long *prof = nullptr;
    if (getenv("SMX_DEBUG")) {
        if (int rc = dalloc(ctx, &prof, 8)) return rc;
        HIPCHK(hipMemsetAsync(prof, 0, 64, ctx->stream));
    }
    tbegin(ctx, "skm_dedupe");
    hipLaunchKernelGGL((k_skm_dedupe<NW>), dim3(std::min<uint32_t>(nitems, 256 * 8)), dim3(BLK), lds, ctx->stream, (const uint64_t *)slots,
                       (const unsigned long long *)soff, K, nitems, cap, T, scap, (void *)*out, (unsigned long long)nwin, ocount, ocount + 1, prof);
    HIPCHK(hipGetLastError());
    tend(ctx);
    if (prof) {
        unsigned long long hp[6];
        HIPCHK(hipMemcpy(hp, prof, 48, hipMemcpyDeviceToHost));
        fprintf(stderr, "[smx] dedupe chunks=%llu slots/chunk=%.1f; 100MHz ticks per chunk: stage+clear %.1f, scan+plan %.1f, insert %.1f, output %.1f\n",
                hp[4], hp[4] ? (double)hp[5] / hp[4] : 0.0, hp[4] ? (double)hp[0] / hp[4] : 0.0, hp[4] ? (double)hp[1] / hp[4] : 0.0,
                hp[4] ? (double)hp[2] / hp[4] : 0.0, hp[4] ? (double)hp[3] / hp[4] : 0.0);
    }
    unsigned long long nn[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(nn, ocount, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (nn[0] + nn[1] > nwin) return fail(ctx, SMX_DEVICE_ERROR, "pre-deduplication produced %llu records from %llu windows", nn[0] + nn[1], (unsigned long long)nwin);
    unsigned long long n = nn[0];
    if (nn[1]) {  // survivors of cut keys: sort + unique them on their own, then the whole array is exactly distinct
        if (int rc = run_count<NW>(ctx, K, SMX_MODE_ALL, 1, *out + (nwin - nn[1]), nn[1])) return rc;
        HIPCHK(hipMemcpyAsync(*out + nn[0], ctx->d_result_buf, ctx->n_records * sizeof(Rec<NW>), hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        n += ctx->n_records;
        ctx->d_result_buf = ctx->d_result = nullptr;  // stays in the temp list
        ctx->n_records = 0;
    }
    *n_out = n;
    if (getenv("SMX_DEBUG"))
        fprintf(stderr, "[smx] prededupe: %llu windows -> %llu super-k-mers -> %llu canonical records (%llu from cut keys before their unique pass)\n",
                (unsigned long long)nwin, nslots, n, nn[1]);
    return 0;
}

// One pipeline run over a selection of the resident reads: straight from the windows, or through the pre-dedupe stage.
template <int NW>
int count_selection(smx_ctx *ctx, unsigned K, int mode, unsigned B, const ReadSel &sel) {
    const uint64_t nwin = mode == SMX_MODE_ALL ? sel.nrec / 2 : sel.nrec;
    const bool possible = K >= 21 && nwin > 0;
    const bool use = possible && (ctx->opt_prededupe > 0 || (ctx->opt_prededupe < 0 && nwin >= (1u << 20)));
    if (!use) return run_count<NW>(ctx, K, mode, B, nullptr, 0, &sel);
    Rec<NW> *recs = nullptr;
    uint64_t n = 0;
    if (int rc = run_prededupe<NW>(ctx, K, sel, nwin, &recs, &n)) return rc;
    free_temps(ctx, recs);
    ctx->temps.push_back(recs);
    if (int rc = run_count<NW>(ctx, K, mode, B, recs, n, nullptr, false, mode == SMX_MODE_ALL, /*distinct_hint=*/true)) return rc;
    ctx->n_instances = sel.nrec;
    return 0;
}

// Counting from the resident reads with HBM-bounded batches (the reference's dump + merge, kmer_splitter.hpp:123-170 +
// kmer_index_builder.hpp:346-430): when two record buffers of the whole batch do not fit the budget, the position
// space of every read chunk is cut into ranges; each range is counted on its own (sorted-unique run), and runs are
// folded into the accumulated set by concatenation + one more pass of the same pipeline from records (= k-way
// merge-unique; the pipeline is a sort, so equal keys of different runs meet in the same leaf).
template <int NW>
int count_reads(smx_ctx *ctx, unsigned K, int mode, unsigned B) {
    std::vector<uint64_t *> masks;
    uint64_t nwin = 0;
    tbegin(ctx, "mark_windows");
    int rc = mark_windows(ctx, K, masks, &nwin, /*temp_masks=*/false);
    tend(ctx);
    auto drop_masks = [&]() {
        for (auto *m : masks) arena_put(ctx, m);
    };
    if (rc) {
        drop_masks();
        return rc;
    }
    const int rpp = mode == SMX_MODE_ALL ? 2 : 1;
    const uint64_t nrec = nwin * rpp;
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    size_t cached = 0;
    for (auto &b : ctx->arena_free) cached += b.second;
    const uint64_t budget = ctx->budget ? ctx->budget : (uint64_t)((free_b + cached) * 0.92);
    // per record: two ping-pong copies + ~1/4 record of bin bookkeeping
    uint64_t max_batch = budget / ((uint64_t)NW * 8 * 2 + 8);
    if (ctx->opt_batch_records > 0) max_batch = (uint64_t)ctx->opt_batch_records;
    uint64_t nbatch = nrec ? (nrec + max_batch - 1) / max_batch : 1;
    if (nbatch > 1) nbatch = (2 * nrec + max_batch - 1) / max_batch;  // keep half of the budget for the accumulated set + merge
    if (nbatch <= 1) {
        ReadSel sel;
        sel.masks = &masks;
        sel.nrec = nrec;
        rc = count_selection<NW>(ctx, K, mode, B, sel);
        drop_masks();
        return rc;
    }
    void *acc = nullptr;
    uint64_t nacc = 0;
    unsigned long long *d_cnt = nullptr;
    if ((rc = dalloc(ctx, &d_cnt, 1, false))) {
        drop_masks();
        return rc;
    }
    auto cleanup = [&](int code) {
        drop_masks();
        arena_put(ctx, d_cnt);
        if (acc && acc != ctx->d_result_buf) arena_put(ctx, acc);
        return code;
    };
    uint64_t total_inst = 0;
    for (uint64_t bi = 0; bi < nbatch; ++bi) {
        std::vector<std::pair<uint64_t, uint64_t>> ranges(ctx->chunks.size());
        HIPCHK(hipMemsetAsync(d_cnt, 0, 8, ctx->stream));
        for (size_t ci = 0; ci < ctx->chunks.size(); ++ci) {
            const uint64_t G = ctx->chunks[ci].n_bases;
            const uint64_t words = (G + 63) / 64;
            const uint64_t w0 = words * bi / nbatch, w1 = words * (bi + 1) / nbatch;
            ranges[ci] = {w0 * 64, std::min<uint64_t>(w1 * 64, G)};
            if (masks[ci] && w1 > w0) {
                hipLaunchKernelGGL(k_count_range, dim3((unsigned)std::min<uint64_t>((w1 - w0 + BLK - 1) / BLK, 4096)), dim3(BLK), 0, ctx->stream,
                                   (const unsigned long long *)masks[ci], w0, w1, d_cnt);
                if (hipGetLastError() != hipSuccess) return cleanup(fail(ctx, SMX_DEVICE_ERROR, "k_count_range launch failed"));
            }
        }
        unsigned long long nw_b = 0;
        if (hipMemcpyAsync(&nw_b, d_cnt, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
            return cleanup(fail(ctx, SMX_DEVICE_ERROR, "batch window count failed"));
        ReadSel sel;
        sel.masks = &masks;
        sel.ranges = &ranges;
        sel.nrec = nw_b * rpp;
        total_inst += sel.nrec;
        if ((rc = count_selection<NW>(ctx, K, mode, B, sel))) return cleanup(rc);
        void *run = ctx->d_result_buf;
        const uint64_t nrun = ctx->n_records;
        ctx->d_result_buf = ctx->d_result = nullptr;
        free_temps(ctx, run);
        if (!acc) {
            acc = run;
            nacc = nrun;
            continue;
        }
        // fold: acc U run -> acc
        Rec<NW> *cat;
        if ((rc = dalloc(ctx, &cat, nacc + nrun))) {
            arena_put(ctx, run);
            return cleanup(rc);
        }
        hipError_t e1 = hipMemcpyAsync(cat, acc, nacc * sizeof(Rec<NW>), hipMemcpyDeviceToDevice, ctx->stream);
        hipError_t e2 = hipMemcpyAsync(cat + nacc, run, nrun * sizeof(Rec<NW>), hipMemcpyDeviceToDevice, ctx->stream);
        hipError_t e3 = hipStreamSynchronize(ctx->stream);
        arena_put(ctx, acc);
        arena_put(ctx, run);
        acc = nullptr;
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return cleanup(fail(ctx, SMX_DEVICE_ERROR, "run concatenation failed"));
        tbegin(ctx, "merge_runs");
        tend(ctx);
        if ((rc = run_count<NW>(ctx, K, mode, B, cat, nacc + nrun, nullptr, /*recs_reusable=*/true))) return cleanup(rc);
        acc = ctx->d_result_buf;
        nacc = ctx->n_records;
        if (bi + 1 < nbatch) {
            ctx->d_result_buf = ctx->d_result = nullptr;
            free_temps(ctx, acc);
        }
    }
    if (ctx->d_result_buf != acc) {  // last batch was the first (cannot happen for nbatch > 1) or a fold result: install it
        ctx->d_result_buf = ctx->d_result = acc;
    }
    ctx->n_instances = total_inst;
    acc = nullptr;
    return cleanup(0);
}

int dispatch_count(smx_ctx *ctx, unsigned K, int mode, unsigned B, const void *d_recs, uint64_t n_in) {
    if (K < 1 || K > 128) return fail(ctx, SMX_INVALID_PARAMETER, "K=%u out of range [1,128]", K);
    if (B < 1) return fail(ctx, SMX_INVALID_PARAMETER, "num_buckets must be >= 1");
    if (mode != SMX_MODE_ALL && mode != SMX_MODE_CANONICAL) return fail(ctx, SMX_INVALID_PARAMETER, "bad mode %d", mode);
    HIPCHK(hipSetDevice(ctx->device));
    int rc;
    const bool reads = d_recs == nullptr;
    switch ((K + 31) / 32) {
        case 1: rc = reads ? count_reads<1>(ctx, K, mode, B) : run_count<1>(ctx, K, mode, B, d_recs, n_in); break;
        case 2: rc = reads ? count_reads<2>(ctx, K, mode, B) : run_count<2>(ctx, K, mode, B, d_recs, n_in); break;
        case 3: rc = reads ? count_reads<3>(ctx, K, mode, B) : run_count<3>(ctx, K, mode, B, d_recs, n_in); break;
        default: rc = reads ? count_reads<4>(ctx, K, mode, B) : run_count<4>(ctx, K, mode, B, d_recs, n_in); break;
    }
    if (rc == 0) {
        WallTrace wt;
        tcollect(ctx);
        free_temps(ctx, ctx->d_result_buf);
        wt.mark(ctx, "free");
    } else {
        (void)hipStreamSynchronize(ctx->stream);
        for (auto &t : ctx->timings) {
            (void)hipEventDestroy(t.e0);
            (void)hipEventDestroy(t.e1);
        }
        ctx->timings.clear();
        free_temps(ctx);
        ctx->d_result_buf = ctx->d_result = nullptr;
        ctx->n_records = 0;
    }
    return rc;
}

}  // namespace

namespace {
template <int NW>
int run_extract_partition(smx_ctx *ctx, unsigned K, int mode, unsigned B, unsigned world, void *d_records, uint64_t capacity,
                          uint64_t *counts) {
    std::vector<uint64_t *> masks;
    uint64_t nwin = 0;
    if (int rc = mark_windows(ctx, K, masks, &nwin)) return rc;
    uint64_t nrec = mode == SMX_MODE_ALL ? 2 * nwin : nwin;
    if (nrec > capacity) return fail(ctx, SMX_MEMORY_LIMIT_EXCEEDED, "record buffer too small: need %llu", (unsigned long long)nrec);
    // Local pre-dedupe first ("optional local sort-unique per destination to cut volume by ~coverage"): the
    // exchange then carries every distinct k-mer of this rank once instead of every instance.
    Rec<NW> *recs = nullptr;
    uint64_t n_dedup = 0;
    const bool dedupe = K >= 21 && nwin > 0 && (ctx->opt_prededupe > 0 || (ctx->opt_prededupe < 0 && nwin >= (1u << 20)));
    if (dedupe) {
        ReadSel sel;
        sel.masks = &masks;
        sel.nrec = nrec;
        if (int rc = run_prededupe<NW>(ctx, K, sel, nwin, &recs, &n_dedup)) return rc;
        nrec = mode == SMX_MODE_ALL ? 2 * n_dedup : n_dedup;
    }
    unsigned long long *hist, *off, *cur, *seg1 = nullptr, *tcnt = nullptr, *tstart = nullptr;
    if (int rc = dalloc(ctx, &hist, world)) return rc;
    if (int rc = dalloc(ctx, &off, world + 1)) return rc;
    if (int rc = dalloc(ctx, &cur, world)) return rc;
    HIPCHK(hipMemsetAsync(hist, 0, (size_t)world * 8, ctx->stream));
    PassArgs a{};
    a.K = K;
    a.num_buckets = B;
    a.world = world;
    a.F = world;
    a.hist = hist;
    if (dedupe) {
        if (int rc = dalloc(ctx, &seg1, 2)) return rc;
        if (int rc = dalloc(ctx, &tcnt, 2)) return rc;
        if (int rc = dalloc(ctx, &tstart, 3)) return rc;
        unsigned long long h2[2] = {0, nrec};
        HIPCHK(hipMemcpyAsync(seg1, h2, 16, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        a.recs = recs;
        a.seg_off = seg1;
        a.nseg = 1;
        a.expand = mode == SMX_MODE_ALL ? 1u : 0u;
    }
    tbegin(ctx, "x_hist");
    if (nrec) {
        if (dedupe) {
            if (int rc = pass_recs<NW, BIN_OWNER>(ctx, false, a, nrec, tcnt, tstart)) return rc;
        } else {
            if (int rc = pass_reads<NW, BIN_OWNER>(ctx, mode, false, a, masks)) return rc;
        }
    }
    tend(ctx);
    if (int rc = scan_u64(ctx, hist, off, world)) return rc;
    HIPCHK(hipMemcpyAsync(cur, off, (size_t)world * 8, hipMemcpyDeviceToDevice, ctx->stream));
    a.cursor = cur;
    a.out = d_records;
    tbegin(ctx, "x_scatter");
    if (nrec) {
        if (dedupe) {
            if (int rc = pass_recs<NW, BIN_OWNER>(ctx, true, a, nrec, tcnt, tstart)) return rc;
        } else {
            if (int rc = pass_reads<NW, BIN_OWNER>(ctx, mode, true, a, masks)) return rc;
        }
    }
    tend(ctx);
    std::vector<unsigned long long> h(world);
    HIPCHK(hipMemcpyAsync(h.data(), hist, (size_t)world * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (unsigned i = 0; i < world; ++i) counts[i] = h[i];
    return 0;
}
}  // namespace


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
        if (int rc = run_count<NW>(ctx, k, SMX_MODE_ALL, B, derived, 2 * nkpo)) return rc;
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

// ============================================================================ C ABI
extern "C" {

const char *smx_version(void) { return "spades-mi355x 0.1 (gfx950)"; }

int smx_create(smx_ctx **out, int device, size_t hbm_budget_bytes) {
    if (!out) return SMX_INVALID_PARAMETER;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return SMX_DEVICE_ERROR;
    if (device < 0 || device >= ndev) return SMX_INVALID_PARAMETER;
    smx_ctx *ctx = new smx_ctx();
    ctx->device = device;
    ctx->budget = hbm_budget_bytes;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&ctx->stream) != hipSuccess) {
        delete ctx;
        return SMX_DEVICE_ERROR;
    }
    *out = ctx;
    if (const char *o = getenv("SMX_OPTS")) {  // experiments: SMX_OPTS="key=value,key=value" applied to every new context
        std::string str(o);
        size_t p = 0;
        while (p < str.size()) {
            size_t e = str.find(',', p);
            if (e == std::string::npos) e = str.size();
            size_t q = str.find('=', p);
            if (q != std::string::npos && q < e) (void)smx_set_option(ctx, str.substr(p, q - p).c_str(), atoll(str.substr(q + 1, e - q - 1).c_str()));
            p = e + 1;
        }
    }
    return SMX_OK;
}

void smx_destroy(smx_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    smx_reads_clear(ctx);
    clear_graph(ctx);
    clear_result(ctx);
    free_temps(ctx);
    arena_release(ctx);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *smx_last_error(const smx_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int smx_set_option(smx_ctx *ctx, const char *key, int64_t value) {
    if (!ctx || !key) return SMX_INVALID_PARAMETER;
    if (!strcmp(key, "leaf_cap")) ctx->opt_leaf_cap = value;
    else if (!strcmp(key, "s1")) ctx->opt_s1 = value;
    else if (!strcmp(key, "s2")) ctx->opt_s2 = value;
    else if (!strcmp(key, "leaf_target")) ctx->opt_leaf_target = value;
    else if (!strcmp(key, "batch_records")) ctx->opt_batch_records = value;
    else if (!strcmp(key, "sort_edges")) ctx->opt_sort_edges = value;
    else if (!strcmp(key, "leaf_grid")) ctx->opt_leaf_grid = value;
    else if (!strcmp(key, "leaf_tab")) ctx->opt_leaf_tab = value;
    else if (!strcmp(key, "prededupe")) ctx->opt_prededupe = value;
    else if (!strcmp(key, "joint_hist")) ctx->opt_joint_hist = value;
    else if (!strcmp(key, "device_links")) ctx->opt_device_links = value;
    else if (!strcmp(key, "skm_stage")) ctx->opt_skm_stage = value;
    else if (!strcmp(key, "early_tip_bound")) ctx->opt_early_tip_bound = value;
    else if (!strcmp(key, "early_at_remover")) ctx->opt_early_at = value;
    else if (!strcmp(key, "submit_contigs")) ctx->opt_submit_contigs = value;
    else if (!strcmp(key, "flank_range")) ctx->opt_flank_range = value;
    else if (!strcmp(key, "skm_cap")) ctx->opt_skm_cap = value;
    else if (!strcmp(key, "skm_scap")) ctx->opt_skm_scap = value;
    else if (!strcmp(key, "keep_perfect_loops")) ctx->opt_keep_loops = value;
    else return fail(ctx, SMX_INVALID_PARAMETER, "unknown option %s", key);
    return SMX_OK;
}

int smx_reads_clear(smx_ctx *ctx) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    (void)hipSetDevice(ctx->device);
    for (auto &c : ctx->chunks)
        if (c.owned) {
            arena_put(ctx, c.d_words);
            arena_put(ctx, c.d_start);
            arena_put(ctx, c.d_len);
        }
    ctx->chunks.clear();
    return SMX_OK;
}

int smx_submit_reads_packed(smx_ctx *ctx, const uint64_t *words, uint64_t n_words, const uint64_t *start,
                            const uint32_t *len, uint64_t n_reads) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (n_reads == 0) return SMX_OK;
    if (!words || !start || !len) return fail(ctx, SMX_INVALID_PARAMETER, "null read arrays");
    uint64_t nb = 0;
    for (uint64_t i = 0; i < n_reads; ++i) {
        uint64_t e = start[i] + len[i];
        if (e > n_words * 32) return fail(ctx, SMX_INVALID_INPUT_FORMAT, "read %llu exceeds the packed stream", (unsigned long long)i);
        nb = std::max(nb, e);
    }
    HIPCHK(hipSetDevice(ctx->device));
    ReadChunk c;
    c.n_words = n_words;
    c.n_reads = n_reads;
    c.n_bases = nb;
    if (int rc = dalloc(ctx, &c.d_words, n_words + 8, false)) return rc;
    if (int rc = dalloc(ctx, &c.d_start, n_reads, false)) return rc;
    if (int rc = dalloc(ctx, &c.d_len, n_reads, false)) return rc;
    HIPCHK(hipMemsetAsync(c.d_words + n_words, 0, 64, ctx->stream));
    HIPCHK(hipMemcpyAsync(c.d_words, words, n_words * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(c.d_start, start, n_reads * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(c.d_len, len, n_reads * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    c.contigs = ctx->opt_submit_contigs != 0;
    ctx->chunks.push_back(c);
    return SMX_OK;
}

int smx_submit_reads_ascii(smx_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n_reads) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (n_reads == 0) return SMX_OK;
    if (!bases || !offsets) return fail(ctx, SMX_INVALID_PARAMETER, "null read arrays");
    // Ingest on the device: the ASCII bytes and offsets are uploaded as they are (relative to offsets[0]); the longest-valid
    // rule and the 2-bit packing run as two small kernels (k_longest_valid, k_pack_ascii). The host only checks lengths.
    const uint64_t base0 = offsets[0], nbases = offsets[n_reads] - base0;
    for (uint64_t r = 0; r < n_reads; ++r) {
        if (offsets[r + 1] < offsets[r]) return fail(ctx, SMX_INVALID_INPUT_FORMAT, "read offsets must be non-decreasing");
        if (offsets[r + 1] - offsets[r] > 0xFFFFFFFFull)
            return fail(ctx, SMX_INVALID_INPUT_FORMAT, "read %llu longer than 2^32-1", (unsigned long long)r);
    }
    HIPCHK(hipSetDevice(ctx->device));
    ReadChunk c;
    c.n_reads = n_reads;
    c.n_bases = nbases;
    c.n_words = (nbases + 31) / 32 + 1;
    char *d_bases;
    unsigned long long *d_off;
    if (int rc = dalloc(ctx, &d_bases, nbases + 1)) return rc;
    if (int rc = dalloc(ctx, &d_off, n_reads + 1)) return rc;
    if (int rc = dalloc(ctx, &c.d_words, c.n_words + 8, false)) return rc;
    if (int rc = dalloc(ctx, &c.d_start, n_reads, false)) return rc;
    if (int rc = dalloc(ctx, &c.d_len, n_reads, false)) return rc;
    std::vector<unsigned long long> rel(n_reads + 1);
    for (uint64_t r = 0; r <= n_reads; ++r) rel[r] = offsets[r] - base0;
    if (nbases) HIPCHK(hipMemcpyAsync(d_bases, bases + base0, nbases, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_off, rel.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(c.d_words + c.n_words, 0, 64, ctx->stream));
    hipLaunchKernelGGL(k_longest_valid, dim3((unsigned)((n_reads + BLK - 1) / BLK)), dim3(BLK), 0, ctx->stream, (const char *)d_bases,
                       (const unsigned long long *)d_off, n_reads, c.d_start, c.d_len);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_pack_ascii, dim3((unsigned)std::min<uint64_t>((c.n_words + BLK - 1) / BLK, 1u << 16)), dim3(BLK), 0, ctx->stream,
                       (const char *)d_bases, nbases, c.d_words, c.n_words);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    free_temps(ctx);
    c.contigs = ctx->opt_submit_contigs != 0;
    ctx->chunks.push_back(c);
    return SMX_OK;
}

// SPAdes binary reads (ReadConverter::ConvertToBinary output, the input of the Construction stage inside spades.py runs):
// ReadStreamStat header {read_count, max_len, total_len} (io/reads/read_stream.hpp:21-38), then per read
// Sequence::BinWrite (size_t length + ceil(len/32) 2-bit words, sequence.hpp:797-830; for 32 < len < 60 the top byte of
// word 1 carries the short-sequence metadata and is masked here), uint16 left/right offsets and a uint64 tag
// (io/reads/single_read.hpp:317-353). Paired files are the same records, two per pair (paired_read.hpp:102-108).
// The reads were already cut to their longest ACGT run by the converter (read_converter.cpp:107-118, handle_Ns = true).
int smx_submit_reads_binary(smx_ctx *ctx, const char *seq_path) {
    if (!ctx || !seq_path) return SMX_INVALID_PARAMETER;
    FILE *f = fopen(seq_path, "rb");
    if (!f) return fail(ctx, SMX_INPUT_FILE_NOT_FOUND, "File %s doesn't exist or can't be read!", seq_path);
    uint64_t hdr[3];
    if (fread(hdr, 8, 3, f) != 3) {
        fclose(f);
        return fail(ctx, SMX_INVALID_INPUT_FORMAT, "%s: truncated binary read header", seq_path);
    }
    std::vector<uint64_t> words, start;
    std::vector<uint32_t> len;
    for (;;) {
        uint64_t n;
        if (fread(&n, 8, 1, f) != 1) break;  // EOF
        if (n > 0xFFFFFFFFull) {
            fclose(f);
            return fail(ctx, SMX_INVALID_INPUT_FORMAT, "%s: corrupt read length", seq_path);
        }
        const size_t nw = (size_t)((n + 31) / 32), w0 = words.size();
        words.resize(w0 + nw);
        uint16_t offs[2];
        uint64_t tag;
        if ((nw && fread(words.data() + w0, 8, nw, f) != nw) || fread(offs, 2, 2, f) != 2 || fread(&tag, 8, 1, f) != 1) {
            fclose(f);
            return fail(ctx, SMX_INVALID_INPUT_FORMAT, "%s: truncated read record", seq_path);
        }
        if (n & 31) words[w0 + nw - 1] &= (1ull << ((n & 31) << 1)) - 1;  // tail (and the short-sequence metadata byte)
        start.push_back((uint64_t)w0 * 32);
        len.push_back((uint32_t)n);
    }
    fclose(f);
    if (words.empty()) words.push_back(0);
    return smx_submit_reads_packed(ctx, words.data(), words.size(), start.data(), len.data(), start.size());
}

// Strict 4-line FASTQ text (uncompressed file bytes) -> read batch, parsed on the device (smx_ingest.hip). `text` may end in the
// middle of a record: only complete records are taken and *consumed tells where the next chunk has to start. is_final: the text
// ends the file (a missing last newline is tolerated).
int smx_submit_fastq_text(smx_ctx *ctx, const char *text, uint64_t n_bytes, int is_final, uint64_t *n_reads, uint64_t *consumed) {
    if (!ctx || (n_bytes && !text)) return SMX_INVALID_PARAMETER;
    if (n_reads) *n_reads = 0;
    if (consumed) *consumed = 0;
    if (n_bytes == 0) return SMX_OK;
    HIPCHK(hipSetDevice(ctx->device));
    auto bail = [&](int code) {
        free_temps(ctx);
        return code;
    };
    char *d_text;
    unsigned long long *cnt, *boff, *d_cons;
    uint32_t *bad;
    const uint64_t nblk = (n_bytes + FQ_BLOCK - 1) / FQ_BLOCK;
    if (nblk > 0x7FFFFFFFull) return fail(ctx, SMX_INVALID_PARAMETER, "FASTQ chunk too large (%llu bytes)", (unsigned long long)n_bytes);
    if (int rc = dalloc(ctx, &d_text, n_bytes + 16)) return bail(rc);
    if (int rc = dalloc(ctx, &cnt, nblk)) return bail(rc);
    if (int rc = dalloc(ctx, &boff, nblk + 1)) return bail(rc);
    if (int rc = dalloc(ctx, &d_cons, 1)) return bail(rc);
    if (int rc = dalloc(ctx, &bad, 1)) return bail(rc);
    HIPCHK(hipMemcpyAsync(d_text, text, n_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(bad, 0, 4, ctx->stream));
    HIPCHK(hipMemsetAsync(d_cons, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_fq_count, dim3((unsigned)nblk), dim3(BLK), 0, ctx->stream, (const char *)d_text, n_bytes, cnt);
    HIPCHK(hipGetLastError());
    if (int rc = scan_u64(ctx, cnt, boff, nblk)) return bail(rc);
    unsigned long long n_nl = 0;
    HIPCHK(hipMemcpyAsync(&n_nl, boff + nblk, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const bool virt = is_final && text[n_bytes - 1] != '\n';
    const uint64_t n_rec = (n_nl + (virt ? 1 : 0)) / 4;
    if (n_rec == 0) {
        if (is_final && consumed) *consumed = n_bytes;
        return bail(SMX_OK);
    }
    unsigned long long *sstart, *send, *wcnt, *woff;
    if (int rc = dalloc(ctx, &sstart, n_rec)) return bail(rc);
    if (int rc = dalloc(ctx, &send, n_rec)) return bail(rc);
    if (int rc = dalloc(ctx, &wcnt, n_rec)) return bail(rc);
    if (int rc = dalloc(ctx, &woff, n_rec + 1)) return bail(rc);
    hipLaunchKernelGGL(k_fq_lines, dim3((unsigned)nblk), dim3(BLK), 0, ctx->stream, (const char *)d_text, n_bytes, (const unsigned long long *)boff,
                       n_rec, sstart, send, d_cons, bad);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_fq_words, dim3((unsigned)((n_rec + BLK - 1) / BLK)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)sstart,
                       (const unsigned long long *)send, n_rec, wcnt, bad);
    HIPCHK(hipGetLastError());
    if (int rc = scan_u64(ctx, wcnt, woff, n_rec)) return bail(rc);
    unsigned long long n_words = 0, cons = 0;
    uint32_t h_bad = 0;
    HIPCHK(hipMemcpyAsync(&n_words, woff + n_rec, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(&cons, d_cons, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(&h_bad, bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (h_bad) return bail(fail(ctx, SMX_INVALID_INPUT_FORMAT, "not a 4-line FASTQ (%u structure violations)", h_bad));
    ReadChunk c;
    c.n_reads = n_rec;
    c.n_words = n_words + 1;
    c.n_bases = c.n_words * 32;
    if (int rc = dalloc(ctx, &c.d_words, c.n_words + 8, false)) return bail(rc);
    if (int rc = dalloc(ctx, &c.d_start, n_rec, false)) {
        arena_put(ctx, c.d_words);
        return bail(rc);
    }
    if (int rc = dalloc(ctx, &c.d_len, n_rec, false)) {
        arena_put(ctx, c.d_words);
        arena_put(ctx, c.d_start);
        return bail(rc);
    }
    HIPCHK(hipMemsetAsync(c.d_words + n_words, 0, 72, ctx->stream));
    hipLaunchKernelGGL(k_fq_pack, dim3((unsigned)((n_rec + BLK - 1) / BLK)), dim3(BLK), 0, ctx->stream, (const char *)d_text,
                       (const unsigned long long *)sstart, (const unsigned long long *)send, (const unsigned long long *)woff, n_rec, c.d_words,
                       c.d_start, c.d_len);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    free_temps(ctx);
    c.contigs = ctx->opt_submit_contigs != 0;
    ctx->chunks.push_back(c);
    if (n_reads) *n_reads = n_rec;
    if (consumed) *consumed = (virt && n_rec * 4 == n_nl + 1) ? n_bytes : cons;
    return SMX_OK;
}

// page-locked host memory for the FASTQ chunks (hipHostMalloc): pageable buffers cap the upload at a fraction of the PCIe rate
void *smx_pinned_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
void smx_pinned_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int smx_submit_reads_device(smx_ctx *ctx, const void *d_words, uint64_t n_words, const void *d_start, const void *d_len,
                            uint64_t n_reads) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (n_reads == 0) return SMX_OK;
    if (!d_words || !d_start || !d_len) return fail(ctx, SMX_INVALID_PARAMETER, "null device read arrays");
    ReadChunk c;
    c.d_words = (uint64_t *)d_words;
    c.d_start = (uint64_t *)d_start;
    c.d_len = (uint32_t *)d_len;
    c.n_words = n_words;
    c.n_reads = n_reads;
    c.n_bases = n_words * 32;
    c.owned = false;
    c.contigs = ctx->opt_submit_contigs != 0;
    ctx->chunks.push_back(c);
    return SMX_OK;
}

int smx_reads_info(const smx_ctx *ctx, uint64_t *n_reads, uint64_t *n_bases) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    uint64_t r = 0, b = 0;
    for (auto &c : ctx->chunks) {
        r += c.n_reads;
        b += c.n_bases;
    }
    if (n_reads) *n_reads = r;
    if (n_bases) *n_bases = b;
    return SMX_OK;
}

int smx_count(smx_ctx *ctx, unsigned K, int mode, unsigned num_buckets) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    ctx->xnames.clear();
    ctx->xms.clear();
    return dispatch_count(ctx, K, mode, num_buckets, nullptr, 0);
}

int smx_count_records(smx_ctx *ctx, unsigned K, unsigned num_buckets, const void *d_records, uint64_t n_records) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (n_records && !d_records) return fail(ctx, SMX_INVALID_PARAMETER, "null records");
    static const uint64_t dummy = 0;
    return dispatch_count(ctx, K, SMX_MODE_ALL, num_buckets, n_records ? d_records : (const void *)&dummy, n_records);
}

int smx_count_info(const smx_ctx *ctx, uint64_t *n_records, unsigned *words_per_record, uint64_t *n_kmer_instances) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (n_records) *n_records = ctx->n_records;
    if (words_per_record) *words_per_record = ctx->nw;
    if (n_kmer_instances) *n_kmer_instances = ctx->n_instances;
    return SMX_OK;
}

int smx_bucket_sizes(const smx_ctx *ctx, uint64_t *sizes) {
    if (!ctx || !sizes) return SMX_INVALID_PARAMETER;
    for (unsigned b = 0; b < ctx->num_buckets; ++b) sizes[b] = ctx->bucket_off[b + 1] - ctx->bucket_off[b];
    return SMX_OK;
}

int smx_copy_bucket(const smx_ctx *cctx, unsigned bucket, void *host_dst) {
    smx_ctx *ctx = const_cast<smx_ctx *>(cctx);
    if (!ctx || !host_dst) return SMX_INVALID_PARAMETER;
    if (bucket >= ctx->num_buckets) return fail(ctx, SMX_INVALID_PARAMETER, "bucket %u out of range", bucket);
    const uint64_t o = ctx->bucket_off[bucket], n = ctx->bucket_off[bucket + 1] - o;
    if (n == 0) return SMX_OK;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t w = (size_t)ctx->nw * 8;
    HIPCHK(hipMemcpy(host_dst, (const char *)ctx->d_result + o * w, n * w, hipMemcpyDeviceToHost));
    return SMX_OK;
}

int smx_copy_final_kmers(const smx_ctx *cctx, void *host_dst) {
    smx_ctx *ctx = const_cast<smx_ctx *>(cctx);
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (ctx->n_records == 0) return SMX_OK;
    if (!host_dst) return SMX_INVALID_PARAMETER;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpy(host_dst, ctx->d_result, ctx->n_records * ctx->nw * 8, hipMemcpyDeviceToHost));
    return SMX_OK;
}

int smx_write_final_kmers(const smx_ctx *cctx, const char *path) {
    smx_ctx *ctx = const_cast<smx_ctx *>(cctx);
    if (!ctx || !path) return SMX_INVALID_PARAMETER;
    FILE *f = fopen(path, "wb");
    if (!f) return fail(ctx, SMX_IO_ERROR, "Cannot open %s for writing", path);
    const size_t w = (size_t)ctx->nw * 8;
    const size_t chunk = (size_t)64 << 20;
    std::vector<char> buf(std::min<size_t>(chunk, std::max<size_t>(ctx->n_records * w, 1)));
    (void)hipSetDevice(ctx->device);
    for (size_t o = 0; o < ctx->n_records * w; o += chunk) {
        size_t n = std::min(chunk, ctx->n_records * w - o);
        if (hipMemcpy(buf.data(), (const char *)ctx->d_result + o, n, hipMemcpyDeviceToHost) != hipSuccess) {
            fclose(f);
            return fail(ctx, SMX_DEVICE_ERROR, "device read-back failed");
        }
        if (fwrite(buf.data(), 1, n, f) != n) {
            fclose(f);
            return fail(ctx, SMX_IO_ERROR, "I/O error! Incomplete write to %s", path);
        }
    }
    if (fclose(f) != 0) return fail(ctx, SMX_IO_ERROR, "I/O error closing %s", path);
    return SMX_OK;
}

const void *smx_device_kmers(const smx_ctx *ctx) { return ctx ? ctx->d_result : nullptr; }

unsigned smx_rank_first_bucket(unsigned num_buckets, unsigned world, unsigned rank) {
    return (unsigned)(((uint64_t)rank * num_buckets + world - 1) / world);
}

int smx_extract_count(smx_ctx *ctx, unsigned K, int mode, uint64_t *n_records) {
    if (!ctx || !n_records) return SMX_INVALID_PARAMETER;
    if (K < 1 || K > 128) return fail(ctx, SMX_INVALID_PARAMETER, "K=%u out of range [1,128]", K);
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<uint64_t *> masks;
    uint64_t nwin = 0;
    int rc = mark_windows(ctx, K, masks, &nwin);
    free_temps(ctx);
    if (rc) return rc;
    *n_records = mode == SMX_MODE_ALL ? 2 * nwin : nwin;
    return SMX_OK;
}


int smx_extract_partition(smx_ctx *ctx, unsigned K, int mode, unsigned num_buckets, unsigned world, void *d_records,
                          uint64_t capacity_records, uint64_t *counts) {
    if (!ctx || !counts) return SMX_INVALID_PARAMETER;
    if (K < 1 || K > 128) return fail(ctx, SMX_INVALID_PARAMETER, "K=%u out of range [1,128]", K);
    if (world < 1 || world > 4096 || num_buckets < 1) return fail(ctx, SMX_INVALID_PARAMETER, "bad world/num_buckets");
    HIPCHK(hipSetDevice(ctx->device));
    int rc;
    switch ((K + 31) / 32) {
        case 1: rc = run_extract_partition<1>(ctx, K, mode, num_buckets, world, d_records, capacity_records, counts); break;
        case 2: rc = run_extract_partition<2>(ctx, K, mode, num_buckets, world, d_records, capacity_records, counts); break;
        case 3: rc = run_extract_partition<3>(ctx, K, mode, num_buckets, world, d_records, capacity_records, counts); break;
        default: rc = run_extract_partition<4>(ctx, K, mode, num_buckets, world, d_records, capacity_records, counts); break;
    }
    (void)hipStreamSynchronize(ctx->stream);
    if (rc == 0) {
        tcollect(ctx);
        ctx->xnames = ctx->tnames;
        ctx->xms = ctx->tms;
        ctx->tnames.clear();
        ctx->tms.clear();
    } else {
        for (auto &t : ctx->timings) {
            (void)hipEventDestroy(t.e0);
            (void)hipEventDestroy(t.e1);
        }
        ctx->timings.clear();
    }
    free_temps(ctx);
    return rc;
}

static int build_graph_impl(smx_ctx *ctx, unsigned k, unsigned num_buckets, const void *recs, uint64_t nrecs) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    // gbuilder.cpp:130-135: MIN_K <= k < MAX_K(128), k odd
    if (k < 1) return fail(ctx, SMX_INVALID_PARAMETER, "k-mer size %u is too low", k);
    if (k >= 128) return fail(ctx, SMX_INVALID_PARAMETER, "k-mer size %u is too high", k);
    if (k % 2 == 0) return fail(ctx, SMX_INVALID_PARAMETER, "k-mer size must be odd");
    if (num_buckets < 1) return fail(ctx, SMX_INVALID_PARAMETER, "num_buckets must be >= 1");
    HIPCHK(hipSetDevice(ctx->device));
    ctx->xnames.clear();
    ctx->xms.clear();
    int rc;
    switch ((k + 32) / 32) {  // words of k+1 (== words of k because k is odd)
        case 1: rc = run_graph<1>(ctx, k, num_buckets, recs, nrecs); break;
        case 2: rc = run_graph<2>(ctx, k, num_buckets, recs, nrecs); break;
        case 3: rc = run_graph<3>(ctx, k, num_buckets, recs, nrecs); break;
        default: rc = run_graph<4>(ctx, k, num_buckets, recs, nrecs); break;
    }
    (void)hipStreamSynchronize(ctx->stream);
    if (rc == 0) {
        tcollect(ctx);
    } else {
        for (auto &t : ctx->timings) {
            (void)hipEventDestroy(t.e0);
            (void)hipEventDestroy(t.e1);
        }
        ctx->timings.clear();
    }
    free_temps(ctx);
    if (rc) clear_graph(ctx);
    return rc;
}

int smx_build_graph(smx_ctx *ctx, unsigned k, unsigned num_buckets) { return build_graph_impl(ctx, k, num_buckets, nullptr, 0); }

int smx_build_graph_from_records(smx_ctx *ctx, unsigned k, unsigned num_buckets, const void *d_kpomers, uint64_t n_records) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (n_records && !d_kpomers) return fail(ctx, SMX_INVALID_PARAMETER, "null records");
    static const uint64_t dummy = 0;
    return build_graph_impl(ctx, k, num_buckets, n_records ? d_kpomers : (const void *)&dummy, n_records);
}

int smx_graph_set_coverage(smx_ctx *ctx, const uint32_t *raw_coverage, uint64_t n_edges) {
    if (!ctx || !ctx->g_ready) return SMX_INVALID_PARAMETER;
    if (n_edges != ctx->gh.n_edges()) return fail(ctx, SMX_INVALID_PARAMETER, "coverage array has %llu entries, graph has %llu unitigs",
                                                  (unsigned long long)n_edges, (unsigned long long)ctx->gh.n_edges());
    if (n_edges && !raw_coverage) return SMX_INVALID_PARAMETER;
    ctx->gh.ecov.assign(raw_coverage, raw_coverage + n_edges);
    return SMX_OK;
}

int smx_copy_kmers_device(const smx_ctx *cctx, void *d_dst) {
    smx_ctx *ctx = const_cast<smx_ctx *>(cctx);
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (ctx->n_records == 0) return SMX_OK;
    if (!d_dst) return SMX_INVALID_PARAMETER;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(d_dst, ctx->d_result, ctx->n_records * ctx->nw * 8, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SMX_OK;
}

int smx_graph_info(const smx_ctx *ctx, uint64_t *info /* [8] */) {
    if (!ctx || !info) return SMX_INVALID_PARAMETER;
    if (!ctx->g_ready) return SMX_INVALID_PARAMETER;
    info[0] = ctx->g_nkpo;
    info[1] = ctx->g_nkmers;
    info[2] = ctx->gh.n_edges();
    info[3] = ctx->gh.n_loops;
    info[4] = ctx->gh.n_vertices;
    info[5] = ctx->gh.n_links;
    info[6] = ctx->gh.seq.size();
    info[7] = ctx->g_nw;
    return SMX_OK;
}

int smx_graph_tip_stats(const smx_ctx *ctx, uint64_t *stats /* [4] */) {
    if (!ctx || !stats || !ctx->g_ready) return SMX_INVALID_PARAMETER;
    stats[0] = ctx->g_tip_kmers;
    stats[1] = ctx->g_tips;
    stats[2] = ctx->g_at_edges;
    stats[3] = ctx->g_at_tip_kmers;
    return SMX_OK;
}

int smx_graph_copy_kmers(const smx_ctx *cctx, void *kmers_host, uint8_t *masks_host) {
    smx_ctx *ctx = const_cast<smx_ctx *>(cctx);
    if (!ctx || !ctx->g_ready) return SMX_INVALID_PARAMETER;
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->g_nkmers == 0) return SMX_OK;
    if (kmers_host) HIPCHK(hipMemcpy(kmers_host, ctx->g_kmers, ctx->g_nkmers * ctx->g_nw * 8, hipMemcpyDeviceToHost));
    if (masks_host) HIPCHK(hipMemcpy(masks_host, ctx->g_mask, ctx->g_nkmers, hipMemcpyDeviceToHost));
    return SMX_OK;
}

int smx_graph_copy_unitigs(const smx_ctx *ctx, uint64_t *offsets, char *seq) {
    if (!ctx || !ctx->g_ready) return SMX_INVALID_PARAMETER;
    if (offsets) memcpy(offsets, ctx->gh.eoff.data(), ctx->gh.eoff.size() * 8);
    if (seq && !ctx->gh.seq.empty()) memcpy(seq, ctx->gh.seq.data(), ctx->gh.seq.size());
    return SMX_OK;
}

int smx_graph_fill_coverage(smx_ctx *ctx) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (!ctx->g_ready) return fail(ctx, SMX_INVALID_PARAMETER, "no graph built");
    HIPCHK(hipSetDevice(ctx->device));
    int rc;
    switch (ctx->g_nw) {
        case 1: rc = run_coverage<1>(ctx); break;
        case 2: rc = run_coverage<2>(ctx); break;
        case 3: rc = run_coverage<3>(ctx); break;
        default: rc = run_coverage<4>(ctx); break;
    }
    (void)hipStreamSynchronize(ctx->stream);
    if (rc == 0) tcollect(ctx);
    else {
        for (auto &t : ctx->timings) {
            (void)hipEventDestroy(t.e0);
            (void)hipEventDestroy(t.e1);
        }
        ctx->timings.clear();
        ctx->gh.ecov.clear();
    }
    free_temps(ctx);
    return rc;
}

int smx_graph_copy_flanking(const smx_ctx *ctx, uint32_t *flank_edge, uint32_t *flank_conjugate) {
    if (!ctx || !ctx->g_ready) return SMX_INVALID_PARAMETER;
    const size_t ne = ctx->gh.n_edges();
    if (ctx->gh.eflank_s.size() != ne || ctx->gh.eflank_e.size() != ne) return SMX_INVALID_PARAMETER;
    if (ne && flank_edge) memcpy(flank_edge, ctx->gh.eflank_s.data(), ne * 4);
    if (ne && flank_conjugate) memcpy(flank_conjugate, ctx->gh.eflank_e.data(), ne * 4);
    return SMX_OK;
}

int smx_graph_copy_coverage(const smx_ctx *ctx, uint32_t *raw_coverage) {
    if (!ctx || !ctx->g_ready || !raw_coverage) return SMX_INVALID_PARAMETER;
    if (ctx->gh.ecov.size() != ctx->gh.n_edges()) return SMX_INVALID_PARAMETER;
    if (!ctx->gh.ecov.empty()) memcpy(raw_coverage, ctx->gh.ecov.data(), ctx->gh.ecov.size() * 4);
    return SMX_OK;
}

int smx_graph_write_gfa(smx_ctx *ctx, const char *path, const char *flavour_version) {
    if (!ctx || !path) return SMX_INVALID_PARAMETER;
    if (!ctx->g_ready) return fail(ctx, SMX_INVALID_PARAMETER, "no graph built");
    FILE *f = fopen(path, "wb");
    if (!f) return fail(ctx, SMX_IO_ERROR, "Cannot open %s for writing", path);
    bool ok = smxh::write_gfa(ctx->gh, f, flavour_version ? flavour_version : "SPAdes-4.3.0-dev");
    if (fclose(f) != 0) ok = false;
    return ok ? SMX_OK : fail(ctx, SMX_IO_ERROR, "I/O error writing %s", path);
}

int smx_graph_write_fastg(smx_ctx *ctx, const char *path) {
    if (!ctx || !path) return SMX_INVALID_PARAMETER;
    if (!ctx->g_ready) return fail(ctx, SMX_INVALID_PARAMETER, "no graph built");
    FILE *f = fopen(path, "wb");
    if (!f) return fail(ctx, SMX_IO_ERROR, "Cannot open %s for writing", path);
    bool ok = smxh::write_fastg(ctx->gh, f);
    if (fclose(f) != 0) ok = false;
    return ok ? SMX_OK : fail(ctx, SMX_IO_ERROR, "I/O error writing %s", path);
}

int smx_graph_write_spades(smx_ctx *ctx, const char *basename) {
    if (!ctx || !basename) return SMX_INVALID_PARAMETER;
    if (!ctx->g_ready) return fail(ctx, SMX_INVALID_PARAMETER, "no graph built");
    for (int part = 0; part < 2; ++part) {
        std::string path = std::string(basename) + (part ? ".cvr" : ".grseq");
        FILE *f = fopen(path.c_str(), "wb");
        if (!f) return fail(ctx, SMX_IO_ERROR, "Cannot open %s for writing", path.c_str());
        bool ok = part ? smxh::write_cvr(ctx->gh, f) : smxh::write_grseq(ctx->gh, f);
        if (fclose(f) != 0) ok = false;
        if (!ok) return fail(ctx, SMX_IO_ERROR, "I/O error writing %s", path.c_str());
    }
    return SMX_OK;
}

// Host-only entry point (no GPU, no context): links + writers on caller-provided unitigs. Lets the reference-side code reuse
// the writers for edges it computed itself, and lets the CPU test tier cover smx_graph_host.hpp.
int smx_host_write_graph(unsigned k, uint64_t n_edges, const uint64_t *offsets, const char *seq, const uint32_t *start_node,
                         const uint32_t *end_node, const uint32_t *raw_coverage, int sort_edges, int format, const char *path,
                         const char *flavour_version) {
    if (!offsets || !path || (n_edges && (!seq || !start_node || !end_node))) return SMX_INVALID_PARAMETER;
    smxh::GraphHost g;
    g.k = k;
    g.eoff.assign(offsets, offsets + n_edges + 1);
    g.seq.assign(seq ? seq : "", (size_t)offsets[n_edges]);
    g.estart.assign(start_node, start_node + n_edges);
    g.eend.assign(end_node, end_node + n_edges);
    g.eself.resize(n_edges);
    for (uint64_t i = 0; i < n_edges; ++i) {
        std::string s = g.seq.substr((size_t)offsets[i], (size_t)(offsets[i + 1] - offsets[i]));
        g.eself[i] = s == smxh::revcomp(s) ? 1 : 0;
    }
    if (sort_edges) {
        if (raw_coverage) return SMX_INVALID_PARAMETER;  // coverage is per edge of the given order
        smxh::sort_edges_raw(g);
    }
    if (raw_coverage) g.ecov.assign(raw_coverage, raw_coverage + n_edges);
    smxh::build_links(g);
    bool ok = true;
    if (format == 3) {  // .grseq + .cvr
        for (int part = 0; part < 2 && ok; ++part) {
            std::string p = std::string(path) + (part ? ".cvr" : ".grseq");
            FILE *f = fopen(p.c_str(), "wb");
            if (!f) return SMX_IO_ERROR;
            ok = part ? smxh::write_cvr(g, f) : smxh::write_grseq(g, f);
            if (fclose(f) != 0) ok = false;
        }
        return ok ? SMX_OK : SMX_IO_ERROR;
    }
    FILE *f = fopen(path, "wb");
    if (!f) return SMX_IO_ERROR;
    if (format == 0) ok = smxh::write_unitigs_fasta(g, f);
    else if (format == 1) ok = smxh::write_gfa(g, f, flavour_version ? flavour_version : "SPAdes-4.3.0-dev");
    else if (format == 2) ok = smxh::write_fastg(g, f);
    else ok = false;
    if (fclose(f) != 0) ok = false;
    return ok ? SMX_OK : (format > 3 || format < 0 ? SMX_INVALID_PARAMETER : SMX_IO_ERROR);
}

int smx_graph_write_unitigs(smx_ctx *ctx, const char *path) {
    if (!ctx || !path) return SMX_INVALID_PARAMETER;
    if (!ctx->g_ready) return fail(ctx, SMX_INVALID_PARAMETER, "no graph built");
    FILE *f = fopen(path, "wb");
    if (!f) return fail(ctx, SMX_IO_ERROR, "Cannot open %s for writing", path);
    bool ok = smxh::write_unitigs_fasta(ctx->gh, f);
    if (fclose(f) != 0) ok = false;
    return ok ? SMX_OK : fail(ctx, SMX_IO_ERROR, "I/O error writing %s", path);
}

int smx_last_timings(const smx_ctx *ctx, const char **names, float *ms, int cap) {
    if (!ctx) return 0;
    int n = 0;
    for (size_t i = 0; i < ctx->xnames.size(); ++i, ++n)
        if (n < cap) {
            if (names) names[n] = ctx->xnames[i].c_str();
            if (ms) ms[n] = ctx->xms[i];
        }
    for (size_t i = 0; i < ctx->tnames.size(); ++i, ++n)
        if (n < cap) {
            if (names) names[n] = ctx->tnames[i].c_str();
            if (ms) ms[n] = ctx->tms[i];
        }
    return n;
}

}  // extern "C"
