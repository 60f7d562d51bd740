// spades_amd/csrc/smx_api.hip — C ABI (include/smx.h) + host-side pipeline driving the gfx950 kernels.
// Built by hipcc into libspades_mi355x.so; no torch / no reference headers involved.
#include "../../include/smx.h"
#include "smx_kernels.hip"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace smx;

namespace {

struct ReadChunk {
    uint64_t *d_words = nullptr;
    uint64_t *d_start = nullptr;
    uint32_t *d_len = nullptr;
    uint64_t n_words = 0, n_reads = 0, n_bases = 0;
    bool owned = true;
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
    int64_t opt_leaf_cap = 0, opt_s1 = -1, opt_s2 = -1;
    // timings
    std::vector<Timing> timings;
    std::vector<std::string> tnames;
    std::vector<float> tms;
    std::vector<void *> temps;  // allocations of the pipeline in flight
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

template <typename T>
int dalloc(smx_ctx *ctx, T **p, size_t count, bool temp = true) {
    void *q = nullptr;
    size_t bytes = std::max<size_t>(count * sizeof(T), 256);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess)
        return fail(ctx, SMX_MEMORY_LIMIT_EXCEEDED, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    if (temp) ctx->temps.push_back(q);
    *p = (T *)q;
    return 0;
}

void free_temps(smx_ctx *ctx, void *keep = nullptr) {
    for (void *p : ctx->temps)
        if (p != keep) (void)hipFree(p);
    ctx->temps.clear();
}

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
    static constexpr int RPT = (NW == 1) ? 16 : 8;                       // records per thread in a scatter tile
    static constexpr uint32_t CAP = (NW == 1) ? 8192 : (NW == 2 ? 4096 : 2048);  // LDS-sortable leaf
};

template <int NW, int RPT>
size_t scatter_lds(uint32_t F) {
    return (size_t)RPT * BLK * NW * 8 + (size_t)F * 8 + (size_t)F * 4 + (size_t)RPT * BLK * 2;
}

// ---- level-1 passes over the resident read chunks (hist or scatter) ----
template <int NW, int BINF>
int pass_reads(smx_ctx *ctx, int mode, bool scatter, PassArgs a, const std::vector<uint64_t *> &masks) {
    constexpr int RPT = Tune<NW>::RPT;
    for (size_t ci = 0; ci < ctx->chunks.size(); ++ci) {
        const ReadChunk &ch = ctx->chunks[ci];
        if (ch.n_bases == 0) continue;
        a.seq = ch.d_words;
        a.mask = masks[ci];
        a.G = ch.n_bases;
        const int rpp = mode == SMX_MODE_ALL ? 2 : 1;
        const uint64_t tp = (uint64_t)(RPT / rpp) * BLK;
        const uint64_t ntiles = (a.G + tp - 1) / tp;
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
            size_t lds = scatter_lds<NW, RPT>(a.F);
            if (mode == SMX_MODE_ALL) {
                if (int rc = set_lds(ctx, k_scatter<NW, SRC_READS_ALL, BINF, RPT>, lds)) return rc;
                hipLaunchKernelGGL((k_scatter<NW, SRC_READS_ALL, BINF, RPT>), dim3((unsigned)ntiles), dim3(BLK), lds, ctx->stream, a);
            } else {
                if (int rc = set_lds(ctx, k_scatter<NW, SRC_READS_CANON, BINF, RPT>, lds)) return rc;
                hipLaunchKernelGGL((k_scatter<NW, SRC_READS_CANON, BINF, RPT>), dim3((unsigned)ntiles), dim3(BLK), lds, ctx->stream, a);
            }
        }
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// ---- passes over records already in HBM, segmented by a.seg_off ----
template <int NW, int BINF>
int pass_recs(smx_ctx *ctx, bool scatter, PassArgs a, uint64_t nrec, uint32_t *d_tile_start) {
    constexpr int RPT = Tune<NW>::RPT;
    const uint32_t tile = scatter ? RPT * BLK : 16 * RPT * BLK;
    hipLaunchKernelGGL(k_tile_prefix, dim3(1), dim3(BLK), 0, ctx->stream, a.seg_off, a.nseg, tile, d_tile_start);
    HIPCHK(hipGetLastError());
    a.tile_start = d_tile_start;
    a.tile_recs = tile;
    const uint64_t grid = nrec / tile + a.nseg + 1;
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
int mark_windows(smx_ctx *ctx, unsigned K, std::vector<uint64_t *> &masks, uint64_t *total) {
    unsigned long long *d_total;
    if (int rc = dalloc(ctx, &d_total, 1)) return rc;
    HIPCHK(hipMemsetAsync(d_total, 0, 8, ctx->stream));
    masks.assign(ctx->chunks.size(), nullptr);
    for (size_t ci = 0; ci < ctx->chunks.size(); ++ci) {
        const ReadChunk &ch = ctx->chunks[ci];
        if (ch.n_reads == 0) continue;
        size_t mw = (size_t)(ch.n_bases / 64 + 2);
        if (int rc = dalloc(ctx, &masks[ci], mw)) return rc;
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
    if (ctx->d_result_buf) (void)hipFree(ctx->d_result_buf);
    ctx->d_result_buf = ctx->d_result = nullptr;
    ctx->n_records = 0;
    ctx->bucket_off.clear();
}

// The whole count: from reads (d_recs == nullptr) or from records already in HBM.
template <int NW>
int run_count(smx_ctx *ctx, unsigned K, int mode, unsigned B, const void *d_recs, uint64_t n_in) {
    constexpr int RPT = Tune<NW>::RPT;
    uint32_t cap = Tune<NW>::CAP;
    if (ctx->opt_leaf_cap > 0) cap = (uint32_t)std::min<int64_t>(ctx->opt_leaf_cap, cap);
    const bool from_reads = d_recs == nullptr;
    clear_result(ctx);
    ctx->K = K;
    ctx->nw = NW;
    ctx->num_buckets = B;
    ctx->bucket_off.assign(B + 1, 0);

    std::vector<uint64_t *> masks;
    uint64_t nrec = n_in;
    if (from_reads) {
        tbegin(ctx, "mark_windows");
        uint64_t nwin = 0;
        int rc = mark_windows(ctx, K, masks, &nwin);
        tend(ctx);
        if (rc) return rc;
        nrec = mode == SMX_MODE_ALL ? 2 * nwin : nwin;
    }
    ctx->n_instances = nrec;
    if (nrec == 0) return 0;

    // ---- choose the MSD split -------------------------------------------------------------
    const unsigned avail = std::min(32u, 2 * K);
    const uint64_t leaf = std::max<uint32_t>(cap / 4, 1);
    const uint64_t fneed = (nrec + leaf - 1) / leaf;
    unsigned bits = fneed > B ? ceil_log2((fneed + B - 1) / B) : 0;
    bits = std::min(bits, avail);
    const unsigned lb = ceil_log2(B);
    unsigned s1 = std::min(bits, lb >= 10 ? 0u : 10u - lb);
    while (s1 > 0 && ((uint64_t)B << s1) > 4096) --s1;
    unsigned s2 = std::min(bits - s1, 11u);
    if (ctx->opt_s1 >= 0) s1 = (unsigned)std::min<int64_t>(ctx->opt_s1, avail);
    if (ctx->opt_s2 >= 0) s2 = (unsigned)std::min<int64_t>(ctx->opt_s2, avail - std::min(avail, s1));
    if (((uint64_t)B << s1) > 4096)
        return fail(ctx, SMX_INVALID_PARAMETER, "num_buckets=%u too large (level-1 fan-out limit 4096)", B);
    const uint32_t F1 = B << s1, F2 = 1u << s2;
    const uint64_t nb = (uint64_t)F1 * F2;  // fine bins

    // ---- allocations ------------------------------------------------------------------------
    Rec<NW> *bufA, *bufB;
    if (int rc = dalloc(ctx, &bufA, nrec)) return rc;
    if (int rc = dalloc(ctx, &bufB, nrec)) return rc;
    unsigned long long *hist1, *off1, *cur1, *hist2 = nullptr, *off2 = nullptr, *cur2 = nullptr, *ucount, *uoff, *bucket_off;
    uint32_t *tile_start, *biglist, *bigcount, *runlen;
    if (int rc = dalloc(ctx, &hist1, F1)) return rc;
    if (int rc = dalloc(ctx, &off1, F1 + 1)) return rc;
    if (int rc = dalloc(ctx, &cur1, F1)) return rc;
    if (int rc = dalloc(ctx, &tile_start, F1 + 2)) return rc;
    if (s2) {
        if (int rc = dalloc(ctx, &hist2, nb)) return rc;
        if (int rc = dalloc(ctx, &off2, nb + 1)) return rc;
        if (int rc = dalloc(ctx, &cur2, nb)) return rc;
    }
    if (int rc = dalloc(ctx, &ucount, nb)) return rc;
    if (int rc = dalloc(ctx, &uoff, nb + 1)) return rc;
    if (int rc = dalloc(ctx, &biglist, nb)) return rc;
    if (int rc = dalloc(ctx, &bigcount, 1)) return rc;
    if (int rc = dalloc(ctx, &runlen, nrec / cap + nb + 2)) return rc;
    if (int rc = dalloc(ctx, &bucket_off, B + 1)) return rc;
    HIPCHK(hipMemsetAsync(hist1, 0, (size_t)F1 * 8, ctx->stream));
    HIPCHK(hipMemsetAsync(bigcount, 0, 4, ctx->stream));

    PassArgs a{};
    a.K = K;
    a.num_buckets = B;
    a.s1 = s1;
    a.s2 = s2;
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
    a.hist = hist1;
    tbegin(ctx, "l1_hist");
    if (from_reads) {
        if (int rc = pass_reads<NW, BIN_L1>(ctx, mode, false, a, masks)) return rc;
    } else {
        a.recs = d_recs;
        a.seg_off = seg1;
        a.nseg = 1;
        if (int rc = pass_recs<NW, BIN_L1>(ctx, false, a, nrec, tile_start)) return rc;
    }
    tend(ctx);
    tbegin(ctx, "l1_scan");
    if (int rc = scan_u64(ctx, hist1, off1, F1)) return rc;
    HIPCHK(hipMemcpyAsync(cur1, off1, (size_t)F1 * 8, hipMemcpyDeviceToDevice, ctx->stream));
    tend(ctx);
    a.cursor = cur1;
    a.out = bufA;
    tbegin(ctx, "l1_scatter");
    if (from_reads) {
        if (int rc = pass_reads<NW, BIN_L1>(ctx, mode, true, a, masks)) return rc;
    } else {
        if (int rc = pass_recs<NW, BIN_L1>(ctx, true, a, nrec, tile_start)) return rc;
    }
    tend(ctx);

    // ---- level 2 ----------------------------------------------------------------------------
    Rec<NW> *sortbuf = bufA, *other = bufB;
    const unsigned long long *fine_off = off1;
    if (s2) {
        HIPCHK(hipMemsetAsync(hist2, 0, (size_t)nb * 8, ctx->stream));
        a.recs = bufA;
        a.seg_off = off1;
        a.nseg = F1;
        a.F = F2;
        a.hist = hist2;
        tbegin(ctx, "l2_hist");
        if (int rc = pass_recs<NW, BIN_L2>(ctx, false, a, nrec, tile_start)) return rc;
        tend(ctx);
        tbegin(ctx, "l2_scan");
        if (int rc = scan_u64(ctx, hist2, off2, nb)) return rc;
        HIPCHK(hipMemcpyAsync(cur2, off2, (size_t)nb * 8, hipMemcpyDeviceToDevice, ctx->stream));
        tend(ctx);
        a.cursor = cur2;
        a.out = bufB;
        tbegin(ctx, "l2_scatter");
        if (int rc = pass_recs<NW, BIN_L2>(ctx, true, a, nrec, tile_start)) return rc;
        tend(ctx);
        sortbuf = bufB;
        other = bufA;
        fine_off = off2;
    }

    // ---- leaf sort + unique -----------------------------------------------------------------
    {
        size_t lds = (size_t)cap * NW * 8;
        if (int rc = set_lds(ctx, k_sort_small<NW>, lds)) return rc;
        if (int rc = set_lds(ctx, k_sort_big<NW>, lds)) return rc;
        tbegin(ctx, "sort_unique");
        hipLaunchKernelGGL((k_sort_small<NW>), dim3((unsigned)nb), dim3(BLK), lds, ctx->stream, (void *)sortbuf, fine_off,
                           (uint32_t)nb, cap, ucount, biglist, bigcount);
        HIPCHK(hipGetLastError());
        tend(ctx);
        tbegin(ctx, "sort_big");
        hipLaunchKernelGGL((k_sort_big<NW>), dim3(1024), dim3(BLK), lds, ctx->stream, (void *)sortbuf, (void *)other, fine_off, cap,
                           ucount, (const uint32_t *)biglist, (const uint32_t *)bigcount, runlen);
        HIPCHK(hipGetLastError());
        tend(ctx);
    }
    // ---- compact ----------------------------------------------------------------------------
    tbegin(ctx, "compact");
    if (int rc = scan_u64(ctx, ucount, uoff, nb)) return rc;
    hipLaunchKernelGGL((k_compact<NW>), dim3((unsigned)std::min<uint64_t>(nb, 1u << 20)), dim3(BLK), 0, ctx->stream,
                       (const void *)sortbuf, fine_off, (const unsigned long long *)ucount, (const unsigned long long *)uoff,
                       (uint32_t)nb, (void *)other);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_bucket_offsets, dim3((B + 1 + BLK - 1) / BLK), dim3(BLK), 0, ctx->stream,
                       (const unsigned long long *)uoff, B, (uint32_t)(nb / B), bucket_off);
    HIPCHK(hipGetLastError());
    tend(ctx);
    std::vector<unsigned long long> h(B + 1);
    HIPCHK(hipMemcpyAsync(h.data(), bucket_off, (size_t)(B + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (unsigned i = 0; i <= B; ++i) ctx->bucket_off[i] = h[i];
    ctx->n_records = h[B];
    ctx->d_result_buf = other;
    ctx->d_result = other;
    return 0;
}

int dispatch_count(smx_ctx *ctx, unsigned K, int mode, unsigned B, const void *d_recs, uint64_t n_in) {
    if (K < 1 || K > 128) return fail(ctx, SMX_INVALID_PARAMETER, "K=%u out of range [1,128]", K);
    if (B < 1) return fail(ctx, SMX_INVALID_PARAMETER, "num_buckets must be >= 1");
    if (mode != SMX_MODE_ALL && mode != SMX_MODE_CANONICAL) return fail(ctx, SMX_INVALID_PARAMETER, "bad mode %d", mode);
    HIPCHK(hipSetDevice(ctx->device));
    int rc;
    switch ((K + 31) / 32) {
        case 1: rc = run_count<1>(ctx, K, mode, B, d_recs, n_in); break;
        case 2: rc = run_count<2>(ctx, K, mode, B, d_recs, n_in); break;
        case 3: rc = run_count<3>(ctx, K, mode, B, d_recs, n_in); break;
        default: rc = run_count<4>(ctx, K, mode, B, d_recs, n_in); break;
    }
    if (rc == 0) {
        tcollect(ctx);
        free_temps(ctx, ctx->d_result_buf);
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
    const uint64_t nrec = mode == SMX_MODE_ALL ? 2 * nwin : nwin;
    if (nrec > capacity) return fail(ctx, SMX_MEMORY_LIMIT_EXCEEDED, "record buffer too small: need %llu", (unsigned long long)nrec);
    unsigned long long *hist, *off, *cur;
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
    if (nrec) {
        if (int rc = pass_reads<NW, BIN_OWNER>(ctx, mode, false, a, masks)) return rc;
    }
    if (int rc = scan_u64(ctx, hist, off, world)) return rc;
    HIPCHK(hipMemcpyAsync(cur, off, (size_t)world * 8, hipMemcpyDeviceToDevice, ctx->stream));
    a.cursor = cur;
    a.out = d_records;
    if (nrec) {
        if (int rc = pass_reads<NW, BIN_OWNER>(ctx, mode, true, a, masks)) return rc;
    }
    std::vector<unsigned long long> h(world);
    HIPCHK(hipMemcpyAsync(h.data(), hist, (size_t)world * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (unsigned i = 0; i < world; ++i) counts[i] = h[i];
    return 0;
}
}  // namespace

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
    return SMX_OK;
}

void smx_destroy(smx_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    smx_reads_clear(ctx);
    clear_result(ctx);
    free_temps(ctx);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *smx_last_error(const smx_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int smx_set_option(smx_ctx *ctx, const char *key, int64_t value) {
    if (!ctx || !key) return SMX_INVALID_PARAMETER;
    if (!strcmp(key, "leaf_cap")) ctx->opt_leaf_cap = value;
    else if (!strcmp(key, "s1")) ctx->opt_s1 = value;
    else if (!strcmp(key, "s2")) ctx->opt_s2 = value;
    else return fail(ctx, SMX_INVALID_PARAMETER, "unknown option %s", key);
    return SMX_OK;
}

int smx_reads_clear(smx_ctx *ctx) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    (void)hipSetDevice(ctx->device);
    for (auto &c : ctx->chunks)
        if (c.owned) {
            (void)hipFree(c.d_words);
            (void)hipFree(c.d_start);
            (void)hipFree(c.d_len);
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
    ctx->chunks.push_back(c);
    return SMX_OK;
}

int smx_submit_reads_ascii(smx_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n_reads) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (n_reads == 0) return SMX_OK;
    if (!bases || !offsets) return fail(ctx, SMX_INVALID_PARAMETER, "null read arrays");
    // host-side read preprocessing (the reference keeps parsing / N handling on the CPU as well):
    // longest run of ACGTacgt, first one on ties (longest_valid_wrapper.hpp:16-43)
    auto is_nucl = [](char ch) {
        switch (ch) {
            case 'A': case 'C': case 'G': case 'T': case 'a': case 'c': case 'g': case 't': return true;
            default: return false;
        }
    };
    std::vector<uint64_t> start(n_reads);
    std::vector<uint32_t> len(n_reads);
    std::vector<uint64_t> words;
    words.reserve((size_t)((offsets[n_reads] - offsets[0]) / 32 + 2));
    uint64_t pos = 0, cur = 0;
    unsigned fill = 0;
    for (uint64_t r = 0; r < n_reads; ++r) {
        const char *s = bases + offsets[r];
        const uint64_t n = offsets[r + 1] - offsets[r];
        uint64_t best_len = 0, best_pos = 0, run = 0;
        for (uint64_t i = 0; i <= n; ++i) {
            if (i < n && is_nucl(s[i])) {
                ++run;
            } else {
                if (run > best_len) {
                    best_len = run;
                    best_pos = i - run;
                }
                run = 0;
            }
        }
        if (best_len > 0xFFFFFFFFull) return fail(ctx, SMX_INVALID_INPUT_FORMAT, "read %llu longer than 2^32-1", (unsigned long long)r);
        start[r] = pos;
        len[r] = (uint32_t)best_len;
        for (uint64_t i = 0; i < best_len; ++i) {
            char ch = s[best_pos + i];
            uint64_t code = (ch == 'A' || ch == 'a') ? 0 : (ch == 'C' || ch == 'c') ? 1 : (ch == 'G' || ch == 'g') ? 2 : 3;
            cur |= code << fill;
            fill += 2;
            if (fill == 64) {
                words.push_back(cur);
                cur = 0;
                fill = 0;
            }
        }
        pos += best_len;
    }
    if (fill) words.push_back(cur);
    if (words.empty()) words.push_back(0);
    return smx_submit_reads_packed(ctx, words.data(), words.size(), start.data(), len.data(), n_reads);
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
    free_temps(ctx);
    return rc;
}

int smx_last_timings(const smx_ctx *ctx, const char **names, float *ms, int cap) {
    if (!ctx) return 0;
    int n = (int)ctx->tnames.size();
    for (int i = 0; i < n && i < cap; ++i) {
        if (names) names[i] = ctx->tnames[i].c_str();
        if (ms) ms[i] = ctx->tms[i];
    }
    return n;
}

}  // extern "C"
