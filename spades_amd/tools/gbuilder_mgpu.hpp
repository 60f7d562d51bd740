// spades_amd/tools/gbuilder_mgpu.hpp — spades-gbuilder on N GPUs of one node, C++ host over librccl (precedent for a
// construction spread over processes: hpcspades/mpi/stages/construction_mpi.cpp:303-412). One process per GPU, forked by the tool
// before any HIP call; the same steps, on the same C entry points, as spades_amd/dist.py: sharded_build_graph(walks = "gathered"):
//   every rank reads ITS share of the input (a byte range of an uncompressed 4-line FASTQ file, cut at records; every n-th read of
//   anything else) ->
//   route "ext" (k whose record has 8 spare bits): smx_extract_kmers_ext_owned -> ONE exchange -> smx_graph_shard_from_ext;
//   route "kpomers" (any k): sharded count of the canonical (k+1)-mers (smx_extract_partition_owned -> exchange -> smx_count_records),
//     smx_graph_shard_updates -> second exchange -> smx_graph_shard_build;
//   -> the owners' shards of {k-mer file, InOutMask bytes} (bucket ranges: rank order is file order) are gathered on the ranks that
//   build the graph — rank 0, or every rank with -c, whose coverage pass counts each rank's own reads against ONE OWNER'S SHARD of the
//   (k+1)-mer file at a time (never the whole file on a rank) and sums the raw edge coverages (ncclAllReduce) —
//   smx_build_graph_from_kmers; rank 0 writes the output.
// Exchanges are grouped ncclSend / ncclRecv between all pairs (xGMI is point to point: every pair has its own link, no ring).
// A graph whose gathered structure exceeds the HBM of the rank that would build it (BASELINE configs 4 and 5) takes the DISTRIBUTED WALKS
// instead (round 6; SMX_MGPU_WALKS=distributed forces them, =gathered refuses such a graph with the memory-limit code as rounds 4-5 did): the
// k-mer file stays sharded, smx_shard_walks does lookups / pointer doubling / chain fetch on the device with THIS host's collectives between
// its kernels (walk_collectives below), only the unitigs (2 bits per base + 25 B each: ~3 % of the k-mer file) and the k-mers of perfect loops
// are gathered, and smx_build_graph_from_unitigs derives link records and vertices. Same GFA, byte for byte.
#pragma once
#include <sys/prctl.h>
#include "rank_watchdog.hpp"
#include <sys/stat.h>
#include <sys/wait.h>
#include <signal.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "../../include/smx.h"
#include "read_share.hpp"

namespace smxtool {

#define GM_HIP(call)                                                                                \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            fprintf(stderr, "[rank %d] %s failed: %s\n", c.rank, #call, hipGetErrorString(e_));     \
            return SMX_DEVICE_ERROR;                                                                \
        }                                                                                           \
    } while (0)
#define GM_NCCL(call)                                                                               \
    do {                                                                                            \
        ncclResult_t r_ = (call);                                                                   \
        if (r_ != ncclSuccess) {                                                                    \
            fprintf(stderr, "[rank %d] %s failed: %s\n", c.rank, #call, ncclGetErrorString(r_));    \
            return SMX_DEVICE_ERROR;                                                                \
        }                                                                                           \
    } while (0)
#define GM_SMX(call)                                                                                \
    do {                                                                                            \
        const int rc_ = (call);                                                                     \
        if (rc_) {                                                                                  \
            fprintf(stderr, "[rank %d] %s\n", c.rank, smx_last_error(ctx));                         \
            return rc_;                                                                             \
        }                                                                                           \
    } while (0)

struct GbOptions {
    unsigned k = 21, nthreads = 1;
    bool coverage = false;
    int mode = 0;  // 0 unitigs, 1 GFA, 2 .grseq/.cvr, 3 FASTG
    std::string outfile;
    std::vector<std::string> files;
};

#define GM_MARK(what)                                                                              \
    smxtool::RankWatch::mark(what);                                                                \
    if (getenv("SMX_DEBUG")) {                                                                    \
        fprintf(stderr, "[rank %d] %s\n", c.rank, what);                                          \
        fflush(stderr);                                                                            \
    }

struct RankComm {
    int rank = 0, world = 1;
    ncclComm_t comm{};
    hipStream_t stream{};
    uint64_t *d_scratch = nullptr;  // device words for the small collectives (counts, flags): allocated once, not per call
    size_t scratch_words = 0;
};

// words per pair and round of an exchange (1 GiB; SMX_MGPU_ROUND_WORDS: a test hook that makes small inputs take several rounds)
inline uint64_t round_limit_words() {
    if (const char *e = getenv("SMX_MGPU_ROUND_WORDS")) return (uint64_t)std::max(1LL, atoll(e));
    return (uint64_t)1 << 27;
}

// ---- communicator ----------------------------------------------------------------------------------------------------------------
inline int comm_init(RankComm &c, const std::string &idfile) {
    ncclUniqueId id;
    if (c.rank == 0) {
        GM_NCCL(ncclGetUniqueId(&id));
        const std::string tmp = idfile + ".tmp";
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(&id, sizeof id, 1, f) != 1) return SMX_IO_ERROR;
        fclose(f);
        if (rename(tmp.c_str(), idfile.c_str()) != 0) return SMX_IO_ERROR;
    } else {
        bool got = false;
        for (int t = 0; t < 12000 && !got; ++t) {  // up to 2 minutes
            FILE *f = fopen(idfile.c_str(), "rb");
            if (f) {
                got = fread(&id, sizeof id, 1, f) == 1;
                fclose(f);
            }
            if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        if (!got) {
            fprintf(stderr, "[rank %d] no communicator id from rank 0\n", c.rank);
            return SMX_IO_ERROR;
        }
    }
    GM_HIP(hipSetDevice(c.rank));
    GM_NCCL(ncclCommInitRank(&c.comm, c.world, id, c.rank));
    GM_HIP(hipStreamCreate(&c.stream));
    return 0;
}

// every rank's `n` words -> all[world * n] on every rank
inline int scratch_words(RankComm &c, size_t words) {
    if (words <= c.scratch_words) return 0;
    if (c.d_scratch) (void)hipFree(c.d_scratch);
    c.d_scratch = nullptr;
    c.scratch_words = 0;
    const size_t want = std::max<size_t>(words, 4096);
    GM_HIP(hipMalloc((void **)&c.d_scratch, want * 8));
    c.scratch_words = want;
    return 0;
}
inline int all_gather_words(RankComm &c, const uint64_t *mine, size_t n, std::vector<uint64_t> &all) {
    if (int rc = scratch_words(c, n * (size_t)(c.world + 1))) return rc;
    uint64_t *d_in = c.d_scratch, *d_out = d_in + n;
    GM_HIP(hipMemcpy(d_in, mine, n * 8, hipMemcpyHostToDevice));
    GM_NCCL(ncclAllGather(d_in, d_out, n, ncclUint64, c.comm, c.stream));
    GM_HIP(hipStreamSynchronize(c.stream));
    all.resize(n * (size_t)c.world);
    GM_HIP(hipMemcpy(all.data(), d_out, all.size() * 8, hipMemcpyDeviceToHost));
    return 0;
}

// What a rank-local step returned, heard by every rank (as dist.py: _guarded): 0 when it went well everywhere; otherwise the largest
// code that is not the memory limit if there is one, and *memory_only says whether every failure was the memory limit (then all
// ranks may take another route together).
inline int agree(RankComm &c, int rc_local, bool *memory_only) {
    const uint64_t mine = (uint64_t)rc_local;
    std::vector<uint64_t> all;
    if (int rc = all_gather_words(c, &mine, 1, all)) return rc;
    int worst = 0, genuine = 0;
    for (uint64_t v : all)
        if (v) {
            worst = std::max(worst, (int)v);
            if ((int)v != SMX_MEMORY_LIMIT_EXCEEDED) genuine = std::max(genuine, (int)v);
        }
    *memory_only = worst != 0 && genuine == 0;
    return genuine ? genuine : worst;  // (a memory limit next to another error is not what the user has to hear about)
}

// ONE all-to-all of records of `wpr` words: counts[p] records of d_send (grouped by destination) go to rank p. The receive side
// comes from the library's pool (consumed by the call that follows) or from hipMalloc (*d_recv then belongs to the caller).
inline int exchange(RankComm &c, smx_ctx *ctx, const uint64_t *d_send, const std::vector<uint64_t> &counts, unsigned wpr, bool pool,
                    uint64_t **d_recv, uint64_t *n_recv) {
    std::vector<uint64_t> all;
    if (int rc = all_gather_words(c, counts.data(), (size_t)c.world, all)) return rc;
    std::vector<uint64_t> soff(c.world + 1, 0), roff(c.world + 1, 0);
    for (int p = 0; p < c.world; ++p) {
        soff[p + 1] = soff[p] + counts[p];
        roff[p + 1] = roff[p] + all[(size_t)p * c.world + c.rank];  // what rank p sends to me
    }
    *n_recv = roff[c.world];
    *d_recv = nullptr;
    if (pool) {
        void *p = nullptr;
        GM_SMX(smx_exchange_buffer(ctx, std::max<uint64_t>(*n_recv * wpr, 1), &p));
        *d_recv = (uint64_t *)p;
    } else {
        GM_HIP(hipMalloc((void **)d_recv, std::max<size_t>((size_t)*n_recv * wpr * 8, 8)));
    }
    // Rounds of <= 1 GiB per pair (one transfer of tens of GB was seen to stop short on this stack: dist.py, _a2a); every pair moves
    // each word once, on views of the two buffers. The segment that stays on this rank is a device copy (SMX_MGPU_SELF_RCCL=1 sends it
    // through ncclSend / ncclRecv as well: the one-rank tests exercise the RCCL calls that way).
    const bool self_rccl = getenv("SMX_MGPU_SELF_RCCL") != nullptr;
    const uint64_t LIM = round_limit_words();
    uint64_t mx = 0;
    for (size_t i = 0; i < all.size(); ++i) mx = std::max<uint64_t>(mx, all[i] * wpr);
    const uint64_t rounds = std::max<uint64_t>(1, (mx + LIM - 1) / LIM);
    GM_MARK(self_rccl ? "exchange: own segment through RCCL" : "exchange: own segment as a device copy")
    for (uint64_t r = 0; r < rounds && (c.world > 1 || self_rccl); ++r) {
        GM_NCCL(ncclGroupStart());
        for (int p = 0; p < c.world; ++p) {
            if (p == c.rank && !self_rccl) continue;
            const uint64_t s1 = soff[p + 1] * wpr, a = std::min(soff[p] * wpr + r * LIM, s1), b = std::min(a + LIM, s1);
            const uint64_t r1 = roff[p + 1] * wpr, e = std::min(roff[p] * wpr + r * LIM, r1), f = std::min(e + LIM, r1);
            if (b > a) GM_NCCL(ncclSend(d_send + a, b - a, ncclUint64, p, c.comm, c.stream));
            if (f > e) GM_NCCL(ncclRecv(*d_recv + e, f - e, ncclUint64, p, c.comm, c.stream));
        }
        GM_NCCL(ncclGroupEnd());
        smxtool::RankWatch::tick();
    }
    if (!self_rccl && counts[c.rank])
        GM_HIP(hipMemcpyAsync(*d_recv + roff[c.rank] * wpr, d_send + soff[c.rank] * wpr, (size_t)counts[c.rank] * wpr * 8, hipMemcpyDeviceToDevice, c.stream));
    GM_HIP(hipStreamSynchronize(c.stream));
    return 0;
}

// The shards (per[p] units of `unit` bytes on rank p, this rank's at d_mine) side by side in rank order on rank 0, or on every rank
// (to_all). *d_full (hipMalloc, the caller's) stays NULL on the ranks that receive nothing.
inline int gather_shards(RankComm &c, const void *d_mine, const std::vector<uint64_t> &per, size_t unit, bool to_all, void **d_full) {
    *d_full = nullptr;
    std::vector<uint64_t> off(c.world + 1, 0);
    for (int p = 0; p < c.world; ++p) off[p + 1] = off[p] + per[p];
    const bool dest = to_all || c.rank == 0;
    if (dest) {
        GM_HIP(hipMalloc(d_full, std::max<size_t>((size_t)off[c.world] * unit, 8)));
        if (per[c.rank]) GM_HIP(hipMemcpy((char *)*d_full + off[c.rank] * unit, d_mine, (size_t)per[c.rank] * unit, hipMemcpyDeviceToDevice));
        GM_HIP(hipDeviceSynchronize());  // (the library reads the buffer on a stream of its own)
    }
    if (c.world == 1) return 0;
    const uint64_t LIM = round_limit_words() * 8;  // bytes per pair and round, as in exchange()
    uint64_t mx = 0;
    for (int p = 0; p < c.world; ++p) mx = std::max<uint64_t>(mx, per[p] * unit);
    const uint64_t rounds = std::max<uint64_t>(1, (mx + LIM - 1) / LIM);
    for (uint64_t r = 0; r < rounds; ++r) {
        GM_NCCL(ncclGroupStart());
        for (int p = 0; p < c.world; ++p) {
            if (p == c.rank) continue;
            const uint64_t mine_b = per[c.rank] * unit, a = std::min(r * LIM, mine_b), b = std::min(a + LIM, mine_b);
            if (b > a && (to_all || p == 0)) GM_NCCL(ncclSend((const char *)d_mine + a, b - a, ncclUint8, p, c.comm, c.stream));
            const uint64_t theirs = per[p] * unit, e = std::min(r * LIM, theirs), f = std::min(e + LIM, theirs);
            if (dest && f > e) GM_NCCL(ncclRecv((char *)*d_full + off[p] * unit + e, f - e, ncclUint8, p, c.comm, c.stream));
        }
        GM_NCCL(ncclGroupEnd());
        smxtool::RankWatch::tick();
    }
    GM_HIP(hipStreamSynchronize(c.stream));
    return 0;
}

// `bytes` bytes at d_buf of rank `root` to d_buf of every other rank: grouped ncclSend / ncclRecv in rounds, like the gathers
inline int bcast_from(RankComm &c, int root, void *d_buf, uint64_t bytes) {
    if (c.world == 1 || bytes == 0) return 0;
    const uint64_t LIM = round_limit_words() * 8;
    for (uint64_t a = 0; a < bytes; a += LIM) {
        const uint64_t n = std::min(LIM, bytes - a);
        GM_NCCL(ncclGroupStart());
        if (c.rank == root) {
            for (int p = 0; p < c.world; ++p)
                if (p != root) GM_NCCL(ncclSend((const char *)d_buf + a, n, ncclUint8, p, c.comm, c.stream));
        } else {
            GM_NCCL(ncclRecv((char *)d_buf + a, n, ncclUint8, root, c.comm, c.stream));
        }
        GM_NCCL(ncclGroupEnd());
        smxtool::RankWatch::tick();
    }
    GM_HIP(hipStreamSynchronize(c.stream));
    return 0;
}

// ---- the collectives smx_shard_walks asks its caller for (include/smx.h: smx_collectives) ------------------------------------------------------
// exchange_counts: one all-gather of every rank's row of counts (world words each), this rank reads its column;
// alltoallv: grouped ncclSend / ncclRecv between all pairs, in rounds of <= 1 GiB per pair (both ends of a pair know that pair's size, so they
//   cut it into the same pieces: no agreement on a global number of rounds is needed), the segment that stays here as a device copy;
// allreduce_u64: ncclAllReduce of the few words on the scratch block.
struct WalkColl {
    RankComm *c;
};
inline int walk_cb_counts(void *user, const uint64_t *send_counts, uint64_t *recv_counts) {
    RankComm &c = *((WalkColl *)user)->c;
    std::vector<uint64_t> all;
    if (int rc = all_gather_words(c, send_counts, (size_t)c.world, all)) return rc;
    for (int p = 0; p < c.world; ++p) recv_counts[p] = all[(size_t)p * c.world + c.rank];
    smxtool::RankWatch::tick();
    return 0;
}
inline int walk_cb_alltoallv(void *user, const void *d_send, const uint64_t *send_counts, void *d_recv, const uint64_t *recv_counts, unsigned unit) {
    RankComm &c = *((WalkColl *)user)->c;
    const bool self_rccl = getenv("SMX_MGPU_SELF_RCCL") != nullptr;
    const uint64_t LIM = round_limit_words() * 8;  // bytes per pair and round
    std::vector<uint64_t> soff(c.world + 1, 0), roff(c.world + 1, 0);
    uint64_t mx = 0;
    for (int p = 0; p < c.world; ++p) {
        soff[p + 1] = soff[p] + send_counts[p] * unit;
        roff[p + 1] = roff[p] + recv_counts[p] * unit;
        if (p != c.rank || self_rccl) mx = std::max<uint64_t>(mx, std::max(send_counts[p], recv_counts[p]) * unit);
    }
    const uint64_t rounds = (mx + LIM - 1) / LIM;
    for (uint64_t r = 0; r < rounds; ++r) {
        GM_NCCL(ncclGroupStart());
        for (int p = 0; p < c.world; ++p) {
            if (p == c.rank && !self_rccl) continue;
            const uint64_t a = std::min(soff[p] + r * LIM, soff[p + 1]), b = std::min(a + LIM, soff[p + 1]);
            const uint64_t e = std::min(roff[p] + r * LIM, roff[p + 1]), f = std::min(e + LIM, roff[p + 1]);
            if (b > a) GM_NCCL(ncclSend((const char *)d_send + a, b - a, ncclUint8, p, c.comm, c.stream));
            if (f > e) GM_NCCL(ncclRecv((char *)d_recv + e, f - e, ncclUint8, p, c.comm, c.stream));
        }
        GM_NCCL(ncclGroupEnd());
        smxtool::RankWatch::tick();
    }
    if (!self_rccl && send_counts[c.rank])
        GM_HIP(hipMemcpyAsync((char *)d_recv + roff[c.rank], (const char *)d_send + soff[c.rank], (size_t)send_counts[c.rank] * unit, hipMemcpyDeviceToDevice, c.stream));
    GM_HIP(hipStreamSynchronize(c.stream));
    return 0;
}
inline int walk_cb_allreduce(void *user, uint64_t *values, unsigned n, int op) {
    RankComm &c = *((WalkColl *)user)->c;
    if (int rc = scratch_words(c, n)) return rc;
    GM_HIP(hipMemcpy(c.d_scratch, values, (size_t)n * 8, hipMemcpyHostToDevice));
    GM_NCCL(ncclAllReduce(c.d_scratch, c.d_scratch, n, ncclUint64, op == 1 ? ncclMax : ncclSum, c.comm, c.stream));
    GM_HIP(hipStreamSynchronize(c.stream));
    GM_HIP(hipMemcpy(values, c.d_scratch, (size_t)n * 8, hipMemcpyDeviceToHost));
    smxtool::RankWatch::tick();
    return 0;
}

// The graph from the DISTRIBUTED walks: this rank's shard of {k-mer file, masks} is in the context; afterwards the ranks that build (rank 0, or
// every rank with -c) hold the whole graph — link records and vertices from the gathered unitigs — and no rank ever held more than its bucket
// range of the k-mer file.
inline int graph_by_distributed_walks(RankComm &c, smx_ctx *ctx, unsigned k, unsigned nb, const std::vector<uint64_t> &kmers_per, uint64_t total_kmers,
                                      uint64_t n_kpo_all, bool to_all, bool builds) {
    const unsigned nwk = (k + 31) / 32;
    WalkColl wc{&c};
    smx_collectives co{};
    co.user = &wc;
    co.rank = (unsigned)c.rank;
    co.world = (unsigned)c.world;
    co.exchange_counts = walk_cb_counts;
    co.alltoallv = walk_cb_alltoallv;
    co.allreduce_u64 = walk_cb_allreduce;
    uint64_t winfo[4] = {0, 0, 0, 0};
    GM_MARK("distributed walks: smx_shard_walks")
    GM_SMX(smx_shard_walks(ctx, kmers_per.data(), &co, winfo));  // (a failure on any rank comes back on every rank: nobody is left in a collective)
    const uint64_t ne = winfo[0], nwd = winfo[1], nl = winfo[2];
    if (c.rank == 0 && getenv("SMX_DEBUG")) fprintf(stderr, "[rank 0] distributed walks: %llu doubling rounds\n", (unsigned long long)winfo[3]);
    uint64_t me2[3] = {ne, nwd, nl};
    std::vector<uint64_t> every2;
    if (int rc = all_gather_words(c, me2, 3, every2)) return rc;
    std::vector<uint64_t> ne_r(c.world), nwd_r(c.world), nl_r(c.world), first(c.world + 1, 0);
    uint64_t ne_all = 0, nwd_all = 0, nl_all = 0;
    for (int p = 0; p < c.world; ++p) {
        ne_all += (ne_r[p] = every2[(size_t)p * 3]);
        nwd_all += (nwd_r[p] = every2[(size_t)p * 3 + 1]);
        nl_all += (nl_r[p] = every2[(size_t)p * 3 + 2]);
        first[p + 1] = first[p] + kmers_per[p];
    }
    uint64_t *d_words = nullptr, *d_len = nullptr, *d_st = nullptr, *d_en = nullptr, *d_loops = nullptr, *d_lk = nullptr;
    uint8_t *d_sf = nullptr, *d_lm = nullptr;
    GM_HIP(hipMalloc((void **)&d_words, std::max<size_t>((size_t)nwd * 8, 8)));
    GM_HIP(hipMalloc((void **)&d_len, std::max<size_t>((size_t)ne * 8, 8)));
    GM_HIP(hipMalloc((void **)&d_st, std::max<size_t>((size_t)ne * 8, 8)));
    GM_HIP(hipMalloc((void **)&d_en, std::max<size_t>((size_t)ne * 8, 8)));
    GM_HIP(hipMalloc((void **)&d_sf, std::max<size_t>((size_t)ne, 8)));
    GM_HIP(hipMalloc((void **)&d_loops, std::max<size_t>((size_t)nl * 8, 8)));
    GM_HIP(hipMalloc((void **)&d_lk, std::max<size_t>((size_t)nl * nwk * 8, 8)));
    GM_HIP(hipMalloc((void **)&d_lm, std::max<size_t>((size_t)nl, 8)));
    GM_SMX(smx_shard_unitigs_copy(ctx, d_words, d_len, d_st, d_en, d_sf));
    GM_SMX(smx_shard_walk_loops(ctx, d_loops));
    GM_SMX(smx_shard_gather_kmers(ctx, d_loops, nl, d_lk, d_lm));
    void *g_words = nullptr, *g_len = nullptr, *g_st = nullptr, *g_en = nullptr, *g_sf = nullptr, *g_loops = nullptr, *g_lk = nullptr, *g_lm = nullptr;
    GM_MARK("distributed walks: gather of the unitigs")
    if (int rc = gather_shards(c, d_words, nwd_r, 8, to_all, &g_words)) return rc;
    if (int rc = gather_shards(c, d_len, ne_r, 8, to_all, &g_len)) return rc;
    if (int rc = gather_shards(c, d_st, ne_r, 8, to_all, &g_st)) return rc;
    if (int rc = gather_shards(c, d_en, ne_r, 8, to_all, &g_en)) return rc;
    if (int rc = gather_shards(c, d_sf, ne_r, 1, to_all, &g_sf)) return rc;
    if (int rc = gather_shards(c, d_loops, nl_r, 8, to_all, &g_loops)) return rc;
    if (int rc = gather_shards(c, d_lk, nl_r, (size_t)nwk * 8, to_all, &g_lk)) return rc;
    if (int rc = gather_shards(c, d_lm, nl_r, 1, to_all, &g_lm)) return rc;
    for (void *p : {(void *)d_words, (void *)d_len, (void *)d_st, (void *)d_en, (void *)d_sf, (void *)d_loops, (void *)d_lk, (void *)d_lm}) (void)hipFree(p);
    if (builds) {
        // the loop k-mers go in as host arrays in k-mer-file order with GLOBAL ranks (rank order is file order: the shards are bucket ranges)
        std::vector<uint64_t> h_lr(std::max<uint64_t>(nl_all, 1)), h_lk(std::max<uint64_t>(nl_all * nwk, 1));
        std::vector<uint8_t> h_lm(std::max<uint64_t>(nl_all, 1));
        if (nl_all) {
            GM_HIP(hipMemcpy(h_lr.data(), g_loops, (size_t)nl_all * 8, hipMemcpyDeviceToHost));
            GM_HIP(hipMemcpy(h_lk.data(), g_lk, (size_t)nl_all * nwk * 8, hipMemcpyDeviceToHost));
            GM_HIP(hipMemcpy(h_lm.data(), g_lm, (size_t)nl_all, hipMemcpyDeviceToHost));
            uint64_t o = 0;
            for (int p = 0; p < c.world; ++p)
                for (uint64_t i = 0; i < nl_r[p]; ++i) h_lr[o++] += first[p];
        }
        GM_SMX(smx_build_graph_from_unitigs(ctx, k, nb, total_kmers, n_kpo_all, (const uint64_t *)g_words, nwd_all, (const uint64_t *)g_len, (const uint64_t *)g_st,
                                            (const uint64_t *)g_en, (const uint8_t *)g_sf, ne_all, h_lr.data(), h_lk.data(), h_lm.data(), nl_all));
    } else {
        GM_SMX(smx_graph_clear(ctx));  // (this rank's shard has done its part)
    }
    for (void *p : {g_words, g_len, g_st, g_en, g_sf, g_loops, g_lk, g_lm})
        if (p) (void)hipFree(p);
    return 0;
}

// sharded count of the canonical K-mers: afterwards the context's count result is this rank's bucket range of the file
inline int sharded_count_canonical(RankComm &c, smx_ctx *ctx, unsigned K, unsigned nb, uint64_t *n_mine, std::vector<uint64_t> &sizes) {
    const unsigned nw = (K + 31) / 32;
    std::vector<uint64_t> counts(c.world, 0);
    const void *p = nullptr;
    GM_SMX(smx_extract_partition_owned(ctx, K, SMX_MODE_CANONICAL, nb, (unsigned)c.world, &p, counts.data()));
    uint64_t *d_recv = nullptr, n_recv = 0;
    if (int rc = exchange(c, ctx, (const uint64_t *)p, counts, nw, true, &d_recv, &n_recv)) return rc;
    GM_SMX(smx_extract_release(ctx));
    GM_SMX(smx_count_records(ctx, K, nb, d_recv, n_recv));
    GM_SMX(smx_count_info(ctx, n_mine, nullptr, nullptr));
    sizes.assign(nb, 0);
    GM_SMX(smx_bucket_sizes(ctx, sizes.data()));
    return 0;
}

inline int gb_rank_main(int rank, int world, const GbOptions &o, const std::string &idfile) {
    RankComm c;
    c.rank = rank;
    c.world = world;
    smxtool::RankWatch::arm(rank);
    smxtool::RankWatch::mark("smx_create");
    smx_ctx *ctx = nullptr;
    if (int rc = smx_create(&ctx, rank, 0)) {
        fprintf(stderr, "[rank %d] no usable MI355X device %d (smx_create -> %d)\n", rank, rank, rc);
        return rc;
    }
    smxtool::RankWatch::mark("communicator: ncclGetUniqueId / ncclCommInitRank");
    if (int rc = comm_init(c, idfile)) return rc;
    GM_MARK("communicator up")
    // input
    unsigned sub = 1;
    if (const char *e = getenv("SMX_MGPU_PARTS")) sub = (unsigned)std::max(1, atoi(e));
    for (const std::string &fn : o.files)
        for (unsigned i = 0; i < sub; ++i) {
            int rc;
            try {
                rc = submit_share(ctx, fn, (unsigned)rank * sub + i, (unsigned)world * sub);
            } catch (const std::string &s) {
                fprintf(stderr, "%s\n", s.c_str());
                return SMX_INVALID_INPUT_FORMAT;
            }
            if (rc == -1) {
                fprintf(stderr, "File %s doesn't exist or can't be read!\n", fn.c_str());
                return SMX_INPUT_FILE_NOT_FOUND;
            }
            if (rc) {
                fprintf(stderr, "[rank %d] %s\n", rank, smx_last_error(ctx));
                return rc;
            }
        }
    GM_MARK("input submitted")
    const unsigned k = o.k, K1 = k + 1, nb = 10 * o.nthreads, nw = (K1 + 31) / 32;  // (an odd k and k + 1 take the same number of words)
    const bool ext = smx_kmers_with_masks_supported(k) != 0 && !getenv("SMX_MGPU_KPOMERS");
    const bool cov = o.coverage && o.mode != 0;  // (-c does nothing for --unitigs: gbuilder.cpp:191-199)
    uint64_t n_kpo = 0, n_kmers = 0, stats[2] = {0, 0};
    std::vector<uint64_t> kpo_sizes(nb, 0), ksizes(nb, 0);
    uint64_t *d_kpo_mine = nullptr;  // this rank's bucket range of the (k+1)-mer file, kept for -c
    if (cov || !ext) {
        if (int rc = sharded_count_canonical(c, ctx, K1, nb, &n_kpo, kpo_sizes)) return rc;
        if (cov) {
            GM_HIP(hipMalloc((void **)&d_kpo_mine, std::max<size_t>((size_t)n_kpo * nw * 8, 8)));
            if (n_kpo) GM_SMX(smx_copy_kmers_device(ctx, d_kpo_mine));
        }
    }
    bool ext_now = ext;
    if (ext) {
        // The one-exchange route keeps every distinct k-mer of the rank's reads and then of its bucket range in HBM at once; where that
        // does not fit on SOME rank, ALL ranks hear of it and take the (k+1)-mer route together (the single-GPU library falls back the
        // same way). Any other failure ends the run on every rank.
        std::vector<uint64_t> counts(world, 0);
        const void *p = nullptr;
        bool memory_only = false;
        int rc_local = smx_extract_kmers_ext_owned(ctx, k, nb, (unsigned)world, &p, counts.data());
        if (rc_local) fprintf(stderr, "[rank %d] %s\n", rank, smx_last_error(ctx));
        int worst = agree(c, rc_local, &memory_only);
        if (!worst) {
            uint64_t *d_recv = nullptr, n_recv = 0;
            if (int rc = exchange(c, ctx, (const uint64_t *)p, counts, nw, true, &d_recv, &n_recv)) return rc;
            GM_SMX(smx_extract_release(ctx));
            rc_local = smx_graph_shard_from_ext(ctx, k, nb, (unsigned)world, (unsigned)rank, d_recv, n_recv);
            if (rc_local) fprintf(stderr, "[rank %d] %s\n", rank, smx_last_error(ctx));
            worst = agree(c, rc_local, &memory_only);
        }
        if (worst) {
            if (!memory_only) return worst;
            if (rank == 0) fprintf(stderr, "the one-exchange route does not fit on some rank: all ranks take the route by the (k+1)-mer count\n");
            (void)smx_extract_release(ctx);   // whatever the abandoned route left behind goes before the retry: the send buffer,
            (void)smx_exchange_release(ctx);  // an unconsumed receive buffer,
            (void)smx_graph_clear(ctx);       // the shard of a rank whose own step had gone well
            ext_now = false;
            // (also when -c had counted the (k+1)-mers before: that result went when the abandoned route extracted its k-mers)
            if (int rc = sharded_count_canonical(c, ctx, K1, nb, &n_kpo, kpo_sizes)) return rc;
        } else {
            GM_SMX(smx_graph_shard_ext_stats(ctx, stats));
        }
    }
    if (!ext_now) {
        uint64_t *d_upd = nullptr;
        GM_HIP(hipMalloc((void **)&d_upd, std::max<size_t>((size_t)2 * n_kpo * (nw + 1) * 8, 8)));
        std::vector<uint64_t> ucounts(world, 0);
        GM_SMX(smx_graph_shard_updates(ctx, k, nb, (unsigned)world, d_upd, 2 * n_kpo, ucounts.data()));
        uint64_t *d_recv = nullptr, n_recv = 0;
        if (int rc = exchange(c, ctx, d_upd, ucounts, nw + 1, false, &d_recv, &n_recv)) return rc;
        (void)hipFree(d_upd);
        GM_SMX(smx_graph_shard_build(ctx, k, nb, (unsigned)world, (unsigned)rank, d_recv, n_recv));
        (void)hipFree(d_recv);
    }
    GM_SMX(smx_graph_shard_info(ctx, &n_kmers, ksizes.data()));
    GM_MARK("owner-side shard built")
    // what every rank has: [k-mers, (k+1)-mers, extension bits, palindromic (k+1)-mers, k-mer bucket sizes, (k+1)-mer bucket sizes]
    std::vector<uint64_t> me(4 + 2 * (size_t)nb), every;
    me[0] = n_kmers, me[1] = n_kpo, me[2] = stats[0], me[3] = stats[1];
    for (unsigned b = 0; b < nb; ++b) me[4 + b] = ksizes[b], me[4 + nb + b] = kpo_sizes[b];
    if (int rc = all_gather_words(c, me.data(), me.size(), every)) return rc;
    std::vector<uint64_t> kmers_per(world), kpo_per(world), g_ksizes(nb, 0), g_psizes(nb, 0);
    uint64_t total_kmers = 0, total_kpo = 0, bits = 0;
    for (int p = 0; p < world; ++p) {
        const uint64_t *e = every.data() + (size_t)p * me.size();
        kmers_per[p] = e[0], kpo_per[p] = e[1];
        total_kmers += e[0], total_kpo += e[1], bits += e[2] + e[3];
        for (unsigned b = 0; b < nb; ++b) g_ksizes[b] += e[4 + b], g_psizes[b] += e[4 + nb + b];
    }
    uint64_t n_kpo_all = total_kpo;
    if (ext_now) {  // every non-palindromic (k+1)-mer set two extension bits somewhere, a palindromic one a single bit
        if (bits % 2) {
            fprintf(stderr, "[rank %d] odd number of extension bits over all shards\n", rank);
            return SMX_DEVICE_ERROR;
        }
        n_kpo_all = bits / 2;
        if (cov && n_kpo_all != total_kpo) {
            fprintf(stderr, "[rank %d] the masks (%llu (k+1)-mers) and the (k+1)-mer count (%llu) disagree\n", rank, (unsigned long long)n_kpo_all,
                    (unsigned long long)total_kpo);
            return SMX_DEVICE_ERROR;
        }
    }
    // gather {k-mers, masks} where a graph is built: everywhere with -c (every rank counts its own reads on it), else on rank 0
    const bool builds = cov || rank == 0;
    bool distributed = false;
    {   // Does the gathered structure fit where it is built? The gathered copy and the library's own cost (8 * words + 1) B per k-mer of
        // the WHOLE graph each, the successor table 16 B more (dist.py decides by the same figure). Every rank hears the answer: where it
        // does not fit on some rank, ALL ranks take the distributed walks together (SMX_MGPU_WALKS=gathered: all leave with the reference's
        // "memory limit" code instead, as rounds 4-5 did; =distributed: the walks whatever the size).
        size_t dev_free = 0, dev_total = 0, arena_free = 0;
        GM_HIP(hipMemGetInfo(&dev_free, &dev_total));
        (void)smx_arena_free_bytes(ctx, &arena_free);
        const double need = (double)total_kmers * (2.0 * (8.0 * nw + 1.0) + 16.0), have = (double)dev_free + (double)arena_free;
        uint64_t fits = (!builds || need <= have) ? 1 : 0;
        if (getenv("SMX_MGPU_ASSUME_FREE_BYTES")) fits = (!builds || need <= atof(getenv("SMX_MGPU_ASSUME_FREE_BYTES"))) ? 1 : 0;  // (test hook)
        const char *wenv = getenv("SMX_MGPU_WALKS");
        const bool force_d = wenv && !strcmp(wenv, "distributed"), force_g = wenv && !strcmp(wenv, "gathered");
        std::vector<uint64_t> all_fit;
        if (int rc = all_gather_words(c, &fits, 1, all_fit)) return rc;
        distributed = force_d;
        for (int p = 0; p < world && !distributed; ++p)
            if (!all_fit[p]) {
                if (force_g) {
                    if (rank == p || (rank == 0 && all_fit[0]))
                        fprintf(stderr, "[rank %d] the graph's k-mers and masks (%llu k-mers, %.1f GB to build from) do not fit rank %d's free device memory%s "
                                        "and SMX_MGPU_WALKS=gathered forbids the distributed walks\n",
                                rank, (unsigned long long)total_kmers, need / 1e9, p, rank == p ? "" : " (reported by that rank)");
                    return SMX_MEMORY_LIMIT_EXCEEDED;
                }
                if (rank == 0)
                    fprintf(stderr, "the graph's k-mers and masks (%llu k-mers, %.1f GB to build from) do not fit rank %d's free device memory: the k-mer file stays "
                                    "sharded, unitigs by distributed walks\n", (unsigned long long)total_kmers, need / 1e9, p);
                distributed = true;
            }
    }
    if (distributed) {
        if (int rc = graph_by_distributed_walks(c, ctx, k, nb, kmers_per, total_kmers, n_kpo_all, cov, builds)) return rc;
    } else {
        uint64_t *d_my_k = nullptr;
        uint8_t *d_my_m = nullptr;
        GM_HIP(hipMalloc((void **)&d_my_k, std::max<size_t>((size_t)n_kmers * nw * 8, 8)));
        GM_HIP(hipMalloc((void **)&d_my_m, std::max<size_t>((size_t)n_kmers, 8)));
        if (n_kmers) GM_SMX(smx_graph_shard_copy(ctx, d_my_k, d_my_m));
        void *d_full_k = nullptr, *d_full_m = nullptr;
        if (int rc = gather_shards(c, d_my_k, kmers_per, (size_t)nw * 8, cov, &d_full_k)) return rc;
        if (int rc = gather_shards(c, d_my_m, kmers_per, 1, cov, &d_full_m)) return rc;
        (void)hipFree(d_my_k);
        (void)hipFree(d_my_m);
        if (builds) GM_SMX(smx_build_graph_from_kmers(ctx, k, nb, d_full_k, d_full_m, total_kmers, g_ksizes.data(), n_kpo_all));
        if (d_full_k) (void)hipFree(d_full_k);
        if (d_full_m) (void)hipFree(d_full_m);
    }
    uint64_t info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (builds) GM_SMX(smx_graph_info(ctx, info));
    GM_MARK(distributed ? "graph built from the gathered unitigs (distributed walks)" : "graph built from the gathered shards")
    if (cov) {
        // The counters of the coverage pass are keyed by (k+1)-mer, and no rank may hold that whole file: shard by shard (dist.py does the
        // same) — the owner of a bucket range sends its shard to everybody, every rank installs it as a (k+1)-mer file of those buckets
        // alone (the lookups of the coverage kernels miss everything else), counts ITS reads against it and sums what the unitigs'
        // (k+1)-mers of the shard collected. Partial sums add up (mod 2^32, like the reference's counters).
        if (rank == 0) printf("Filling coverage index\n");
        const uint64_t ne = info[2];
        std::vector<uint32_t> raw(std::max<uint64_t>(ne, 1), 0), part(std::max<uint64_t>(ne, 1), 0);
        for (int s_ = 0; s_ < world; ++s_) {
            const uint64_t n_s = kpo_per[s_];
            if (!n_s) continue;
            const unsigned b0 = smx_rank_first_bucket(nb, (unsigned)world, (unsigned)s_), b1 = smx_rank_first_bucket(nb, (unsigned)world, (unsigned)s_ + 1);
            std::vector<uint64_t> sizes_s(nb, 0);
            uint64_t sum_s = 0;
            for (unsigned b = b0; b < b1; ++b) sum_s += (sizes_s[b] = g_psizes[b]);
            if (sum_s != n_s) {
                fprintf(stderr, "[rank %d] rank %d owns %llu (k+1)-mers, its buckets hold %llu\n", rank, s_, (unsigned long long)n_s, (unsigned long long)sum_s);
                return SMX_DEVICE_ERROR;
            }
            void *d_shard = d_kpo_mine;
            if (s_ != rank) GM_HIP(hipMalloc(&d_shard, (size_t)n_s * nw * 8));
            if (int rc = bcast_from(c, s_, d_shard, n_s * (uint64_t)nw * 8)) return rc;
            GM_SMX(smx_graph_set_kpomers(ctx, d_shard, n_s, sizes_s.data()));
            if (s_ != rank) (void)hipFree(d_shard);
            GM_SMX(smx_graph_fill_coverage(ctx));  // this rank's reads against that shard
            GM_SMX(smx_graph_copy_coverage(ctx, part.data()));
            for (uint64_t e = 0; e < ne; ++e) raw[e] += part[e];
        }
        (void)hipFree(d_kpo_mine);
        uint32_t *d_cov = nullptr;
        GM_HIP(hipMalloc((void **)&d_cov, raw.size() * 4));
        GM_HIP(hipMemcpy(d_cov, raw.data(), raw.size() * 4, hipMemcpyHostToDevice));
        GM_NCCL(ncclAllReduce(d_cov, d_cov, raw.size(), ncclUint32, ncclSum, c.comm, c.stream));  // (wraps like the reference's uint32 counters)
        GM_HIP(hipStreamSynchronize(c.stream));
        GM_HIP(hipMemcpy(raw.data(), d_cov, raw.size() * 4, hipMemcpyDeviceToHost));
        (void)hipFree(d_cov);
        GM_SMX(smx_graph_set_coverage(ctx, raw.data(), ne));
    }
    if (rank == 0) {
        printf("Extracting unbranching paths finished. %llu sequences extracted\n", (unsigned long long)(info[2] - info[3]));
        printf("Collecting perfect loops finished. %llu loops collected\n", (unsigned long long)info[3]);
        printf("Saving %s to %s\n", o.mode == 1 ? "graph" : "unitigs", o.outfile.c_str());
        GM_SMX(o.mode == 1   ? smx_graph_write_gfa(ctx, o.outfile.c_str(), "SPAdes-4.3.0-dev")
               : o.mode == 2 ? smx_graph_write_spades(ctx, o.outfile.c_str())
               : o.mode == 3 ? smx_graph_write_fastg(ctx, o.outfile.c_str())
                             : smx_graph_write_unitigs(ctx, o.outfile.c_str()));
        unlink(idfile.c_str());
    }
    GM_MARK("output written")
    // The work is done and every collective has completed; what follows only gives resources back. Should that not finish (a
    // communicator teardown that waits for ever has been seen on other stacks), the rank leaves with success after a grace period
    // that lets the other ranks finish theirs.
    smxtool::RankWatch::teardown_begins();
    smxtool::RankWatch::mark("teardown: ncclCommDestroy");
    ncclCommDestroy(c.comm);
    smxtool::RankWatch::mark("teardown: hipStreamDestroy");
    (void)hipStreamDestroy(c.stream);
    if (c.d_scratch) (void)hipFree(c.d_scratch);
    smxtool::RankWatch::mark("teardown: smx_destroy");
    smx_destroy(ctx);
    smxtool::RankWatch::done();
    GM_MARK("rank done")
    return 0;
}

// fork one process per GPU (nothing of HIP has been touched yet in this process) and wait for them; the first rank that fails takes
// the others down (a rank that leaves between two collectives would leave them waiting in the next one for ever)
inline int gb_run_sharded(int world, const GbOptions &o) {
    for (const std::string &f : o.files) {
        FILE *t = fopen(f.c_str(), "rb");
        if (!t) {
            fprintf(stderr, "File %s doesn't exist or can't be read!\n", f.c_str());
            return SMX_INPUT_FILE_NOT_FOUND;
        }
        fclose(t);
    }
    const std::string idfile = o.outfile + ".smx_nccl_id";
    // RCCL between processes of one node shares device memory through dmabuf handles: the host driver of these boxes supports nothing else
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    unlink(idfile.c_str());
    std::vector<pid_t> kids;
    for (int r = 0; r < world; ++r) {
        const pid_t pid = fork();
        if (pid < 0) {
            for (pid_t kk : kids) kill(kk, SIGKILL);
            for (pid_t kk : kids) waitpid(kk, nullptr, 0);
            return SMX_DEVICE_ERROR;
        }
        if (pid == 0) {
            prctl(PR_SET_PDEATHSIG, SIGKILL);  // a rank never outlives the tool (a killed tool would leave its ranks on the GPUs)
            int rc;
            try {
                rc = gb_rank_main(r, world, o, idfile);
            } catch (const std::string &s) {
                fprintf(stderr, "%s\n", s.c_str());
                rc = SMX_INVALID_INPUT_FORMAT;
            }
            fflush(stdout);
            fflush(stderr);
            _exit(rc);
        }
        kids.push_back(pid);
    }
    int rc = 0;
    size_t left = kids.size();
    while (left) {
        int st = 0;
        const pid_t pid = waitpid(-1, &st, 0);
        if (pid < 0) break;
        bool ours = false;
        for (pid_t &kk : kids)
            if (kk == pid) {
                kk = -1;
                ours = true;
            }
        if (!ours) continue;
        --left;
        const int code = WIFEXITED(st) ? WEXITSTATUS(st) : SMX_DEVICE_ERROR;
        if (code && !rc) {
            rc = code;
            fprintf(stderr, "a rank failed with code %d: stopping the other ranks\n", code);
            for (pid_t kk : kids)
                if (kk > 0) kill(kk, SIGKILL);
        }
    }
    unlink(idfile.c_str());
    return rc;
}

}  // namespace smxtool
