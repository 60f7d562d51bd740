// spades_amd/csrc/smx_ctx.hpp — the context behind the C ABI: resident read chunks, result view, options, grow-only device
// arena, temp bookkeeping, stage timers (included by smx_api.hip; one translation unit).
#pragma once
#include <atomic>
#include <mutex>
#include <thread>

namespace {

struct ReadChunk {
    uint64_t *d_words = nullptr;
    uint64_t *d_start = nullptr;
    uint32_t *d_len = nullptr;
    uint64_t n_words = 0, n_reads = 0, n_bases = 0;
    bool owned = true;
    bool contigs = false;  // takes part in the construction, not in the coverage (trusted / previous-k contigs)
    // asynchronous submission (option "async_upload"): (start, len) first, then the 2-bit stream in pieces on the context's copy stream.
    // ev_meta: start / len are there and checked (h_ext: pinned, [0] furthest nucleotide, [1] reads beyond the stream);
    // piece_ev[p]: words [0, piece_end[p]) are there. Consumers make their stream wait (mark_windows; run_prededupe range by range).
    hipEvent_t ev_meta = nullptr;
    unsigned long long *h_ext = nullptr;
    unsigned long long *d_ext = nullptr;  // device side of h_ext: lives as long as the chunk (the copy stream writes it long after the submit returned)
    std::vector<uint64_t> piece_end;
    std::vector<hipEvent_t> piece_ev;
};

struct Timing {
    std::string name;
    hipEvent_t e0, e1;
};

}  // namespace

struct Arena {  // one virtual range, physically backed up to `mapped`
    bool tried = false, vmm = false;
    char *base = nullptr;
    size_t reserved = 0, gran = 0;
    size_t lo = 0, hi = 0;  // [0, lo) and [hi, reserved) are backed by physical memory
    struct Chunk {
        hipMemGenericAllocationHandle_t h{};
        bool mapped = false;
    };
    std::vector<Chunk> chunk;  // one entry per A.gran of the reserved range
    std::map<size_t, size_t> free_blocks;                                     // offset -> bytes, inside [0, mapped)
    std::unordered_map<void *, size_t> live;                                  // block -> bytes
    std::string last_err;                                                     // why the last growth failed
};

// Partition-major construction route (smx_pm.hip / smx_pm.hpp): side arrays the dedupe stage leaves next to its records, and the
// sorted tail of the k-mers of cut partitions.
struct PmState {
    bool active = false;  // the dedupe stage in flight writes partition-major output (set by the route, like smx_ctx::ext_mode)
    bool nx = false;      // ... with PLAIN k-mer records (no room for the byte in the last word): the bytes go to `mask` alone, those of cut partitions' survivors too
    unsigned long long *pinfo = nullptr, *cinfo = nullptr;
    uint32_t *meta = nullptr, *overflow = nullptr, *llink = nullptr;
    unsigned long long *pals = nullptr;
    uint8_t *mask = nullptr;
    // the node table of the clean chunks written by the dedupe stage itself (PmOut::tab, smx_superkmer.hip): asked for by the route (fuse_tab), allocated by
    // run_prededupe at the output's capacity, taken over by pm_route
    bool fuse_tab = false;
    bool keep_links = false;  // ... and the local links are written all the same (an early clipper will isolate whole chains: k_pm_isolate_chains follows them, 4 B per k-mer)
    unsigned long long *tab = nullptr, *tab_stats = nullptr;
    uint32_t *jmp = nullptr, *rbits = nullptr;
    uint32_t max_chunks = 0, nchunks = 0, T = 0, nkey = 0;
    unsigned m = 0, w = 0, pshift = 0;
    uint64_t nclean = 0, ndirty = 0;
    uint64_t nslots = 0, nfolded = 0;   // super-k-mer slots of the count behind this graph; k-mer instances in slots that were folded away
    unsigned dirty_B = 1;               // the dirty region is sorted bucket-major in this many hash buckets
    std::vector<uint64_t> dirty_boff;   // their offsets
    void *dk = nullptr;  // the k-mers of the dirty region without their bytes (sorted: what its rank directory indexes)
    smx::RankDir ddir{};
};

struct smx_ctx {
    int device = 0;
    Arena arena;
    // the arena is shared with ONE helper: the prewarm thread (smx_prewarm) maps chunks ahead of the calls that will need them while the
    // caller's thread reads its input; every entry of the allocator takes this lock, the helper holds it for one chunk at a time
    std::recursive_mutex arena_mu;
    std::thread prewarm_thr;
    std::atomic<bool> prewarm_stop{false};
    double arena_map_s = 0, arena_prewarm_s = 0;          // wall time inside hipMemCreate / hipMemMap / hipMemSetAccess: all of it, the helper's share
    size_t arena_map_bytes = 0, arena_prewarm_bytes = 0;  // ... and the bytes mapped
    size_t budget = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;  // uploads of asynchronous submissions
    hipStream_t side_stream = nullptr;  // construction, route 0: the successor table (k_pm_tab, k_pm_remote) runs here while the junction k-mers are sorted on `stream`
    int64_t opt_pm_fuse_tab = 1;        // route 0: 1 = the dedupe stage writes the node table of its chunks from LDS (no link array, no k_pm_tab pass over the clean chunks); 0 = k_pm_tab afterwards
    int64_t opt_pm_remote_mirror = 1;   // route 0: 1 = a successor outside its chunk is looked up from one end of the edge for both (k_pm_remote modes 1 + 2), 0 = from each end. (Measured: without the per-wave queue no gain — half-empty waves; with it 41.5-41.8 against 43.8-44.3 ms)
    int64_t opt_walk_pack = 1;          // route 0: 1 = word offset and edge index of the kept paths from one scan, packed in one word (smx_construct.hpp)
    int64_t opt_pm_full_retab = 0;      // route 0 with early clippers: 1 = the whole node table is made again after an edit (k_pm_tab + k_pm_remote), 0 = the edited k-mers' entries only
    int64_t opt_pm_overlap = 0;         // ... 1: measured (profiles/r06/bench_config3_successor_table_on_side_stream.json): side by side both get slower by what the
                                        // other takes (80 -> 92 ms, 48 -> 120 ms; step 470-479 ms either way) — both are bound by the fabric's random-sector rate
    int64_t opt_async_upload = 0;       // smx_submit_reads_packed returns before the copy is done (the host arrays stay valid until the reads are used)
    std::string err;
    std::vector<ReadChunk> chunks;
    // result of the last count
    void *d_result_buf = nullptr;  // allocation holding the result
    void *d_result = nullptr;
    // ... or, when the sorted-unique set does not fit the HBM budget, in host memory: chunks of whole buckets in file order
    struct HostChunk {
        char *data = nullptr;
        uint64_t n = 0;  // records
    };
    std::vector<HostChunk> h_result;
    bool result_on_host = false;
    // ... or, for a count started by smx_count_to_file, in the FILE: the out-of-core merge streams every merged bucket range to its place in the file
    // instead of keeping it (the reference's merge does: kmer_index_builder.hpp:346-430, fwrite in 1 Mi-record chunks) — host memory then holds the
    // spilled runs alone, and those shrink as the merge consumes them
    int sink_fd = -1;
    std::string sink_path;
    bool result_on_file = false;
    // ... or as the two strands of a both-strands count that is too large to hold merged (smx_pipeline.hpp: two_strand_finish): the
    // sorted canonical set and the sorted set of its reverse complements, both bucket-major; the accessors merge bucket by bucket
    struct TwoStrand {
        bool active = false;
        void *c = nullptr;                 // the canonical set, bucket-major
        uint64_t nc = 0, nr = 0;
        std::vector<uint64_t> boff_c;      // its bucket offsets
        std::vector<void *> rseg;          // the reverse complements, sorted in a few bucket ranges of their own (one block each)
        std::vector<std::pair<unsigned, unsigned>> rseg_range;  // ... the buckets [first, last) each of them holds
        std::vector<const void *> rb_ptr;  // per bucket: where its reverse-complement records start ...
        std::vector<uint64_t> rb_n;        // ... and how many there are
    } ts;
    uint64_t g_ext_bits = 0, g_ext_pals = 0;  // extension bits / palindromic (k+1)-mers among them in the k-mer file or shard built from EXT records
    PmState pm;               // partition-major construction route
    uint64_t g_route_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // smx_graph_route_stats of the last build
    bool g_pm = false;        // g_kmers holds EXT records in partition-major order (no sorted k-mer file yet: made on demand)
    bool g_pm_clipped = false;  // ... EXT records whose bytes are the UNCLIPPED masks (an early clipper edited g_mask after they were written): synced before the file is made
    bool g_pm_nx = false;     // ... plain k-mer records instead (the k-mer leaves the last word no 8 spare bits): the bytes live in g_mask alone
    bool pm_view_pending = false;  // the count-result view (smx_copy_final_kmers, smx_bucket_sizes, ...) stands for the k-mer file of that
                                   // graph, not made yet; any later count owns the view again (clear_result)
    int64_t opt_pm_route = -1;  // construction without the sort of the k-mers: -1 where it applies, 0 never, 1 = -1
    int64_t opt_nx_route = 1;   // ... also for k whose record has no 8 spare bits, on plain k-mer records (0: those k take the (k+1)-mer route as until round 5)
    bool ext_mode = false;    // the count in flight carries extension bytes in its records (EXT layout, smx_device.hpp): set by the construction
    std::set<void *> pool_blocks;  // blocks handed to the caller by smx_pool_alloc (returned at smx_destroy at the latest)
    void *x_owned = nullptr;  // output of smx_extract_partition_owned (released by the next extract / smx_extract_release)
    void *x_recv = nullptr;   // smx_exchange_buffer: receive side of the exchange, consumed by smx_count_records
    bool single_batch_only = false;  // count_reads: fail (memory limit) rather than cut the input into batches
    uint64_t n_records = 0, n_instances = 0;
    unsigned nw = 0, K = 0, num_buckets = 0;
    std::vector<uint64_t> bucket_off;
    // tuning / test hooks
    int64_t opt_leaf_cap = 0, opt_leaf_target = 0, opt_s1 = -1, opt_s2 = -1, opt_batch_records = 0;
    int64_t opt_sort_edges = 0, opt_keep_loops = 1;
    int64_t opt_skm_nkey_log2 = 0;  // > 0: the super-k-mer stage starts from 2^this minimizer partitions instead of 2^24 (12..28; it still doubles with the input)
    int64_t opt_device_loops = 1;  // 1: perfect loops by the kernels of smx_loops.hip (odd k; even k always on the host); 0: on the host (smx_loops_host.hpp).
                                   // Default 1 since round 5: on the MI355X the 9 937-loop golden of the real spades-gbuilder (6 M reads from 10 000 plasmids)
                                   // is written byte for byte on all three routes, the six plasmid tests take 10 s against 63 s through the host collector
    int64_t opt_flank_range = 50;     // FlankingCoverage averaging range ((k+1)-mers at either end of an edge)
    int64_t opt_submit_contigs = 0;   // reads submitted while this is 1 are contigs: construction yes, coverage no
    int64_t opt_early_at = 0;         // 1: the early A/T remover of the RNA pipelines before the tip clipper
    int64_t opt_early_tip_bound = 0;  // > 0: spades-core's early tip clipper with this length bound (RL - K there) before condensation
    int64_t opt_skm_stage = 1;     // pass 0 of the super-k-mer scan stages its output so that placing it needs no second scan
    int64_t opt_device_gfa = 1;    // GFA text formatted on the device when the link records live there (0: always the host writer)
    int64_t opt_device_links = 1;  // link records + vertices of the graph on the device (0: host, 2: also for tiny graphs)
    int64_t opt_two_strand = -1;  // both-strands count as canonical set + its reverse complements: -1 when the direct expansion does not fit HBM, 0 never,
                                  // 1 always (merged into one array when that fits), 2 always and left unmerged (tests of the bucket-wise accessors)
    int64_t opt_two_strand_parts = 0;  // > 0: the reverse complements of a two-strand count are sorted in this many bucket ranges (tests; 0 = as HBM requires)
    int64_t opt_single_batch = 0;  // 1: a count that does not fit one batch fails with the memory limit instead of taking batches / spilling (probes at size)
    int64_t opt_spill_merge_max = 0;  // > 0: the merge of the spilled runs takes at most this many records at once (tests: drives small inputs through the key-range split of a bucket)
    int64_t opt_spill = -1;  // sorted runs to host memory + merge by bucket ranges: -1 when the accumulated set outgrows HBM, 1 always (tests)
    int64_t opt_verify_lookups = 0;  // 1: rank lookups of k-mers that are known to be present still compare the record
    int64_t opt_dir_slots = -1;        // rank directory: slots per record (-1: 2, or 1 next to a resident (k+1)-mer file)
    int64_t opt_ext_presort = 1;       // ext route: merge the survivors of cut partitions before the sort (0: after it — the general merge; tests)
    int64_t opt_ext_route = -1;        // construction: k-mers AND their extension masks from one count of the reads (-1 when it applies, 0 never, 1 = -1)
    int64_t opt_kmers_from_reads = 1;  // construction: k-mer file counted from the resident reads (0: derived from the (k+1)-mer file)
    int64_t opt_derive_batches = 0;  // > 1: derive the k-mer file in this many bucket ranges (tests; 0 = as HBM requires)
    int64_t opt_keep_kpo = -1;       // keep the (k+1)-mer file after the masks are filled: -1 = if HBM allows, 0 = drop (coverage recounts)
    int64_t opt_joint_hist = 1;  // fuse the level-2 histogram into the level-1 histogram pass (records source)
    int64_t opt_prededupe = -1;  // super-k-mer pre-deduplication: -1 auto, 0 off, 1 on whenever K allows it
    int64_t opt_skm_cap = 0;     // instances per LDS dedupe chunk (0 = default)
    int64_t opt_skm_scap = 0;    // slots staged per chunk (0 = default)
    int64_t opt_skm_fold = 1;    // identical super-k-mers are folded by the chunk plan before they are expanded (0: tests)
    int64_t opt_leaf_grid = 0, opt_leaf_tab = 0;  // tuning experiments (tools/sweep.py)  // spades-core construction variant (debruijn_graph_constructor.hpp:590-604)
    // timings
    std::vector<Timing> timings;
    std::string tprefix;  // prepended to stage names (which part of a construction a pipeline run belongs to)
    std::vector<std::string> tnames, xnames;  // last count stages / last extract_partition stages
    std::vector<float> tms, xms;
    std::vector<void *> temps;  // allocations of the pipeline in flight
    // grow-only device arena: blocks are recycled across calls (hipMalloc/hipFree of tens of GB stalls for seconds)
    std::vector<std::pair<void *, size_t>> arena_free;  // cached blocks
    std::unordered_map<void *, size_t> arena_size;      // every live or cached block -> bytes
    size_t arena_live = 0;                              // bytes handed out and not yet returned
    // construction state (smx_build_graph)
    void *g_kpo = nullptr, *g_kmers = nullptr;
    uint8_t *g_mask = nullptr;
    uint64_t g_nkpo = 0, g_nkmers = 0;
    unsigned g_k = 0, g_nw = 0, g_B = 0;
    std::vector<uint64_t> g_kboff, g_kpoboff;
    smx::RankDir g_dir_kmers{}, g_dir_kpo{};  // .dir / .boff owned by the graph state
    bool g_ready = false;
    // distributed walks (smx_dwalk.hpp): the start de-edges of this rank's shard in k-mer-file order, the requests its chain k-mers have
    unsigned long long *dw_cand = nullptr;
    uint64_t dw_ncand = 0, dw_nchain = 0;
    bool dw_ready = false;
    unsigned long long *dw_loops = nullptr;  // after smx_shard_walks: local ranks of the shard's k-mers on perfect loops, ascending
    uint64_t dw_nloops = 0;
    int64_t opt_walk_chunk = 0, opt_walk_start_chunk = 0;  // smx_shard_walks: oriented nodes / start de-edges per exchange round (0: 2^26 / 2^22; tests make them small)
    int64_t opt_walk_fail_at = 0;                          // test hook: this rank's local step fails in that phase (1..5) of smx_shard_walks
    int64_t opt_walk_hop_bits = 24;                        // ... bits of a node's word that count the steps to its pointer (tests: chains at the limit)
    bool g_sharded_file = false;  // the graph was built from gathered unitigs: no k-mer file, no masks on this rank (smx_build_graph_from_unitigs)
    // the graph itself, resident in HBM (unitigs 2-bit packed, word-aligned starts; edges in the reference's enumeration order)
    uint64_t *g_uwords = nullptr;                          // [g_nuwords + 8]
    unsigned long long *g_eoffw = nullptr, *g_elen = nullptr;  // [g_ne] word offset / nucleotides of every unitig
    smx::node_t *g_estart = nullptr, *g_eend = nullptr;    // [g_ne] node of the first / last k-mer
    uint8_t *g_eself = nullptr;                            // [g_ne] s == RC(s)
    uint64_t g_ne = 0, g_nuwords = 0, g_nbases = 0, g_npaths = 0, g_nloops = 0;
    smx::Rec<2> *g_lrecs = nullptr;                        // sorted link records (rank << g_lsh, EdgeAndMask), [g_nlrec]
    unsigned long long *g_vstart = nullptr;                // first record of every vertex, in vertex-id order, [g_nv]
    uint64_t g_nlrec = 0, g_nv = 0;
    unsigned g_lsh = 0;
    bool g_links_dev = false;   // link records / vertices live in g_lrecs / g_vstart (else gh holds them)
    bool g_host_valid = false;  // gh mirrors the device graph
    bool g_dev_valid = false;   // the device arrays above describe the graph (false after a host-side edge sort until re-uploaded)
    uint64_t g_tip_kmers = 0, g_tips = 0;  // early tip clipper: k-mers isolated, tips removed
    uint64_t g_at_edges = 0, g_at_tip_kmers = 0;  // early A/T remover: length-1 edges marked, tip k-mers isolated
    uint64_t g_nkpo_total = 0;  // g_nkpo of the whole (k+1)-mer file while a shard installed by smx_graph_set_kpomers stands in for it
    std::vector<uint64_t> g_cov_hist;  // [c] = canonical (k+1)-mers with multiplicity c (after smx_graph_fill_coverage)
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

// ---- device arena -----------------------------------------------------------------------------------------------------------
// hipMalloc / hipFree of tens of GB stall for seconds, and a construction at BASELINE-config-3 scale asks for a few hundred blocks
// of every size up to 80 GB. So the context owns ONE virtual address range of the size of the device memory (HIP virtual memory
// management: hipMemAddressReserve / hipMemCreate / hipMemMap) and backs it with physical chunks the first time an address is
// needed; blocks are carved out first-fit with coalescing of neighbours, which costs nothing on the device. Mapping is slow too
// (measured ~17 ms per GiB), so nothing is ever unmapped before smx_destroy(): a second call with the same shape maps nothing.
// Placement keeps the range unfragmented: pipeline temporaries (freed wholesale between phases) grow from the bottom, long-lived
// blocks (reads, the k-mer files, the graph) from the top; a pipeline result that becomes long-lived is moved to the top
// (adopt_result). SMX_ARENA=malloc (or a HIP runtime without the VMM calls) falls back to a cache of hipMalloc'ed blocks.
constexpr size_t ARENA_ALIGN = 256;

// Arenas of destroyed contexts wait here for the next context on the same device instead of being unmapped: tearing a range down
// (hipMemUnmap / hipMemAddressFree) crashed inside the HIP runtime once in a few hundred context lifetimes on this stack (native
// backtraces from smx_destroy, immediately with MALLOC_PERTURB_: the runtime touches freed host memory), and a process that makes
// context after context — a multi-k pipeline, the tests — also saves the mapping time (~17 ms per GiB). A pooled arena keeps its
// physical memory; it is handed to the next context whose budget class matches (no budget: any unbudgeted arena of the device).
// SMX_ARENA_POOL=0 restores the teardown.
struct PooledArena {
    int device;
    size_t budget_reserved;  // 0 = made by a context without a budget, else its reserved size
    Arena a;
};
inline std::vector<PooledArena> &arena_pool() {
    static std::vector<PooledArena> pool;
    return pool;
}
inline std::mutex &arena_pool_mutex() {
    static std::mutex m;
    return m;
}
inline bool arena_pool_enabled() {
    const char *e = getenv("SMX_ARENA_POOL");
    return !(e && !strcmp(e, "0"));
}

// unmap and release every chunk, free the address range
inline void arena_teardown(Arena &A) {
    if (!A.vmm) return;
    (void)hipDeviceSynchronize();
    for (size_t ci = 0; ci < A.chunk.size(); ++ci)
        if (A.chunk[ci].mapped) {
            (void)hipMemUnmap(A.base + ci * A.gran, A.gran);
            (void)hipMemRelease(A.chunk[ci].h);
        }
    (void)hipMemAddressFree(A.base, A.reserved);
    A = Arena();
}

bool arena_vmm_init(smx_ctx *ctx) {
    Arena &A = ctx->arena;
    if (A.tried) return A.vmm;
    A.tried = true;
    const char *e = getenv("SMX_ARENA");
    if (e && !strcmp(e, "malloc")) return false;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
    if (arena_pool_enabled()) {
        const size_t g = getenv("SMX_ARENA_CHUNK_MB") ? std::max<size_t>((size_t)atoll(getenv("SMX_ARENA_CHUNK_MB")), 2) << 20 : (size_t)512 << 20;
        const size_t want = ctx->budget ? (ctx->budget + g - 1) / g * g + g : 0;
        std::lock_guard<std::mutex> lk(arena_pool_mutex());
        auto &pool = arena_pool();
        for (size_t i = 0; i < pool.size(); ++i)
            if (pool[i].device == ctx->device && pool[i].a.gran == g && pool[i].budget_reserved == want) {
                A = std::move(pool[i].a);
                pool.erase(pool.begin() + i);
                A.tried = A.vmm = true;
                A.last_err.clear();
                return true;
            }
        // a parked arena of this device that does not fit this context would keep its physical memory for nothing: it is torn down
        // before a new one is made (the pool never holds more than one arena per device)
        for (size_t i = 0; i < pool.size();)
            if (pool[i].device == ctx->device) {
                arena_teardown(pool[i].a);
                pool.erase(pool.begin() + i);
            } else {
                ++i;
            }
        (void)hipMemGetInfo(&free_b, &total_b);
    }
    // Every physical chunk has the same size: on this stack hipMemSetAccess rejects a mapping whose size differs from its
    // neighbour's in many combinations (tools/vmm_probe.hip: 2 MiB, 6 MiB, 64 MiB or 1 GiB chunks back to back all work, mixed
    // sizes fail with "invalid argument"), and the reported granularity (4 KiB) says nothing about it.
    A.gran = (size_t)512 << 20;
    if (const char *c = getenv("SMX_ARENA_CHUNK_MB")) A.gran = std::max<size_t>((size_t)atoll(c), 2) << 20;  // experiments
    // Never more than 94 % of what is free now: a box whose VRAM is mapped to the last chunk dies instead of returning an error
    // (measured the hard way: the page tables of these very mappings need VRAM too).
    A.reserved = (size_t)((double)free_b * 0.94) / A.gran * A.gran;
    if (A.reserved < A.gran) return false;
    if (ctx->budget) A.reserved = std::min(A.reserved, (ctx->budget + A.gran - 1) / A.gran * A.gran + A.gran);
    void *base = nullptr;
    if (hipMemAddressReserve(&base, A.reserved, 0, nullptr, 0) != hipSuccess || !base) {
        (void)hipGetLastError();
        return false;
    }
    A.base = (char *)base;
    A.chunk.assign(A.reserved / A.gran, Arena::Chunk{});
    A.lo = 0;
    A.hi = A.reserved;
    A.vmm = true;
    return true;
}
// back the chunks [c0, c1) with physical memory
double wall_now();
bool arena_map_chunks(smx_ctx *ctx, size_t c0, size_t c1) {
    Arena &A = ctx->arena;
    struct MapTimer {
        smx_ctx *c;
        double t0;
        size_t bytes;
        ~MapTimer() {
            c->arena_map_s += wall_now() - t0;
            c->arena_map_bytes += bytes;
        }
    } map_timer{ctx, wall_now(), (c1 - c0) * A.gran};
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = ctx->device;
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    size_t ci = c0;
    for (; ci < c1; ++ci) {
        char *at = A.base + ci * A.gran;
        hipMemGenericAllocationHandle_t h;
        hipError_t e = hipMemCreate(&h, A.gran, &prop, 0);
        if (e != hipSuccess) {
            A.last_err = std::string("hipMemCreate: ") + hipGetErrorString(e);
            break;
        }
        if ((e = hipMemMap(at, A.gran, 0, h, 0)) != hipSuccess) {
            A.last_err = std::string("hipMemMap: ") + hipGetErrorString(e);
            (void)hipMemRelease(h);
            break;
        }
        if ((e = hipMemSetAccess(at, A.gran, &acc, 1)) != hipSuccess) {
            A.last_err = std::string("hipMemSetAccess: ") + hipGetErrorString(e);
            (void)hipMemUnmap(at, A.gran);
            (void)hipMemRelease(h);
            break;
        }
        A.chunk[ci].h = h;
        A.chunk[ci].mapped = true;
    }
    if (ci < c1) {  // out of device memory: undo
        (void)hipGetLastError();
        while (ci > c0) {
            --ci;
            (void)hipMemUnmap(A.base + ci * A.gran, A.gran);
            (void)hipMemRelease(A.chunk[ci].h);
            A.chunk[ci].mapped = false;
        }
        return false;
    }
    return true;
}
void arena_add_free(Arena &A, size_t off, size_t sz) {  // with coalescing
    auto nx = A.free_blocks.lower_bound(off);
    if (nx != A.free_blocks.end() && off + sz == nx->first) {
        sz += nx->second;
        nx = A.free_blocks.erase(nx);
    }
    if (nx != A.free_blocks.begin()) {
        auto pv = std::prev(nx);
        if (pv->first + pv->second == off) {
            off = pv->first;
            sz += pv->second;
            A.free_blocks.erase(pv);
        }
    }
    A.free_blocks[off] = sz;
}
// make `want` free bytes end at the bottom mark (grow it upwards) or start at the top mark (grow it downwards)
bool arena_grow(smx_ctx *ctx, size_t want, bool top) {
    Arena &A = ctx->arena;
    A.last_err.clear();
    size_t have = 0;  // free bytes already adjacent to the mark
    if (!top) {
        auto it = A.free_blocks.lower_bound(A.lo);
        if (it != A.free_blocks.begin()) {
            auto pv = std::prev(it);
            if (pv->first + pv->second == A.lo) have = pv->second;
        }
    } else {
        auto it = A.free_blocks.find(A.hi);
        if (it != A.free_blocks.end()) have = it->second;
    }
    if (have >= want) return true;
    const size_t bytes = (want - have + A.gran - 1) / A.gran * A.gran;
    if (A.lo + bytes > A.hi) {
        A.last_err = "the arena is full";
        return false;
    }
    {   // somebody else (torch in the same process, another context) may have taken VRAM since the range was sized: never map into the
        // last 5 % of the device (a box whose VRAM is mapped to the last chunk dies instead of returning an error)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < bytes + total_b / 20) {
            A.last_err = "the device has no free memory left for the arena";
            return false;
        }
    }
    if (!top) {
        if (!arena_map_chunks(ctx, A.lo / A.gran, (A.lo + bytes) / A.gran)) return false;
        arena_add_free(A, A.lo, bytes);
        A.lo += bytes;
    } else {
        if (!arena_map_chunks(ctx, (A.hi - bytes) / A.gran, A.hi / A.gran)) return false;
        A.hi -= bytes;
        arena_add_free(A, A.hi, bytes);
    }
    return true;
}

// smx_prewarm: map `bottom` bytes above the bottom mark and `top` bytes below the top mark on a helper thread, one chunk per lock hold (a
// caller that needs a block meanwhile waits for at most one chunk, ~9 ms, and maps what it needs itself). Mapping costs ~17 ms per GiB and a
// process pays it once per address: a tool that builds ONE graph pays it inside its first build — 1.25 s for the 74 GiB of a 20 M-read
// construction — unless it happens while the tool is still reading its input. Nothing is promised: the helper stops
// where the device runs short, and a context with an HBM budget is left alone.
void arena_prewarm_join(smx_ctx *ctx) {
    ctx->prewarm_stop = true;
    if (ctx->prewarm_thr.joinable()) ctx->prewarm_thr.join();
    ctx->prewarm_stop = false;
}
void arena_prewarm_start(smx_ctx *ctx, size_t bottom, size_t top) {
    arena_prewarm_join(ctx);
    if (ctx->budget || (bottom == 0 && top == 0)) return;
    {   // never more than half of the device ahead of need: a hint must not take what another allocator of the process (RCCL's buffers, a framework) will ask for
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return;
        const size_t cap = total_b / 2;
        if (bottom + top > cap) {
            const double f = (double)cap / (double)(bottom + top);
            bottom = (size_t)((double)bottom * f);
            top = (size_t)((double)top * f);
        }
    }
    ctx->prewarm_thr = std::thread([ctx, bottom, top]() {
        if (hipSetDevice(ctx->device) != hipSuccess) return;
        size_t done_b = 0, done_t = 0;
        while (!ctx->prewarm_stop && (done_b < bottom || done_t < top)) {
            std::lock_guard<std::recursive_mutex> lk(ctx->arena_mu);
            if (!arena_vmm_init(ctx)) return;
            Arena &A = ctx->arena;
            const bool at_top = done_t < top && (done_b >= bottom || done_t * (bottom + 1) <= done_b * (top + 1));
            if (A.lo + A.gran > A.hi) return;
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < A.gran + total_b / 10) return;  // (the builds themselves stop at 5 %)
            const double t0 = wall_now();
            if (!at_top) {
                if (!arena_map_chunks(ctx, A.lo / A.gran, A.lo / A.gran + 1)) return;
                arena_add_free(A, A.lo, A.gran);
                A.lo += A.gran;
                done_b += A.gran;
            } else {
                if (!arena_map_chunks(ctx, A.hi / A.gran - 1, A.hi / A.gran)) return;
                A.hi -= A.gran;
                arena_add_free(A, A.hi, A.gran);
                done_t += A.gran;
            }
            ctx->arena_prewarm_s += wall_now() - t0;
            ctx->arena_prewarm_bytes += A.gran;
        }
    });
}

void *arena_get_malloc(smx_ctx *ctx, size_t bytes);
void arena_put_malloc(smx_ctx *ctx, void *p);

// first fit among the free blocks that start inside [r0, r1): ascending addresses, block taken from its low end — or descending,
// taken from its high end
void *arena_take(smx_ctx *ctx, size_t bytes, size_t r0, size_t r1, bool descending) {
    Arena &A = ctx->arena;
    size_t off = 0, sz = 0;
    bool found = false;
    if (!descending) {
        for (auto it = A.free_blocks.lower_bound(r0); it != A.free_blocks.end() && it->first < r1; ++it)
            if (it->second >= bytes) {
                off = it->first;
                sz = it->second;
                found = true;
                break;
            }
    } else {
        auto it = A.free_blocks.lower_bound(r1);
        while (it != A.free_blocks.begin()) {
            --it;
            if (it->first < r0) break;
            if (it->second >= bytes) {
                off = it->first;
                sz = it->second;
                found = true;
                break;
            }
        }
    }
    if (!found) return nullptr;
    A.free_blocks.erase(off);
    size_t at = off;
    if (!descending) {
        if (sz > bytes) A.free_blocks[off + bytes] = sz - bytes;
    } else {
        at = off + (sz - bytes);
        if (sz > bytes) A.free_blocks[off] = sz - bytes;
    }
    void *p = A.base + at;
    if (getenv("SMX_ARENA_CHECK")) {  // diagnostics: a block handed out must not overlap a live one ...
        for (auto &lb : A.live) {
            const size_t lo_ = (size_t)((char *)lb.first - A.base), hi_ = lo_ + lb.second;
            if (lo_ < at + bytes && at < hi_)
                fprintf(stderr, "[smx] ARENA: block [%zu, %zu) handed out over the live block [%zu, %zu) (lo %zu hi %zu)\n", at, at + bytes, lo_, hi_, A.lo, A.hi);
        }
    }
    if (getenv("SMX_ARENA_CHECK")) {  // ... and must lie inside mapped chunks
        for (size_t ci = at / A.gran; ci <= (at + bytes - 1) / A.gran; ++ci)
            if (ci >= A.chunk.size() || !A.chunk[ci].mapped) {
                fprintf(stderr, "[smx] ARENA: block [%zu, %zu) handed out over unmapped chunk %zu (lo %zu hi %zu reserved %zu, free block was [%zu, %zu), %s)\n", at,
                        at + bytes, ci, A.lo, A.hi, A.reserved, off, off + sz, descending ? "descending" : "ascending");
                break;
            }
    }
    A.live[p] = bytes;
    ctx->arena_live += bytes;
    if (const char *pz = getenv("SMX_ARENA_POISON")) {  // diagnostics: no block may be read before it is written — fill it with a byte that breaks what does
        (void)hipDeviceSynchronize();
        (void)hipMemset(p, atoi(pz) & 0xFF, bytes);
        (void)hipDeviceSynchronize();
    }
    return p;
}
// top = long-lived block (top region, highest address first); otherwise a temporary (bottom region, lowest address first).
// Each kind spills into the other region's free blocks only when its own region cannot grow any more.
void *arena_get(smx_ctx *ctx, size_t bytes, bool top = false) {
    std::lock_guard<std::recursive_mutex> lk(ctx->arena_mu);
    if (!arena_vmm_init(ctx)) return arena_get_malloc(ctx, bytes);
    Arena &A = ctx->arena;
    bytes = (bytes + ARENA_ALIGN - 1) / ARENA_ALIGN * ARENA_ALIGN;
    void *p;
    if (!top) {
        if ((p = arena_take(ctx, bytes, 0, A.lo, false))) return p;
        if (arena_grow(ctx, bytes, false) && (p = arena_take(ctx, bytes, 0, A.lo, false))) return p;
        if ((p = arena_take(ctx, bytes, A.hi, A.reserved, false))) return p;
        if (arena_grow(ctx, bytes, true) && (p = arena_take(ctx, bytes, A.hi, A.reserved, false))) return p;
    } else {
        if ((p = arena_take(ctx, bytes, A.hi, A.reserved, true))) return p;
        if (arena_grow(ctx, bytes, true) && (p = arena_take(ctx, bytes, A.hi, A.reserved, true))) return p;
        if ((p = arena_take(ctx, bytes, 0, A.lo, true))) return p;
        if (arena_grow(ctx, bytes, false) && (p = arena_take(ctx, bytes, 0, A.lo, true))) return p;
    }
    return nullptr;
}
void arena_put(smx_ctx *ctx, void *p) {
    if (!p) return;
    std::lock_guard<std::recursive_mutex> lk(ctx->arena_mu);
    Arena &A = ctx->arena;
    if (!A.vmm) {
        arena_put_malloc(ctx, p);
        return;
    }
    auto it = A.live.find(p);
    if (it == A.live.end()) return;  // not ours (borrowed pointer), or already returned: one owner only
    const size_t off = (size_t)((char *)p - A.base), sz = it->second;
    ctx->arena_live -= std::min(ctx->arena_live, sz);
    A.live.erase(it);
    arena_add_free(A, off, sz);
}
// a live block gives its tail back (a result buffer sized for the worst case)
void arena_shrink(smx_ctx *ctx, void *p, size_t bytes) {
    std::lock_guard<std::recursive_mutex> lk(ctx->arena_mu);
    Arena &A = ctx->arena;
    if (!p || !A.vmm) return;
    auto it = A.live.find(p);
    if (it == A.live.end()) return;
    bytes = std::max<size_t>((bytes + ARENA_ALIGN - 1) / ARENA_ALIGN * ARENA_ALIGN, ARENA_ALIGN);
    if (bytes >= it->second) return;
    const size_t off = (size_t)((char *)p - A.base), old = it->second;
    it->second = bytes;
    ctx->arena_live -= std::min(ctx->arena_live, old - bytes);
    arena_add_free(A, off + bytes, old - bytes);
}
// give the memory back to the device (smx_destroy: every block has been returned by then)
void arena_release(smx_ctx *ctx) {
    arena_prewarm_join(ctx);
    if (getenv("SMX_DEBUG") && ctx->arena.vmm)
        fprintf(stderr, "[smx] arena: %.1f GiB mapped in %.3f s (%.1f ms/GiB; the prewarm helper: %.1f GiB in %.3f s); bottom mark %.1f GiB, top region %.1f GiB of %.1f reserved\n",
                (double)ctx->arena_map_bytes / (1 << 30), ctx->arena_map_s, ctx->arena_map_bytes ? ctx->arena_map_s * 1e3 / ((double)ctx->arena_map_bytes / (1 << 30)) : 0.0,
                (double)ctx->arena_prewarm_bytes / (1 << 30), ctx->arena_prewarm_s, (double)ctx->arena.lo / (1 << 30),
                (double)(ctx->arena.reserved - ctx->arena.hi) / (1 << 30), (double)ctx->arena.reserved / (1 << 30));
    Arena &A = ctx->arena;
    if (A.vmm) {
        if (!A.live.empty()) {  // somebody still holds a block (a bug of the caller's bookkeeping): the range cannot go, say so
            size_t bytes = 0;
            for (auto &b : A.live) bytes += b.second;
            fprintf(stderr, "[smx] smx_destroy: %zu device blocks (%zu bytes) are still handed out; the context's device arena is left mapped\n", A.live.size(), bytes);
            return;
        }
        (void)hipDeviceSynchronize();
        if (arena_pool_enabled()) {
            // (a budgeted context whose reserved size was cut by the free memory of its day does not match its class: torn down below)
            const size_t cls = ctx->budget ? (ctx->budget + A.gran - 1) / A.gran * A.gran + A.gran : 0;
            if (cls == 0 || cls == A.reserved) {
                std::lock_guard<std::mutex> lk(arena_pool_mutex());
                auto &pool = arena_pool();
                for (size_t i = 0; i < pool.size();)  // one parked arena per device: an older one of another class gives its memory back now
                    if (pool[i].device == ctx->device) {
                        arena_teardown(pool[i].a);
                        pool.erase(pool.begin() + i);
                    } else {
                        ++i;
                    }
                pool.push_back(PooledArena{ctx->device, cls, std::move(A)});
                A = Arena();
                return;
            }
        }
        arena_teardown(A);
        return;
    }
    for (auto &b : ctx->arena_free) {
        ctx->arena_size.erase(b.first);
        (void)hipFree(b.first);
    }
    ctx->arena_free.clear();
}
// smx_trim. The VMM arena does NOT give physical memory back while its context lives: unmapping chunks (hipMemUnmap + hipMemRelease)
// and mapping memory again shortly afterwards — the very next count — lost writes on this stack: kernels ran, their atomics and stores
// never arrived ("0 super-k-mers", garbage masks; tools/trim_probe.py at 2 M reads, every time; gone under SMX_DEBUG's extra
// synchronisation; the same with a whole new address range, so it is the physical pages, not the addresses: most likely the
// driver's deferred clear of released VRAM racing with the next owner). tools/vmm_remap_probe.hip, which lets hipMalloc take the
// released pages first, sees nothing wrong. Until that window is understood the arena only grows; memory goes back at smx_destroy
// (parked arenas: at most one per device). The hipMalloc fallback (SMX_ARENA=malloc) frees its cache. Returns the bytes released.
size_t arena_trim(smx_ctx *ctx) {
    std::lock_guard<std::recursive_mutex> lk(ctx->arena_mu);
    Arena &A = ctx->arena;
    if (A.vmm) return 0;
    size_t freed = 0;
    for (auto &b : ctx->arena_free) {
        freed += b.second;
        ctx->arena_size.erase(b.first);
        (void)hipFree(b.first);
    }
    ctx->arena_free.clear();
    return freed;
}
// HBM still obtainable for new allocations (bytes): free blocks of the arena + what can still be mapped between its two marks, as
// far as the device has it, or what is left of the caller's budget
size_t arena_avail(smx_ctx *ctx) {
    std::lock_guard<std::recursive_mutex> lk(ctx->arena_mu);
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    size_t cached = 0;
    if (ctx->arena.vmm) {
        for (auto &b : ctx->arena.free_blocks) cached += b.second;
        free_b = std::min(free_b, ctx->arena.hi - ctx->arena.lo);
    } else {
        for (auto &b : ctx->arena_free) cached += b.second;
    }
    const size_t dev = (size_t)((double)(free_b + cached) * 0.94);
    if (!ctx->budget) return dev;
    return std::min(dev, ctx->budget > ctx->arena_live ? ctx->budget - ctx->arena_live : (size_t)0);
}

// largest free block of the arena (diagnostics: SMX_DEBUG lines that tell a full arena from a fragmented one)
size_t arena_largest_free(smx_ctx *ctx) {
    std::lock_guard<std::recursive_mutex> lk(ctx->arena_mu);
    size_t mx = 0;
    if (ctx->arena.vmm) {
        for (auto &b : ctx->arena.free_blocks) mx = std::max(mx, b.second);
        mx = std::max(mx, ctx->arena.hi > ctx->arena.lo ? ctx->arena.hi - ctx->arena.lo : (size_t)0);  // (what can still be mapped between the two marks)
    } else {
        for (auto &b : ctx->arena_free) mx = std::max(mx, b.second);
    }
    return mx;
}

// fallback: a cache of hipMalloc'ed blocks, best fit with at most 2x waste
void *arena_get_malloc(smx_ctx *ctx, size_t bytes) {
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
        ctx->arena_live += best;
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
    ctx->arena_live += bytes;
    return q;
}
void arena_put_malloc(smx_ctx *ctx, void *p) {
    auto it = ctx->arena_size.find(p);
    if (it == ctx->arena_size.end()) return;  // not ours
    for (auto &b : ctx->arena_free)
        if (b.first == p) return;  // already returned (an error path released it twice): one owner only
    ctx->arena_free.emplace_back(p, it->second);
    ctx->arena_live -= std::min(ctx->arena_live, it->second);
}

template <typename T>
int dalloc(smx_ctx *ctx, T **p, size_t count, bool temp = true) {
    size_t bytes = std::max<size_t>(count * sizeof(T), 256);
    void *q = arena_get(ctx, bytes, /*top=*/!temp);
    if (!q) return fail(ctx, SMX_MEMORY_LIMIT_EXCEEDED, "device allocation of %zu bytes failed (%s; arena: %zu mapped, %zu in use)", bytes,
                        ctx->arena.last_err.c_str(), ctx->arena.lo + (ctx->arena.reserved - ctx->arena.hi), ctx->arena_live);
    if (temp) ctx->temps.push_back(q);
    *p = (T *)q;
    return 0;
}

// a temp becomes a long-lived block of its new owner (count result, graph state)
void detach_temp(smx_ctx *ctx, void *p) {
    for (size_t i = 0; i < ctx->temps.size(); ++i)
        if (ctx->temps[i] == p) {
            ctx->temps.erase(ctx->temps.begin() + i);
            return;
        }
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
    t.name = ctx->tprefix + name;
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

}  // namespace
