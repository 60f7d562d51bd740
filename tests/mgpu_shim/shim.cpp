// tests/mgpu_shim/shim.cpp — TEST INFRASTRUCTURE (CPU tier). The C++ multi-GPU hosts of the CLI tools (spades_amd/tools/gbuilder_mgpu.hpp,
// kmercount_mgpu.hpp) compiled as they are, with the three libraries they call replaced:
//   hip*   -> host memory (malloc / memcpy; a "device pointer" is a host pointer),
//   nccl*  -> ONE callback into the Python test, which runs the collective over torch.distributed / gloo between the test's ranks,
//   smx_*  -> the same callback, answered by the oracle-backed engine doubles of tests/test_dist_cpu.py.
// What is under test is the hosts' own logic at world size > 1: who reads which part of the input, offsets and counts of the
// exchanges, their rounds, the order of the gathered shards, the bookkeeping of bucket sizes and (k+1)-mer counts, the coverage sum.
// Every forwarded call is (name, up to 10 integer / pointer arguments as long long).
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../spades_amd/tools/gbuilder_mgpu.hpp"
#include "../../spades_amd/tools/kmercount_mgpu.hpp"

typedef long long (*shim_cb_t)(const char *, long long *);
static shim_cb_t g_cb = nullptr;
static std::string g_err;

static long long fwd(const char *name, int n, ...) {
    long long a[12] = {0};
    va_list ap;
    va_start(ap, n);
    for (int i = 0; i < n && i < 12; ++i) a[i] = va_arg(ap, long long);
    va_end(ap);
    if (!g_cb) return SMX_DEVICE_ERROR;
    return g_cb(name, a);
}
#define LL(x) ((long long)(x))

extern "C" {

void mgpu_shim_set_callback(shim_cb_t f) { g_cb = f; }
void mgpu_shim_set_error(const char *s) { g_err = s ? s : ""; }

int mgpu_shim_run_gbuilder(int rank, int world, unsigned k, unsigned nthreads, int coverage, int mode, const char *outfile, const char *files_nl,
                           const char *idfile) {
    smxtool::GbOptions o;
    o.k = k, o.nthreads = nthreads, o.coverage = coverage != 0, o.mode = mode, o.outfile = outfile;
    std::string all = files_nl;
    size_t p = 0;
    while (p < all.size()) {
        size_t e = all.find('\n', p);
        if (e == std::string::npos) e = all.size();
        if (e > p) o.files.push_back(all.substr(p, e - p));
        p = e + 1;
    }
    try {
        return smxtool::gb_rank_main(rank, world, o, idfile);
    } catch (const std::string &s) {
        fprintf(stderr, "%s\n", s.c_str());
        return SMX_INVALID_INPUT_FORMAT;
    }
}

int mgpu_shim_run_kmercount(int rank, int world, unsigned K, const char *workdir, const char *files_nl) {
    std::vector<std::string> files;
    std::string all = files_nl;
    size_t p = 0;
    while (p < all.size()) {
        size_t e = all.find('\n', p);
        if (e == std::string::npos) e = all.size();
        if (e > p) files.push_back(all.substr(p, e - p));
        p = e + 1;
    }
    return smxtool::sharded_rank_main(rank, world, K, workdir, files);
}

// ---- HIP: host memory ---------------------------------------------------------------------------------------------------------------
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t *s) {
    *s = nullptr;
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n) {
    *p = malloc(n ? n : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void *p) {
    free(p);
    return hipSuccess;
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) {
    memmove(d, s, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) {
    memmove(d, s, n);
    return hipSuccess;
}
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) {
    *free_b = (size_t)200 << 30, *total_b = (size_t)288 << 30;
    return hipSuccess;
}
const char *hipGetErrorString(hipError_t) { return "hip (shim)"; }

// ---- RCCL: forwarded ----------------------------------------------------------------------------------------------------------------
static long long elem_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclUint8: return 1;
        case ncclUint32: return 4;
        case ncclUint64: return 8;
        default: return 0;
    }
}
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof *id);
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int world, ncclUniqueId, int rank) {
    *comm = (ncclComm_t)(uintptr_t)1;
    return fwd("ncclCommInitRank", 2, LL(world), LL(rank)) ? ncclInternalError : ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t) { return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t) { return "nccl (shim)"; }
ncclResult_t ncclGroupStart(void) { return fwd("ncclGroupStart", 0) ? ncclInternalError : ncclSuccess; }
ncclResult_t ncclGroupEnd(void) { return fwd("ncclGroupEnd", 0) ? ncclInternalError : ncclSuccess; }
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t, hipStream_t) {
    return fwd("ncclSend", 3, LL(buf), LL(count) * elem_bytes(t), LL(peer)) ? ncclInternalError : ncclSuccess;
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t, hipStream_t) {
    return fwd("ncclRecv", 3, LL(buf), LL(count) * elem_bytes(t), LL(peer)) ? ncclInternalError : ncclSuccess;
}
ncclResult_t ncclAllGather(const void *in, void *out, size_t count, ncclDataType_t t, ncclComm_t, hipStream_t) {
    return fwd("ncclAllGather", 3, LL(in), LL(out), LL(count) * elem_bytes(t)) ? ncclInternalError : ncclSuccess;
}
ncclResult_t ncclAllReduce(const void *in, void *out, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t, hipStream_t) {
    if (t == ncclUint64 && (op == ncclSum || op == ncclMax))  // the few words of the walks' collectives
        return fwd("ncclAllReduceU64", 4, LL(in), LL(out), LL(count), LL(op == ncclMax ? 1 : 0)) ? ncclInternalError : ncclSuccess;
    if (t != ncclUint32 || op != ncclSum) return ncclInvalidArgument;
    return fwd("ncclAllReduceU32Sum", 3, LL(in), LL(out), LL(count)) ? ncclInternalError : ncclSuccess;
}

// ---- the C ABI of the product library: forwarded ----------------------------------------------------------------------------------------
int smx_create(smx_ctx **out, int device, size_t) {
    *out = (smx_ctx *)(uintptr_t)1;
    return (int)fwd("smx_create", 1, LL(device));
}
void smx_destroy(smx_ctx *) {}
int smx_arena_free_bytes(smx_ctx *, size_t *bytes) {
    *bytes = (size_t)10 << 30;
    return SMX_OK;
}
const char *smx_last_error(const smx_ctx *) { return g_err.c_str(); }
void *smx_pinned_alloc(size_t n) { return malloc(n ? n : 1); }
void smx_pinned_free(void *p) { free(p); }
int smx_submit_fastq_text(smx_ctx *, const char *text, uint64_t n, int is_final, uint64_t *n_reads, uint64_t *consumed) {
    return (int)fwd("smx_submit_fastq_text", 5, LL(text), LL(n), LL(is_final), LL(n_reads), LL(consumed));
}
int smx_submit_reads_ascii(smx_ctx *, const char *bases, const uint64_t *off, uint64_t n) {
    return (int)fwd("smx_submit_reads_ascii", 3, LL(bases), LL(off), LL(n));
}
int smx_kmers_with_masks_supported(unsigned k) { return (int)fwd("smx_kmers_with_masks_supported", 1, LL(k)); }
// (pure arithmetic in the library, include/smx.h: the owner of a bucket range)
unsigned smx_rank_first_bucket(unsigned num_buckets, unsigned world, unsigned rank) { return (unsigned)(((uint64_t)rank * num_buckets + world - 1) / world); }
int smx_extract_partition_owned(smx_ctx *, unsigned K, int mode, unsigned nb, unsigned world, const void **d, uint64_t *counts) {
    return (int)fwd("smx_extract_partition_owned", 6, LL(K), LL(mode), LL(nb), LL(world), LL(d), LL(counts));
}
int smx_extract_release(smx_ctx *) { return (int)fwd("smx_extract_release", 0); }
int smx_exchange_release(smx_ctx *) { return (int)fwd("smx_exchange_release", 0); }
int smx_graph_clear(smx_ctx *) { return (int)fwd("smx_graph_clear", 0); }
int smx_exchange_buffer(smx_ctx *, uint64_t n_words, void **d) { return (int)fwd("smx_exchange_buffer", 2, LL(n_words), LL(d)); }
int smx_count_records(smx_ctx *, unsigned K, unsigned nb, const void *d, uint64_t n) { return (int)fwd("smx_count_records", 4, LL(K), LL(nb), LL(d), LL(n)); }
int smx_count_info(const smx_ctx *, uint64_t *n, unsigned *wpr, uint64_t *inst) { return (int)fwd("smx_count_info", 3, LL(n), LL(wpr), LL(inst)); }
int smx_bucket_sizes(const smx_ctx *, uint64_t *sizes) { return (int)fwd("smx_bucket_sizes", 1, LL(sizes)); }
int smx_copy_bucket(const smx_ctx *, unsigned b, void *dst) { return (int)fwd("smx_copy_bucket", 2, LL(b), LL(dst)); }
int smx_copy_kmers_device(const smx_ctx *, void *d) { return (int)fwd("smx_copy_kmers_device", 1, LL(d)); }
int smx_extract_kmers_ext_owned(smx_ctx *, unsigned k, unsigned nb, unsigned world, const void **d, uint64_t *counts) {
    return (int)fwd("smx_extract_kmers_ext_owned", 5, LL(k), LL(nb), LL(world), LL(d), LL(counts));
}
int smx_graph_shard_from_ext(smx_ctx *, unsigned k, unsigned nb, unsigned world, unsigned rank, const void *d, uint64_t n) {
    return (int)fwd("smx_graph_shard_from_ext", 6, LL(k), LL(nb), LL(world), LL(rank), LL(d), LL(n));
}
int smx_graph_shard_ext_stats(const smx_ctx *, uint64_t *st) { return (int)fwd("smx_graph_shard_ext_stats", 1, LL(st)); }
int smx_graph_shard_updates(smx_ctx *, unsigned k, unsigned nb, unsigned world, void *d, uint64_t cap, uint64_t *counts) {
    return (int)fwd("smx_graph_shard_updates", 6, LL(k), LL(nb), LL(world), LL(d), LL(cap), LL(counts));
}
int smx_graph_shard_build(smx_ctx *, unsigned k, unsigned nb, unsigned world, unsigned rank, const void *d, uint64_t n) {
    return (int)fwd("smx_graph_shard_build", 6, LL(k), LL(nb), LL(world), LL(rank), LL(d), LL(n));
}
int smx_graph_shard_info(const smx_ctx *, uint64_t *n, uint64_t *sizes) { return (int)fwd("smx_graph_shard_info", 2, LL(n), LL(sizes)); }
int smx_graph_shard_copy(const smx_ctx *, void *dk, void *dm) { return (int)fwd("smx_graph_shard_copy", 2, LL(dk), LL(dm)); }
int smx_build_graph_from_kmers(smx_ctx *, unsigned k, unsigned nb, const void *dk, const void *dm, uint64_t n, const uint64_t *sizes, uint64_t n_kpo) {
    return (int)fwd("smx_build_graph_from_kmers", 7, LL(k), LL(nb), LL(dk), LL(dm), LL(n), LL(sizes), LL(n_kpo));
}
// distributed walks: the one call is answered by the test with its torch restatement of the algorithm (tests/dwalk_torch_double.py), which moves
// its data through THIS host's collectives (the callbacks in *coll are the host's own functions: counts, grouped ncclSend / ncclRecv, all-reduce)
int smx_shard_walks(smx_ctx *, const uint64_t *kmers_per_rank, const smx_collectives *coll, uint64_t *info) {
    return (int)fwd("smx_shard_walks", 9, LL(kmers_per_rank), LL(coll->user), LL(coll->rank), LL(coll->world), LL(coll->exchange_counts), LL(coll->alltoallv),
                    LL(coll->allreduce_u64), LL(info), LL(0));
}
int smx_shard_unitigs_copy(const smx_ctx *, uint64_t *w, uint64_t *ln, uint64_t *st, uint64_t *en, uint8_t *sf) {
    return (int)fwd("smx_shard_unitigs_copy", 5, LL(w), LL(ln), LL(st), LL(en), LL(sf));
}
int smx_shard_walk_loops(const smx_ctx *, uint64_t *d) { return (int)fwd("smx_shard_walk_loops", 1, LL(d)); }
int smx_shard_gather_kmers(smx_ctx *, const uint64_t *ranks, uint64_t n, void *dk, uint8_t *dm) { return (int)fwd("smx_shard_gather_kmers", 4, LL(ranks), LL(n), LL(dk), LL(dm)); }
int smx_build_graph_from_unitigs(smx_ctx *, unsigned k, unsigned nb, uint64_t n_kmers, uint64_t n_kpo, const uint64_t *words, uint64_t n_words, const uint64_t *ln,
                                 const uint64_t *st, const uint64_t *en, const uint8_t *sf, uint64_t ne, const uint64_t *lr, const uint64_t *lk, const uint8_t *lm, uint64_t nl) {
    long long a[15] = {LL(k), LL(nb), LL(n_kmers), LL(n_kpo), LL(words), LL(n_words), LL(ln), LL(st), LL(en), LL(sf), LL(ne), LL(lr), LL(lk), LL(lm), LL(nl)};
    return (int)fwd("smx_build_graph_from_unitigs", 1, LL(a));
}
int smx_graph_info(const smx_ctx *, uint64_t *info) { return (int)fwd("smx_graph_info", 1, LL(info)); }
int smx_graph_set_kpomers(smx_ctx *, const void *d, uint64_t n, const uint64_t *sizes) { return (int)fwd("smx_graph_set_kpomers", 3, LL(d), LL(n), LL(sizes)); }
int smx_graph_fill_coverage(smx_ctx *) { return (int)fwd("smx_graph_fill_coverage", 0); }
int smx_graph_copy_coverage(const smx_ctx *, uint32_t *raw) { return (int)fwd("smx_graph_copy_coverage", 1, LL(raw)); }
int smx_graph_set_coverage(smx_ctx *, const uint32_t *raw, uint64_t n) { return (int)fwd("smx_graph_set_coverage", 2, LL(raw), LL(n)); }
int smx_graph_write_gfa(smx_ctx *, const char *path, const char *) { return (int)fwd("smx_graph_write", 2, LL(path), LL(1)); }
int smx_graph_write_spades(smx_ctx *, const char *path) { return (int)fwd("smx_graph_write", 2, LL(path), LL(2)); }
int smx_graph_write_fastg(smx_ctx *, const char *path) { return (int)fwd("smx_graph_write", 2, LL(path), LL(3)); }
int smx_graph_write_unitigs(smx_ctx *, const char *path) { return (int)fwd("smx_graph_write", 2, LL(path), LL(0)); }

}  // extern "C"
