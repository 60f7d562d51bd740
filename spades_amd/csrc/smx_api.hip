// spades_amd/csrc/smx_api.hip — C ABI (include/smx.h) + host-side pipeline driving the gfx950 kernels.
// Built by hipcc into libspades_mi355x.so; no torch / no reference headers involved.
#include "../../include/smx.h"
#include "smx_kernels.hip"
#include "smx_superkmer.hip"
#include "smx_ingest.hip"
#include "smx_graph.hip"
#include "smx_pm.hip"
#include "smx_loops.hip"
#include "smx_dwalk.hip"
#include "smx_gfa.hip"
#include "smx_graph_host.hpp"
#include "smx_loops_host.hpp"
#include "smx_file_sink.hpp"

#include <algorithm>
#include <execinfo.h>
#include <csignal>
#include <unistd.h>
#include <fcntl.h>
#include <atomic>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <ctime>
#include <cstdlib>
#include <map>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

using namespace smx;

#include "smx_ctx.hpp"
#include "smx_spill_split.hpp"
#include "smx_pipeline.hpp"
#include "smx_construct.hpp"
#include "smx_pm.hpp"
#include "smx_dwalk.hpp"

// ============================================================================ C ABI
extern "C" {

const char *smx_version(void) { return "spades-mi355x 0.1 (gfx950)"; }

// SMX_SEGV_TRACE=1: a native backtrace on SIGSEGV / SIGABRT (module + offset per frame; addr2line -e <module> <offset>), then the
// default action. Diagnostics only.
static void segv_trace(int sig) {
    const char msg[] = "[smx] fatal signal, native frames:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    void *frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

int smx_create(smx_ctx **out, int device, size_t hbm_budget_bytes) {
    static bool traced = false;
    if (!traced && getenv("SMX_SEGV_TRACE")) {
        traced = true;
        {
            void *warm[4];
            (void)backtrace(warm, 4);  // loads the unwinder now: no allocation inside the handler
        }
        static char altstack[1 << 16];
        stack_t ss{};
        ss.ss_sp = altstack;
        ss.ss_size = sizeof(altstack);
        (void)sigaltstack(&ss, nullptr);
        struct sigaction sa{};
        sa.sa_handler = segv_trace;
        sa.sa_flags = SA_ONSTACK | SA_NODEFER;
        sigemptyset(&sa.sa_mask);
        (void)sigaction(SIGSEGV, &sa, nullptr);
        (void)sigaction(SIGABRT, &sa, nullptr);
        (void)sigaction(SIGBUS, &sa, nullptr);
    }
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
    (void)smx_extract_release(ctx);
    (void)smx_exchange_release(ctx);
    clear_graph(ctx);
    clear_result(ctx);
    free_temps(ctx);
    for (void *p : ctx->pool_blocks) arena_put(ctx, p);  // (a caller's blocks it did not return: they end with the context)
    ctx->pool_blocks.clear();
    arena_release(ctx);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    delete ctx;
}

int smx_trim(smx_ctx *ctx, size_t *bytes_returned) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    HIPCHK(hipSetDevice(ctx->device));
    const size_t n = arena_trim(ctx);
    if (bytes_returned) *bytes_returned = n;
    return SMX_OK;
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
    else if (!strcmp(key, "skm_fold")) ctx->opt_skm_fold = value;
    else if (!strcmp(key, "two_strand")) ctx->opt_two_strand = value;
    else if (!strcmp(key, "two_strand_parts")) ctx->opt_two_strand_parts = value;
    else if (!strcmp(key, "device_gfa")) ctx->opt_device_gfa = value;
    else if (!strcmp(key, "keep_perfect_loops")) ctx->opt_keep_loops = value;
    else if (!strcmp(key, "device_loops")) ctx->opt_device_loops = value;
    else if (!strcmp(key, "nx_route")) ctx->opt_nx_route = value;
    else if (!strcmp(key, "skm_nkey_log2")) ctx->opt_skm_nkey_log2 = value;
    else if (!strcmp(key, "derive_batches")) ctx->opt_derive_batches = value;
    else if (!strcmp(key, "keep_kpo")) ctx->opt_keep_kpo = value;
    else if (!strcmp(key, "pm_overlap")) ctx->opt_pm_overlap = value;
    else if (!strcmp(key, "pm_full_retab")) ctx->opt_pm_full_retab = value;
    else if (!strcmp(key, "pm_fuse_tab")) ctx->opt_pm_fuse_tab = value;
    else if (!strcmp(key, "walk_pack")) ctx->opt_walk_pack = value;
    else if (!strcmp(key, "pm_remote_mirror")) ctx->opt_pm_remote_mirror = value;
    else if (!strcmp(key, "walk_chunk")) ctx->opt_walk_chunk = value;
    else if (!strcmp(key, "walk_start_chunk")) ctx->opt_walk_start_chunk = value;
    else if (!strcmp(key, "walk_hop_bits")) ctx->opt_walk_hop_bits = value;
    else if (!strcmp(key, "walk_fail_at")) ctx->opt_walk_fail_at = value;
    else if (!strcmp(key, "verify_lookups")) ctx->opt_verify_lookups = value;
    else if (!strcmp(key, "spill")) ctx->opt_spill = value;
    else if (!strcmp(key, "spill_merge_max")) ctx->opt_spill_merge_max = value;
    else if (!strcmp(key, "single_batch")) ctx->opt_single_batch = value;
    else if (!strcmp(key, "kmers_from_reads")) ctx->opt_kmers_from_reads = value;
    else if (!strcmp(key, "ext_route")) ctx->opt_ext_route = value;
    else if (!strcmp(key, "pm_route")) ctx->opt_pm_route = value;
    else if (!strcmp(key, "async_upload")) ctx->opt_async_upload = value;
    else if (!strcmp(key, "dir_slots")) ctx->opt_dir_slots = value;
    else if (!strcmp(key, "ext_presort")) ctx->opt_ext_presort = value;
    else return fail(ctx, SMX_INVALID_PARAMETER, "unknown option %s", key);
    return SMX_OK;
}

int smx_reads_clear(smx_ctx *ctx) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    (void)hipSetDevice(ctx->device);
    for (auto &c : ctx->chunks) {
        if (c.ev_meta) {  // an asynchronous submission: its copies must be over before the buffers go
            (void)hipEventSynchronize(c.piece_ev.empty() ? c.ev_meta : c.piece_ev.back());
            (void)hipEventDestroy(c.ev_meta);
            for (auto e : c.piece_ev) (void)hipEventDestroy(e);
            if (c.h_ext) (void)hipHostFree(c.h_ext);
            arena_put(ctx, c.d_ext);
        }
        if (c.owned) {
            arena_put(ctx, c.d_words);
            arena_put(ctx, c.d_start);
            arena_put(ctx, c.d_len);
        }
    }
    ctx->chunks.clear();
    return SMX_OK;
}

// smx_submit_reads_packed with option "async_upload": nothing is waited for. (start, len) go first (the window marks need only them),
// then the stream in pieces; every piece has an event, so that the first scan of the reads can follow the upload piece by piece
// (run_prededupe) instead of waiting for the last byte. The caller's arrays must stay valid until the reads have been used.
static int submit_packed_async(smx_ctx *ctx, const uint64_t *words, uint64_t n_words, const uint64_t *start, const uint32_t *len, uint64_t n_reads) {
    if (!ctx->copy_stream) HIPCHK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    ReadChunk c;
    c.n_words = n_words;
    c.n_reads = n_reads;
    c.n_bases = n_words * 32;  // until the check says how far the reads really go
    unsigned long long *d_ext = nullptr;
    auto bail = [&](int code) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        if (c.ev_meta) (void)hipEventDestroy(c.ev_meta);
        for (auto e : c.piece_ev) (void)hipEventDestroy(e);
        if (c.h_ext) (void)hipHostFree(c.h_ext);
        arena_put(ctx, c.d_words);
        arena_put(ctx, c.d_start);
        arena_put(ctx, c.d_len);
        arena_put(ctx, d_ext);
        return code;
    };
    if (int rc = dalloc(ctx, &c.d_words, n_words + 8, false)) return bail(rc);
    if (int rc = dalloc(ctx, &c.d_start, n_reads, false)) return bail(rc);
    if (int rc = dalloc(ctx, &c.d_len, n_reads, false)) return bail(rc);
    if (int rc = dalloc(ctx, &d_ext, 2, false)) return bail(rc);
    hipStream_t cs = ctx->copy_stream;
    hipError_t e = hipHostMalloc((void **)&c.h_ext, 16, hipHostMallocDefault);
    if (e == hipSuccess) {
        c.h_ext[0] = 0;
        c.h_ext[1] = 0;
        e = hipMemsetAsync(d_ext, 0, 16, cs);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(c.d_start, start, n_reads * 8, hipMemcpyHostToDevice, cs);
    if (e == hipSuccess) e = hipMemcpyAsync(c.d_len, len, n_reads * 4, hipMemcpyHostToDevice, cs);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_reads_extent, dim3((unsigned)std::min<uint64_t>((n_reads + BLK - 1) / BLK, 2048)), dim3(BLK), 0, cs, (const uint64_t *)c.d_start,
                           (const uint32_t *)c.d_len, n_reads, (uint64_t)n_words * 32, d_ext);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(c.h_ext, d_ext, 16, hipMemcpyDeviceToHost, cs);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c.ev_meta, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(c.ev_meta, cs);
    if (e == hipSuccess) e = hipMemsetAsync(c.d_words + n_words, 0, 64, cs);
    const uint64_t npieces = std::min<uint64_t>(16, std::max<uint64_t>(2, n_words / ((uint64_t)32 << 20)));  // pieces of >= 256 MB
    for (uint64_t p = 0; p < npieces && e == hipSuccess; ++p) {
        const uint64_t w0 = n_words * p / npieces / 2 * 2, w1 = p + 1 == npieces ? n_words : n_words * (p + 1) / npieces / 2 * 2;  // multiples of 64 positions
        if (w1 > w0) e = hipMemcpyAsync(c.d_words + w0, words + w0, (w1 - w0) * 8, hipMemcpyHostToDevice, cs);
        hipEvent_t ev = nullptr;
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess) {
            c.piece_ev.push_back(ev);
            c.piece_end.push_back(w1);
            e = hipEventRecord(ev, cs);
        }
    }
    if (e != hipSuccess) return bail(fail(ctx, SMX_DEVICE_ERROR, "read upload failed: %s", hipGetErrorString(e)));
    c.d_ext = d_ext;  // released with the chunk (smx_reads_clear, after its events): a free_temps() of some other call must not hand it out
    c.contigs = ctx->opt_submit_contigs != 0;
    ctx->chunks.push_back(c);
    return SMX_OK;
}

int smx_submit_reads_packed(smx_ctx *ctx, const uint64_t *words, uint64_t n_words, const uint64_t *start,
                            const uint32_t *len, uint64_t n_reads) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (n_reads == 0) return SMX_OK;
    if (!words || !start || !len) return fail(ctx, SMX_INVALID_PARAMETER, "null read arrays");
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->opt_async_upload > 0 && n_words >= ((uint64_t)1 << 24)) return submit_packed_async(ctx, words, n_words, start, len, n_reads);
    ReadChunk c;
    c.n_words = n_words;
    c.n_reads = n_reads;
    unsigned long long *d_ext;
    auto bail = [&](int code) {  // (every early return: the long-lived blocks taken so far go back)
        arena_put(ctx, c.d_words);
        arena_put(ctx, c.d_start);
        arena_put(ctx, c.d_len);
        free_temps(ctx);
        return code;
    };
    if (int rc = dalloc(ctx, &c.d_words, n_words + 8, false)) return bail(rc);
    if (int rc = dalloc(ctx, &c.d_start, n_reads, false)) return bail(rc);
    if (int rc = dalloc(ctx, &c.d_len, n_reads, false)) return bail(rc);
    if (int rc = dalloc(ctx, &d_ext, 2)) return bail(rc);
    hipError_t e = hipMemsetAsync(c.d_words + n_words, 0, 64, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_ext, 0, 16, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c.d_words, words, n_words * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c.d_start, start, n_reads * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c.d_len, len, n_reads * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) return bail(fail(ctx, SMX_DEVICE_ERROR, "read upload failed: %s", hipGetErrorString(e)));
    // the reference's streams cannot hand over a read that is not there; here the (start, len) pairs are checked against the stream
    hipLaunchKernelGGL(k_reads_extent, dim3((unsigned)std::min<uint64_t>((n_reads + BLK - 1) / BLK, 2048)), dim3(BLK), 0, ctx->stream,
                       (const uint64_t *)c.d_start, (const uint32_t *)c.d_len, n_reads, (uint64_t)n_words * 32, d_ext);
    unsigned long long ext[2] = {0, 0};
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(ext, d_ext, 16, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return bail(fail(ctx, SMX_DEVICE_ERROR, "read upload failed: %s", hipGetErrorString(e)));
    if (ext[1]) return bail(fail(ctx, SMX_INVALID_INPUT_FORMAT, "%llu reads exceed the packed stream", ext[1]));
    c.n_bases = ext[0];
    free_temps(ctx);
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
    auto bail = [&](int code) {
        arena_put(ctx, c.d_words);
        arena_put(ctx, c.d_start);
        arena_put(ctx, c.d_len);
        free_temps(ctx);
        return code;
    };
    if (int rc = dalloc(ctx, &d_bases, nbases + 1)) return bail(rc);
    if (int rc = dalloc(ctx, &d_off, n_reads + 1)) return bail(rc);
    if (int rc = dalloc(ctx, &c.d_words, c.n_words + 8, false)) return bail(rc);
    if (int rc = dalloc(ctx, &c.d_start, n_reads, false)) return bail(rc);
    if (int rc = dalloc(ctx, &c.d_len, n_reads, false)) return bail(rc);
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
    clear_result(ctx);  // the previous count's result goes first: the HBM plan of this one is made from what is free
    return dispatch_count(ctx, K, mode, num_buckets, nullptr, 0);
}

// KMerDiskCounter::Count + KMerDiskStorage::merge in one call with the destination known from the start (kmer_index_builder.hpp:306-332,190-203):
// a count that has to go out of core streams every merged bucket range to its place in the file instead of holding the merged result in host
// memory next to the spilled runs (the reference's merge writes as it goes, :346-430); a result that stays in HBM is written as
// smx_write_final_kmers writes it. Afterwards the figures of the count (smx_count_info, smx_bucket_sizes) are there; the records are in the file.
int smx_count_to_file(smx_ctx *ctx, unsigned K, int mode, unsigned num_buckets, const char *path) {
    if (!ctx || !path) return SMX_INVALID_PARAMETER;
    ctx->xnames.clear();
    ctx->xms.clear();
    clear_result(ctx);
    const int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return fail(ctx, SMX_IO_ERROR, "Cannot open %s for writing", path);
    ctx->sink_fd = fd;
    ctx->sink_path = path;
    int rc = dispatch_count(ctx, K, mode, num_buckets, nullptr, 0);
    ctx->sink_fd = -1;
    if (rc == 0 && !ctx->result_on_file) {  // the result is resident (one array, or two strands): written the usual way
        (void)close(fd);
        return smx_write_final_kmers(ctx, path);
    }
    if (rc == 0 && ftruncate(fd, (off_t)(ctx->n_records * (size_t)ctx->nw * 8)) != 0) rc = fail(ctx, SMX_IO_ERROR, "I/O error writing %s", path);
    if (close(fd) != 0 && rc == 0) rc = fail(ctx, SMX_IO_ERROR, "I/O error closing %s", path);
    if (rc) {
        (void)unlink(path);
        ctx->result_on_file = false;
    }
    return rc;
}
static int on_file(smx_ctx *ctx) {
    return fail(ctx, SMX_INVALID_PARAMETER, "the records of this count were streamed to %s (smx_count_to_file): they are not held by the context", ctx->sink_path.c_str());
}

int smx_count_records(smx_ctx *ctx, unsigned K, unsigned num_buckets, const void *d_records, uint64_t n_records) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (n_records && !d_records) return fail(ctx, SMX_INVALID_PARAMETER, "null records");
    static const uint64_t dummy = 0;
    // records in the library's own exchange buffer are consumed: the buffer serves as one of the sort's ping-pong buffers
    const bool own = ctx->x_recv && d_records == ctx->x_recv;
    const int rc = dispatch_count(ctx, K, SMX_MODE_ALL, num_buckets, n_records ? d_records : (const void *)&dummy, n_records, own);
    if (own) {
        if (ctx->d_result_buf != ctx->x_recv) arena_put(ctx, ctx->x_recv);
        ctx->x_recv = nullptr;
    }
    return rc;
}

int smx_exchange_buffer(smx_ctx *ctx, uint64_t n_words, void **d_buf) {
    if (!ctx || !d_buf) return SMX_INVALID_PARAMETER;
    HIPCHK(hipSetDevice(ctx->device));
    arena_put(ctx, ctx->x_recv);
    ctx->x_recv = nullptr;
    *d_buf = nullptr;
    void *p = arena_get(ctx, std::max<uint64_t>(n_words, 1) * 8, false);
    if (!p) return fail(ctx, SMX_MEMORY_LIMIT_EXCEEDED, "no room for an exchange buffer of %llu words", (unsigned long long)n_words);
    *d_buf = ctx->x_recv = p;
    return SMX_OK;
}

int smx_count_info(const smx_ctx *ctx, uint64_t *n_records, unsigned *words_per_record, uint64_t *n_kmer_instances) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (n_records) *n_records = ctx->n_records;
    if (words_per_record) *words_per_record = ctx->nw;
    if (n_kmer_instances) *n_kmer_instances = ctx->n_instances;
    return SMX_OK;
}

// Device memory -> file: a ring of page-locked buffers; while the calling thread hands one chunk to the file, the copies of the next
// ones run. (On tmpfs — where the tools' outputs are measured — 8 pwrite threads on one file reached 3.4 GB/s, a single thread 6.4:
// pwrite allocates pages under the inode lock. smx_file_sink.hpp therefore maps a tmpfs output and fills it with several threads;
// everything else stays on one pwrite thread.) Used by the k-mer file and the GFA text writers.
struct DevToFile {
    static constexpr int NBUF = 4;
    size_t chunk = (size_t)64 << 20;
    char *buf[NBUF] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[NBUF] = {nullptr, nullptr, nullptr, nullptr};
    bool wok = true;
    int nbuf = 0;
    size_t seq = 0;  // chunks issued so far (a buffer per chunk, round robin)
    bool init(size_t total) {
        chunk = std::min(chunk, std::max<size_t>(total, 1));
        for (; nbuf < NBUF && (size_t)nbuf * chunk < total + chunk; ++nbuf)
            if (hipHostMalloc((void **)&buf[nbuf], chunk, hipHostMallocDefault) != hipSuccess || hipEventCreate(&ev[nbuf]) != hipSuccess) {
                (void)hipGetLastError();
                if (buf[nbuf]) (void)hipHostFree(buf[nbuf]);
                buf[nbuf] = nullptr;
                break;
            }
        return nbuf > 0;
    }
    // bytes [0, n) of device block src -> file offset off; returns false on a device error (I/O errors: wok)
    bool send(hipStream_t st, smxio::FileSink &sink, off_t off, const char *src, size_t n) {
        const size_t nchunks = (n + chunk - 1) / chunk;
        size_t issued = 0;
        std::vector<int> bof(nchunks);
        auto issue = [&](size_t c) {  // (its buffer was written out by this thread before: free)
            const int b = (int)(seq++ % (size_t)nbuf);
            bof[c] = b;
            const size_t o = c * chunk, m = std::min(chunk, n - o);
            return hipMemcpyAsync(buf[b], src + o, m, hipMemcpyDeviceToHost, st) == hipSuccess && hipEventRecord(ev[b], st) == hipSuccess;
        };
        for (size_t c = 0; c < nchunks; ++c) {
            while (issued < nchunks && issued < c + (size_t)nbuf)
                if (!issue(issued++)) return false;
            const int b = bof[c];
            if (hipEventSynchronize(ev[b]) != hipSuccess) return false;
            const size_t o = c * chunk, m = std::min(chunk, n - o);
            if (wok && !sink.put(buf[b], m, off + (off_t)o)) wok = false;
        }
        return true;
    }
    bool finish() {
        for (int i = 0; i < NBUF; ++i) {
            if (ev[i]) (void)hipEventDestroy(ev[i]);
            if (buf[i]) (void)hipHostFree(buf[i]);
            ev[i] = nullptr;
            buf[i] = nullptr;
        }
        return wok;
    }
};

// Two-strand result (smx_ctx::TwoStrand): bucket b merged into a device block of its own (caller releases it with arena_put)
static int ts_bucket_block(smx_ctx *ctx, unsigned b, void **blk) {
    const uint64_t n = ctx->bucket_off[b + 1] - ctx->bucket_off[b];
    uint64_t *p = nullptr;
    if (int rc = dalloc(ctx, &p, std::max<uint64_t>(n * ctx->nw, 1), false)) return rc;
    int rc = ts_merge_bucket_any(ctx, b, p);
    if (rc == 0 && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, SMX_DEVICE_ERROR, "two-strand merge failed");
    if (rc) {
        arena_put(ctx, p);
        return rc;
    }
    *blk = p;
    return SMX_OK;
}

int smx_bucket_sizes(const smx_ctx *ctx, uint64_t *sizes) {
    if (!ctx || !sizes) return SMX_INVALID_PARAMETER;
    if (int rc = ensure_kmer_file(const_cast<smx_ctx *>(ctx))) return rc;  // (a graph built without a sorted k-mer file: made now)
    for (unsigned b = 0; b < ctx->num_buckets; ++b) sizes[b] = ctx->bucket_off[b + 1] - ctx->bucket_off[b];
    return SMX_OK;
}

int smx_copy_bucket(const smx_ctx *cctx, unsigned bucket, void *host_dst) {
    smx_ctx *ctx = const_cast<smx_ctx *>(cctx);
    if (!ctx || !host_dst) return SMX_INVALID_PARAMETER;
    if (int rc = ensure_kmer_file(ctx)) return rc;
    if (bucket >= ctx->num_buckets) return fail(ctx, SMX_INVALID_PARAMETER, "bucket %u out of range", bucket);
    const uint64_t o = ctx->bucket_off[bucket], n = ctx->bucket_off[bucket + 1] - o;
    if (n == 0) return SMX_OK;
    const size_t w = (size_t)ctx->nw * 8;
    if (ctx->result_on_file) return on_file(ctx);
    if (ctx->result_on_host) {  // spilled result: chunks of whole buckets in file order
        uint64_t base = 0;
        for (auto &c : ctx->h_result) {
            if (o >= base && o + n <= base + c.n) {
                memcpy(host_dst, c.data + (o - base) * w, n * w);
                return SMX_OK;
            }
            base += c.n;
        }
        return fail(ctx, SMX_DEVICE_ERROR, "bucket %u is not inside one chunk of the spilled result", bucket);
    }
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->ts.active) {  // the bucket is the merge of its two strands
        void *blk = nullptr;
        if (int rc = ts_bucket_block(ctx, bucket, &blk)) return rc;
        const hipError_t e = hipMemcpy(host_dst, blk, n * w, hipMemcpyDeviceToHost);
        arena_put(ctx, blk);
        if (e != hipSuccess) return fail(ctx, SMX_DEVICE_ERROR, "device read-back failed: %s", hipGetErrorString(e));
        return SMX_OK;
    }
    HIPCHK(hipMemcpy(host_dst, (const char *)ctx->d_result + o * w, n * w, hipMemcpyDeviceToHost));
    return SMX_OK;
}

int smx_copy_final_kmers(const smx_ctx *cctx, void *host_dst) {
    smx_ctx *ctx = const_cast<smx_ctx *>(cctx);
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (ctx->n_records == 0) return SMX_OK;
    if (!host_dst) return SMX_INVALID_PARAMETER;
    if (int rc = ensure_kmer_file(ctx)) return rc;
    if (ctx->result_on_file) return on_file(ctx);
    if (ctx->result_on_host) {
        char *dst = (char *)host_dst;
        for (auto &c : ctx->h_result) {
            memcpy(dst, c.data, c.n * ctx->nw * 8);
            dst += c.n * ctx->nw * 8;
        }
        return SMX_OK;
    }
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->ts.active) {
        for (unsigned b = 0; b < ctx->num_buckets; ++b) {
            const uint64_t o = ctx->bucket_off[b], n = ctx->bucket_off[b + 1] - o;
            if (!n) continue;
            void *blk = nullptr;
            if (int rc = ts_bucket_block(ctx, b, &blk)) return rc;
            const hipError_t e = hipMemcpy((char *)host_dst + o * ctx->nw * 8, blk, n * ctx->nw * 8, hipMemcpyDeviceToHost);
            arena_put(ctx, blk);
            if (e != hipSuccess) return fail(ctx, SMX_DEVICE_ERROR, "device read-back failed: %s", hipGetErrorString(e));
        }
        return SMX_OK;
    }
    HIPCHK(hipMemcpy(host_dst, ctx->d_result, ctx->n_records * ctx->nw * 8, hipMemcpyDeviceToHost));
    return SMX_OK;
}

int smx_write_final_kmers(const smx_ctx *cctx, const char *path) {
    smx_ctx *ctx = const_cast<smx_ctx *>(cctx);
    if (!ctx || !path) return SMX_INVALID_PARAMETER;
    if (int rc = ensure_kmer_file(ctx)) return rc;
    if (ctx->result_on_file) return ctx->sink_path == path ? SMX_OK : on_file(ctx);  // (that very file holds them already)
    FILE *f = fopen(path, "w+b");  // (read access too: a tmpfs output is filled through a shared mapping, smx_file_sink.hpp)
    if (!f) return fail(ctx, SMX_IO_ERROR, "Cannot open %s for writing", path);
    if (ctx->result_on_host) {
        int hrc = SMX_OK;
        for (auto &c : ctx->h_result) {
            const size_t nb_ = c.n * (size_t)ctx->nw * 8;
            if (nb_ && fwrite(c.data, 1, nb_, f) != nb_) hrc = fail(ctx, SMX_IO_ERROR, "I/O error! Incomplete write to %s", path);
        }
        if (fclose(f) != 0 && hrc == SMX_OK) hrc = fail(ctx, SMX_IO_ERROR, "I/O error closing %s", path);
        return hrc;
    }
    const size_t total = ctx->n_records * (size_t)ctx->nw * 8;
    (void)hipSetDevice(ctx->device);
    if (fflush(f) != 0) {
        fclose(f);
        return fail(ctx, SMX_IO_ERROR, "I/O error writing %s", path);
    }
    const int fd = fileno(f);
    int rc = SMX_OK;
    DevToFile d2f;
    smxio::FileSink sink;
    if (total && !d2f.init(total)) rc = fail(ctx, SMX_MEMORY_LIMIT_EXCEEDED, "no page-locked memory for the read-back");
    if (rc == SMX_OK && total) sink.begin(fd, total);
    if (rc == SMX_OK && total) {
        if (ctx->ts.active) {  // two strands: bucket after bucket, each merged on the device first (kmer_index_builder.hpp:190-203 concatenates buckets too)
            for (unsigned b = 0; b < ctx->num_buckets && rc == SMX_OK; ++b) {
                const uint64_t n = ctx->bucket_off[b + 1] - ctx->bucket_off[b];
                if (!n) continue;
                void *blk = nullptr;
                rc = ts_bucket_block(ctx, b, &blk);
                if (rc == SMX_OK) {
                    if (!d2f.send(ctx->stream, sink, (off_t)(ctx->bucket_off[b] * (size_t)ctx->nw * 8), (const char *)blk, n * (size_t)ctx->nw * 8))
                        rc = fail(ctx, SMX_DEVICE_ERROR, "device read-back failed");
                    (void)hipStreamSynchronize(ctx->stream);  // (the block goes back to the arena: its copies must be over)
                    arena_put(ctx, blk);
                }
            }
        } else if (!d2f.send(ctx->stream, sink, 0, (const char *)ctx->d_result, total)) {
            rc = fail(ctx, SMX_DEVICE_ERROR, "device read-back failed");
        }
    }
    (void)hipStreamSynchronize(ctx->stream);
    const bool sunk = sink.end();
    if ((!d2f.finish() || !sunk) && rc == SMX_OK) rc = fail(ctx, SMX_IO_ERROR, "I/O error! Incomplete write to %s", path);
    if (fclose(f) != 0 && rc == SMX_OK) rc = fail(ctx, SMX_IO_ERROR, "I/O error closing %s", path);
    return rc;
}

const void *smx_device_kmers(const smx_ctx *ctx) {
    if (!ctx || ensure_kmer_file(const_cast<smx_ctx *>(ctx))) return nullptr;
    return ctx->d_result;  // (NULL for a result that is not one resident array: spilled to the host, or held as two strands)
}

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


static int extract_partition_impl(smx_ctx *ctx, unsigned K, int mode, unsigned num_buckets, unsigned world, void *d_records,
                                  uint64_t capacity_records, uint64_t *counts, void **owned, unsigned min_len = 0) {
    if (!ctx || !counts) return SMX_INVALID_PARAMETER;
    if (K < 1 || K > 128) return fail(ctx, SMX_INVALID_PARAMETER, "K=%u out of range [1,128]", K);
    if (world < 1 || world > 4096 || num_buckets < 1) return fail(ctx, SMX_INVALID_PARAMETER, "bad world/num_buckets");
    HIPCHK(hipSetDevice(ctx->device));
    arena_put(ctx, ctx->x_owned);
    ctx->x_owned = nullptr;
    if (owned) {  // the library plans the HBM of this step itself: the previous step's result and graph make room
        clear_graph(ctx);
        clear_result(ctx);
    }
    int rc;
    switch ((K + 31) / 32) {
        case 1: rc = run_extract_partition<1>(ctx, K, mode, num_buckets, world, d_records, capacity_records, counts, owned, min_len); break;
        case 2: rc = run_extract_partition<2>(ctx, K, mode, num_buckets, world, d_records, capacity_records, counts, owned, min_len); break;
        case 3: rc = run_extract_partition<3>(ctx, K, mode, num_buckets, world, d_records, capacity_records, counts, owned, min_len); break;
        default: rc = run_extract_partition<4>(ctx, K, mode, num_buckets, world, d_records, capacity_records, counts, owned, min_len); break;
    }
    (void)hipStreamSynchronize(ctx->stream);
    if (rc == 0) {
        tcollect(ctx);
        ctx->xnames = ctx->tnames;
        ctx->xms = ctx->tms;
        ctx->tnames.clear();
        ctx->tms.clear();
        if (owned) ctx->x_owned = *owned;
    } else {
        for (auto &t : ctx->timings) {
            (void)hipEventDestroy(t.e0);
            (void)hipEventDestroy(t.e1);
        }
        ctx->timings.clear();
        if (owned && *owned) {
            arena_put(ctx, *owned);
            *owned = nullptr;
        }
    }
    free_temps(ctx);
    return rc;
}

int smx_extract_partition(smx_ctx *ctx, unsigned K, int mode, unsigned num_buckets, unsigned world, void *d_records,
                          uint64_t capacity_records, uint64_t *counts) {
    return extract_partition_impl(ctx, K, mode, num_buckets, world, d_records, capacity_records, counts, nullptr);
}

int smx_extract_partition_owned(smx_ctx *ctx, unsigned K, int mode, unsigned num_buckets, unsigned world, const void **d_records,
                                uint64_t *counts) {
    if (!d_records) return SMX_INVALID_PARAMETER;
    void *p = nullptr;
    const int rc = extract_partition_impl(ctx, K, mode, num_buckets, world, nullptr, 0, counts, &p);
    *d_records = p;
    return rc;
}

int smx_kmers_with_masks_supported(unsigned k) { return (k >= 21 && k < 128 && (k & 1) && ext_layout_fits(k, (int)((k + 31) / 32))) ? 1 : 0; }

int smx_extract_kmers_ext_owned(smx_ctx *ctx, unsigned k, unsigned num_buckets, unsigned world, const void **d_records, uint64_t *counts) {
    if (!ctx || !d_records) return SMX_INVALID_PARAMETER;
    if (!smx_kmers_with_masks_supported(k)) return fail(ctx, SMX_INVALID_PARAMETER, "k=%u: no room for an extension byte in the k-mer record (or k < 21)", k);
    void *p = nullptr;
    ctx->ext_mode = true;
    const int rc = extract_partition_impl(ctx, k, SMX_MODE_CANONICAL, num_buckets, world, nullptr, 0, counts, &p, k + 1);
    ctx->ext_mode = false;
    *d_records = p;
    return rc;
}

int smx_extract_release(smx_ctx *ctx) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    arena_put(ctx, ctx->x_owned);
    ctx->x_owned = nullptr;
    return SMX_OK;
}

int smx_exchange_release(smx_ctx *ctx) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    arena_put(ctx, ctx->x_recv);
    ctx->x_recv = nullptr;
    return SMX_OK;
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

int smx_prewarm(smx_ctx *ctx, size_t bottom_bytes, size_t top_bytes) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    arena_prewarm_start(ctx, bottom_bytes, top_bytes);
    return SMX_OK;
}

int smx_arena_free_bytes(smx_ctx *ctx, size_t *bytes) {
    if (!ctx || !bytes) return SMX_INVALID_PARAMETER;
    std::lock_guard<std::recursive_mutex> lk(ctx->arena_mu);
    size_t cached = 0;
    if (ctx->arena.vmm) {
        for (auto &b : ctx->arena.free_blocks) cached += b.second;
    } else {
        for (auto &b : ctx->arena_free) cached += b.second;
    }
    *bytes = cached;
    return SMX_OK;
}

int smx_pool_alloc(smx_ctx *ctx, size_t bytes, void **d_block) {
    if (!ctx || !d_block) return SMX_INVALID_PARAMETER;
    *d_block = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    void *p = arena_get(ctx, std::max<size_t>(bytes, 256), /*top=*/true);
    if (!p) return fail(ctx, SMX_MEMORY_LIMIT_EXCEEDED, "no room for a caller's block of %zu bytes in the device arena (%.1f GB obtainable)", bytes, (double)arena_avail(ctx) / 1e9);
    ctx->pool_blocks.insert(p);
    *d_block = p;
    return SMX_OK;
}
int smx_pool_free(smx_ctx *ctx, void *d_block) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (!d_block) return SMX_OK;
    auto it = ctx->pool_blocks.find(d_block);
    if (it == ctx->pool_blocks.end()) return fail(ctx, SMX_INVALID_PARAMETER, "smx_pool_free: not a block smx_pool_alloc handed out");
    // the block was the CALLER's to write, on streams the library knows nothing about (torch's, RCCL's): all of the device's work ends before
    // the arena may hand the memory to somebody else (a stream-only wait made a free with work in flight a silent overwrite)
    (void)hipDeviceSynchronize();
    ctx->pool_blocks.erase(it);
    arena_put(ctx, d_block);
    return SMX_OK;
}

int smx_graph_clear(smx_ctx *ctx) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->pm_view_pending || (ctx->d_result && ctx->d_result == ctx->g_kmers)) {  // the count-result view showed (or stood for) the graph's k-mer file
        ctx->pm_view_pending = false;
        ctx->d_result = nullptr;
        ctx->n_records = 0;
        ctx->bucket_off.assign(ctx->num_buckets + 1, 0);
    }
    clear_graph(ctx);
    return SMX_OK;
}

int smx_build_graph(smx_ctx *ctx, unsigned k, unsigned num_buckets) { return build_graph_impl(ctx, k, num_buckets, nullptr, 0); }

int smx_build_graph_from_records(smx_ctx *ctx, unsigned k, unsigned num_buckets, const void *d_kpomers, uint64_t n_records) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (n_records && !d_kpomers) return fail(ctx, SMX_INVALID_PARAMETER, "null records");
    static const uint64_t dummy = 0;
    return build_graph_impl(ctx, k, num_buckets, n_records ? d_kpomers : (const void *)&dummy, n_records);
}

static int finish_call(smx_ctx *ctx, int rc, bool graph) {
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
    if (rc && graph) clear_graph(ctx);
    return rc;
}

int smx_graph_shard_updates(smx_ctx *ctx, unsigned k, unsigned num_buckets, unsigned world, void *d_updates, uint64_t capacity_records,
                            uint64_t *counts) {
    if (!ctx || !counts) return SMX_INVALID_PARAMETER;
    if (k < 1 || k >= 128 || k % 2 == 0) return fail(ctx, SMX_INVALID_PARAMETER, "k-mer size must be odd and below 128");
    if (world < 1 || world > 1024 || num_buckets < 1) return fail(ctx, SMX_INVALID_PARAMETER, "bad world/num_buckets");
    if (ctx->n_records && (ctx->K != k + 1 || ctx->num_buckets != num_buckets || !ctx->d_result))
        return fail(ctx, SMX_INVALID_PARAMETER, "the context does not hold a count of canonical %u-mers in %u buckets", k + 1, num_buckets);
    if (ctx->n_records && !d_updates) return fail(ctx, SMX_INVALID_PARAMETER, "null update buffer");
    HIPCHK(hipSetDevice(ctx->device));
    int rc;
    switch ((k + 32) / 32) {
        case 1: rc = shard_updates<1>(ctx, k, num_buckets, world, d_updates, capacity_records, counts); break;
        case 2: rc = shard_updates<2>(ctx, k, num_buckets, world, d_updates, capacity_records, counts); break;
        case 3: rc = shard_updates<3>(ctx, k, num_buckets, world, d_updates, capacity_records, counts); break;
        default: rc = shard_updates<4>(ctx, k, num_buckets, world, d_updates, capacity_records, counts); break;
    }
    (void)hipStreamSynchronize(ctx->stream);
    free_temps(ctx);
    return rc;
}

int smx_graph_shard_build(smx_ctx *ctx, unsigned k, unsigned num_buckets, unsigned world, unsigned rank, const void *d_updates, uint64_t n_updates) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (k < 1 || k >= 128 || k % 2 == 0) return fail(ctx, SMX_INVALID_PARAMETER, "k-mer size must be odd and below 128");
    if (world < 1 || rank >= world || num_buckets < 1) return fail(ctx, SMX_INVALID_PARAMETER, "bad world/rank/num_buckets");
    if (n_updates && !d_updates) return fail(ctx, SMX_INVALID_PARAMETER, "null updates");
    HIPCHK(hipSetDevice(ctx->device));
    ctx->xnames.clear();
    ctx->xms.clear();
    int rc;
    switch ((k + 32) / 32) {
        case 1: rc = shard_build<1>(ctx, k, num_buckets, world, rank, d_updates, n_updates); break;
        case 2: rc = shard_build<2>(ctx, k, num_buckets, world, rank, d_updates, n_updates); break;
        case 3: rc = shard_build<3>(ctx, k, num_buckets, world, rank, d_updates, n_updates); break;
        default: rc = shard_build<4>(ctx, k, num_buckets, world, rank, d_updates, n_updates); break;
    }
    return finish_call(ctx, rc, true);
}

int smx_graph_shard_from_ext(smx_ctx *ctx, unsigned k, unsigned num_buckets, unsigned world, unsigned rank, const void *d_records, uint64_t n_records) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (k < 1 || k >= 128 || k % 2 == 0) return fail(ctx, SMX_INVALID_PARAMETER, "k-mer size must be odd and below 128");
    if (world < 1 || rank >= world || num_buckets < 1) return fail(ctx, SMX_INVALID_PARAMETER, "bad world/rank/num_buckets");
    if (n_records && !d_records) return fail(ctx, SMX_INVALID_PARAMETER, "null records");
    HIPCHK(hipSetDevice(ctx->device));
    ctx->xnames.clear();
    ctx->xms.clear();
    // records in the library's own exchange buffer are consumed: the buffer becomes one of the sort's temporaries
    const bool own = ctx->x_recv && d_records == ctx->x_recv;
    if (own) {
        ctx->temps.push_back(ctx->x_recv);
        ctx->x_recv = nullptr;
    }
    int rc;
    switch ((k + 31) / 32) {
        case 1: rc = shard_from_ext<1>(ctx, k, num_buckets, world, rank, d_records, n_records, own); break;
        case 2: rc = shard_from_ext<2>(ctx, k, num_buckets, world, rank, d_records, n_records, own); break;
        case 3: rc = shard_from_ext<3>(ctx, k, num_buckets, world, rank, d_records, n_records, own); break;
        default: rc = shard_from_ext<4>(ctx, k, num_buckets, world, rank, d_records, n_records, own); break;
    }
    return finish_call(ctx, rc, true);
}

int smx_graph_fingerprint(const smx_ctx *cctx, uint64_t *out) {
    smx_ctx *ctx = const_cast<smx_ctx *>(cctx);
    if (!ctx || !out) return SMX_INVALID_PARAMETER;
    for (int i = 0; i < 16; ++i) out[i] = 0;
    if (!ctx->g_ready) return fail(ctx, SMX_INVALID_PARAMETER, "no graph");
    if (ctx->g_nkmers == 0) return SMX_OK;
    if (!ctx->g_dev_valid) return fail(ctx, SMX_INVALID_PARAMETER, "the graph is not resident on the device");
    if (ctx->g_sharded_file) return fail(ctx, SMX_INVALID_PARAMETER, "this graph was built from gathered unitigs: the k-mer file is sharded over the ranks; compare builds with smx_graph_fingerprint_portable");
    if (ctx->g_pm) return fail(ctx, SMX_INVALID_PARAMETER, "this graph was built without a sorted k-mer file (partition-major route): its k-mer and node arrays "
                                                           "are numbered differently; compare builds with smx_graph_fingerprint_portable");
    HIPCHK(hipSetDevice(ctx->device));
    unsigned long long *d;
    if (int rc = dalloc(ctx, &d, 16)) return rc;
    HIPCHK(hipMemsetAsync(d, 0, 128, ctx->stream));
    auto run64 = [&](const void *p, uint64_t n, int slot) {
        if (p && n) hipLaunchKernelGGL((k_fingerprint<unsigned long long>), dim3(grid_for(n)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)p, n, d + 2 * slot);
    };
    run64(ctx->g_kmers, ctx->g_nkmers * ctx->g_nw, 0);
    hipLaunchKernelGGL((k_fingerprint<uint8_t>), dim3(grid_for(ctx->g_nkmers)), dim3(BLK), 0, ctx->stream, (const uint8_t *)ctx->g_mask, ctx->g_nkmers, d + 2);
    run64(ctx->g_uwords, ctx->g_nuwords, 2);
    run64(ctx->g_elen, ctx->g_ne, 3);
    run64(ctx->g_estart, ctx->g_ne, 4);
    run64(ctx->g_eend, ctx->g_ne, 5);
    if (ctx->g_links_dev) {
        run64(ctx->g_lrecs, ctx->g_nlrec * 2, 6);
        run64(ctx->g_vstart, ctx->g_nv, 7);
    }
    HIPCHK(hipGetLastError());
    unsigned long long h[16];
    HIPCHK(hipMemcpyAsync(h, d, 128, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    free_temps(ctx);
    for (int i = 0; i < 16; ++i) out[i] = h[i];
    return SMX_OK;
}

int smx_graph_fingerprint_portable(const smx_ctx *cctx, uint64_t *out) {
    smx_ctx *ctx = const_cast<smx_ctx *>(cctx);
    if (!ctx || !out) return SMX_INVALID_PARAMETER;
    for (int i = 0; i < 8; ++i) out[i] = 0;
    if (!ctx->g_ready) return fail(ctx, SMX_INVALID_PARAMETER, "no graph");
    if (ctx->g_nkmers == 0) return SMX_OK;
    if (!ctx->g_dev_valid || !ctx->g_links_dev) return fail(ctx, SMX_INVALID_PARAMETER, "the graph (with its link records) is not resident on the device");
    HIPCHK(hipSetDevice(ctx->device));
    unsigned long long *d;
    if (int rc = dalloc(ctx, &d, 8)) return rc;
    HIPCHK(hipMemsetAsync(d, 0, 64, ctx->stream));
    if (ctx->g_nuwords) hipLaunchKernelGGL((k_fingerprint<unsigned long long>), dim3(grid_for(ctx->g_nuwords)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)ctx->g_uwords, ctx->g_nuwords, d);
    if (ctx->g_ne) {
        hipLaunchKernelGGL((k_fingerprint<unsigned long long>), dim3(grid_for(ctx->g_ne)), dim3(BLK), 0, ctx->stream, (const unsigned long long *)ctx->g_elen, ctx->g_ne, d + 2);
        hipLaunchKernelGGL((k_fingerprint<uint8_t>), dim3(grid_for(ctx->g_ne)), dim3(BLK), 0, ctx->stream, (const uint8_t *)ctx->g_eself, ctx->g_ne, d + 4);
    }
    if (ctx->g_nv)
        hipLaunchKernelGGL(k_pm_link_fingerprint, dim3(grid_for(ctx->g_nv)), dim3(BLK), 0, ctx->stream, (const Rec<2> *)ctx->g_lrecs, ctx->g_nlrec,
                           (const unsigned long long *)ctx->g_vstart, ctx->g_nv, d + 6);
    HIPCHK(hipGetLastError());
    unsigned long long h[8];
    HIPCHK(hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    free_temps(ctx);
    for (int i = 0; i < 8; ++i) out[i] = h[i];
    return SMX_OK;
}

int smx_graph_shard_ext_stats(const smx_ctx *ctx, uint64_t *stats) {
    if (!ctx || !stats) return SMX_INVALID_PARAMETER;
    stats[0] = ctx->g_ext_bits;
    stats[1] = ctx->g_ext_pals;
    return SMX_OK;
}

int smx_graph_shard_info(const smx_ctx *ctx, uint64_t *n_kmers, uint64_t *bucket_sizes) {
    if (!ctx || !n_kmers) return SMX_INVALID_PARAMETER;
    *n_kmers = ctx->g_nkmers;
    if (bucket_sizes)
        for (unsigned b = 0; b < ctx->g_B; ++b) bucket_sizes[b] = ctx->g_kboff.size() > b + 1 ? ctx->g_kboff[b + 1] - ctx->g_kboff[b] : 0;
    return SMX_OK;
}

int smx_graph_shard_copy(const smx_ctx *cctx, void *d_kmers, void *d_masks) {
    smx_ctx *ctx = const_cast<smx_ctx *>(cctx);
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (ctx->g_nkmers == 0) return SMX_OK;
    if (!d_kmers || !d_masks) return SMX_INVALID_PARAMETER;
    if (ctx->g_sharded_file) return fail(ctx, SMX_INVALID_PARAMETER, "this graph was built from gathered unitigs: the k-mer file is sharded over the ranks");
    if (int rc = ensure_kmer_file(ctx, /*view=*/false)) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(d_kmers, ctx->g_kmers, ctx->g_nkmers * ctx->g_nw * 8, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_masks, ctx->g_mask, ctx->g_nkmers, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SMX_OK;
}

int smx_build_graph_from_kmers(smx_ctx *ctx, unsigned k, unsigned num_buckets, const void *d_kmers, const void *d_masks, uint64_t n_kmers,
                               const uint64_t *bucket_sizes, uint64_t n_kpomers) {
    if (!ctx || !bucket_sizes) return SMX_INVALID_PARAMETER;
    if (k < 1 || k >= 128 || k % 2 == 0) return fail(ctx, SMX_INVALID_PARAMETER, "k-mer size must be odd and below 128");
    if (n_kmers && (!d_kmers || !d_masks)) return fail(ctx, SMX_INVALID_PARAMETER, "null k-mer file");
    HIPCHK(hipSetDevice(ctx->device));
    ctx->xnames.clear();
    ctx->xms.clear();
    int rc;
    switch ((k + 32) / 32) {
        case 1: rc = run_graph_from_kmers<1>(ctx, k, num_buckets, d_kmers, d_masks, n_kmers, bucket_sizes); break;
        case 2: rc = run_graph_from_kmers<2>(ctx, k, num_buckets, d_kmers, d_masks, n_kmers, bucket_sizes); break;
        case 3: rc = run_graph_from_kmers<3>(ctx, k, num_buckets, d_kmers, d_masks, n_kmers, bucket_sizes); break;
        default: rc = run_graph_from_kmers<4>(ctx, k, num_buckets, d_kmers, d_masks, n_kmers, bucket_sizes); break;
    }
    rc = finish_call(ctx, rc, true);
    if (rc == 0) ctx->g_nkpo = n_kpomers;
    return rc;
}

// ---- distributed walks (smx_dwalk.hpp) -----------------------------------------------------------------------
static int dw_check_shard(smx_ctx *ctx) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (ctx->g_nw < 1 || ctx->g_nw > 4 || ctx->g_k == 0)
        return fail(ctx, SMX_INVALID_PARAMETER, "no shard of the k-mer file in this context (smx_graph_shard_from_ext / smx_graph_shard_build first)");
    return SMX_OK;
}

int smx_shard_walk_counts(smx_ctx *ctx, uint64_t *n_chain_requests, uint64_t *n_start_requests) {
    if (int rc = dw_check_shard(ctx)) return rc;
    if (!n_chain_requests || !n_start_requests) return SMX_INVALID_PARAMETER;
    HIPCHK(hipSetDevice(ctx->device));
    int rc;
    switch (ctx->g_nw) {
        case 1: rc = dw_prepare<1>(ctx); break;
        case 2: rc = dw_prepare<2>(ctx); break;
        case 3: rc = dw_prepare<3>(ctx); break;
        default: rc = dw_prepare<4>(ctx); break;
    }
    rc = finish_call(ctx, rc, false);
    *n_chain_requests = ctx->dw_nchain;
    *n_start_requests = ctx->dw_ncand;
    return rc;
}

int smx_shard_walks(smx_ctx *ctx, const uint64_t *kmers_per_rank, const smx_collectives *coll, uint64_t *info) {
    if (int rc = dw_check_shard(ctx)) return rc;
    if (!kmers_per_rank || !coll || !info) return fail(ctx, SMX_INVALID_PARAMETER, "smx_shard_walks: null argument");
    HIPCHK(hipSetDevice(ctx->device));
    info[0] = info[1] = info[2] = info[3] = 0;
    int rc;
    switch (ctx->g_nw) {
        case 1: rc = dw_walks<1>(ctx, kmers_per_rank, coll, info); break;
        case 2: rc = dw_walks<2>(ctx, kmers_per_rank, coll, info); break;
        case 3: rc = dw_walks<3>(ctx, kmers_per_rank, coll, info); break;
        default: rc = dw_walks<4>(ctx, kmers_per_rank, coll, info); break;
    }
    return finish_call(ctx, rc, false);
}

int smx_shard_walk_loops(const smx_ctx *cctx, uint64_t *d_local_ranks) {
    smx_ctx *ctx = const_cast<smx_ctx *>(cctx);
    if (!ctx || !ctx->dw_ready) return SMX_INVALID_PARAMETER;
    if (ctx->dw_nloops == 0) return SMX_OK;
    if (!d_local_ranks || !ctx->dw_loops) return SMX_INVALID_PARAMETER;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(d_local_ranks, ctx->dw_loops, ctx->dw_nloops * 8, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SMX_OK;
}

int smx_shard_walk_requests_range(smx_ctx *ctx, int starts, unsigned world, uint64_t first_item, uint64_t n_items, void *d_records, uint64_t *d_tags,
                                  uint64_t *counts) {
    if (int rc = dw_check_shard(ctx)) return rc;
    if (!counts || world < 1 || world > 1024) return fail(ctx, SMX_INVALID_PARAMETER, "bad world / null counts");
    HIPCHK(hipSetDevice(ctx->device));
    int rc;
    switch (ctx->g_nw) {
        case 1: rc = dw_requests<1>(ctx, starts != 0, world, d_records, (unsigned long long *)d_tags, counts, first_item, n_items); break;
        case 2: rc = dw_requests<2>(ctx, starts != 0, world, d_records, (unsigned long long *)d_tags, counts, first_item, n_items); break;
        case 3: rc = dw_requests<3>(ctx, starts != 0, world, d_records, (unsigned long long *)d_tags, counts, first_item, n_items); break;
        default: rc = dw_requests<4>(ctx, starts != 0, world, d_records, (unsigned long long *)d_tags, counts, first_item, n_items); break;
    }
    return finish_call(ctx, rc, false);
}

int smx_shard_walk_requests(smx_ctx *ctx, int starts, unsigned world, void *d_records, uint64_t *d_tags, uint64_t *counts) {
    return smx_shard_walk_requests_range(ctx, starts, world, 0, ~0ull, d_records, d_tags, counts);
}

int smx_shard_walk_starts(const smx_ctx *cctx, uint64_t *d_starts) {
    smx_ctx *ctx = const_cast<smx_ctx *>(cctx);
    if (!ctx || !ctx->dw_ready) return SMX_INVALID_PARAMETER;
    if (ctx->dw_ncand == 0) return SMX_OK;
    if (!d_starts) return SMX_INVALID_PARAMETER;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(d_starts, ctx->dw_cand, ctx->dw_ncand * 8, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SMX_OK;
}

int smx_shard_lookup(smx_ctx *ctx, const void *d_records, uint64_t n, uint64_t *d_reply) {
    if (int rc = dw_check_shard(ctx)) return rc;
    if (n && (!d_records || !d_reply)) return fail(ctx, SMX_INVALID_PARAMETER, "null lookup buffers");
    HIPCHK(hipSetDevice(ctx->device));
    int rc;
    switch (ctx->g_nw) {
        case 1: rc = dw_lookup<1>(ctx, d_records, n, (unsigned long long *)d_reply); break;
        case 2: rc = dw_lookup<2>(ctx, d_records, n, (unsigned long long *)d_reply); break;
        case 3: rc = dw_lookup<3>(ctx, d_records, n, (unsigned long long *)d_reply); break;
        default: rc = dw_lookup<4>(ctx, d_records, n, (unsigned long long *)d_reply); break;
    }
    return finish_call(ctx, rc, false);
}

int smx_shard_gather_kmers(smx_ctx *ctx, const uint64_t *d_local_ranks, uint64_t n, void *d_kmers, uint8_t *d_masks) {
    if (int rc = dw_check_shard(ctx)) return rc;
    if (n == 0) return SMX_OK;
    if (!d_local_ranks || !d_kmers || !d_masks || !ctx->g_kmers || !ctx->g_mask) return fail(ctx, SMX_INVALID_PARAMETER, "null buffers / no shard");
    HIPCHK(hipSetDevice(ctx->device));
    switch (ctx->g_nw) {
        case 1: hipLaunchKernelGGL((k_gather_kmers<1>), dim3(grid_for(n)), dim3(BLK), 0, ctx->stream, (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask, (const unsigned long long *)d_local_ranks, n, d_kmers, d_masks); break;
        case 2: hipLaunchKernelGGL((k_gather_kmers<2>), dim3(grid_for(n)), dim3(BLK), 0, ctx->stream, (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask, (const unsigned long long *)d_local_ranks, n, d_kmers, d_masks); break;
        case 3: hipLaunchKernelGGL((k_gather_kmers<3>), dim3(grid_for(n)), dim3(BLK), 0, ctx->stream, (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask, (const unsigned long long *)d_local_ranks, n, d_kmers, d_masks); break;
        default: hipLaunchKernelGGL((k_gather_kmers<4>), dim3(grid_for(n)), dim3(BLK), 0, ctx->stream, (const void *)ctx->g_kmers, (const uint8_t *)ctx->g_mask, (const unsigned long long *)d_local_ranks, n, d_kmers, d_masks); break;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SMX_OK;
}

int smx_shard_unitigs(smx_ctx *ctx, uint64_t first_rank, const uint64_t *d_steps, const uint64_t *d_last, const uint64_t *d_base_off, const uint8_t *d_bases,
                      uint64_t *n_kept, uint64_t *n_words) {
    if (int rc = dw_check_shard(ctx)) return rc;
    if (!n_kept || !n_words) return SMX_INVALID_PARAMETER;
    HIPCHK(hipSetDevice(ctx->device));
    int rc;
    switch (ctx->g_nw) {
        case 1: rc = dw_unitigs<1>(ctx, first_rank, (const unsigned long long *)d_steps, (const unsigned long long *)d_last, (const unsigned long long *)d_base_off, d_bases, n_kept, n_words); break;
        case 2: rc = dw_unitigs<2>(ctx, first_rank, (const unsigned long long *)d_steps, (const unsigned long long *)d_last, (const unsigned long long *)d_base_off, d_bases, n_kept, n_words); break;
        case 3: rc = dw_unitigs<3>(ctx, first_rank, (const unsigned long long *)d_steps, (const unsigned long long *)d_last, (const unsigned long long *)d_base_off, d_bases, n_kept, n_words); break;
        default: rc = dw_unitigs<4>(ctx, first_rank, (const unsigned long long *)d_steps, (const unsigned long long *)d_last, (const unsigned long long *)d_base_off, d_bases, n_kept, n_words); break;
    }
    return finish_call(ctx, rc, false);
}

int smx_shard_unitigs_copy(const smx_ctx *cctx, uint64_t *d_words, uint64_t *d_len, uint64_t *d_start, uint64_t *d_end, uint8_t *d_self) {
    smx_ctx *ctx = const_cast<smx_ctx *>(cctx);
    if (!ctx || !ctx->dw_ready) return SMX_INVALID_PARAMETER;
    HIPCHK(hipSetDevice(ctx->device));
    const uint64_t ne = ctx->g_ne;
    if (ctx->g_nuwords && d_words) HIPCHK(hipMemcpyAsync(d_words, ctx->g_uwords, ctx->g_nuwords * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (ne && d_len) HIPCHK(hipMemcpyAsync(d_len, ctx->g_elen, ne * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (ne && d_start) HIPCHK(hipMemcpyAsync(d_start, ctx->g_estart, ne * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (ne && d_end) HIPCHK(hipMemcpyAsync(d_end, ctx->g_eend, ne * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (ne && d_self) HIPCHK(hipMemcpyAsync(d_self, ctx->g_eself, ne, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SMX_OK;
}

int smx_build_graph_from_unitigs(smx_ctx *ctx, unsigned k, unsigned num_buckets, uint64_t n_kmers, uint64_t n_kpomers, const uint64_t *d_words, uint64_t n_words,
                                 const uint64_t *d_len, const uint64_t *d_start, const uint64_t *d_end, const uint8_t *d_self, uint64_t n_unitigs,
                                 const uint64_t *loop_ranks, const uint64_t *loop_kmers, const uint8_t *loop_masks, uint64_t n_loop_kmers) {
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (k < 1 || k >= 128 || k % 2 == 0) return fail(ctx, SMX_INVALID_PARAMETER, "k-mer size must be odd and below 128");
    if (n_unitigs && (!d_words || !d_len || !d_start || !d_end || !d_self)) return fail(ctx, SMX_INVALID_PARAMETER, "null unitig arrays");
    if (n_loop_kmers && (!loop_ranks || !loop_kmers || !loop_masks)) return fail(ctx, SMX_INVALID_PARAMETER, "null loop k-mer arrays");
    HIPCHK(hipSetDevice(ctx->device));
    ctx->xnames.clear();
    ctx->xms.clear();
    int rc;
#define SMX_GFU(N) graph_from_unitigs<N>(ctx, k, num_buckets, n_kmers, n_kpomers, d_words, n_words, (const unsigned long long *)d_len, (const unsigned long long *)d_start, \
                                         (const unsigned long long *)d_end, d_self, n_unitigs, loop_ranks, loop_kmers, loop_masks, n_loop_kmers)
    switch ((k + 32) / 32) {
        case 1: rc = SMX_GFU(1); break;
        case 2: rc = SMX_GFU(2); break;
        case 3: rc = SMX_GFU(3); break;
        default: rc = SMX_GFU(4); break;
    }
#undef SMX_GFU
    return finish_call(ctx, rc, true);
}

// (k+1)-mer file for the coverage pass of a graph that was built without one on this rank (sharded construction): sorted, bucket-major
int smx_graph_set_kpomers(smx_ctx *ctx, const void *d_kpomers, uint64_t n, const uint64_t *bucket_sizes) {
    if (!ctx || !ctx->g_ready || !bucket_sizes) return SMX_INVALID_PARAMETER;
    if (n && !d_kpomers) return SMX_INVALID_PARAMETER;
    HIPCHK(hipSetDevice(ctx->device));
    {   // (checked before anything of the context changes: a refused call leaves the graph's bookkeeping as it was)
        uint64_t sum = 0;
        for (unsigned b = 0; b < ctx->g_B; ++b) sum += bucket_sizes[b];
        if (sum != n) return fail(ctx, SMX_INVALID_PARAMETER, "bucket sizes do not add up to the number of (k+1)-mers");
    }
    drop_kpo(ctx);
    // A SHARD of the file is about to stand for it (fewer records than the graph's (k+1)-mer count): smx_graph_info keeps reporting the whole
    // count and smx_graph_set_coverage puts it back. A file of the full size is a replacement, not a shard: it ends any latch.
    if (n < (ctx->g_nkpo_total ? ctx->g_nkpo_total : ctx->g_nkpo)) {
        if (!ctx->g_nkpo_total) ctx->g_nkpo_total = ctx->g_nkpo;
    } else {
        ctx->g_nkpo_total = 0;
    }
    ctx->g_kpoboff.assign(ctx->g_B + 1, 0);
    for (unsigned b = 0; b < ctx->g_B; ++b) ctx->g_kpoboff[b + 1] = ctx->g_kpoboff[b] + bucket_sizes[b];
    ctx->g_nkpo = n;
    if (n == 0) return SMX_OK;
    if (int rc = dalloc(ctx, (uint64_t **)&ctx->g_kpo, (size_t)n * ctx->g_nw, false)) return rc;
    HIPCHK(hipMemcpyAsync(ctx->g_kpo, d_kpomers, (size_t)n * ctx->g_nw * 8, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SMX_OK;
}

int smx_graph_set_coverage(smx_ctx *ctx, const uint32_t *raw_coverage, uint64_t n_edges) {
    if (!ctx || !ctx->g_ready) return SMX_INVALID_PARAMETER;
    if (n_edges != ctx->g_ne) return fail(ctx, SMX_INVALID_PARAMETER, "coverage array has %llu entries, graph has %llu unitigs",
                                          (unsigned long long)n_edges, (unsigned long long)ctx->g_ne);
    if (n_edges && !raw_coverage) return SMX_INVALID_PARAMETER;
    ctx->gh.ecov.assign(raw_coverage, raw_coverage + n_edges);
    // The caller summed the edge coverages itself (over ranks, or over shards of the (k+1)-mer file installed one after the other): what the
    // last smx_graph_fill_coverage left next to them — flanking coverage and the multiplicity histogram — saw one rank's reads against one
    // shard only. They are dropped rather than served (smx_graph_copy_flanking / smx_graph_coverage_histogram then say so), and the
    // (k+1)-mer count of the whole file comes back after a shard stood in for it.
    ctx->gh.eflank_s.clear();
    ctx->gh.eflank_e.clear();
    ctx->g_cov_hist.clear();
    if (ctx->g_nkpo_total) {
        ctx->g_nkpo = ctx->g_nkpo_total;
        ctx->g_nkpo_total = 0;
    }
    return SMX_OK;
}

// one bucket of the result into caller-owned HBM (a two-strand result merges the bucket's strands straight into the block)
int smx_copy_bucket_device(const smx_ctx *cctx, unsigned bucket, void *d_dst) {
    smx_ctx *ctx = const_cast<smx_ctx *>(cctx);
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (int rc = ensure_kmer_file(ctx)) return rc;
    if (bucket >= ctx->num_buckets) return fail(ctx, SMX_INVALID_PARAMETER, "bucket %u out of range", bucket);
    const uint64_t o = ctx->bucket_off[bucket], n = ctx->bucket_off[bucket + 1] - o;
    if (n == 0) return SMX_OK;
    if (!d_dst) return SMX_INVALID_PARAMETER;
    if (ctx->result_on_file) return on_file(ctx);
    if (ctx->result_on_host) return fail(ctx, SMX_MEMORY_LIMIT_EXCEEDED, "the result was spilled to host memory (it does not fit the HBM budget)");
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->ts.active) {
        if (int rc = ts_merge_bucket_any(ctx, bucket, d_dst)) return rc;
    } else {
        HIPCHK(hipMemcpyAsync(d_dst, (const char *)ctx->d_result + o * (size_t)ctx->nw * 8, n * (size_t)ctx->nw * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SMX_OK;
}

int smx_copy_kmers_device(const smx_ctx *cctx, void *d_dst) {
    smx_ctx *ctx = const_cast<smx_ctx *>(cctx);
    if (!ctx) return SMX_INVALID_PARAMETER;
    if (ctx->n_records == 0) return SMX_OK;
    if (!d_dst) return SMX_INVALID_PARAMETER;
    if (int rc = ensure_kmer_file(ctx)) return rc;
    if (ctx->result_on_file) return on_file(ctx);
    if (ctx->result_on_host) return fail(ctx, SMX_MEMORY_LIMIT_EXCEEDED, "the result was spilled to host memory (it does not fit the HBM budget)");
    HIPCHK(hipSetDevice(ctx->device));
    if (ctx->ts.active) {  // merged straight into the caller's block
        for (unsigned b = 0; b < ctx->num_buckets; ++b)
            if (int rc = ts_merge_bucket_any(ctx, b, (char *)d_dst + ctx->bucket_off[b] * ctx->nw * 8)) return rc;
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return SMX_OK;
    }
    HIPCHK(hipMemcpyAsync(d_dst, ctx->d_result, ctx->n_records * ctx->nw * 8, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return SMX_OK;
}

int smx_graph_info(const smx_ctx *ctx, uint64_t *info /* [8] */) {
    if (!ctx || !info) return SMX_INVALID_PARAMETER;
    if (!ctx->g_ready) return SMX_INVALID_PARAMETER;
    info[0] = ctx->g_nkpo;
    info[1] = ctx->g_nkmers;
    info[2] = ctx->g_ne;
    info[3] = ctx->g_nloops;
    info[4] = ctx->g_links_dev ? ctx->g_nv : ctx->gh.n_vertices;
    info[5] = ctx->gh.n_links;
    info[6] = ctx->g_nbases;
    info[7] = ctx->g_nw;
    return SMX_OK;
}

int smx_graph_route_stats(const smx_ctx *ctx, uint64_t *stats /* [8] */) {
    if (!ctx || !stats || !ctx->g_ready) return SMX_INVALID_PARAMETER;
    for (int i = 0; i < 8; ++i) stats[i] = ctx->g_route_stats[i];
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
    if (ctx->g_sharded_file) return fail(ctx, SMX_INVALID_PARAMETER, "this graph was built from gathered unitigs: the k-mer file is sharded over the ranks");
    if (int rc = ensure_kmer_file(ctx, /*view=*/false)) return rc;
    if (kmers_host) HIPCHK(hipMemcpy(kmers_host, ctx->g_kmers, ctx->g_nkmers * ctx->g_nw * 8, hipMemcpyDeviceToHost));
    if (masks_host) HIPCHK(hipMemcpy(masks_host, ctx->g_mask, ctx->g_nkmers, hipMemcpyDeviceToHost));
    return SMX_OK;
}

int smx_graph_copy_unitigs(const smx_ctx *cctx, uint64_t *offsets, char *seq) {
    smx_ctx *ctx = const_cast<smx_ctx *>(cctx);
    if (!ctx || !ctx->g_ready) return SMX_INVALID_PARAMETER;
    HIPCHK(hipSetDevice(ctx->device));
    if (int rc = materialize_host(ctx)) return rc;
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
    const size_t ne = ctx->g_ne;
    if (ctx->gh.eflank_s.size() != ne || ctx->gh.eflank_e.size() != ne)
        return fail(const_cast<smx_ctx *>(ctx), SMX_INVALID_PARAMETER, "no flanking coverage: smx_graph_fill_coverage has not run on this graph, or its coverage was set from outside (smx_graph_set_coverage)");
    if (ne && flank_edge) memcpy(flank_edge, ctx->gh.eflank_s.data(), ne * 4);
    if (ne && flank_conjugate) memcpy(flank_conjugate, ctx->gh.eflank_e.data(), ne * 4);
    return SMX_OK;
}

int smx_graph_coverage_histogram(const smx_ctx *ctx, uint64_t *hist, uint64_t capacity, uint64_t *n_entries) {
    if (!ctx || !ctx->g_ready || !n_entries) return SMX_INVALID_PARAMETER;
    *n_entries = ctx->g_cov_hist.size();
    if (hist)
        for (uint64_t c = 0; c < std::min<uint64_t>(capacity, ctx->g_cov_hist.size()); ++c) hist[c] = ctx->g_cov_hist[c];
    return SMX_OK;
}

int smx_graph_copy_coverage(const smx_ctx *ctx, uint32_t *raw_coverage) {
    if (!ctx || !ctx->g_ready || !raw_coverage) return SMX_INVALID_PARAMETER;
    if (ctx->gh.ecov.size() != ctx->g_ne) return SMX_INVALID_PARAMETER;
    if (!ctx->gh.ecov.empty()) memcpy(raw_coverage, ctx->gh.ecov.data(), ctx->gh.ecov.size() * 4);
    return SMX_OK;
}

// Device-formatted GFA (smx_gfa.hip). Returns 1 when this writer does not apply (link records on the host, no room for the text):
// the caller then takes the host writer.
static int write_gfa_device(smx_ctx *ctx, const char *path, const char *flavour) {
    const uint64_t ne = ctx->g_ne, nv = ctx->g_nv;
    if (!ctx->g_links_dev || ne == 0 || ctx->opt_device_gfa == 0) return 1;
    const bool cov = ctx->gh.ecov.size() == ne;
    std::vector<void *> mine;  // device blocks of this call
    auto done = [&](int code) {
        (void)hipStreamSynchronize(ctx->stream);
        for (void *p : mine) arena_put(ctx, p);
        return code;
    };
    // (a device error leaves through done() as well: HIPCHK's plain return would keep this call's blocks out of the arena — ADVICE r4)
#define GFACHK(call)                                                                                                                    \
    do {                                                                                                                                \
        hipError_t e_ = (call);                                                                                                         \
        if (e_ != hipSuccess)                                                                                                           \
            return done(fail(ctx, e_ == hipErrorOutOfMemory ? SMX_MEMORY_LIMIT_EXCEEDED : SMX_DEVICE_ERROR, "%s failed: %s (%s:%d)", #call, \
                             hipGetErrorString(e_), __FILE__, __LINE__));                                                               \
    } while (0)
    auto take = [&](auto **p, size_t n) {
        const int rc = dalloc(ctx, p, std::max<size_t>(n, 1), false);
        if (rc == 0) mine.push_back((void *)*p);
        return rc;
    };
    unsigned long long *soff, *loff, *d_nl;
    uint32_t *d_taglen = nullptr;
    unsigned long long *d_tagoff = nullptr;
    char *d_pool = nullptr;
    if (take(&soff, ne + 1) || take(&loff, nv + 1) || take(&d_nl, 1)) {
        ctx->err.clear();
        return done(1);
    }
    // coverage tags: "\tDP:f:%g\tKC:i:%u\n" per edge, formatted by the host threads ("DP:f:" << float(cov): ostream default = %g)
    if (cov) {
        std::vector<unsigned long long> elen;
        if (int rc = d2h(ctx, elen, ctx->g_elen, ne)) return done(rc);
        std::vector<uint32_t> tl(ne);
        std::vector<unsigned long long> to(ne + 1, 0);
        const unsigned k = ctx->g_k;
        const uint32_t *ec = ctx->gh.ecov.data();
        auto fmt = [&](size_t i, char *t) {
            const double c = (double)ec[i] / (double)(elen[i] - k);
            return snprintf(t, 64, "\tDP:f:%g\tKC:i:%u\n", (double)(float)c, ec[i]);
        };
        smxh::parallel_blocks(ne, (size_t)1 << 16, [&](size_t b, size_t e) {
            char t[64];
            for (size_t i = b; i < e; ++i) tl[i] = (uint32_t)fmt(i, t);
        });
        for (size_t i = 0; i < ne; ++i) to[i + 1] = to[i] + tl[i];
        std::vector<char> pool(to[ne] + 64);
        smxh::parallel_blocks(ne, (size_t)1 << 16, [&](size_t b, size_t e) {
            char t[64];
            for (size_t i = b; i < e; ++i) {
                fmt(i, t);
                memcpy(pool.data() + to[i], t, tl[i]);
            }
        });
        if (take(&d_taglen, ne) || take(&d_tagoff, ne + 1) || take(&d_pool, pool.size())) {
            ctx->err.clear();
            return done(1);
        }
        GFACHK(hipMemcpyAsync(d_taglen, tl.data(), ne * 4, hipMemcpyHostToDevice, ctx->stream));
        GFACHK(hipMemcpyAsync(d_tagoff, to.data(), (ne + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        GFACHK(hipMemcpyAsync(d_pool, pool.data(), pool.size(), hipMemcpyHostToDevice, ctx->stream));
        GFACHK(hipStreamSynchronize(ctx->stream));
    }
    const unsigned g1 = (unsigned)std::min<uint64_t>((ne + BLK - 1) / BLK, 256 * 16), g2 = (unsigned)std::min<uint64_t>((nv + BLK - 1) / BLK, 256 * 16);
    hipLaunchKernelGGL(k_gfa_s_len, dim3(g1), dim3(BLK), 0, ctx->stream, (const unsigned long long *)ctx->g_elen, ne, (const uint32_t *)d_taglen, soff);
    GFACHK(hipGetLastError());
    GFACHK(hipMemsetAsync(d_nl, 0, 8, ctx->stream));
    if (nv) {
        hipLaunchKernelGGL((k_gfa_l<false>), dim3(std::max(g2, 1u)), dim3(BLK), 0, ctx->stream, (const Rec<2> *)ctx->g_lrecs, ctx->g_nlrec, ctx->g_lsh,
                           (const unsigned long long *)ctx->g_vstart, nv, (const uint8_t *)ctx->g_eself, ctx->g_k, loff, d_nl, (const unsigned long long *)nullptr, (char *)nullptr);
        GFACHK(hipGetLastError());
    }
    if (int rc = scan_u64(ctx, soff, soff, ne)) return done(rc);
    if (nv)
        if (int rc = scan_u64(ctx, loff, loff, nv)) return done(rc);
    unsigned long long ts = 0, tl_ = 0, nlinks = 0;
    GFACHK(hipMemcpyAsync(&ts, soff + ne, 8, hipMemcpyDeviceToHost, ctx->stream));
    if (nv) GFACHK(hipMemcpyAsync(&tl_, loff + nv, 8, hipMemcpyDeviceToHost, ctx->stream));
    GFACHK(hipMemcpyAsync(&nlinks, d_nl, 8, hipMemcpyDeviceToHost, ctx->stream));
    GFACHK(hipStreamSynchronize(ctx->stream));
    free_temps(ctx);  // (the scans' scratch)
    const size_t total = (size_t)ts + (size_t)tl_;
    char *text;
    if (take(&text, total + 64)) {  // no room for the text next to the graph: the host writer streams it block by block
        ctx->err.clear();
        return done(1);
    }
    hipLaunchKernelGGL(k_gfa_s_write, dim3(std::max(g1, 1u)), dim3(BLK), 0, ctx->stream, (const uint64_t *)ctx->g_uwords, (const unsigned long long *)ctx->g_eoffw,
                       (const unsigned long long *)ctx->g_elen, ne, (const unsigned long long *)soff, (const char *)d_pool, (const unsigned long long *)d_tagoff, text);
    GFACHK(hipGetLastError());
    if (nv) {
        hipLaunchKernelGGL((k_gfa_l<true>), dim3(std::max(g2, 1u)), dim3(BLK), 0, ctx->stream, (const Rec<2> *)ctx->g_lrecs, ctx->g_nlrec, ctx->g_lsh,
                           (const unsigned long long *)ctx->g_vstart, nv, (const uint8_t *)ctx->g_eself, ctx->g_k, (unsigned long long *)nullptr, (unsigned long long *)nullptr,
                           (const unsigned long long *)loff, text + ts);
        GFACHK(hipGetLastError());
    }
    GFACHK(hipStreamSynchronize(ctx->stream));
    // to the file: the header, then the text through a ring of page-locked buffers; every buffer is written by several pwrite threads
    const int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);  // (read access too: a tmpfs output is filled through a shared mapping)
    if (fd < 0) return done(fail(ctx, SMX_IO_ERROR, "Cannot open %s for writing", path));
    const std::string head = std::string("H\tsp:Z:") + flavour + "\n";
    smxio::FileSink sink;
    sink.begin(fd, head.size() + total);
    bool ok = sink.put(head.data(), head.size(), 0);
    DevToFile d2f;
    if (!d2f.init(total)) ok = false;
    if (ok && !d2f.send(ctx->stream, sink, (off_t)head.size(), text, total)) ok = false;
    (void)hipStreamSynchronize(ctx->stream);
    if (!d2f.finish()) ok = false;
    if (!sink.end()) ok = false;
    if (close(fd) != 0) ok = false;
    ctx->gh.n_links = nlinks;
    if (!ok) return done(fail(ctx, SMX_IO_ERROR, "I/O error writing %s", path));
    return done(SMX_OK);
}
#undef GFACHK

int smx_graph_write_gfa(smx_ctx *ctx, const char *path, const char *flavour_version) {
    if (!ctx || !path) return SMX_INVALID_PARAMETER;
    if (!ctx->g_ready) return fail(ctx, SMX_INVALID_PARAMETER, "no graph built");
    (void)hipSetDevice(ctx->device);
    const char *fv = flavour_version ? flavour_version : "SPAdes-4.3.0-dev";
    const double t0 = wall_now();
    {
        const int rc = write_gfa_device(ctx, path, fv);  // the text formatted on the device; 1: this graph takes the host writer
        if (rc != 1) {
            if (getenv("SMX_DEBUG")) fprintf(stderr, "[smx] write_gfa: formatted on the device, %.3f s\n", wall_now() - t0);
            return rc;
        }
    }
    if (int rc = materialize_host(ctx)) return rc;
    const double t1 = wall_now();
    FILE *f = fopen(path, "wb");
    if (!f) return fail(ctx, SMX_IO_ERROR, "Cannot open %s for writing", path);
    bool ok = smxh::write_gfa(ctx->gh, f, fv);
    if (fclose(f) != 0) ok = false;
    if (getenv("SMX_DEBUG")) fprintf(stderr, "[smx] write_gfa: graph to the host %.3f s, text %.3f s\n", t1 - t0, wall_now() - t1);
    return ok ? SMX_OK : fail(ctx, SMX_IO_ERROR, "I/O error writing %s", path);
}

int smx_graph_write_fastg(smx_ctx *ctx, const char *path) {
    if (!ctx || !path) return SMX_INVALID_PARAMETER;
    if (!ctx->g_ready) return fail(ctx, SMX_INVALID_PARAMETER, "no graph built");
    (void)hipSetDevice(ctx->device);
    if (int rc = materialize_host(ctx)) return rc;
    FILE *f = fopen(path, "wb");
    if (!f) return fail(ctx, SMX_IO_ERROR, "Cannot open %s for writing", path);
    bool ok = smxh::write_fastg(ctx->gh, f);
    if (fclose(f) != 0) ok = false;
    return ok ? SMX_OK : fail(ctx, SMX_IO_ERROR, "I/O error writing %s", path);
}

int smx_graph_write_spades(smx_ctx *ctx, const char *basename) {
    if (!ctx || !basename) return SMX_INVALID_PARAMETER;
    if (!ctx->g_ready) return fail(ctx, SMX_INVALID_PARAMETER, "no graph built");
    (void)hipSetDevice(ctx->device);
    if (int rc = materialize_host(ctx)) return rc;
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
int smx_host_write_graph(unsigned k, uint64_t n_edges, const uint64_t *offsets, const char *seq, const uint64_t *start_node,
                         const uint64_t *end_node, const uint32_t *raw_coverage, int sort_edges, int format, const char *path,
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
    (void)hipSetDevice(ctx->device);
    if (int rc = materialize_host(ctx)) return rc;
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
