/* include/smx.h — C ABI of libspades_mi355x.so
 *
 * MI355X-native (gfx950) replacement for the k-mer counting / de Bruijn construction hot path of
 * SPAdes. The reference has no C ABI for this path; its seams are C++ virtuals and on-disk files
 * (SURVEY.md §8b). Every entry point below names the reference interface it stands in for
 * (paths relative to /root/reference/src). INTEGRATION.md shows the reference-side binding.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ / torch types cross the boundary;
 *   - every function returns 0 or one of the reference's process exit codes
 *     (common/utils/logger/error_codes.hpp:14-20), never throws; text via smx_last_error();
 *   - one host thread per context (thread-compatible, like one KMerDiskCounter object);
 *   - k-mer record = ceil(K/32) little-endian uint64, nucleotide i at bits 2*(i mod 32) of word
 *     i/32, A=0 C=1 G=2 T=3, pad bits zero (common/sequence/rtseq.hpp:131-151,379-382);
 *   - "device" pointers are HBM addresses on the context's GPU.
 */
#ifndef SMX_H
#define SMX_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct smx_ctx smx_ctx;

enum { /* common/utils/logger/error_codes.hpp:14-20 */
    SMX_OK = 0,
    SMX_INVALID_INPUT_FORMAT = 64,
    SMX_INPUT_FILE_NOT_FOUND = 65,
    SMX_IO_ERROR = 66,
    SMX_INVALID_PARAMETER = 67,
    SMX_MEMORY_LIMIT_EXCEEDED = 68,
    SMX_DEVICE_ERROR = 70 /* HIP runtime failure (no reference equivalent) */
};

enum { /* which k-mers a read contributes */
    SMX_MODE_ALL = 0,      /* 'A': every K-mer of read and RC(read) — spades-kmercount,
                              projects/spades_tools/kmercount.cpp:65-83 */
    SMX_MODE_CANONICAL = 1 /* 'B': only K-mers with IsMinimal() — construction,
                              common/kmer_index/kmer_mph/kmer_splitters.hpp:28-44 +
                              common/kmer_index/ph_map/storing_traits.hpp:92-101 */
};

/* ---- context ------------------------------------------------------------------------------
 * Replaces the pair (workdir, KMerDiskCounter object): kmer_index_builder.hpp:284-304.
 * device = HIP device ordinal; hbm_budget_bytes = 0 -> use what the device has free. */
int smx_create(smx_ctx **out, int device, size_t hbm_budget_bytes);
/* Device memory after smx_destroy: the context's arena (one virtual range, physically backed as far as it was ever used) is NOT
 * unmapped — it is parked for the next context of this process on the same device with the same budget class (mapping costs
 * ~17 ms per GiB; tearing a range down has crashed inside the HIP runtime). At most one arena per device is parked: a context with
 * another budget class, or the next destroy, tears the parked one down. SMX_ARENA_POOL=0 in the environment unmaps at every destroy. */
void smx_destroy(smx_ctx *ctx);
/* Asks the context to give unused device memory back now, e.g. before another allocator of the same process (a framework's caching
 * allocator) needs the room; *bytes_returned (may be NULL) says how much went. With the default arena (one virtual range, mapped in
 * 512 MiB chunks) that is currently always 0: releasing chunks and mapping memory again shortly afterwards lost writes on this stack
 * (arena_trim in smx_ctx.hpp has the measurements), so the arena only grows while its context lives and goes back at smx_destroy; the
 * hipMalloc fallback (SMX_ARENA=malloc) frees its cached blocks. The context stays usable either way. */
int smx_trim(smx_ctx *ctx, size_t *bytes_returned);
/* Device memory the context holds but does not use right now (free blocks inside its arena, bytes): a neighbour that sizes its own
 * work from the device's free memory (hipMemGetInfo) must add this — the arena only grows, so what the library released after a big
 * step is invisible to the device-level figure. No reference equivalent (the reference has no device). */
int smx_arena_free_bytes(smx_ctx *ctx, size_t *bytes);
/* Returns at once; a helper thread of the context backs `bottom_bytes` of the arena's temporary region and `top_bytes` of its long-lived region
 * with physical memory (512 MiB chunks, ~17 ms per GiB on this stack) while the caller goes on — a tool that builds ONE graph per process calls
 * it before it reads its input, so that the mapping its first build would pay for (1.25 s of a 1.39 s "build graph" at 20 M reads) overlaps the
 * parse. A hint, never a promise: the helper stops where the device runs short, leaves a context with an HBM budget alone, and any call that needs
 * memory meanwhile maps what it needs itself. The reference's counterpart is the page cache warming up under `-tmp-dir`: no API there. */
int smx_prewarm(smx_ctx *ctx, size_t bottom_bytes, size_t top_bytes);
/* A block of the context's device arena for the CALLER's own use (bytes, 256-byte aligned), and its return. For a neighbour whose working arrays
 * must live next to what the library holds — the walk state of spades_amd.dist.distributed_walks, 9–10 B per oriented node of the rank's shard: the
 * arena only grows and gives nothing back before smx_destroy, so after the library's big steps (the sharded count peaks at ~2x the shard) a framework
 * allocator finds the device full although half of the arena is free. The block stays valid until smx_pool_free or smx_destroy; the library never
 * reads or writes it. SMX_MEMORY_LIMIT_EXCEEDED when the arena (and the HBM budget) cannot hold it. No reference equivalent. */
int smx_pool_alloc(smx_ctx *ctx, size_t bytes, void **d_block);
int smx_pool_free(smx_ctx *ctx, void *d_block);
const char *smx_last_error(const smx_ctx *ctx);
const char *smx_version(void);
/* Options. Behaviour switches of the reference's spades-core Construction stage (defaults reproduce spades-gbuilder):
 *   "sort_edges" = 1          unitigs ordered by Sequence::RawCompare before ids are assigned — DeBruijnGraphExtentionConstructor::
 *                             ConstructGraph, common/assembly_graph/construction/debruijn_graph_constructor.hpp:590-604 (thread-independent ids)
 *   "keep_perfect_loops" = 0  drop perfect loops (config key construction.keep_perfect_loops; same function)
 *   "early_tip_bound" = N     EarlyTipClipperProcessor(index, N) before condensation (construction/early_simplification.hpp:38-162;
 *                             N = RL - K in the reference, stages/construction.cpp:296-301); 0 = off
 *   "early_at_remover" = 1    EarlyLowComplexityClipperProcessor(index, 0.8, 10, 200) of the RNA pipelines (:164-347; construction.cpp:317-326)
 *   "submit_contigs" = 1      reads submitted while set are contigs: construction yes, coverage no (construction.cpp:108-117)
 *   "flank_range" = 50        FlankingCoverage averaging range (graph_support/detail_coverage.hpp:69-76)
 * Engine knobs (no reference equivalent; closest is the -b buffer-size knob of kmercount.cpp:139); results never depend on them:
 *   "prededupe" (-1 auto, 0 direct pipeline, 1 force the super-k-mer stage), "skm_cap", "skm_scap", "skm_stage" (its chunk sizes / staging),
 *   "skm_nkey_log2" (the stage starts from 2^this minimizer partitions instead of 2^24; 12..28, tests),
 *   "batch_records" (force HBM-bounded batches), "leaf_cap", "leaf_target", "leaf_grid", "leaf_tab", "s1", "s2" (leaf / MSD split geometry),
 *   "joint_hist" (level-2 histogram counted with level 1), "device_links" (0 host, 1 device from 65 536 edges, 2 always),
 *   "derive_batches" (k-mer file of the construction in this many bucket ranges; 0 = as HBM requires), "keep_kpo" (-1 keep the
 *   (k+1)-mer file after the masks if HBM allows, 0 drop it: the coverage pass recounts), "verify_lookups" (1: rank lookups of
 *   k-mers that are present by construction still compare the record), "spill" (1: always keep sorted runs in host memory and
 *   merge them by bucket ranges; -1 only when the set outgrows the HBM budget), "device_loops" (1: the perfect loops — CollectLoops,
 *   debruijn_graph_constructor.hpp:359-397 — are made by kernels on the walks' successor table, smx_loops.hip, odd k; 0: by all host
 *   cores on the gathered left-over k-mers, smx_loops_host.hpp; same result), "spill_merge_max" (> 0: that merge takes at most
 *   this many records at once, so a small input goes through the key-range split of a bucket), "ext_route" (-1 / 1: the construction takes k-mers and
 *   InOutMask bytes from ONE count of the reads where the record has 8 spare bits, the pre-dedupe stage applies and one batch fits;
 *   0: always the (k+1)-mer file first, as the reference does), "ext_presort" (0: copies of a k-mer from cut partitions are merged
 *   after the sort instead of before it), "kmers_from_reads" (0: the k-mer file of the second route is derived from the (k+1)-mer
 *   file, not counted from the reads), "dir_slots" (rank directory: slots per record, 1..8; -1 = 2, or 1 next to a resident
 *   (k+1)-mer file), "pm_route" (-1 / 1: where "ext_route" applies and no early clipper is asked for, the construction does not sort
 *   the k-mers at all — nodes are numbered by minimizer partition, only the junction k-mers are put into k-mer-file order to number
 *   the unitigs, and the sorted k-mer file is made the first time smx_copy_final_kmers / smx_bucket_sizes / smx_graph_copy_kmers ask
 *   for it; 0: the k-mers are sorted first, as in the reference). The early clippers run on that route too since round 6 ("pm_full_retab" = 1: the whole
 *   node table is made again after an edit instead of the edited k-mers' entries). "pm_fuse_tab" (1: the dedupe stage writes the node table of its chunks from
 *   LDS; 0: link array + a pass of its own afterwards — what data that cut many partitions get either way), "walk_pack" (1: word offset and edge index of the
 *   kept paths from one scan), "pm_remote_mirror" (1: a successor outside its chunk is looked up from one end of the edge for both), "pm_overlap" (1: the
 *   successor table on a side stream beside the junction order) — measured variants of that route's kernels.
 * SMX_OPTS="key=value,..." in the environment applies options to every new context.
 *
 * HBM budget (smx_create): the context never holds more device memory than hbm_budget_bytes (0 = what the device has). A count whose
 * sorted-unique set does not fit is cut into batches whose runs are folded on the device or, when even that does not fit, kept in
 * host memory and merged one bucket range at a time (the reference's dump + merge, kmer_splitter.hpp:123-170,
 * kmer_index_builder.hpp:346-430) — a single bucket that is larger than what can be merged at once is cut by key range (every run's
 * slice of it is sorted: splitter keys cut them all, the parts are merged one after the other); the result is then served from host memory: smx_copy_bucket / smx_copy_final_kmers /
 * smx_write_final_kmers work as usual, smx_device_kmers returns NULL and smx_build_graph refuses (it needs the file resident). */
int smx_set_option(smx_ctx *ctx, const char *key, int64_t value);

/* ---- reads -> HBM -------------------------------------------------------------------------
 * Replaces the read streams the splitters pull from: io::EasyStream(file, followed_by_rc=true,
 * handle_Ns=true) (common/io/reads/io_helper.cpp:21-34) for mode A, and the binary read streams
 * (common/io/reads/binary_streams.hpp:54-102) for mode B. Reverse complements are NOT submitted:
 * the kernels generate them. Submissions append to the context's resident batch. */
int smx_reads_clear(smx_ctx *ctx);
/* ASCII reads, read i = bases[offsets[i] .. offsets[i+1]). The bytes are uploaded as they are; the reference's N rule
 * (each read is cut to its longest run of ACGTacgt, first one on ties: common/io/reads/longest_valid_wrapper.hpp:16-53)
 * and the 2-bit packing (common/io/reads/binary_converter.cpp:83-151 keeps the same 2-bit words) run on the device. */
int smx_submit_reads_ascii(smx_ctx *ctx, const char *bases, const uint64_t *offsets, uint64_t n_reads);
/* Already-packed reads in host memory: one 2-bit stream (layout as a k-mer record, arbitrarily
 * long), read i occupies nucleotides [start[i], start[i]+len[i]). Mirrors Sequence::BinWrite's
 * payload (common/sequence/sequence.hpp BinWrite: size_t len + words). The (start, len) pairs are checked against the stream on
 * the device; page-locked arrays (smx_pinned_alloc) upload at the PCIe rate. With option "async_upload" = 1 and a stream of
 * >= 2^24 words the call returns before the copies are over: (start, len) go first, the stream follows in pieces on a copy stream and
 * the first scan of the next smx_count / smx_build_graph follows the upload piece by piece; the three arrays must then stay valid
 * (and page-locked, to overlap at all) until that call returns, and a (start, len) pair that leaves the stream is reported by that
 * call (SMX_INVALID_PARAMETER) instead of this one. */
int smx_submit_reads_packed(smx_ctx *ctx, const uint64_t *words, uint64_t n_words,
                            const uint64_t *start, const uint32_t *len, uint64_t n_reads);
/* SPAdes' own binary read format: one <prefix>.seq file written by io::ReadConverter::ConvertToBinary
 * (common/io/reads/binary_converter.cpp:83-151, read back by common/io/reads/binary_streams.hpp:54-102) — what the
 * Construction stage streams inside a spades.py run. Single or paired file; reads arrive already N-trimmed. */
int smx_submit_reads_binary(smx_ctx *ctx, const char *seq_path);
/* Uncompressed strict 4-line FASTQ, parsed ON THE DEVICE (newline scan -> line roles -> per-read longest ACGT run + 2-bit packing):
 * replaces io::FastaFastaGzParser + LongestValidWrap + the binary conversion for the common case (common/io/reads/parser.cpp,
 * longest_valid_wrapper.hpp:16-53, binary_converter.cpp:83-151). `text` is a chunk of file bytes starting at a record; it may end
 * inside a record: only complete records are taken, *consumed = offset of the first unconsumed byte (carry the tail into the next
 * chunk). is_final = the chunk ends the file. Multi-line FASTQ, FASTA or any other structure -> SMX_INVALID_INPUT_FORMAT with
 * nothing submitted (use the host parser + smx_submit_reads_ascii then). */
int smx_submit_fastq_text(smx_ctx *ctx, const char *text, uint64_t n_bytes, int is_final, uint64_t *n_reads, uint64_t *consumed);
/* page-locked host buffers for such chunks (optional; any host memory works, pinned memory uploads at the PCIe rate) */
void *smx_pinned_alloc(size_t bytes);
void smx_pinned_free(void *p);
/* Same, but the three arrays already live in HBM (benchmark path: inputs resident before the
 * timed region). The context borrows the pointers until smx_reads_clear()/smx_destroy();
 * d_words must be readable for n_words + 8 words (tail pad). */
int smx_submit_reads_device(smx_ctx *ctx, const void *d_words, uint64_t n_words,
                            const void *d_start, const void *d_len, uint64_t n_reads);
/* Stream contract of every entry point that takes device pointers (smx_submit_reads_device, smx_count_records, smx_graph_shard_build,
 * smx_build_graph_from_kmers, smx_graph_set_kpomers, ...): the library works on a stream of its own. Whatever filled those buffers
 * (a kernel or a collective on another stream) must have completed before the call — synchronise that stream first, as
 * spades_amd/dist.py does after its exchanges; results handed out through device pointers are complete when the call returns. */
int smx_reads_info(const smx_ctx *ctx, uint64_t *n_reads, uint64_t *n_bases);

/* ---- counting -----------------------------------------------------------------------------
 * Replaces KMerCounter<RtSeq>::Count(num_buckets, num_threads)
 * (common/kmer_index/kmer_mph/kmer_index_builder.hpp:273,306-332) together with the splitter it
 * drives (KMerSplitter<RtSeq>::Split, kmer_mph/kmer_splitter.hpp:38). Result, resident in HBM:
 * num_buckets sorted-unique runs, bucket b = mulhi64(XXH3_64(record), num_buckets)
 * (kmer_mph/kmer_buckets.hpp:47-52), records strictly increasing as (w0,w1,..) tuples inside a
 * bucket — i.e. exactly the bytes of the reference's bucket files kmers_XXXXXX.<b>.
 * num_buckets: 16 for spades-kmercount (kmercount.cpp:220), 10*nthreads for construction
 * (extension_index/kmer_extension_index_builder.hpp:75). It is an input of the ORDER, not of the
 * parallelism. */
int smx_count(smx_ctx *ctx, unsigned K, int mode, unsigned num_buckets);
/* KMerDiskStorage::total_kmers / bucket_size (kmer_index_builder.hpp:139-150,176-178) */
int smx_count_info(const smx_ctx *ctx, uint64_t *n_records, unsigned *words_per_record,
                   uint64_t *n_kmer_instances);
int smx_bucket_sizes(const smx_ctx *ctx, uint64_t *sizes /* [num_buckets] */);
/* KMerDiskStorage::bucket(i) contents (kmer_index_builder.hpp:180-190) -> host memory */
int smx_copy_bucket(const smx_ctx *ctx, unsigned bucket, void *host_dst);
/* KMerDiskStorage::merge() (kmer_index_builder.hpp:190-203): buckets 0..B-1 concatenated =
 * the bytes of <workdir>/final_kmers (kmercount.cpp:221-223) */
int smx_copy_final_kmers(const smx_ctx *ctx, void *host_dst);
int smx_write_final_kmers(const smx_ctx *ctx, const char *path);
/* Count and file in one call, the destination known from the start: KMerDiskCounter::Count + KMerDiskStorage::merge (kmer_index_builder.hpp:306-332,
 * 190-203). A count that goes out of core (the sorted-unique set does not fit the HBM budget) streams every merged bucket range to its place in
 * `path` as the reference's merge does (:346-430, fwrite in 1 Mi-record chunks) instead of keeping the merged result in host memory next to the
 * spilled runs, and gives the consumed parts of the runs back as it goes: host memory holds the runs alone, shrinking. A result that stays
 * resident is written as smx_write_final_kmers writes it. Afterwards smx_count_info / smx_bucket_sizes answer; accessors of the RECORDS
 * (smx_copy_bucket, smx_copy_final_kmers, ...) refuse with SMX_INVALID_PARAMETER when they were streamed: they are in the file. */
int smx_count_to_file(smx_ctx *ctx, unsigned K, int mode, unsigned num_buckets, const char *path);
/* device view of the same bytes (valid until the next smx_count / smx_destroy) */
const void *smx_device_kmers(const smx_ctx *ctx);
/* device-to-device copy of the same array into caller-owned HBM (e.g. a torch tensor about to enter a collective) */
int smx_copy_kmers_device(const smx_ctx *ctx, void *d_dst);
/* ... and of ONE bucket (KMerDiskStorage::bucket(i), kmer_index_builder.hpp:180-190) into caller-owned HBM of smx_bucket_sizes()[bucket] records: what
 * a consumer that works bucket by bucket — the reference's index builder does, kmer_index_builder.hpp:470-500 — reads instead of the whole file;
 * a both-strands result that is held as two strands (smx_device_kmers() == NULL) is merged straight into the block, bucket by bucket. */
int smx_copy_bucket_device(const smx_ctx *ctx, unsigned bucket, void *d_dst);

/* ---- multi-GPU sharding ---------------------------------------------------
 * Bucket ownership is a contiguous bucket range per rank. smx_extract_partition runs the
 * extract + hash stage on the local read shard and groups the records by owner rank into a
 * caller-provided HBM buffer (the caller performs the RCCL all-to-all; the library never touches
 * the communicator). smx_count_records then sorts/uniques records the caller received. */
int smx_extract_count(smx_ctx *ctx, unsigned K, int mode, uint64_t *n_records);
int smx_extract_partition(smx_ctx *ctx, unsigned K, int mode, unsigned num_buckets, unsigned world,
                          void *d_records, uint64_t capacity_records, uint64_t *counts /* [world] */);
/* Same, into a buffer the library allocates itself, sized to what the local pre-dedupe leaves (one record per window instance — all
 * the caller could provide for without knowing the data — is 150 GB at 100 M PE150 reads): *d_records is valid until the next
 * smx_extract_partition* / smx_extract_release / smx_destroy. Release it before smx_count_records when HBM is short. */
int smx_extract_partition_owned(smx_ctx *ctx, unsigned K, int mode, unsigned num_buckets, unsigned world, const void **d_records,
                                uint64_t *counts /* [world] */);
int smx_extract_release(smx_ctx *ctx);
/* Receive side of the exchange in the library's HBM pool (what a framework allocator caches after the step is out of the pool's
 * reach): n_words 64-bit words, valid until smx_count_records is given exactly this pointer — which consumes it: the records are
 * sorted in place, one record buffer less than for caller memory — or smx_exchange_release / smx_destroy.
 * smx_extract_partition_owned drops the context's previous count result and graph (the next step needs their room). */
int smx_exchange_buffer(smx_ctx *ctx, uint64_t n_words, void **d_buf);
int smx_exchange_release(smx_ctx *ctx);
int smx_count_records(smx_ctx *ctx, unsigned K, unsigned num_buckets, const void *d_records,
                      uint64_t n_records);
/* first bucket owned by rank r of world (rank r owns [first(r), first(r+1))) */
unsigned smx_rank_first_bucket(unsigned num_buckets, unsigned world, unsigned rank);

/* ---- de Bruijn construction (spades-gbuilder path) ---------------------------------------------
 * smx_build_graph replaces, on the resident reads:
 *   DeBruijnExtensionIndexBuilder::BuildExtensionIndexFromStream (canonical (k+1)-mers -> k-mers -> in/out masks;
 *     common/kmer_index/extension_index/kmer_extension_index_builder.hpp:63-107),
 *   UnbranchingPathExtractor::ExtractUnbranchingPathsAndLoops
 *     (common/assembly_graph/construction/debruijn_graph_constructor.hpp:399-406),
 *   FastGraphFromSequencesConstructor::ConstructGraph (same file :506-567).
 * num_buckets = 10 * nthreads of the reference run being reproduced (kmer_extension_index_builder.hpp:75): the GFA
 * of spades-gbuilder depends on it (SURVEY.md finding 3). k odd, 1 <= k < 128 (projects/spades_tools/gbuilder.cpp:130-135).
 * After the call smx_copy_final_kmers()/smx_bucket_sizes() describe the canonical k-mer file (made at that moment when the
 * construction did not need it: option "pm_route"). Node ids (smx_host_write_graph's start_node / end_node) are opaque. */
int smx_build_graph(smx_ctx *ctx, unsigned k, unsigned num_buckets);
/* Drops the graph (and the HBM it holds: k-mers, masks, unitigs, link records) without destroying the context; the resident reads
 * stay. The reference frees its extension index and graph when their owners go out of scope (projects/spades_tools/gbuilder.cpp:232). */
int smx_graph_clear(smx_ctx *ctx);
/* Multi-GPU construction ("replicated lookup"; precedent: hpcspades/mpi/stages/construction_mpi.cpp:343-354,
 * count per node + merge): the same steps on a canonical (k+1)-mer file that the caller gathered from its owner ranks
 * (HBM pointer, any order, duplicates allowed) instead of on the resident reads. The resident reads are still what
 * smx_graph_fill_coverage counts; per-rank raw coverages add up (smx_graph_copy_coverage -> all-reduce SUM ->
 * smx_graph_set_coverage), because edge coverage is a sum of per-(k+1)-mer counts (coverage_filling.hpp:17-55). */
int smx_build_graph_from_records(smx_ctx *ctx, unsigned k, unsigned num_buckets, const void *d_kpomers, uint64_t n_records);
/* Sharded construction (owner-side mask fill; precedent: hpcSPAdes counts per node and OR-reduces the masks,
 * hpcspades/mpi/stages/construction_mpi.cpp:343-354, mpi/pipeline/partask_mpi.hpp:293). No rank holds the whole (k+1)-mer file:
 *   1. sharded count of the canonical (k+1)-mers (smx_extract_partition -> all-to-all -> smx_count_records): the rank's shard is the
 *      context's count result;
 *   2. smx_graph_shard_updates: the extension updates of the shard — one (canonical k-mer, InOutMask bit) record of words_per_kmer + 1
 *      uint64 per k-mer of every (k+1)-mer (DeBruijnKMerKMerSplitter + FillExtensionsFromIndex in one stream,
 *      kmer_mph/kmer_splitters.hpp:138-207, extension_index/kmer_extension_index_builder.hpp:45-60) — grouped by the rank that owns the
 *      k-mer's bucket, into a caller-provided HBM buffer of >= 2 x shard records; the caller runs the all-to-all;
 *   3. smx_graph_shard_build: the owner sorts/uniques the k-mers it received (its bucket range of the k-mer file) and ORs the bits
 *      into their InOutMask bytes; smx_graph_shard_info / smx_graph_shard_copy hand the shard out for the gather;
 *   4. smx_build_graph_from_kmers: every rank builds the graph from the gathered compact structure {k-mer file, mask bytes}
 *      (17 B per k-mer at k = 55): unitigs, loops, link records as in smx_build_graph. n_kpomers only feeds smx_graph_info.
 * smx_graph_set_kpomers installs a (k+1)-mer file (sorted, bucket-major) for smx_graph_fill_coverage on a graph built this way. */
int smx_graph_shard_updates(smx_ctx *ctx, unsigned k, unsigned num_buckets, unsigned world, void *d_updates, uint64_t capacity_records,
                            uint64_t *counts /* [world] */);
int smx_graph_shard_build(smx_ctx *ctx, unsigned k, unsigned num_buckets, unsigned world, unsigned rank, const void *d_updates,
                          uint64_t n_updates);
/* The same owner-side shard by ONE exchange, for k whose record has 8 spare bits (smx_kmers_with_masks_supported; the single-GPU
 * construction takes this route by itself): smx_extract_kmers_ext_owned hands out the canonical k-mers of this rank's reads (those
 * with a (k+1)-mer), each with the InOutMask byte ITS reads give it, as records of ceil(k/32) words whose last word is
 * (k-mer bits << 8 | byte), grouped by owner rank like smx_extract_partition_owned; after the all-to-all smx_graph_shard_from_ext
 * sorts what arrived, ORs the bytes of the copies of a k-mer and leaves the shard where smx_graph_shard_info / _copy find it
 * (records in the smx_exchange_buffer are consumed). smx_graph_shard_ext_stats: [0] extension bits set in the shard, [1] those
 * whose (k+1)-mer is its own reverse complement — the graph has (sum over ranks of [0] + [1]) / 2 canonical (k+1)-mers, the
 * n_kpomers of smx_build_graph_from_kmers. (kmer_extension_index_builder.hpp:45-60: a (k+1)-mer of the reads is the window behind
 * its prefix instance and before its suffix instance.) */
int smx_kmers_with_masks_supported(unsigned k);
int smx_extract_kmers_ext_owned(smx_ctx *ctx, unsigned k, unsigned num_buckets, unsigned world, const void **d_records, uint64_t *counts /* [world] */);
int smx_graph_shard_from_ext(smx_ctx *ctx, unsigned k, unsigned num_buckets, unsigned world, unsigned rank, const void *d_records,
                             uint64_t n_records);
int smx_graph_shard_ext_stats(const smx_ctx *ctx, uint64_t *stats /* [2] */);
/* Distributed walks: unitigs of a graph whose k-mer file stays sharded over the ranks, for graphs that do not
 * fit one GPU as a gathered {k-mer file, masks} structure. The shard is what smx_graph_shard_from_ext / smx_graph_shard_build left in
 * the context. A walk of the reference (UnbranchingPathExtractor::ConstructSequenceWithEdge,
 * assembly_graph/construction/debruijn_graph_constructor.hpp:264-273) would leave its rank at every step, so no rank walks:
 *   1. smx_shard_walk_counts / smx_shard_walk_requests(starts = 0): every oriented non-junction k-mer of the shard (a node: 2 * local
 *      rank + orientation) asks for its unique successor: canonical k-mer records of ceil(k/32) words grouped by the rank that owns
 *      the successor's bucket, counts[world]; d_tags (same order, stays here) = node << 4 | rc << 2 | nucleotide, rc = the successor
 *      node is the reverse complement of the stored k-mer. With starts = 1 the same for the start de-edges of the junction k-mers
 *      (AddStartDeEdges, :203-226) in k-mer-file order: tag = index << 4 | 8 | rc << 2 | nucleotide; smx_shard_walk_starts copies
 *      them out as (local rank << 3 | orientation << 2 | nucleotide).
 *   2. all-to-all; smx_shard_lookup on the owner: reply = local rank << 1 | (junction k-mer), all ones if the k-mer is not in the shard;
 *      replies go back in request order.
 *   3. the caller ranks the chains of non-junction k-mers by pointer doubling over the exchange (integers only, spades_amd/dist.py) and
 *      delivers, per start de-edge, the number of chain k-mers behind it, the node the walk ends at and the chain's nucleotides;
 *   4. smx_shard_unitigs assembles start (k+1)-mer + chain nucleotides, keeps a unitig iff !(s < !s) (:305-306) and leaves the kept
 *      ones — this rank's part of the edge list, in the reference's order — for smx_shard_unitigs_copy (2-bit packed, every unitig on
 *      a word boundary; start / end = global nodes, first_rank = global rank of the shard's first k-mer);
 *   5. smx_build_graph_from_unitigs on every rank that wants the graph: the ranks' unitigs concatenated in rank order, plus the k-mers
 *      that no chain reached (perfect loops, CollectLoops :359-397; smx_shard_gather_kmers fetches them by local rank; host arrays
 *      in k-mer-file order with GLOBAL ranks) -> link records, vertices, every writer and smx_graph_fill_coverage as after
 *      smx_build_graph. That graph has no k-mer file: smx_graph_copy_kmers / smx_graph_fingerprint refuse. */
int smx_shard_walk_counts(smx_ctx *ctx, uint64_t *n_chain_requests, uint64_t *n_start_requests);
/* Steps 1-4 behind ONE call (round 6): the caller brings its collectives, the library does everything else on the device — successor
 * lookups range by range, pointer doubling, chain nucleotides to the heads of their chains, the chains of this rank's start de-edges,
 * assembly. The library still never touches a communicator: spades-gbuilder-mi355x --gpus N hands in grouped ncclSend / ncclRecv
 * (tools/gbuilder_mgpu.hpp), spades_amd.dist torch.distributed; an MPI host would hand in MPI_Alltoallv. Role model in the reference tree:
 * hpcspades/mpi/stages/construction_mpi.cpp:303-412 (the construction's phases over processes, collectives between them).
 * Callbacks return 0 or non-zero (the call then fails with SMX_DEVICE_ERROR); they are called on the caller's thread, between kernels: the
 * library's stream is idle when one is entered, and the data a callback moved must be complete when it returns.
 *   exchange_counts: send_counts[world] of this rank -> recv_counts[p] = what rank p's send_counts[this rank] was (an all-to-all of one word
 *                    per pair; all ones in any entry = that rank has failed: every rank then leaves the call together);
 *   alltoallv:       segments of `unit_bytes`-byte elements between DEVICE buffers, grouped by rank in rank order on both sides
 *                    (send_counts / recv_counts in elements; the segment a rank keeps is copied like any other);
 *   allreduce_u64:   n host words reduced over all ranks in place, op 0 = sum, 1 = max.
 * kmers_per_rank[world]: k-mers of every rank's shard (smx_graph_shard_info on each, all-gathered). On success info[4] = this rank's kept
 * unitigs, their 2-bit words, its k-mers on perfect loops, doubling rounds; the unitigs wait for smx_shard_unitigs_copy, the loop k-mers'
 * local ranks (ascending) for smx_shard_walk_loops. Collective: every rank calls it; a rank-local failure reaches every rank at its next
 * collective and all return non-zero. Options "walk_chunk" / "walk_start_chunk" (oriented nodes / start de-edges per exchange round,
 * default 2^26 / 2^22) bound the temporaries; "walk_hop_bits" (24) is the width of a node's step counter. */
typedef struct smx_collectives {
    void *user;
    unsigned rank, world;
    int (*exchange_counts)(void *user, const uint64_t *send_counts, uint64_t *recv_counts);
    int (*alltoallv)(void *user, const void *d_send, const uint64_t *send_counts, void *d_recv, const uint64_t *recv_counts, unsigned unit_bytes);
    int (*allreduce_u64)(void *user, uint64_t *values, unsigned n, int op);
} smx_collectives;
int smx_shard_walks(smx_ctx *ctx, const uint64_t *kmers_per_rank /* [world] */, const smx_collectives *coll, uint64_t *info /* [4] */);
int smx_shard_walk_loops(const smx_ctx *ctx, uint64_t *d_local_ranks);
int smx_shard_walk_requests(smx_ctx *ctx, int starts, unsigned world, void *d_records, uint64_t *d_tags, uint64_t *counts /* [world] */);
/* ... the same for the items [first_item, first_item + n_items) only (oriented nodes 2 * local rank + orientation, or start de-edges in
 * k-mer-file order): at most one request per item, so buffers of n_items records / tags suffice — a caller that walks a shard of
 * billions of k-mers asks range by range and never holds more than one range's requests (16 + 8 B per oriented node otherwise). */
int smx_shard_walk_requests_range(smx_ctx *ctx, int starts, unsigned world, uint64_t first_item, uint64_t n_items, void *d_records, uint64_t *d_tags,
                                  uint64_t *counts /* [world] */);
int smx_shard_walk_starts(const smx_ctx *ctx, uint64_t *d_starts);
int smx_shard_lookup(smx_ctx *ctx, const void *d_records, uint64_t n, uint64_t *d_reply);
int smx_shard_gather_kmers(smx_ctx *ctx, const uint64_t *d_local_ranks, uint64_t n, void *d_kmers, uint8_t *d_masks);
int smx_shard_unitigs(smx_ctx *ctx, uint64_t first_rank, const uint64_t *d_steps, const uint64_t *d_last, const uint64_t *d_base_off /* [n + 1] */,
                      const uint8_t *d_bases, uint64_t *n_kept, uint64_t *n_words);
int smx_shard_unitigs_copy(const smx_ctx *ctx, uint64_t *d_words, uint64_t *d_len, uint64_t *d_start, uint64_t *d_end, uint8_t *d_self);
int smx_build_graph_from_unitigs(smx_ctx *ctx, unsigned k, unsigned num_buckets, uint64_t n_kmers, uint64_t n_kpomers, const uint64_t *d_words, uint64_t n_words,
                                 const uint64_t *d_len, const uint64_t *d_start, const uint64_t *d_end, const uint8_t *d_self, uint64_t n_unitigs,
                                 const uint64_t *loop_ranks, const uint64_t *loop_kmers, const uint8_t *loop_masks, uint64_t n_loop_kmers);
/* Fingerprint of the device-resident graph, for comparing two builds that are too big to leave the device: for each of the arrays
 * k-mer file, InOutMask bytes, packed unitig words, unitig lengths, start nodes, end nodes, sorted link records, vertex starts —
 * out[2i] = sum of the elements, out[2i+1] = sum of element * (2 * index + 1), both mod 2^64 (order-sensitive). */
int smx_graph_fingerprint(const smx_ctx *ctx, uint64_t *out /* [16] */);
/* The part of it that does not depend on how the k-mers are numbered (k-mer indices never reach the output,
 * debruijn_graph_constructor.hpp:540-547; the construction route "pm_route" numbers them by minimizer partition): packed unitig words,
 * unitig lengths, self-conjugate flags (same sums as above), and the link structure as the writers see it — out[6] = sum of the
 * EdgeAndMask words of all link records, out[7] = sum of word * (2 * (64 * vertex + place) + 1), vertices in id order, records of a
 * vertex in their sorted order. Equal between the construction routes on the same input (needs the link records on the device). */
int smx_graph_fingerprint_portable(const smx_ctx *ctx, uint64_t *out /* [8] */);
int smx_graph_shard_info(const smx_ctx *ctx, uint64_t *n_kmers, uint64_t *bucket_sizes /* [num_buckets] or NULL */);
int smx_graph_shard_copy(const smx_ctx *ctx, void *d_kmers, void *d_masks);
int smx_build_graph_from_kmers(smx_ctx *ctx, unsigned k, unsigned num_buckets, const void *d_kmers, const void *d_masks, uint64_t n_kmers,
                               const uint64_t *bucket_sizes /* [num_buckets] */, uint64_t n_kpomers);
int smx_graph_set_kpomers(smx_ctx *ctx, const void *d_kpomers, uint64_t n_kpomers, const uint64_t *bucket_sizes /* [num_buckets] */);
int smx_graph_set_coverage(smx_ctx *ctx, const uint32_t *raw_coverage, uint64_t n_unitigs);
/* info[8] = { #canonical (k+1)-mers, #canonical k-mers, #unitigs (incl. loops), #perfect loops, #vertices, #links
 * (valid after a GFA was written), total unitig nucleotides, words per k-mer } */
int smx_graph_info(const smx_ctx *ctx, uint64_t *info);
/* spades-core's early tip clipper (Construction phase "Early tip clipping", stages/construction.cpp:289-305;
 * EarlyTipClipperProcessor, assembly_graph/construction/early_simplification.hpp:38-162) runs inside smx_build_graph between the
 * extension masks and the unitigs when the option "early_tip_bound" is > 0 (the reference uses RL - K). spades-gbuilder never runs it.
 * Option "early_at_remover" = 1 runs the early A/T remover of the RNA pipelines before it (EarlyLowComplexityClipperProcessor(index,
 * 0.8, 10, 200).RemoveATEdges() + RemoveATTips(), early_simplification.hpp:164-347, stages/construction.cpp:317-326, 446-448).
 * stats[4] = { k-mers isolated by the tip clipper, tips removed, length-1 A/T edges removed (counted from both ends), k-mers
 * isolated by the A/T tip remover } of the last build. */
int smx_graph_tip_stats(const smx_ctx *ctx, uint64_t *stats);
/* How the last smx_build_graph went (engine diagnostics, no reference equivalent): stats[8] = { route: 0 no sort of the k-mers
 * ("pm_route"), 1 k-mers + masks from one count, sorted ("ext_route"), 2 (k+1)-mer file first; route 0: k-mers in chunks of whole
 * minimizer partitions, k-mers of cut partitions (sorted tail), chunks; junction k-mers; start de-edges; route 0: super-k-mer slots of
 * the count behind the graph; k-mer instances in super-k-mers that were folded away as identical copies }. */
int smx_graph_route_stats(const smx_ctx *ctx, uint64_t *stats /* [8] */);
/* k-mer file order: records [n_kmers * words] and InOutMask bytes (extension_index/inout_mask.hpp:55-221) */
int smx_graph_copy_kmers(const smx_ctx *ctx, void *kmers_host, uint8_t *masks_host);
/* unitigs in the reference's enumeration order: offsets [n_unitigs+1], ACGT bytes */
int smx_graph_copy_unitigs(const smx_ctx *ctx, uint64_t *offsets, char *seq);
/* -c of spades-gbuilder: second pass over the resident reads. Replaces CoverageHashMapBuilder::BuildIndex
 * (common/kmer_index/ph_map/coverage_hash_map_builder.hpp:16-57) + FillCoverageAndFlankingFromPHM
 * (common/assembly_graph/graph_support/coverage_filling.hpp:17-96): per-(k+1)-mer uint32 multiplicities over the read+RC
 * stream, summed per edge. Afterwards smx_graph_write_gfa emits DP:f:<raw/len>  KC:i:<raw> instead of zeros. */
/* Reads submitted while the option "submit_contigs" is 1 (trusted contigs, contigs of the previous k) take part in the construction
 * but are not counted here — "Has to be separate stream for not counting it in coverage" (stages/construction.cpp:108-117). */
int smx_graph_fill_coverage(smx_ctx *ctx);
int smx_graph_copy_coverage(const smx_ctx *ctx, uint32_t *raw_coverage /* [n_unitigs] */);
/* Multiplicity histogram of the canonical (k+1)-mers after smx_graph_fill_coverage: hist[c] = how many have been seen c times
 * (hist[0] = those met only in contig streams). What PHMCoverageFiller gives to GenomicInfo::set_cov_histogram
 * (stages/construction.cpp:414-431; there hist[c-1] += 2 per canonical record). *n_entries = largest multiplicity + 1. */
int smx_graph_coverage_histogram(const smx_ctx *ctx, uint64_t *hist, uint64_t capacity, uint64_t *n_entries);

/* Flanking raw coverage, filled by the same pass (FillCoverageAndFlankingFromPHM, graph_support/coverage_filling.hpp:40-44,89-96;
 * omnigraph::FlankingCoverage, detail_coverage.hpp:22-100): sum of the counters of the first `flank_range` (option, default 50 as in
 * spades-core) (k+1)-mers of every canonical edge, and of its conjugate (= the last ones). Values restated from the reference source,
 * not pinned by a reference run (spades-gbuilder does not compute them). */
int smx_graph_copy_flanking(const smx_ctx *ctx, uint32_t *flank_edge /* [n_unitigs] */, uint32_t *flank_conjugate /* [n_unitigs] */);
/* gfa::GFAWriter::WriteSegmentsAndLinks (common/io/graph/gfa_writer.cpp); flavour_version fills "H\tsp:Z:<..>" */
int smx_graph_write_gfa(smx_ctx *ctx, const char *path, const char *flavour_version);
/* gbuilder --fastg (gbuilder.cpp:226-228): io::FastgWriter::WriteSegmentsAndLinks (common/io/graph/fastg_writer.cpp:21-48) */
int smx_graph_write_fastg(smx_ctx *ctx, const char *path);
/* gbuilder --spades (gbuilder.cpp:229-230): io::binary::BasicGraphIO::Save -> <basename>.grseq (common/io/binary/graph.hpp:27-74)
 * + <basename>.cvr (common/io/binary/coverage.hpp:23-29) — the graph_pack surface spades-core loads. */
int smx_graph_write_spades(smx_ctx *ctx, const char *basename);
/* gbuilder --unitigs (gbuilder.cpp:191-200): >EDGE_<i>_length_<len>, wrapped at 60 */
int smx_graph_write_unitigs(smx_ctx *ctx, const char *path);

/* Host-only (no GPU, no context): FastGraphFromSequencesConstructor::ConstructGraph + the writers on caller-provided unitigs.
 * offsets[n+1]/seq: ACGT unitigs in id order; start_node/end_node = 2*rank + rc of the first / last k-mer, rank being ANY injective
 * id of the canonical k-mer (the reference uses the MPHF index only to group records: debruijn_graph_constructor.hpp:540-547);
 * raw_coverage may be NULL; sort_edges as the option of the same name; format 0 unitig FASTA, 1 GFA, 2 FASTG, 3 .grseq+.cvr. */
int smx_host_write_graph(unsigned k, uint64_t n_edges, const uint64_t *offsets, const char *seq, const uint64_t *start_node,
                         const uint64_t *end_node, const uint32_t *raw_coverage, int sort_edges, int format, const char *path,
                         const char *flavour_version);

/* ---- instrumentation -----------------------------------------------------------------------
 * Per-stage GPU time of the last smx_count in milliseconds (HIP events on the library's stream).
 * names/ms arrays of capacity cap; returns number of stages. Stands where the reference has
 * TIME_TRACE_SCOPE("KMerDiskCounter::Split"/"::Count") (kmer_index_builder.hpp:309-324). */
int smx_last_timings(const smx_ctx *ctx, const char **names, float *ms, int cap);

#ifdef __cplusplus
}
#endif
#endif
