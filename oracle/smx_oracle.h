/* oracle/smx_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference algorithm for SPAdes' k-mer counting /
 * de Bruijn construction hot path. It exists so that tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg can check the HIP product path; NOTHING under spades_amd/
 * may include, link or call it.
 *
 * Parity status: PINNED. The restatement is checked (tests/test_oracle_*.py) against
 *   - the reference's own worked byte example (kmercount.cpp:160-170),
 *   - golden final_kmers / bucket-size fixtures produced by the REAL reference classes
 *     (oracle/_ref/ref_kmercount, built by oracle/ref_recipe/Makefile from the sources where
 *     they lie under /root/reference) and by the reference binaries spades-kmercount /
 *     spades-gbuilder (tests/golden/, generator script tests/golden/make_golden.py),
 *   - the six k=5 known-answer graph tests of src/test/debruijn/construction_test.cpp:30-64.
 */
#ifndef SMX_ORACLE_H
#define SMX_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_WORDS 4 /* RtSeq = RuntimeSeq<128,uint64_t>: rtseq.hpp:768, seq_common.hpp:19-43 */

/* ---- k-mer value type (rtseq.hpp) ---- */
unsigned orc_words(unsigned K);                                   /* GetDataSize, rtseq.hpp:131-133 */
void orc_from_string(uint64_t *w, unsigned K, const char *s);    /* init, rtseq.hpp:166-187 */
void orc_to_string(const uint64_t *w, unsigned K, char *s);      /* str, rtseq.hpp:629-635 */
void orc_shl(uint64_t *w, unsigned K, unsigned c);               /* operator<<=, rtseq.hpp:459-476 */
void orc_rc(const uint64_t *w, unsigned K, uint64_t *out);       /* FastRC, rtseq.hpp:81-117 */
int  orc_is_minimal(const uint64_t *w, unsigned K);              /* IsMinimal, rtseq.hpp:409-417 */
int  orc_less_nucl(const uint64_t *a, const uint64_t *b, unsigned K); /* operator<, rtseq.hpp:742-750 */
uint64_t orc_xxh3_64(const void *data, size_t len);             /* XXH3_64bits_withSeed(.,.,0), len in {8,16,24,32} */
uint64_t orc_bucket(const uint64_t *w, unsigned K, uint64_t num_buckets); /* kmer_buckets.hpp:47-52 */

/* ---- read preprocessing (longest_valid_wrapper.hpp:16-43) ---- */
void orc_longest_valid(const char *s, size_t n, size_t *from, size_t *to);

/* ---- counting (kmercount.cpp:65-83 mode 'A'; kmer_splitters.hpp:28-44 mode 'B') ----
 * reads: concatenated ASCII, read i = bases[off[i] .. off[i+1]).
 * Returns number of distinct records; *out is malloc'ed (caller frees with orc_free):
 * buckets 0..B-1 concatenated, strictly increasing (w0,w1,..) inside a bucket
 * (kmer_splitter.hpp:140-141, kmer_index_builder.hpp:346-430,190-203). bucket_sizes[B]. */
int64_t orc_count(char mode, unsigned K, unsigned num_buckets,
                  const char *bases, const uint64_t *off, uint64_t nreads,
                  uint64_t **out, uint64_t *bucket_sizes);
void orc_free(void *p);

/* ---- graph construction (spades-gbuilder path; rows a13-a19) ----
 * reads -> canonical (k+1)-mers (B = num_buckets = 10*threads) -> canonical k-mers in k-mer-file order
 * (kmer_extension_index_builder.hpp:83-107) -> in/out masks (:45-60, inout_mask.hpp:117-131) ->
 * unbranching paths + perfect loops in the reference's enumeration order
 * (debruijn_graph_constructor.hpp:184-410) -> edge/vertex ids + links (:412-568) -> GFA text
 * (io/graph/gfa_writer.cpp:19-47,73-87,113-116). k must be odd (gbuilder.cpp:134-135). */
typedef struct {
    uint64_t n_kpomers, n_kmers;   /* distinct canonical (k+1)-mers / k-mers */
    uint64_t *kmers;               /* [n_kmers * words(k)] k-mer-file order */
    uint8_t *masks;                /* [n_kmers] InOutMask bytes BEFORE RemoveSequences */
    uint64_t n_unitigs, n_loops;   /* loops are the last n_loops unitigs */
    uint64_t *unitig_off;          /* [n_unitigs+1] offsets into unitig_seq */
    char *unitig_seq;              /* ACGT, concatenated */
    uint64_t n_vertices, n_links;
    char *gfa;                     /* NUL-terminated GFA1 text */
    uint64_t gfa_len;
} orc_graph;
orc_graph *orc_build_graph(unsigned k, unsigned num_buckets, const char *bases, const uint64_t *off,
                           uint64_t nreads, const char *flavour_version /* e.g. "SPAdes-4.3.0-dev" */);
/* with_cov != 0: spades-gbuilder -c (coverage_hash_map_builder.hpp:18-39, coverage_filling.hpp:46-62): DP:f / KC:i tags */
orc_graph *orc_build_graph_cov(unsigned k, unsigned num_buckets, const char *bases, const uint64_t *off,
                               uint64_t nreads, const char *flavour_version, int with_cov);
/* spades-core variant (DeBruijnGraphExtentionConstructor, debruijn_graph_constructor.hpp:590-604): sort_edges = unitigs
 * ordered by Sequence::RawCompare before ids; keep_loops = keep_perfect_loops. PARITY UNPINNED for this variant: no
 * spades-core build is available to produce a golden; the ordering rule itself is restated from sequence.hpp:605-624. */
/* spades-core only: EarlyTipClipperProcessor(index, bound).ClipTips() between the extension index and the unitigs
 * (early_simplification.hpp:38-162, stages/construction.cpp:289-305); 0 = off. Applies to the following orc_build_graph_* calls. */
void orc_set_early_tip_bound(uint64_t bound);
/* RNA pipelines only: EarlyLowComplexityClipperProcessor(index, 0.8, 10, 200).RemoveATEdges() + RemoveATTips() before the tip clipper
 * (early_simplification.hpp:164-347, stages/construction.cpp:317-326, 446-448) */
void orc_set_early_at_remover(int on);
/* coverage from the first n reads only (0 = all): contigs of a previous k / trusted contigs shape the graph but are not counted
 * (stages/construction.cpp:108-117) */
void orc_set_coverage_reads(uint64_t n);
orc_graph *orc_build_graph_ex(unsigned k, unsigned num_buckets, const char *bases, const uint64_t *off, uint64_t nreads,
                              const char *flavour_version, int with_cov, int sort_edges, int keep_loops);
void orc_graph_free(orc_graph *g);

#ifdef __cplusplus
}
#endif
#endif
