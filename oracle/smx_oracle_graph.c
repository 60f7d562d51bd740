/* oracle/smx_oracle_graph.c — TEST INFRASTRUCTURE ONLY (see smx_oracle.h).
 * Plain-C restatement of the construction path of spades-gbuilder: extension index, unbranching
 * paths + perfect loops, graph ids/links, GFA text. Scalar, sorted-array lookups instead of the MPHF
 * (MPHF values never reach the output: debruijn_graph_constructor.hpp:540-547). */
#include "smx_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- small helpers */
typedef struct { unsigned k, nw, nb; uint64_t n; uint64_t *recs; uint64_t *boff; uint8_t *mask; } kidx;

static unsigned nucl_at(const uint64_t *w, unsigned i) { return (unsigned)((w[i >> 5] >> ((i & 31) << 1)) & 3); }
static int rec_cmp(const uint64_t *a, const uint64_t *b, unsigned nw) {
    for (unsigned i = 0; i < nw; ++i) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
}
static int rec_eq(const uint64_t *a, const uint64_t *b, unsigned nw) { return rec_cmp(a, b, nw) == 0; }

/* position of a CANONICAL k-mer in the k-mer file (stands in for KMerIndex::seq_idx, kmer_index.hpp:88-100) */
static int64_t kidx_find(const kidx *ix, const uint64_t *canon) {
    uint64_t b = orc_bucket(canon, ix->k, ix->nb);
    uint64_t lo = ix->boff[b], hi = ix->boff[b + 1];
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        int c = rec_cmp(ix->recs + mid * ix->nw, canon, ix->nw);
        if (c == 0) return (int64_t)mid;
        if (c < 0) lo = mid + 1; else hi = mid;
    }
    return -1;
}

/* KeyWithHash: oriented k-mer + whether it is the stored (minimal) orientation (key_with_hash.hpp:109-208) */
typedef struct { uint64_t w[ORC_MAX_WORDS]; int64_t idx; int minimal; } kwh;
static kwh make_kwh(const kidx *ix, const uint64_t *w) {
    kwh r;
    memset(&r, 0, sizeof r);
    for (unsigned i = 0; i < ix->nw; ++i) r.w[i] = w[i];
    r.minimal = orc_is_minimal(r.w, ix->k);
    if (r.minimal) r.idx = kidx_find(ix, r.w);
    else { uint64_t rc[ORC_MAX_WORDS]; orc_rc(r.w, ix->k, rc); r.idx = kidx_find(ix, rc); }
    return r;
}
static uint8_t invert_byte(uint8_t a) { /* inout_mask.hpp:18-39 */
    uint8_t r = 0;
    for (int i = 0; i < 8; ++i) { r = (uint8_t)((r << 1) | (a & 1)); a >>= 1; }
    return r;
}
/* get_value for the key's orientation: InvertableStoring::get_value (storing_traits.hpp) + InOutMask::conjugate */
static uint8_t get_mask(const kidx *ix, const kwh *h) {
    uint8_t m = ix->mask[h->idx];
    return h->minimal ? m : invert_byte(m);
}
static int uniq4(unsigned m) { return m == 1 || m == 2 || m == 4 || m == 8; }
static unsigned uniq_nucl(unsigned m) { return m == 1 ? 0 : m == 2 ? 1 : m == 4 ? 2 : 3; }
static int is_junction(uint8_t m) { return !uniq4(m & 15) || !uniq4((m >> 4) & 15); } /* inout_mask.hpp:157-159 */
static kwh kwh_shl(const kidx *ix, const kwh *h, unsigned c) { /* GetOutgoing: kwh << nucl */
    uint64_t w[ORC_MAX_WORDS] = {0, 0, 0, 0};
    for (unsigned i = 0; i < ix->nw; ++i) w[i] = h->w[i];
    orc_shl(w, ix->k, c);
    return make_kwh(ix, w);
}
static kwh kwh_rc(const kidx *ix, const kwh *h) {
    uint64_t w[ORC_MAX_WORDS];
    orc_rc(h->w, ix->k, w);
    return make_kwh(ix, w);
}

/* growable byte sequences (codes 0..3) */
typedef struct { unsigned char *d; size_t n, cap; } bseq;
static void bs_push(bseq *s, unsigned c) {
    if (s->n == s->cap) { s->cap = s->cap ? s->cap * 2 : 256; s->d = (unsigned char *)realloc(s->d, s->cap); }
    s->d[s->n++] = (unsigned char)c;
}
typedef struct { unsigned char **seq; size_t *len; size_t n, cap; } seqlist;
static void sl_push(seqlist *l, const unsigned char *d, size_t n) {
    if (l->n == l->cap) {
        l->cap = l->cap ? l->cap * 2 : 1024;
        l->seq = (unsigned char **)realloc(l->seq, l->cap * sizeof *l->seq);
        l->len = (size_t *)realloc(l->len, l->cap * sizeof *l->len);
    }
    l->seq[l->n] = (unsigned char *)malloc(n ? n : 1);
    memcpy(l->seq[l->n], d, n);
    l->len[l->n++] = n;
}
static void seq_rc(const unsigned char *s, size_t n, unsigned char *out) { for (size_t i = 0; i < n; ++i) out[i] = (unsigned char)(3 - s[n - 1 - i]); }
/* Sequence::operator<, sequence/sequence.hpp:592-600 (equal lengths here) */
static int seq_less(const unsigned char *a, const unsigned char *b, size_t n) {
    for (size_t i = 0; i < n; ++i) if (a[i] != b[i]) return a[i] < b[i];
    return 0;
}
static void kmer_from_codes(const unsigned char *s, unsigned k, uint64_t *w) {
    for (unsigned i = 0; i < ORC_MAX_WORDS; ++i) w[i] = 0;
    for (unsigned i = 0; i < k; ++i) w[i >> 5] |= (uint64_t)s[i] << ((i & 31) << 1);
}

/* ---------------------------------------------------------------- unitigs */
typedef struct { kwh start, end; } deedge;
static int de_eq(const kidx *ix, const deedge *a, const deedge *b) { return rec_eq(a->start.w, b->start.w, ix->nw) && rec_eq(a->end.w, b->end.w, ix->nw); }

/* StepRightIfPossible(DeEdge&), debruijn_graph_constructor.hpp:237-245 */
static int step_right_edge(const kidx *ix, deedge *e) {
    uint8_t m = get_mask(ix, &e->end);
    if (uniq4(m & 15) && uniq4((m >> 4) & 15)) {
        kwh nx = kwh_shl(ix, &e->end, uniq_nucl(m & 15));
        e->start = e->end;
        e->end = nx;
        return 1;
    }
    return 0;
}
/* StepRightIfPossible(KeyWithHash&), :228-235 */
static int step_right_kwh(const kidx *ix, kwh *h) {
    uint8_t m = get_mask(ix, h);
    if (uniq4(m & 15) && uniq4((m >> 4) & 15)) { *h = kwh_shl(ix, h, uniq_nucl(m & 15)); return 1; }
    return 0;
}
/* ConstructSequenceWithEdge, :264-273 */
static void construct_sequence(const kidx *ix, deedge edge, bseq *b) {
    b->n = 0;
    for (unsigned i = 0; i < ix->k; ++i) bs_push(b, nucl_at(edge.start.w, i));
    bs_push(b, nucl_at(edge.end.w, ix->k - 1));
    deedge initial = edge;
    while (step_right_edge(ix, &edge) && !de_eq(ix, &edge, &initial)) bs_push(b, nucl_at(edge.end.w, ix->k - 1));
}
/* IsolateVertex along a sequence: RemoveSequence, kmer_extension_index.hpp:131-139 */
static void remove_sequence(kidx *ix, const unsigned char *s, size_t n) {
    uint64_t w[ORC_MAX_WORDS];
    kmer_from_codes(s, ix->k, w);
    kwh h = make_kwh(ix, w);
    ix->mask[h.idx] = 0;
    for (size_t pos = ix->k; pos < n; ++pos) { h = kwh_shl(ix, &h, s[pos]); ix->mask[h.idx] = 0; }
}

/* ---------------------------------------------------------------- early tip clipper (spades-core only)
 * EarlyTipClipperProcessor, assembly_graph/construction/early_simplification.hpp:38-162, run by the Construction stage between the
 * extension index and the condensation (stages/construction.cpp:289-305) with length_bound = RL - K. Restated sequentially in
 * k-mer-file order (= the reference with one thread). */
typedef struct { int64_t *idx; size_t n, cap; } tiplist;
static void tl_push(tiplist *t, int64_t i) {
    if (t->n == t->cap) { t->cap = t->cap ? t->cap * 2 : 64; t->idx = (int64_t *)realloc(t->idx, t->cap * sizeof *t->idx); }
    t->idx[t->n++] = i;
}
/* FindForward, :102-112: from the second k-mer of a would-be tip; leaves the list empty unless it ends in a dead end within bound */
static void tip_find_forward(const kidx *ix, kwh kh, size_t bound, tiplist *tip) {
    tip->n = 0;
    for (;;) {
        uint8_t m = get_mask(ix, &kh);
        if (!(tip->n < bound && uniq4((m >> 4) & 15) && uniq4(m & 15))) break;
        tl_push(tip, kh.idx);
        kh = kwh_shl(ix, &kh, uniq_nucl(m & 15));
    }
    tl_push(tip, kh.idx);
    uint8_t m = get_mask(ix, &kh);
    if (!uniq4((m >> 4) & 15) || (m & 15) != 0) tip->n = 0; /* branching or too long */
}
/* RemoveForward + RemoveTips, :114-146 */
static size_t tip_remove_forward(kidx *ix, const kwh *kh, size_t bound, tiplist tips[4]) {
    size_t max = 0, removed = 0;
    uint8_t m = get_mask(ix, kh);
    for (unsigned c = 0; c < 4; ++c) {
        tips[c].n = 0;
        if (m & (1u << c)) {
            kwh khc = kwh_shl(ix, kh, c);
            tip_find_forward(ix, khc, bound, &tips[c]);
            size_t len = tips[c].n ? tips[c].n : (size_t)-1;
            if (len > max) max = len;
        }
    }
    for (unsigned c = 0; c < 4; ++c)
        if (tips[c].n < max) {
            for (size_t i = 0; i < tips[c].n; ++i) ix->mask[tips[c].idx[i]] = 0; /* IsolateVertex */
            removed += tips[c].n;
        }
    return removed;
}
/* RemoveInconsistentForwardLinks, :21-36 */
static size_t tip_remove_inconsistent(kidx *ix, const kwh *kh) {
    size_t count = 0;
    uint8_t m = get_mask(ix, kh);
    const unsigned first = nucl_at(kh->w, 0);
    for (unsigned c = 0; c < 4; ++c) {
        if (!(m & (1u << c))) continue;
        kwh nx = kwh_shl(ix, kh, c);
        uint8_t mn = get_mask(ix, &nx);
        if (!(mn & (1u << (4 + first)))) {
            ix->mask[kh->idx] &= (uint8_t)~(1u << (kh->minimal ? c : 7 - c)); /* DeleteOutgoing, inout_mask.hpp:133-139 */
            ++count;
        }
    }
    return count;
}
/* ClipTips, :52-99; returns the number of isolated k-mers */
static size_t early_tip_clip(kidx *ix, size_t bound) {
    tiplist tips[4] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    kwh *tipped = NULL; size_t nt = 0, capt = 0, removed = 0;
    for (uint64_t r = 0; r < ix->n; ++r) {
        kwh side[2];
        side[0] = make_kwh(ix, ix->recs + r * ix->nw);
        side[1] = kwh_rc(ix, &side[0]);
        for (int sd = 0; sd < 2; ++sd) {
            uint8_t m = get_mask(ix, &side[sd]);
            if (__builtin_popcount(m & 15) < 2) continue;
            size_t rm = tip_remove_forward(ix, &side[sd], bound, tips);
            removed += rm;
            if (rm) {
                if (nt == capt) { capt = capt ? capt * 2 : 256; tipped = (kwh *)realloc(tipped, capt * sizeof *tipped); }
                tipped[nt++] = side[sd];
            }
        }
    }
    for (size_t i = 0; i < nt; ++i) tip_remove_inconsistent(ix, &tipped[i]);
    for (unsigned c = 0; c < 4; ++c) free(tips[c].idx);
    free(tipped);
    return removed;
}
/* ---------------------------------------------------------------- early A/T remover (RNA pipelines only)
 * EarlyLowComplexityClipperProcessor(index, 0.8, 10, 200), early_simplification.hpp:164-347; stages/construction.cpp:317-326.
 * math::ls(a, b) = !AlmostEquals(a, b) && a < b with the 4-ULP AlmostEquals of math/xmath.h:284-312. */
static int dbl_almost_eq(double a, double b) {
    int64_t x, y;
    memcpy(&x, &a, 8); memcpy(&y, &b, 8);
    if (x < 0) x = (int64_t)0x8000000000000000ull - x;
    if (y < 0) y = (int64_t)0x8000000000000000ull - y;
    int64_t d = x > y ? x - y : y - x;
    return d <= 4;
}
static int dbl_ls(double a, double b) { return !dbl_almost_eq(a, b) && a < b; }
static kwh kwh_shr(const kidx *ix, const kwh *h, unsigned c) { /* GetIncoming: kwh >> nucl = c + kmer[0..k-2] */
    uint64_t w[ORC_MAX_WORDS] = {0, 0, 0, 0};
    w[0] = c;
    for (unsigned i = 0; i + 1 < ix->k; ++i) w[(i + 1) >> 5] |= (uint64_t)nucl_at(h->w, i) << (((i + 1) & 31) << 1);
    return make_kwh(ix, w);
}
static void mask_del_out(kidx *ix, const kwh *h, unsigned c) { ix->mask[h->idx] &= (uint8_t)~(1u << (h->minimal ? c : 7 - c)); }
static void mask_del_in(kidx *ix, const kwh *h, unsigned c) { ix->mask[h->idx] &= (uint8_t)~(1u << (h->minimal ? c + 4 : 7 - (c + 4))); }
/* RemoveATEdges, :176-259 */
static size_t at_remove_edges(kidx *ix, double ratio) {
    typedef struct { kwh h; unsigned c; } atedge;
    atedge *e = NULL; size_t ne = 0, cap = 0;
    const double thr = ix->k * ratio;
    for (uint64_t r = 0; r < ix->n; ++r) {
        kwh side[2];
        side[0] = make_kwh(ix, ix->recs + r * ix->nw);
        side[1] = kwh_rc(ix, &side[0]);
        for (int sd = 0; sd < 2; ++sd) {
            uint8_t m = get_mask(ix, &side[sd]);
            if (!is_junction(m)) continue;
            size_t counts[4] = {0, 0, 0, 0}, curm = 0;
            for (unsigned i = 0; i < ix->k; ++i) counts[nucl_at(side[sd].w, i)]++;
            for (int t = 0; t < 4; ++t) if (counts[t] > curm) curm = counts[t];
            if (dbl_ls((double)curm, thr)) continue;
            for (unsigned c = 0; c < 4; ++c) {
                if (!(m & (1u << c))) continue;
                kwh nx = kwh_shl(ix, &side[sd], c);
                uint8_t mn = get_mask(ix, &nx);
                if (!is_junction(mn) && (mn & 15) != 0) continue; /* edge of length 1: next is a junction or a dead end */
                if (ne == cap) { cap = cap ? cap * 2 : 256; e = (atedge *)realloc(e, cap * sizeof *e); }
                e[ne].h = side[sd]; e[ne].c = c; ++ne;
            }
        }
    }
    for (size_t i = 0; i < ne; ++i) {
        if (!(get_mask(ix, &e[i].h) & (1u << e[i].c))) continue;
        kwh nx = kwh_shl(ix, &e[i].h, e[i].c);
        mask_del_out(ix, &e[i].h, e[i].c);
        mask_del_in(ix, &nx, nucl_at(e[i].h.w, 0));
    }
    free(e);
    return ne;
}
/* RemoveATTips, :262-338 */
static size_t at_remove_tips(kidx *ix, double ratio, size_t min_len, size_t max_len) {
    kwh *roots = NULL; size_t nr = 0, capr = 0, removed = 0;
    int64_t *tip = (int64_t *)malloc((max_len + 1) * sizeof *tip);
    for (uint64_t r = 0; r < ix->n; ++r) {
        kwh side[2];
        side[0] = make_kwh(ix, ix->recs + r * ix->nw);
        side[1] = kwh_rc(ix, &side[0]);
        for (int sd = 0; sd < 2; ++sd) {
            kwh kh = side[sd];
            uint8_t m = get_mask(ix, &kh);
            if ((m & 15) != 0 || !uniq4((m >> 4) & 15)) continue; /* start from tip ends */
            size_t counts[4] = {0, 0, 0, 0}, n = 0;
            do {
                tip[n++] = kh.idx;
                counts[nucl_at(kh.w, ix->k - 1)]++;
                uint8_t mm = get_mask(ix, &kh);
                kh = kwh_shr(ix, &kh, uniq_nucl((mm >> 4) & 15));
            } while (n < max_len && !is_junction(get_mask(ix, &kh)));
            uint8_t mr = get_mask(ix, &kh);
            if (((mr >> 4) & 15) == 0 || !is_junction(mr)) continue; /* dead start, or the tip is too long */
            for (size_t i = n - 1; i < min_len; ++i) counts[nucl_at(kh.w, ix->k - 1 - (unsigned)i)]++;
            size_t curm = 0;
            for (int t = 0; t < 4; ++t) if (counts[t] > curm) curm = counts[t];
            const double thr = (double)(n > min_len ? n : min_len) * ratio;
            if (dbl_ls((double)curm, thr)) continue;
            if (nr == capr) { capr = capr ? capr * 2 : 256; roots = (kwh *)realloc(roots, capr * sizeof *roots); }
            roots[nr++] = kh;
            removed += n;
            for (size_t i = 0; i < n; ++i) ix->mask[tip[i]] = 0;
        }
    }
    for (size_t i = 0; i < nr; ++i) tip_remove_inconsistent(ix, &roots[i]);
    free(roots);
    free(tip);
    return removed;
}
static int g_early_at = 0;
void orc_set_early_at_remover(int on) { g_early_at = on; }
/* spades-core: trusted / previous-k contigs take part in the construction but are "separate streams for not counting it in
 * coverage" (stages/construction.cpp:108-117, 371-435): only the first g_cov_reads reads are counted (0 = all of them) */
static uint64_t g_cov_reads = 0;
void orc_set_coverage_reads(uint64_t n) { g_cov_reads = n; }

/* 0 = off (spades-gbuilder); set by orc_set_early_tip_bound before orc_build_graph_* for the spades-core variant */
static size_t g_early_tip_bound = 0;
void orc_set_early_tip_bound(uint64_t bound) { g_early_tip_bound = (size_t)bound; }

/* ---------------------------------------------------------------- link records */
typedef struct { uint64_t hash_and_mask; uint64_t edge; } linkrec; /* debruijn_graph_constructor.hpp:422-454 */
static uint64_t lr_hash(const linkrec *r) { return r->hash_and_mask >> 2; }
static uint64_t lr_eam(const linkrec *r) { return (r->edge << 2) | (r->hash_and_mask & 3); }
static int lr_invalid(const linkrec *r) { return r->hash_and_mask + 1 == 0 && r->edge == 0; }
static int lr_cmp(const void *a, const void *b) {
    const linkrec *x = (const linkrec *)a, *y = (const linkrec *)b;
    if (lr_hash(x) != lr_hash(y)) return lr_hash(x) < lr_hash(y) ? -1 : 1;
    if (lr_eam(x) != lr_eam(y)) return lr_eam(x) < lr_eam(y) ? -1 : 1;
    return 0;
}
static const linkrec *g_recs;
static int vtx_cmp(const void *a, const void *b) {
    uint64_t x = lr_eam(&g_recs[*(const size_t *)a]), y = lr_eam(&g_recs[*(const size_t *)b]);
    return x < y ? -1 : x > y;
}
static int u64_cmp(const void *a, const void *b) { uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b; return x < y ? -1 : x > y; }

typedef struct { char *d; size_t n, cap; } sbuf;
static void sb_add(sbuf *b, const char *s, size_t n) {
    if (b->n + n + 1 > b->cap) { while (b->n + n + 1 > b->cap) b->cap = b->cap ? b->cap * 2 : 4096; b->d = (char *)realloc(b->d, b->cap); }
    memcpy(b->d + b->n, s, n);
    b->n += n;
    b->d[b->n] = 0;
}
static void sb_printf_u64(sbuf *b, uint64_t v) { char t[32]; int n = snprintf(t, sizeof t, "%llu", (unsigned long long)v); sb_add(b, t, (size_t)n); }

/* ---------------------------------------------------------------- the whole construction */
/* ASCII -> code for the coverage pass (sequence/nucl.hpp dignucl) */
static unsigned dig(char c) {
    if (c >= 'a' && c <= 't') c = (char)(c - 'a' + 'A');
    return c <= 'C' ? (c == 'A' ? 0u : 1u) : (c == 'G' ? 2u : 3u);
}

/* Sequence::RawCompare (sequence/sequence.hpp:605-624): length, then packed 64-bit words from word 0 */
static const seqlist *g_sl;
static int rawcmp_idx(const void *a, const void *b) {
    size_t i = *(const size_t *)a, j = *(const size_t *)b;
    if (g_sl->len[i] != g_sl->len[j]) return g_sl->len[i] < g_sl->len[j] ? -1 : 1;
    size_t n = g_sl->len[i];
    for (size_t w0 = 0; w0 < n; w0 += 32) {
        uint64_t x = 0, y = 0;
        for (size_t t = w0; t < n && t < w0 + 32; ++t) {
            x |= (uint64_t)g_sl->seq[i][t] << ((t & 31) << 1);
            y |= (uint64_t)g_sl->seq[j][t] << ((t & 31) << 1);
        }
        if (x != y) return x < y ? -1 : 1;
    }
    return 0;
}

orc_graph *orc_build_graph_ex(unsigned k, unsigned num_buckets, const char *bases, const uint64_t *off, uint64_t nreads,
                              const char *flavour_version, int with_cov, int sort_edges, int keep_loops);

orc_graph *orc_build_graph_cov(unsigned k, unsigned num_buckets, const char *bases, const uint64_t *off, uint64_t nreads,
                               const char *flavour_version, int with_cov) {
    return orc_build_graph_ex(k, num_buckets, bases, off, nreads, flavour_version, with_cov, 0, 1);
}

/* sort_edges / keep_loops: DeBruijnGraphExtentionConstructor::ConstructGraph(keep_perfect_loops), the spades-core variant
 * (debruijn_graph_constructor.hpp:590-604): unitigs sorted by Sequence::RawCompare before ids are assigned. */
orc_graph *orc_build_graph_ex(unsigned k, unsigned num_buckets, const char *bases, const uint64_t *off, uint64_t nreads,
                              const char *flavour_version, int with_cov, int sort_edges, int keep_loops) {
    orc_graph *g = (orc_graph *)calloc(1, sizeof *g);
    const unsigned K1 = k + 1, nw1 = orc_words(K1), nw = orc_words(k);
    /* STEP 1a: canonical (k+1)-mers, B buckets (kmer_extension_index_builder.hpp:72-75) */
    uint64_t *kpo = NULL, *kpo_sizes = (uint64_t *)calloc(num_buckets, 8);
    int64_t nkpo = orc_count('B', K1, num_buckets, bases, off, nreads, &kpo, kpo_sizes);
    g->n_kpomers = (uint64_t)nkpo;
    /* STEP 1b: k-mers of every (k+1)-mer and of its RC, IsMinimal filter, same B (DeBruijnKMerKMerSplitter,
     * kmer_splitters.hpp:163-179; :90-98 of the builder) -> sorted unique per bucket */
    size_t cap = (size_t)nkpo * 4 + 1, m = 0;
    uint64_t *cand = (uint64_t *)malloc(cap * (nw + 1) * 8);
    for (int64_t i = 0; i < nkpo; ++i) {
        uint64_t x[2][ORC_MAX_WORDS] = {{0}};
        for (unsigned j = 0; j < nw1; ++j) x[0][j] = kpo[(size_t)i * nw1 + j];
        orc_rc(x[0], K1, x[1]);
        for (int o = 0; o < 2; ++o) {
            for (unsigned p = 0; p < 2; ++p) { /* the two k-mers of the (k+1)-mer: positions 0 and 1 */
                uint64_t km[ORC_MAX_WORDS] = {0, 0, 0, 0};
                for (unsigned t = 0; t < k; ++t) km[t >> 5] |= (uint64_t)nucl_at(x[o], t + p) << ((t & 31) << 1);
                if (!orc_is_minimal(km, k)) continue;
                cand[m * (nw + 1)] = orc_bucket(km, k, num_buckets);
                for (unsigned j = 0; j < nw; ++j) cand[m * (nw + 1) + 1 + j] = km[j];
                ++m;
            }
        }
    }
    /* sort by (bucket, words) and unique */
    {
        /* reuse qsort with element = (nw+1) words; comparator needs nw -> use static */
        extern void orc__sort_records(uint64_t *d, size_t n, unsigned nwp1);
        orc__sort_records(cand, m, nw + 1);
    }
    kidx ix;
    ix.k = k; ix.nw = nw; ix.nb = num_buckets;
    ix.recs = (uint64_t *)malloc((m ? m : 1) * nw * 8);
    ix.boff = (uint64_t *)calloc(num_buckets + 1, 8);
    ix.n = 0;
    for (size_t i = 0; i < m; ++i) {
        const uint64_t *p = cand + i * (nw + 1);
        if (i && rec_cmp(p, p - (nw + 1), nw + 1) == 0) continue;
        for (unsigned j = 0; j < nw; ++j) ix.recs[ix.n * nw + j] = p[1 + j];
        ix.boff[p[0] + 1]++;
        ix.n++;
    }
    for (unsigned b = 0; b < num_buckets; ++b) ix.boff[b + 1] += ix.boff[b];
    free(cand);
    ix.mask = (uint8_t *)calloc(ix.n ? ix.n : 1, 1);
    g->n_kmers = ix.n;
    /* STEP 1c: FillExtensionsFromIndex, kmer_extension_index_builder.hpp:45-60 */
    for (int64_t i = 0; i < nkpo; ++i) {
        uint64_t x[ORC_MAX_WORDS] = {0, 0, 0, 0}, pre[ORC_MAX_WORDS] = {0, 0, 0, 0}, suf[ORC_MAX_WORDS] = {0, 0, 0, 0};
        for (unsigned j = 0; j < nw1; ++j) x[j] = kpo[(size_t)i * nw1 + j];
        unsigned pnucl = nucl_at(x, 0), nnucl = nucl_at(x, K1 - 1);
        for (unsigned t = 0; t < k; ++t) {
            pre[t >> 5] |= (uint64_t)nucl_at(x, t) << ((t & 31) << 1);
            suf[t >> 5] |= (uint64_t)nucl_at(x, t + 1) << ((t & 31) << 1);
        }
        kwh hp = make_kwh(&ix, pre), hs = make_kwh(&ix, suf);
        /* AddOutgoing / AddIncoming with inv_position, inout_mask.hpp:92-94,117-131 */
        ix.mask[hp.idx] |= (uint8_t)(1u << (hp.minimal ? nnucl : 7 - nnucl));
        ix.mask[hs.idx] |= (uint8_t)(1u << (hs.minimal ? pnucl + 4 : 7 - (pnucl + 4)));
    }
    if (g_early_at) { /* EarlyATClipper::run, construction.cpp:324-328 */
        at_remove_edges(&ix, 0.8);
        at_remove_tips(&ix, 0.8, 10, 200);
    }
    if (g_early_tip_bound) early_tip_clip(&ix, g_early_tip_bound);
    g->kmers = (uint64_t *)malloc((ix.n ? ix.n : 1) * nw * 8);
    memcpy(g->kmers, ix.recs, ix.n * nw * 8);
    g->masks = (uint8_t *)malloc(ix.n ? ix.n : 1);
    memcpy(g->masks, ix.mask, ix.n);

    /* STEP 2a: ExtractUnbranchingPaths, debruijn_graph_constructor.hpp:295-350 (chunks are concatenated in file order) */
    seqlist seqs = {0, 0, 0, 0};
    bseq b = {0, 0, 0};
    unsigned char *rcbuf = NULL; size_t rccap = 0;
    for (uint64_t r = 0; r < ix.n; ++r) {
        kwh kh = make_kwh(&ix, ix.recs + r * nw);
        uint8_t ext = get_mask(&ix, &kh);
        if (!is_junction(ext)) continue; /* AddStartDeEdges :215-226 */
        kwh side[2]; uint8_t sm[2]; int nside = 1;
        side[0] = kh; sm[0] = ext;
        kwh inv = kwh_rc(&ix, &kh);
        if (!inv.minimal) { side[1] = inv; sm[1] = get_mask(&ix, &inv); nside = 2; }
        for (int sd = 0; sd < nside; ++sd)
            for (unsigned next = 0; next < 4; ++next) {
                if (!(sm[sd] & (1u << next))) continue;
                deedge e; e.start = side[sd]; e.end = kwh_shl(&ix, &side[sd], next);
                construct_sequence(&ix, e, &b);
                if (b.n > rccap) { rccap = b.n * 2; rcbuf = (unsigned char *)realloc(rcbuf, rccap); }
                seq_rc(b.d, b.n, rcbuf);
                if (seq_less(b.d, rcbuf, b.n)) continue; /* if (s < !s) continue; :305-306 */
                sl_push(&seqs, b.d, b.n);
            }
    }
    /* STEP 2b: RemoveSequences, kmer_extension_index.hpp:141-147 */
    for (size_t i = 0; i < seqs.n; ++i) {
        remove_sequence(&ix, seqs.seq[i], seqs.len[i]);
        if (seqs.len[i] > rccap) { rccap = seqs.len[i] * 2; rcbuf = (unsigned char *)realloc(rcbuf, rccap); }
        seq_rc(seqs.seq[i], seqs.len[i], rcbuf);
        remove_sequence(&ix, rcbuf, seqs.len[i]);
    }
    size_t n_paths = seqs.n;
    /* STEP 2c: CollectLoops, :359-397 */
    if (keep_loops) {
        uint64_t *starts = (uint64_t *)malloc((ix.n ? ix.n : 1) * 8); size_t ns = 0;
        for (uint64_t r = 0; r < ix.n; ++r) if (!is_junction(ix.mask[r])) starts[ns++] = r;
        for (size_t si = 0; si < ns; ++si) {
            kwh st = make_kwh(&ix, ix.recs + starts[si] * nw);
            if (is_junction(get_mask(&ix, &st))) continue;
            /* FindMinimalKMerInLoop :252-262 (min by RtSeq operator<, rtseq.hpp:742-750) */
            kwh strc = kwh_rc(&ix, &st);
            kwh minimal = orc_less_nucl(st.w, strc.w, k) ? st : strc;
            kwh kh = st;
            step_right_kwh(&ix, &kh);
            for (; !rec_eq(st.w, kh.w, nw); step_right_kwh(&ix, &kh)) {
                if (!orc_less_nucl(minimal.w, kh.w, k)) minimal = kh;
                kwh khrc = kwh_rc(&ix, &kh);
                if (!orc_less_nucl(minimal.w, khrc.w, k)) minimal = khrc;
            }
            /* ConstructLoopFromVertex :283-293 */
            deedge bp; bp.start = minimal;
            bp.end = kwh_shl(&ix, &minimal, uniq_nucl(get_mask(&ix, &minimal) & 15));
            construct_sequence(&ix, bp, &b);
            /* look for a self-RC (k+1)-mer */
            long split = -1;
            for (size_t i = k; i < b.n; ++i) {
                uint64_t kp[ORC_MAX_WORDS], kprc[ORC_MAX_WORDS];
                kmer_from_codes(b.d + (i - k), K1, kp);
                orc_rc(kp, K1, kprc);
                if (rec_eq(kp, kprc, nw1)) { split = (long)(i - k); break; }
            }
            unsigned char *parts[2] = {NULL, NULL}; size_t plen[2] = {0, 0}; int np = 1;
            if (split < 0) { parts[0] = (unsigned char *)malloc(b.n); memcpy(parts[0], b.d, b.n); plen[0] = b.n; }
            else { /* SplitLoop :276-280 */
                size_t pos = (size_t)split; np = 2;
                plen[0] = k + 1; parts[0] = (unsigned char *)malloc(plen[0]); memcpy(parts[0], b.d + pos, plen[0]);
                size_t l1 = (b.n - k) - (pos + 1), l2 = pos + k;
                plen[1] = l1 + l2; parts[1] = (unsigned char *)malloc(plen[1] ? plen[1] : 1);
                memcpy(parts[1], b.d + pos + 1, l1); memcpy(parts[1] + l1, b.d, l2);
            }
            for (int p = 0; p < np; ++p) {
                unsigned char *rc = (unsigned char *)malloc(plen[p] ? plen[p] : 1);
                seq_rc(parts[p], plen[p], rc);
                if (seq_less(parts[p], rc, plen[p])) sl_push(&seqs, rc, plen[p]); else sl_push(&seqs, parts[p], plen[p]);
                remove_sequence(&ix, parts[p], plen[p]);
                remove_sequence(&ix, rc, plen[p]);
                free(rc); free(parts[p]);
            }
        }
        free(starts);
    }
    if (sort_edges && seqs.n > 1) {
        size_t *perm = (size_t *)malloc(seqs.n * sizeof *perm);
        for (size_t i = 0; i < seqs.n; ++i) perm[i] = i;
        g_sl = &seqs;
        qsort(perm, seqs.n, sizeof *perm, rawcmp_idx);
        unsigned char **ns = (unsigned char **)malloc(seqs.n * sizeof *ns);
        size_t *nl = (size_t *)malloc(seqs.n * sizeof *nl);
        for (size_t i = 0; i < seqs.n; ++i) { ns[i] = seqs.seq[perm[i]]; nl[i] = seqs.len[perm[i]]; }
        memcpy(seqs.seq, ns, seqs.n * sizeof *ns); memcpy(seqs.len, nl, seqs.n * sizeof *nl);
        free(ns); free(nl); free(perm);
    }
    g->n_unitigs = seqs.n; g->n_loops = seqs.n - n_paths;
    g->unitig_off = (uint64_t *)calloc(seqs.n + 1, 8);
    for (size_t i = 0; i < seqs.n; ++i) g->unitig_off[i + 1] = g->unitig_off[i] + seqs.len[i];
    g->unitig_seq = (char *)malloc(g->unitig_off[seqs.n] + 1);
    for (size_t i = 0; i < seqs.n; ++i) for (size_t j = 0; j < seqs.len[i]; ++j) g->unitig_seq[g->unitig_off[i] + j] = "ACGT"[seqs.seq[i][j]];
    g->unitig_seq[g->unitig_off[seqs.n]] = 0;

    /* STEP 3: FastGraphFromSequencesConstructor::ConstructGraph, :506-567 */
    const uint64_t min_id = 3; /* GraphCore ID_BIAS, assembly_graph/core/graph_core.hpp:234 */
    linkrec *recs = (linkrec *)malloc((seqs.n ? seqs.n * 2 : 1) * sizeof *recs);
    unsigned char *selfconj = (unsigned char *)calloc(seqs.n ? seqs.n : 1, 1);
    for (size_t i = 0; i < seqs.n; ++i) {
        const unsigned char *s = seqs.seq[i]; size_t n = seqs.len[i];
        uint64_t edge = min_id + 2 * i;
        if (n > rccap) { rccap = n * 2; rcbuf = (unsigned char *)realloc(rcbuf, rccap); }
        seq_rc(s, n, rcbuf);
        selfconj[i] = memcmp(s, rcbuf, n) == 0;
        for (int end = 0; end < 2; ++end) {
            linkrec *r = &recs[2 * i + end];
            if (end && selfconj[i]) { r->hash_and_mask = ~0ull; r->edge = 0; continue; } /* LinkRecord() :447-448,478-481 */
            uint64_t w[ORC_MAX_WORDS], wrc[ORC_MAX_WORDS];
            kmer_from_codes(end ? s + n - k : s, k, w);
            orc_rc(w, k, wrc);
            int is_rc = !orc_less_nucl(w, wrc, k); /* StartLink/EndLink :456-471: kmer < kmer_rc ? as-is : rc */
            kwh h = make_kwh(&ix, is_rc ? wrc : w);
            r->hash_and_mask = ((uint64_t)h.idx << 2) | ((uint64_t)is_rc << 1) | (uint64_t)(end ? 0 : 1);
            r->edge = edge;
        }
    }
    size_t nrecs = seqs.n * 2;
    qsort(recs, nrecs, sizeof *recs, lr_cmp);
    size_t *uniq = (size_t *)malloc((nrecs ? nrecs : 1) * sizeof *uniq), nv = 0;
    for (size_t i = 0; i < nrecs; ++i)
        if ((i == 0 || lr_hash(&recs[i]) != lr_hash(&recs[i - 1])) && !lr_invalid(&recs[i])) uniq[nv++] = i;
    g_recs = recs;
    qsort(uniq, nv, sizeof *uniq, vtx_cmp);
    g->n_vertices = nv;

    /* STEP 3b (-c): CoverageHashMapBuilder::FillCoverageFromStream, ph_map/coverage_hash_map_builder.hpp:18-39: every
     * (k+1)-mer instance of the read+RC stream that is minimal increments its counter (uint32); edge raw coverage =
     * sum over the edge's (k+1)-mers (graph_support/coverage_filling.hpp:46-62), 32-bit (core/coverage.hpp:44-65) */
    uint32_t *ecov = (uint32_t *)calloc(seqs.n ? seqs.n : 1, 4);
    if (with_cov && nkpo > 0) {
        kidx kx;
        kx.k = K1; kx.nw = nw1; kx.nb = num_buckets; kx.n = (uint64_t)nkpo; kx.recs = kpo; kx.mask = NULL;
        kx.boff = (uint64_t *)calloc(num_buckets + 1, 8);
        for (unsigned bb = 0; bb < num_buckets; ++bb) kx.boff[bb + 1] = kx.boff[bb] + kpo_sizes[bb];
        uint32_t *cnt = (uint32_t *)calloc((size_t)nkpo, 4);
        unsigned char *fw = NULL, *rc = NULL; size_t bc = 0;
        for (uint64_t r = 0; r < (g_cov_reads ? (g_cov_reads < nreads ? g_cov_reads : nreads) : nreads); ++r) {
            const char *sq = bases + off[r]; size_t n = (size_t)(off[r + 1] - off[r]), from, to;
            orc_longest_valid(sq, n, &from, &to);
            size_t len = to - from;
            if (len < K1) continue;
            if (len > bc) { bc = len * 2; fw = (unsigned char *)realloc(fw, bc); rc = (unsigned char *)realloc(rc, bc); }
            for (size_t i = 0; i < len; ++i) fw[i] = (unsigned char)dig(sq[from + i]);
            seq_rc(fw, len, rc);
            for (int strand = 0; strand < 2; ++strand) {
                const unsigned char *t = strand ? rc : fw;
                for (size_t j = 0; j + K1 <= len; ++j) {
                    uint64_t w[ORC_MAX_WORDS];
                    kmer_from_codes(t + j, K1, w);
                    if (!orc_is_minimal(w, K1)) continue;
                    int64_t ix2 = kidx_find(&kx, w);
                    if (ix2 >= 0) cnt[ix2] += 1;
                }
            }
        }
        free(fw); free(rc);
        for (size_t i = 0; i < seqs.n; ++i) {
            uint32_t raw = 0;
            for (size_t j = 0; j + K1 <= seqs.len[i]; ++j) {
                uint64_t w[ORC_MAX_WORDS], wr[ORC_MAX_WORDS];
                kmer_from_codes(seqs.seq[i] + j, K1, w);
                const uint64_t *c = w;
                if (!orc_is_minimal(w, K1)) { orc_rc(w, K1, wr); c = wr; }
                int64_t ix2 = kidx_find(&kx, c);
                if (ix2 >= 0) raw += cnt[ix2];
            }
            ecov[i] = raw;
        }
        free(cnt); free(kx.boff);
    }
    free(kpo); free(kpo_sizes);

    /* STEP 4: GFA (gfa_writer.cpp) */
    sbuf out = {0, 0, 0};
    sb_add(&out, "H\tsp:Z:", 7); sb_add(&out, flavour_version, strlen(flavour_version)); sb_add(&out, "\n", 1);
    for (size_t i = 0; i < seqs.n; ++i) { /* canonical edges in id order; without -c coverage is 0 */
        sb_add(&out, "S\t", 2); sb_printf_u64(&out, min_id + 2 * i); sb_add(&out, "\t", 1);
        sb_add(&out, g->unitig_seq + g->unitig_off[i], seqs.len[i]);
        if (!with_cov) sb_add(&out, "\tDP:f:0\tKC:i:0\n", 15);
        else { /* "DP:f:" << float(cov) (default ostream formatting = %g, precision 6) << "KC:i:" << raw */
            char t[64];
            double cov = (double)ecov[i] / (double)(seqs.len[i] - k);
            int tn = snprintf(t, sizeof t, "\tDP:f:%g\tKC:i:%u\n", (double)(float)cov, ecov[i]);
            sb_add(&out, t, (size_t)tn);
        }
    }
    for (size_t vn = 0; vn < nv; ++vn) { /* canonical vertices in id order = vertex_num order */
        size_t i0 = uniq[vn];
        uint64_t outv[8], outc[8]; size_t no = 0, nc = 0; /* out edges of v and of conj(v): at most 4 each */
        for (size_t j = i0; j < nrecs && lr_hash(&recs[j]) == lr_hash(&recs[i0]); ++j) {
            uint64_t e = recs[j].edge; size_t ei = (size_t)((e - min_id) >> 1);
            uint64_t ce = selfconj[ei] ? e : e + 1;
            int is_rc = (int)((recs[j].hash_and_mask >> 1) & 1), is_start = (int)(recs[j].hash_and_mask & 1);
            /* LinkEdge :491-500 + LinkOutgoingEdge/LinkIncomingEdge construction_helper.hpp:79-97 */
            if (is_start) { if (!is_rc) outv[no++] = e; else outc[nc++] = e; }
            else { if (!is_rc) outc[nc++] = ce; else outv[no++] = ce; }
        }
        qsort(outv, no, 8, u64_cmp); qsort(outc, nc, 8, u64_cmp); /* sorted out-edge lists, graph_core.hpp:199-202 */
        for (size_t a = 0; a < nc; ++a) { /* IncomingEdges(v) = conj of OutgoingEdges(conj v), graph_core.hpp:625-628 */
            uint64_t oc = outc[a]; size_t ei = (size_t)((oc - min_id) >> 1);
            uint64_t inc = selfconj[ei] ? oc : (((oc - min_id) & 1) ? oc - 1 : oc + 1);
            for (size_t c = 0; c < no; ++c) {
                uint64_t oe = outv[c];
                uint64_t cin = min_id + (((inc - min_id) >> 1) << 1), cout = min_id + (((oe - min_id) >> 1) << 1);
                sb_add(&out, "L\t", 2); sb_printf_u64(&out, cin); sb_add(&out, inc == cin ? "\t+\t" : "\t-\t", 3);
                sb_printf_u64(&out, cout); sb_add(&out, oe == cout ? "\t+\t" : "\t-\t", 3);
                sb_printf_u64(&out, k); sb_add(&out, "M\n", 2);
                g->n_links++;
            }
        }
    }
    if (!out.d) sb_add(&out, "", 0);
    g->gfa = out.d; g->gfa_len = out.n;
    for (size_t i = 0; i < seqs.n; ++i) free(seqs.seq[i]);
    free(seqs.seq); free(seqs.len); free(b.d); free(rcbuf); free(recs); free(uniq); free(selfconj); free(ecov);
    free(ix.recs); free(ix.boff); free(ix.mask);
    return g;
}

orc_graph *orc_build_graph(unsigned k, unsigned num_buckets, const char *bases, const uint64_t *off, uint64_t nreads,
                           const char *flavour_version) {
    return orc_build_graph_cov(k, num_buckets, bases, off, nreads, flavour_version, 0);
}

void orc_graph_free(orc_graph *g) {
    if (!g) return;
    free(g->kmers); free(g->masks); free(g->unitig_off); free(g->unitig_seq); free(g->gfa); free(g);
}
