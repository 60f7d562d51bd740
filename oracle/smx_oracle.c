/* oracle/smx_oracle.c — TEST INFRASTRUCTURE ONLY (see smx_oracle.h).
 * Plain-C restatement of the reference algorithm; every function cites the reference
 * file:line it follows (paths relative to /root/reference/src/common unless noted).
 * Deliberately scalar and simple: clarity over speed. */
#include "smx_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ nucleotides */
/* sequence/nucl.hpp:132-142 dignucl: A/a->0 C/c->1 G/g->2 T/t->3 */
static unsigned dignucl(char c) {
    if (c >= 'a' && c <= 't') c = (char)(c - 'a' + 'A');
    return c <= 'C' ? (c == 'A' ? 0u : 1u) : (c == 'G' ? 2u : 3u);
}
/* sequence/nucl.hpp is_nucl: ACGTacgt */
static int is_nucl(char c) {
    return c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'a' || c == 'c' || c == 'g' || c == 't';
}

/* ------------------------------------------------------------------ RtSeq */
unsigned orc_words(unsigned K) { return (K + 31) >> 5; } /* rtseq.hpp:131-133, TNucl=32 */

static unsigned get_nucl(const uint64_t *w, unsigned i) { /* operator[], rtseq.hpp:379-382 */
    return (unsigned)((w[i >> 5] >> ((i & 31) << 1)) & 3);
}

void orc_from_string(uint64_t *w, unsigned K, const char *s) { /* rtseq.hpp:166-187 */
    unsigned nw = orc_words(K);
    for (unsigned i = 0; i < ORC_MAX_WORDS; ++i) w[i] = 0;
    (void)nw;
    for (unsigned i = 0; i < K; ++i)
        w[i >> 5] |= (uint64_t)dignucl(s[i]) << ((i & 31) << 1);
}

void orc_to_string(const uint64_t *w, unsigned K, char *s) { /* rtseq.hpp:629-635 */
    for (unsigned i = 0; i < K; ++i) s[i] = "ACGT"[get_nucl(w, i)];
    s[K] = 0;
}

/* operator<<=, rtseq.hpp:459-476: every word >>2, low 2 bits of word i+1 carried into bits
 * 62-63 of word i, new nucleotide at 2*((K-1) mod 32) of the last word. */
void orc_shl(uint64_t *w, unsigned K, unsigned c) {
    unsigned nw = orc_words(K);
    if (nw == 0) return;
    for (unsigned i = 0; i + 1 < nw; ++i)
        w[i] = (w[i] >> 2) | ((w[i + 1] & 3) << 62);
    unsigned lastshift = ((K + 32 - 1) & 31) << 1;
    w[nw - 1] = (w[nw - 1] >> 2) | ((uint64_t)c << lastshift);
}

/* operator>>, rtseq.hpp:569-588: push c in front, drop the last nucleotide. */
static void orc_shr(uint64_t *w, unsigned K, unsigned c) {
    unsigned nw = orc_words(K);
    uint64_t rm = c;
    for (unsigned i = 0; i < nw; ++i) {
        uint64_t new_rm = (w[i] >> 62) & 3;
        w[i] = (w[i] << 2) | rm;
        rm = new_rm;
    }
    unsigned nr = K & 31;
    if (nr) w[nw - 1] &= (((uint64_t)1) << (nr << 1)) - 1; /* MaskForLastBucket, rtseq.hpp:146-151 */
}

/* FastRC, rtseq.hpp:81-117: reverse word order with cross-word shift, complement, swap 2-bit
 * groups inside each word with the mask ladder, mask the tail. */
void orc_rc(const uint64_t *w, unsigned K, uint64_t *out) {
    uint64_t res[ORC_MAX_WORDS + 1] = {0, 0, 0, 0, 0};
    const unsigned bit_size = K << 1;
    const unsigned extra = bit_size & 63;
    const unsigned to_extra = 64 - extra;
    const unsigned filled = bit_size >> 6;
    unsigned real_length = filled;
    if (extra == 0) {
        for (unsigned i = 0, j = filled - 1; i < filled; i++, j--) res[i] = w[j];
    } else {
        for (unsigned i = 0, j = filled; i < filled && j > 0; i++, j--)
            res[i] = (w[j] << to_extra) + (w[j - 1] >> extra);
        res[filled] = (w[0] << to_extra);
        real_length++;
    }
    /* mask ladder (ConstructLeftMasks / RightMasks, rtseq.hpp:58-79): swap groups of 2,4,8,16,32 bits */
    static const uint64_t left[6] = {0, 0xCCCCCCCCCCCCCCCCull, 0xF0F0F0F0F0F0F0F0ull, 0xFF00FF00FF00FF00ull,
                                     0xFFFF0000FFFF0000ull, 0xFFFFFFFF00000000ull};
    for (unsigned i = 0; i < real_length; i++) {
        uint64_t v = ~res[i];
        for (unsigned it = 1; it < 6; it++) {
            unsigned shift = 1u << it;
            v = ((v & left[it]) >> shift) ^ ((v & ~left[it]) << shift);
        }
        res[i] = v;
    }
    if (extra != 0) res[real_length - 1] &= (((uint64_t)1) << extra) - 1;
    for (unsigned i = 0; i < ORC_MAX_WORDS; ++i) out[i] = i < real_length ? res[i] : 0;
}

/* IsMinimal, rtseq.hpp:409-417 */
int orc_is_minimal(const uint64_t *w, unsigned K) {
    for (unsigned i = 0; (i << 1) + 1 <= K; ++i) {
        unsigned front = get_nucl(w, i);
        unsigned end = 3 - get_nucl(w, K - 1 - i); /* complement, nucl.hpp */
        if (front != end) return front < end;
    }
    return 1;
}

/* operator<, rtseq.hpp:742-750: nucleotide-lexicographic from position 0 */
int orc_less_nucl(const uint64_t *a, const uint64_t *b, unsigned K) {
    for (unsigned i = 0; i < K; ++i) {
        unsigned x = get_nucl(a, i), y = get_nucl(b, i);
        if (x != y) return x < y;
    }
    return 0;
}

/* ------------------------------------------------------------------ XXH3-64 (xxHash 0.8.2) */
/* ext/include/xxh/xxhash.h:4239-4252 default secret (first 64 bytes are all the short paths use) */
static const uint8_t kSecret[64] = {
    0xb8, 0xfe, 0x6c, 0x39, 0x23, 0xa4, 0x4b, 0xbe, 0x7c, 0x01, 0x81, 0x2c, 0xf7, 0x21, 0xad, 0x1c,
    0xde, 0xd4, 0x6d, 0xe9, 0x83, 0x90, 0x97, 0xdb, 0x72, 0x40, 0xa4, 0xa4, 0xb7, 0xb3, 0x67, 0x1f,
    0xcb, 0x79, 0xe6, 0x4e, 0xcc, 0xc0, 0xe5, 0x78, 0x82, 0x5a, 0xd0, 0x7d, 0xcc, 0xff, 0x72, 0x21,
    0xb8, 0x08, 0x46, 0x74, 0xf7, 0x43, 0x24, 0x8e, 0xe0, 0x35, 0x90, 0xe6, 0x81, 0x3a, 0x26, 0x4c,
};
#define PRIME_MX1 0x165667919E3779F9ull /* xxhash.h:4254 */
#define PRIME_MX2 0x9FB21C651E98DF25ull /* xxhash.h:4255 */
#define PRIME64_1 0x9E3779B185EBCA87ull /* xxhash.h:3353 */

static uint64_t rd64(const uint8_t *p) { /* XXH_readLE64 */
    uint64_t v = 0;
    for (int i = 7; i >= 0; --i) v = (v << 8) | p[i];
    return v;
}
static uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint64_t rotl64(uint64_t v, int r) { return (v << r) | (v >> (64 - r)); }
static uint64_t swap64(uint64_t v) {
    uint64_t r = 0;
    for (int i = 0; i < 8; ++i) r = (r << 8) | ((v >> (8 * i)) & 0xff);
    return r;
}
static uint64_t mul128_fold64(uint64_t a, uint64_t b) { /* xxhash.h XXH3_mul128_fold64 */
    __uint128_t p = (__uint128_t)a * b;
    return (uint64_t)p ^ (uint64_t)(p >> 64);
}
static uint64_t xxh3_avalanche(uint64_t h) { /* xxhash.h XXH3_avalanche */
    h ^= h >> 37; h *= PRIME_MX1; h ^= h >> 32;
    return h;
}
static uint64_t xxh3_rrmxmx(uint64_t h, uint64_t len) { /* xxhash.h XXH3_rrmxmx */
    h ^= rotl64(h, 49) ^ rotl64(h, 24);
    h *= PRIME_MX2;
    h ^= (h >> 35) + len;
    h *= PRIME_MX2;
    return h ^ (h >> 28);
}
static uint64_t mix16B(const uint8_t *in, const uint8_t *sec) { /* XXH3_mix16B, seed 0 */
    return mul128_fold64(rd64(in) ^ rd64(sec), rd64(in + 8) ^ rd64(sec + 8));
}

uint64_t orc_xxh3_64(const void *data, size_t len) {
    const uint8_t *in = (const uint8_t *)data;
    if (len == 8) { /* XXH3_len_4to8_64b, xxhash.h:4537-4551, seed 0 */
        uint32_t input1 = rd32(in), input2 = rd32(in + len - 4);
        uint64_t bitflip = rd64(kSecret + 8) ^ rd64(kSecret + 16);
        uint64_t input64 = input2 + (((uint64_t)input1) << 32);
        return xxh3_rrmxmx(input64 ^ bitflip, len);
    }
    if (len == 16) { /* XXH3_len_9to16_64b, xxhash.h:4553-4568 */
        uint64_t bitflip1 = rd64(kSecret + 24) ^ rd64(kSecret + 32);
        uint64_t bitflip2 = rd64(kSecret + 40) ^ rd64(kSecret + 48);
        uint64_t lo = rd64(in) ^ bitflip1, hi = rd64(in + len - 8) ^ bitflip2;
        uint64_t acc = len + swap64(lo) + hi + mul128_fold64(lo, hi);
        return xxh3_avalanche(acc);
    }
    /* XXH3_len_17to128_64b, xxhash.h:4640-4674, len in (16,32]: one round */
    uint64_t acc = len * PRIME64_1;
    acc += mix16B(in, kSecret);
    acc += mix16B(in + len - 16, kSecret + 16);
    return xxh3_avalanche(acc);
}

/* KMerSegmentPolicy::operator(), kmer_index/kmer_mph/kmer_buckets.hpp:47-52;
 * multiply_high_u64, adt/lemiere_mod_reduce.hpp:18-36; RtSeq::GetHash rtseq.hpp:690-696 */
uint64_t orc_bucket(const uint64_t *w, unsigned K, uint64_t num_buckets) {
    if (num_buckets == 1) return 0;
    uint64_t h = orc_xxh3_64(w, 8 * (size_t)orc_words(K));
    return (uint64_t)(((__uint128_t)h * (__uint128_t)num_buckets) >> 64);
}

/* ------------------------------------------------------------------ reads */
/* LongestValidCoords, io/reads/longest_valid_wrapper.hpp:16-43: longest run of is_nucl, first on ties */
void orc_longest_valid(const char *s, size_t n, size_t *from, size_t *to) {
    const size_t none = (size_t)-1;
    size_t best_len = 0, best_pos = none, pos = none;
    for (size_t i = 0; i <= n; ++i) {
        if (i < n && is_nucl(s[i])) {
            if (pos == none) pos = i;
        } else {
            if (pos != none) {
                size_t len = i - pos;
                if (len > best_len) { best_len = len; best_pos = pos; }
            }
            pos = none;
        }
    }
    if (best_len == 0) { *from = 0; *to = 0; return; }
    *from = best_pos; *to = best_pos + best_len;
}

/* ------------------------------------------------------------------ counting */
typedef struct { uint64_t *d; size_t n, cap; unsigned nw; } recvec; /* records carry bucket in slot 0 */

static void rv_push(recvec *v, uint64_t bucket, const uint64_t *w) {
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 1024;
        v->d = (uint64_t *)realloc(v->d, v->cap * (v->nw + 1) * sizeof(uint64_t));
    }
    uint64_t *p = v->d + v->n * (v->nw + 1);
    p[0] = bucket;
    for (unsigned i = 0; i < v->nw; ++i) p[1 + i] = w[i];
    v->n++;
}

static unsigned g_cmp_nw; /* qsort has no context argument; single-threaded test code */
/* order: bucket (file concatenation order, kmer_index_builder.hpp:190-203), then array_less:
 * lexicographic over uint64 words, word 0 first (adt/array_vector.hpp:332-341, pdqsort_pod.h:725-734) */
static int cmp_rec(const void *a, const void *b) {
    const uint64_t *x = (const uint64_t *)a, *y = (const uint64_t *)b;
    for (unsigned i = 0; i <= g_cmp_nw; ++i)
        if (x[i] != y[i]) return x[i] < y[i] ? -1 : 1;
    return 0;
}

/* one strand of one read: kmercount.cpp:65-83 (mode A) / kmer_splitters.hpp:28-44 (mode B) */
static void fill_from_sequence(recvec *v, char mode, unsigned K, unsigned nb, const unsigned char *seq, size_t len) {
    if (len < K) return;
    uint64_t kmer[ORC_MAX_WORDS] = {0, 0, 0, 0};
    for (unsigned i = 0; i < K; ++i) kmer[i >> 5] |= (uint64_t)seq[i] << ((i & 31) << 1); /* seq.start<RtSeq>(K) */
    orc_shr(kmer, K, 0);                                                                  /* >> 'A' */
    for (size_t j = K - 1; j < len; ++j) {
        orc_shl(kmer, K, seq[j]);
        if (mode == 'B' && !orc_is_minimal(kmer, K)) continue; /* StoringTypeFilter<InvertableStoring>, storing_traits.hpp:92-101 */
        rv_push(v, orc_bucket(kmer, K, nb), kmer);
    }
}

int64_t orc_count(char mode, unsigned K, unsigned num_buckets, const char *bases, const uint64_t *off,
                  uint64_t nreads, uint64_t **out, uint64_t *bucket_sizes) {
    recvec v = {0, 0, 0, orc_words(K)};
    unsigned char *fw = NULL, *rc = NULL;
    size_t bufcap = 0;
    for (uint64_t r = 0; r < nreads; ++r) {
        const char *s = bases + off[r];
        size_t n = (size_t)(off[r + 1] - off[r]), from, to;
        orc_longest_valid(s, n, &from, &to); /* EasyStream(handle_Ns=true), io_helper.cpp:21-34 */
        size_t len = to - from;
        if (len == 0) continue;
        if (len > bufcap) {
            bufcap = len * 2;
            fw = (unsigned char *)realloc(fw, bufcap);
            rc = (unsigned char *)realloc(rc, bufcap);
        }
        for (size_t i = 0; i < len; ++i) fw[i] = (unsigned char)dignucl(s[from + i]);
        for (size_t i = 0; i < len; ++i) rc[i] = (unsigned char)(3 - fw[len - 1 - i]); /* RCWrap, rc_reader_wrapper.hpp:24-53 */
        fill_from_sequence(&v, mode, K, num_buckets, fw, len);
        fill_from_sequence(&v, mode, K, num_buckets, rc, len);
    }
    free(fw); free(rc);
    g_cmp_nw = v.nw;
    if (v.n) qsort(v.d, v.n, (v.nw + 1) * sizeof(uint64_t), cmp_rec); /* pdqsort_pod, kmer_splitter.hpp:140 */
    for (unsigned b = 0; b < num_buckets; ++b) bucket_sizes[b] = 0;
    uint64_t *res = (uint64_t *)malloc((v.n ? v.n : 1) * v.nw * sizeof(uint64_t));
    size_t m = 0;
    for (size_t i = 0; i < v.n; ++i) { /* std::unique, kmer_splitter.hpp:141 ; merge-unique kmer_index_builder.hpp:381-399 */
        const uint64_t *p = v.d + i * (v.nw + 1);
        if (i && cmp_rec(p, p - (v.nw + 1)) == 0) continue;
        for (unsigned k = 0; k < v.nw; ++k) res[m * v.nw + k] = p[1 + k];
        bucket_sizes[p[0]]++;
        m++;
    }
    free(v.d);
    *out = res;
    return (int64_t)m;
}

void orc_free(void *p) { free(p); }

/* shared with smx_oracle_graph.c: sort records of nwp1 words lexicographically (word 0 first) */
void orc__sort_records(uint64_t *d, size_t n, unsigned nwp1) {
    g_cmp_nw = nwp1 - 1;
    if (n) qsort(d, n, nwp1 * sizeof(uint64_t), cmp_rec);
}
