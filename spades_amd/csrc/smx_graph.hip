// spades_amd/csrc/smx_graph.hip — de Bruijn construction on the device results of the counting path.
// Included by smx_api.hip (one translation unit).
//
// Reference rows (paths relative to /root/reference/src/common):
//   a13 DeBruijnKMerKMerSplitter        kmer_index/kmer_mph/kmer_splitters.hpp:138-207      -> k_derive_* + count
//   a14 KMerIndex / PerfectHashMap      kmer_index/kmer_mph/kmer_index.hpp:88-100           -> RankDir + kmer_rank
//   a15 FillExtensionsFromIndex/InOutMask  extension_index/kmer_extension_index_builder.hpp:45-60,
//                                           extension_index/inout_mask.hpp:92-131           -> k_fill_masks
//   a16 UnbranchingPathExtractor        assembly_graph/construction/debruijn_graph_constructor.hpp:184-410
//                                        -> k_cand_*, k_walk_len, k_keep, k_walk_write (+ host loops)
//   a17 FastGraphFromSequencesConstructor  same file :412-568                               -> k_link_keys, k_vertex_*
//   a20 coverage                         ph_map/coverage_hash_map_builder.hpp:16-57, graph_support/coverage_filling.hpp:17-96
//
// Device layout: k-mers = the sorted-unique k-mer "file" (bucket-major, B = 10*threads), rank r = position in it
// (stands in for the MPHF index, whose values never reach the output: debruijn_graph_constructor.hpp:540-547; size_t there,
// 64 bits here). One byte mask per rank. A *node* is an oriented k-mer: node = 2*rank + o, o=1 meaning RC of the stored k-mer.
// Unitigs leave the walk kernels 2-bit packed (the RtSeq / Sequence word layout), every unitig starting on a word boundary.
#pragma once
#include "smx_device.hpp"

namespace smx {

typedef unsigned long long node_t;
constexpr node_t NODE_NONE = ~0ull;
// A successor-table entry: the successor node in the low 62 bits, the nucleotide that leads to it in the top 2; NODE_NONE in the
// entries of junction nodes (written by k_succ_junctions once the masks are final), so that a walk needs ONE read per step.
constexpr node_t NODE_MASK = (1ull << 62) - 1;
__device__ __forceinline__ node_t succ_node(node_t e) { return e == NODE_NONE ? NODE_NONE : (e & NODE_MASK); }
__device__ __forceinline__ unsigned succ_nucl(node_t e) { return (unsigned)(e >> 62); }
// The node table of the default path: one 64-bit entry per oriented k-mer, bits 40..43 = its outgoing extensions (the incoming
// ones of a node are the outgoing ones of its reverse complement, the neighbouring entry), bits 0..39 = the successor node,
// meaningful when exactly one outgoing bit is set (several writers OR their successors together, and nobody reads the mix).
constexpr unsigned TAB_OUT_SHIFT = 40;
constexpr node_t TAB_NODE_MASK = (1ull << TAB_OUT_SHIFT) - 1;
__device__ __forceinline__ unsigned tab_out4(node_t e) { return (unsigned)(e >> TAB_OUT_SHIFT) & 15u; }

template <int NW>
__device__ __forceinline__ Rec<NW> rec_shl(const Rec<NW> &x, unsigned K, unsigned c) {  // operator<<, rtseq.hpp:437-457
    Rec<NW> r;
#pragma unroll
    for (int i = 0; i < NW - 1; ++i) r.w[i] = (x.w[i] >> 2) | (x.w[i + 1] << 62);
    r.w[NW - 1] = (x.w[NW - 1] >> 2) | ((uint64_t)c << (((K - 1) & 31) << 1));
    return r;
}
// first k nucleotides of a (k+1)-mer (k and k+1 need the same number of words because k is odd)
template <int NW>
__device__ __forceinline__ Rec<NW> rec_prefix(const Rec<NW> &x, unsigned k) {
    Rec<NW> r = x;
    r.w[k >> 5] &= ~(3ull << ((k & 31) << 1));
    return r;
}
template <int NW>
__device__ __forceinline__ Rec<NW> rec_suffix(const Rec<NW> &x) {  // nucleotides 1..k of a (k+1)-mer
    Rec<NW> r;
#pragma unroll
    for (int i = 0; i < NW - 1; ++i) r.w[i] = (x.w[i] >> 2) | (x.w[i + 1] << 62);
    r.w[NW - 1] = x.w[NW - 1] >> 2;
    return r;
}
template <int NW>
__device__ __forceinline__ unsigned rec_nucl(const Rec<NW> &x, unsigned i) {
    return (unsigned)((x.w[i >> 5] >> ((i & 31) << 1)) & 3);
}
// canonical representative and whether the input was the non-minimal orientation
template <int NW>
__device__ __forceinline__ Rec<NW> rec_canon(const Rec<NW> &x, unsigned K, unsigned &is_rc) {
    Rec<NW> y = rec_rc<NW>(x, K);
    bool minimal = rc_ge<NW>(y, x);
    is_rc = minimal ? 0u : 1u;
    return minimal ? x : y;
}
// nucleotide-lexicographic compare, nucleotide 0 first (RtSeq operator<, rtseq.hpp:742-750; Sequence operator<, sequence.hpp:592-600)
template <int NW>
__device__ __forceinline__ int rec_lex_cmp(const Rec<NW> &a, const Rec<NW> &b) {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const uint64_t d = a.w[i] ^ b.w[i];
        if (d) {
            const unsigned pos = (unsigned)__builtin_ctzll(d) & ~1u;
            return ((a.w[i] >> pos) & 3) < ((b.w[i] >> pos) & 3) ? -1 : 1;
        }
    }
    return 0;
}

// ---- rank directory -----------------------------------------------------------------------------------------------------------
// Where a canonical k-mer sits in a sorted k-mer file (KMerIndex::seq_idx stand-in, kmer_index.hpp:88-100). Inside a bucket the
// records ascend by key, so slot = floor(key_fraction * SB) ascends too: dir[b][s] = first record of bucket b whose slot is >= s
// (relative to the bucket start, low 32 bits; SB + 1 entries per bucket, the last one = bucket size). A lookup is one hash, one
// 16-byte directory read (this entry and the next) and, where that does not settle it, a scan of the few adjacent records — two
// dependent HBM reads instead of the ~10 of a binary search over the bucket.
// The high 32 bits of the entry of a non-empty slot hold 10-bit fingerprints (low bits of the XXH3 value, which the slot does not
// depend on) of its first three records: a k-mer that is KNOWN to be in the file (PRESENT) and lands in a slot of <= 3 records is
// identified without touching the records when it is alone in the slot or matches exactly ONE of the slot's fingerprints (it is one
// of those records, so the others are ruled out); two records sharing a fingerprint (2^-10 per pair) fall back to the scan.
// ~92 % of the present lookups at one record per slot on average (37 % + 37 % + 18 %): 1.1 random transactions per lookup
// instead of 1.4 with the plain directory at two slots per record (same 8 B per record).
struct RankDir {
    const uint64_t *dir;             // [B * (SB + 1)]
    const unsigned long long *boff;  // [B + 1] bucket offsets of the file
    uint32_t B, SB;
    unsigned K;
    uint32_t verify;  // 1: every lookup compares the record it finds
};
template <int NW>
__device__ __forceinline__ uint64_t dir_slot(const Rec<NW> &x, const RankDir &ix) {
    return __umul64hi(key_top64<NW>(x, ix.K), (uint64_t)ix.SB);
}
constexpr unsigned DIR_FP_BITS = 10;
__device__ __forceinline__ uint32_t dir_fp(uint64_t hash) { return (uint32_t)hash & ((1u << DIR_FP_BITS) - 1); }
// RankDir::verify (option "verify_lookups", the tests) turns the PRESENT shortcuts off.
template <int NW, bool PRESENT = false>
__device__ __forceinline__ node_t kmer_rank(const Rec<NW> *__restrict__ kmers, const RankDir &ix, const Rec<NW> &canon) {
    const uint64_t hash = xxh3_rec<NW>(canon);
    const uint32_t b = bucket_of(hash, ix.B);
    const uint64_t *d = ix.dir + (uint64_t)b * (ix.SB + 1) + dir_slot<NW>(canon, ix);
    const uint64_t base = ix.boff[b];
    const uint64_t e0 = d[0], e1 = d[1];
    uint64_t lo = base + (uint32_t)e0, hi = base + (uint32_t)e1;
    if (PRESENT && !ix.verify) {
        const uint64_t n = hi - lo;
        if (n == 1) return lo;
        if (n == 2 || n == 3) {
            const uint32_t fp = dir_fp(hash), fm = (1u << DIR_FP_BITS) - 1;
            const uint32_t m1 = fp == ((uint32_t)(e0 >> 32) & fm), m2 = fp == ((uint32_t)(e0 >> (32 + DIR_FP_BITS)) & fm),
                           m3 = (n == 3 && fp == ((uint32_t)(e0 >> (32 + 2 * DIR_FP_BITS)) & fm)) ? 1u : 0u;
            if (m1 + m2 + m3 == 1) return lo + (m2 ? 1 : (m3 ? 2 : 0));
        }
    }
    const uint64_t end = hi;
    while (hi - lo > 8) {  // crowded slot (skewed keys): halve first (lower bound: the answer stays in [lo, hi])
        const uint64_t mid = (lo + hi) >> 1;
        if (rec_less<NW>(kmers[mid], canon)) lo = mid + 1; else hi = mid;
    }
    for (; lo < end; ++lo) {
        const Rec<NW> r = kmers[lo];
        if (!rec_less<NW>(r, canon)) return rec_eq<NW>(r, canon) ? lo : NODE_NONE;
    }
    return NODE_NONE;
}
// Directory of a sorted file. One record per lane; a record that opens new slots writes their entries (its own slot's with the
// fingerprints of the slot's first three records, read ahead), long gaps are filled by the whole wave (low-complexity data leaves
// most slots of a bucket empty).
template <int NW>
__global__ void __launch_bounds__(BLK) k_dir_fill(const void *kmers_, uint64_t n, RankDir ix, uint64_t *dir) {
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    const int lane = threadIdx.x & 63;
    for (uint64_t base = (uint64_t)blockIdx.x * BLK; base < n; base += (uint64_t)gridDim.x * BLK) {
        const uint64_t i = base + threadIdx.x;
        uint64_t *d = dir;
        uint64_t t0 = 0, t1 = 0;  // entries [t0, t1) get `val`
        uint32_t val = 0;
        uint64_t u0 = 0, u1 = 0;  // last record of a bucket: entries [u0, u1) get val + 1
        if (i < n) {
            const Rec<NW> x = kmers[i];
            const uint64_t hash = xxh3_rec<NW>(x);
            const uint32_t b = bucket_of(hash, ix.B);
            const uint64_t s = dir_slot<NW>(x, ix), bs = ix.boff[b];
            d = dir + (uint64_t)b * (ix.SB + 1);
            val = (uint32_t)(i - bs);
            const uint64_t sp = i == bs ? ~0ull : dir_slot<NW>(kmers[i - 1], ix);  // slot of the record before (same bucket)
            t0 = i == bs ? 0 : sp + 1;
            t1 = s;  // the empty slots before this record's own
            if (t0 <= s) {  // first record of its slot: offset + the fingerprints of the slot's first three records (one plain store)
                unsigned long long e = (unsigned long long)val | ((unsigned long long)dir_fp(hash) << 32);
                const uint64_t be = ix.boff[b + 1];
                for (int j = 1; j <= 2 && i + j < be; ++j) {
                    const Rec<NW> y = kmers[i + j];
                    if (dir_slot<NW>(y, ix) != s) break;
                    e |= (unsigned long long)dir_fp(xxh3_rec<NW>(y)) << (32 + j * DIR_FP_BITS);
                }
                d[s] = e;
            }
            if (t0 > s) t0 = t1 = 0;
            if (i + 1 == ix.boff[b + 1]) {
                u0 = s + 1;
                u1 = (uint64_t)ix.SB + 1;
            }
        }
        for (int pass = 0; pass < 2; ++pass) {
            const uint64_t a0 = pass ? u0 : t0, a1 = pass ? u1 : t1;
            const uint64_t v = pass ? (uint64_t)val + 1 : (uint64_t)val;
            const bool big = a1 > a0 + 8;
            if (!big)
                for (uint64_t t = a0; t < a1; ++t) d[t] = v;
            unsigned long long m = __ballot(big);
            while (m) {
                const int src = __builtin_ctzll(m);
                m &= m - 1;
                uint64_t *dd = (uint64_t *)__shfl((unsigned long long)(uintptr_t)d, src, 64);
                const uint64_t b0 = __shfl((unsigned long long)a0, src, 64), b1 = __shfl((unsigned long long)a1, src, 64);
                const uint64_t vv = __shfl((unsigned long long)v, src, 64);
                for (uint64_t t = b0 + lane; t < b1; t += 64) dd[t] = vv;
            }
        }
    }
}

__device__ __forceinline__ unsigned brev8(unsigned m) { return __brev(m) >> 24; }  // InOutMask::conjugate, inout_mask.hpp:18-39,112-115
__device__ __forceinline__ bool uniq4(unsigned m) { return m && !(m & (m - 1)); }
__device__ __forceinline__ bool mask_junction(unsigned m) { return !uniq4(m & 15) || !uniq4((m >> 4) & 15); }  // inout_mask.hpp:157-159
// one step of a walk: false at a junction k-mer, else the successor and the nucleotide that leads to it — ONE 16-byte read (the
// entries of both orientations of a k-mer are neighbours)
__device__ __forceinline__ bool tab_step(const node_t *__restrict__ tab, node_t node, node_t &next, unsigned &nuc) {
    const ulonglong2 pr = *reinterpret_cast<const ulonglong2 *>(tab + (node & ~1ull));
    const node_t es = (node & 1) ? pr.y : pr.x, ec = (node & 1) ? pr.x : pr.y;
    const unsigned o = tab_out4(es);
    if (!uniq4(o) || !uniq4(tab_out4(ec))) return false;
    next = es & TAB_NODE_MASK;
    nuc = __ffs(o) - 1;
    return true;
}
template <int NW>
__device__ __forceinline__ Rec<NW> node_kmer(const Rec<NW> *__restrict__ kmers, node_t node, unsigned k) {  // oriented k-mer of a node
    const Rec<NW> x = kmers[node >> 1];
    return (node & 1) ? rec_rc<NW>(x, k) : x;
}

// a13: the 2 k-mers of every canonical (k+1)-mer, canonicalised (the RC (k+1)-mer yields the same two)
template <int NW>
__global__ void __launch_bounds__(BLK) k_derive_kmers(const void *kpo_, uint64_t n, unsigned k, void *out_) {
    const Rec<NW> *kpo = (const Rec<NW> *)kpo_;
    Rec<NW> *out = (Rec<NW> *)out_;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLK) {
        Rec<NW> x = kpo[i];
        unsigned f;
        out[2 * i] = rec_canon<NW>(rec_prefix<NW>(x, k), k, f);
        out[2 * i + 1] = rec_canon<NW>(rec_suffix<NW>(x), k, f);
    }
}
// The same step when 4 record buffers of the whole (k+1)-mer file do not fit HBM: the k-mer file is produced one bucket range at a
// time. The buckets of the k-mer file are disjoint by construction, so each range is final on its own (no merge of runs).
// k_derive_hist: how many derived k-mers land in every bucket (decides the ranges); k_derive_range: emit those of [b0, b1).
template <int NW>
__global__ void __launch_bounds__(BLK) k_derive_hist(const void *kpo_, uint64_t n, unsigned k, uint32_t B, unsigned long long *hist,
                                                     uint32_t *tags /* nullable: bucket of the prefix | bucket of the suffix << 16 */) {
    extern __shared__ uint32_t lh[];
    const Rec<NW> *kpo = (const Rec<NW> *)kpo_;
    for (uint32_t t = threadIdx.x; t < B; t += BLK) lh[t] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLK) {
        const Rec<NW> x = kpo[i];
        unsigned f;
        const uint32_t bp = bucket_of(xxh3_rec<NW>(rec_canon<NW>(rec_prefix<NW>(x, k), k, f)), B);
        const uint32_t bs = bucket_of(xxh3_rec<NW>(rec_canon<NW>(rec_suffix<NW>(x), k, f)), B);
        atomicAdd(&lh[bp], 1u);
        atomicAdd(&lh[bs], 1u);
        if (tags) tags[i] = bp | (bs << 16);
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < B; t += BLK)
        if (lh[t]) atomicAdd(&hist[t], (unsigned long long)lh[t]);
}
// One reservation (a single-address atomic: ~88 per microsecond on this chip) per tile of DR_ITEMS x 256 (k+1)-mers.
constexpr int DR_ITEMS = 16;
template <int NW>
__global__ void __launch_bounds__(BLK) k_derive_range(const void *kpo_, uint64_t n, unsigned k, uint32_t B, uint32_t b0, uint32_t b1,
                                                      const uint32_t *tags, void *out_, unsigned long long *count) {
    static_assert(DR_ITEMS * (BLK / 64) == 64, "one wave scans the run lengths");
    __shared__ uint32_t scratch[BLK / 64 + 2];
    __shared__ uint32_t s_cnt[DR_ITEMS * (BLK / 64)];
    __shared__ unsigned long long s_base;
    const Rec<NW> *kpo = (const Rec<NW> *)kpo_;
    Rec<NW> *out = (Rec<NW> *)out_;
    const uint64_t tile = (uint64_t)DR_ITEMS * BLK;
    for (uint64_t base = (uint64_t)blockIdx.x * tile; base < n; base += (uint64_t)gridDim.x * tile) {
        uint32_t fl = 0, cnt = 0;  // 2 flag bits per item: emit the prefix / the suffix k-mer
#pragma unroll
        for (int j = 0; j < DR_ITEMS; ++j) {
            const uint64_t i = base + (uint64_t)j * BLK + threadIdx.x;
            if (i >= n) continue;
            uint32_t bp, bs;
            if (tags) {  // the buckets were found by the histogram pass
                const uint32_t t = tags[i];
                bp = t & 0xFFFFu;
                bs = t >> 16;
            } else {
                const Rec<NW> x = kpo[i];
                unsigned f;
                bp = bucket_of(xxh3_rec<NW>(rec_canon<NW>(rec_prefix<NW>(x, k), k, f)), B);
                bs = bucket_of(xxh3_rec<NW>(rec_canon<NW>(rec_suffix<NW>(x), k, f)), B);
            }
            const uint32_t kp = (bp >= b0 && bp < b1) ? 1u : 0u, ks = (bs >= b0 && bs < b1) ? 1u : 0u;
            fl |= (kp | (ks << 1)) << (2 * j);
            cnt += kp + ks;
        }
        // offsets per (item, wave): the 64 lanes of a wave then write one contiguous run per item
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int j = 0; j < DR_ITEMS; ++j) {
            const uint32_t f2 = (fl >> (2 * j)) & 3u;
            const unsigned long long bp_ = __ballot(f2 & 1), bs_ = __ballot(f2 & 2);
            if (lane == 0) s_cnt[j * (BLK / 64) + wave] = (uint32_t)(__popcll(bp_) + __popcll(bs_));
        }
        __syncthreads();
        if (threadIdx.x < 64) {  // exclusive scan of the DR_ITEMS * 4 = 64 run lengths by one wave
            const uint32_t v = s_cnt[threadIdx.x];
            const uint32_t inc = wave_incl_scan<uint32_t>(v);
            s_cnt[threadIdx.x] = inc - v;
            if (threadIdx.x == 63) s_base = inc ? atomicAdd(count, (unsigned long long)inc) : 0ull;
        }
        __syncthreads();
        (void)cnt;
        (void)scratch;
#pragma unroll
        for (int j = 0; j < DR_ITEMS; ++j) {
            const uint32_t f2 = (fl >> (2 * j)) & 3u;
            const unsigned long long bp_ = __ballot(f2 & 1), bs_ = __ballot(f2 & 2);
            if (!f2) continue;
            const unsigned long long lt = (1ull << lane) - 1;
            const unsigned long long o = s_base + s_cnt[j * (BLK / 64) + wave];
            const Rec<NW> x = kpo[base + (uint64_t)j * BLK + threadIdx.x];
            unsigned f;
            if (f2 & 1) out[o + __popcll(bp_ & lt)] = rec_canon<NW>(rec_prefix<NW>(x, k), k, f);
            if (f2 & 2) out[o + __popcll(bp_) + __popcll(bs_ & lt)] = rec_canon<NW>(rec_suffix<NW>(x), k, f);
        }
        __syncthreads();
    }
}

// a15: every canonical (k+1)-mer x is a de Bruijn edge P -> S between its oriented prefix and suffix nodes (and S^1 -> P^1 on the
// other strand): out[P] |= bit(x_k), out[S^1] |= bit(complement of x_0) — the reference's out[prefix] / in[suffix] updates
// (kmer_extension_index_builder.hpp:45-60; in-bits = the out-bits of the reverse complement, inout_mask.hpp:92-131) — and the same
// two atomics carry the successor nodes, which the two rank lookups have just produced.
template <int NW>
__global__ void __launch_bounds__(BLK) k_fill_tab(const void *kpo_, uint64_t n, unsigned k, const void *kmers_,
                                                  RankDir ix, node_t *tab, uint32_t *err) {
    const Rec<NW> *kpo = (const Rec<NW> *)kpo_;
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLK) {
        Rec<NW> x = kpo[i];
        const unsigned pn = rec_nucl<NW>(x, 0), nn = rec_nucl<NW>(x, k);
        unsigned prc, src;
        Rec<NW> p = rec_canon<NW>(rec_prefix<NW>(x, k), k, prc);
        Rec<NW> s = rec_canon<NW>(rec_suffix<NW>(x), k, src);
        const node_t rp = kmer_rank<NW, true>(kmers, ix, p), rs = kmer_rank<NW, true>(kmers, ix, s);
        if (rp == NODE_NONE || rs == NODE_NONE) {
            atomicAdd(err, 1u);
            continue;
        }
        const node_t P = (rp << 1) | prc, S = (rs << 1) | src;
        atomicOr(&tab[P], S | (1ull << (TAB_OUT_SHIFT + nn)));
        atomicOr(&tab[S ^ 1], (P ^ 1) | (1ull << (TAB_OUT_SHIFT + 3 - pn)));
    }
}
// InOutMask bytes of the canonical k-mers from the node table: out bits 0-3, in bits 4-7 = the bit-reversed out bits of the RC node
__global__ void k_tab_masks(const node_t *tab, uint64_t D0, uint8_t *mask) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < D0; r += (uint64_t)gridDim.x * blockDim.x)
        mask[r] = (uint8_t)(tab_out4(tab[2 * r]) | brev8(tab_out4(tab[2 * r + 1])));
}
// position-weighted checksum of an array of 64-bit words (or bytes): sum of value * (2 * index + 1) mod 2^64 and the plain sum — a
// fingerprint of the device graph for comparing two builds that are too big to leave the device (smx_graph_fingerprint)
template <typename T>
__global__ void __launch_bounds__(BLK) k_fingerprint(const T *a, uint64_t n, unsigned long long *out) {
    unsigned long long s0 = 0, s1 = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLK) {
        const unsigned long long v = (unsigned long long)a[i];
        s0 += v;
        s1 += v * (2ull * i + 1ull);
    }
    for (int o = 32; o > 0; o >>= 1) {
        s0 += __shfl_down(s0, o, 64);
        s1 += __shfl_down(s1, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], s0);
        atomicAdd(&out[1], s1);
    }
}

// ---- k-mer file + InOutMask bytes straight from a count in the EXT layout (smx_device.hpp) -------------------------------------------
// The sorted records hold (k-mer << 8 | byte) in their last word; copies of a k-mer whose bytes differ (survivors of cut minimizer
// partitions) sit next to each other. A head is the first record of a k-mer; its byte is the OR over its copies.
constexpr int XM_ITEMS = 4;
constexpr int XM_TILE = BLK * XM_ITEMS;
template <int NW>
__device__ __forceinline__ bool range_is_rc_palindrome(const Rec<NW> &x, unsigned a, unsigned len) {  // bases a .. a+len-1, len even
    for (unsigned i = 0; i < len / 2; ++i)
        if (rec_nucl<NW>(x, a + i) + rec_nucl<NW>(x, a + len - 1 - i) != 3) return false;
    return true;
}
template <int NW>
__device__ __forceinline__ bool ext_head(const Rec<NW> *__restrict__ in, uint64_t i) {
    return i == 0 || !rec_eq<NW>(rec_pure<NW>(in[i]), rec_pure<NW>(in[i - 1]));
}
template <int NW>
__global__ void __launch_bounds__(BLK) k_ext_heads(const void *in_, uint64_t n, unsigned long long *tcnt) {
    const Rec<NW> *in = (const Rec<NW> *)in_;
    __shared__ uint32_t s_cnt;
    const uint64_t ntiles = (n + XM_TILE - 1) / XM_TILE;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        uint32_t c = 0;
#pragma unroll
        for (int j = 0; j < XM_ITEMS; ++j) {
            const uint64_t i = tile * XM_TILE + (uint64_t)j * BLK + threadIdx.x;
            if (i < n && ext_head<NW>(in, i)) ++c;
        }
        const uint32_t inc = wave_incl_scan<uint32_t>(c);
        if ((threadIdx.x & 63) == 63 && inc) atomicAdd(&s_cnt, inc);
        __syncthreads();
        if (threadIdx.x == 0) tcnt[tile] = s_cnt;
        __syncthreads();
    }
}
// stats: [0] extension bits set, [1] palindromic (k+1)-mers among them (k+1 is even): with both, the number of canonical (k+1)-mers
// the reads hold = (bits + palindromes) / 2 — every other (k+1)-mer sets one bit at its prefix node and one at the reverse
// complement of its suffix node (k_fill_tab), a palindrome sets the same bit twice.
// SPLIT: k-mer file + bytes + stats; else the merged records stay in the EXT layout (the survivors of cut partitions, before the sort)
template <int NW, bool SPLIT>
__global__ void __launch_bounds__(BLK) k_ext_merge(const void *in_, uint64_t n, const unsigned long long *toff, unsigned k, void *kmers_, uint8_t *mask,
                                                   unsigned long long *stats) {
    const Rec<NW> *in = (const Rec<NW> *)in_;
    Rec<NW> *kmers = (Rec<NW> *)kmers_;
    __shared__ uint32_t s_w[XM_ITEMS * (BLK / 64)];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t ntiles = (n + XM_TILE - 1) / XM_TILE;
    unsigned long long bits = 0, pals = 0;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t fl = 0;
        unsigned long long bal[XM_ITEMS];
#pragma unroll
        for (int j = 0; j < XM_ITEMS; ++j) {
            const uint64_t i = tile * XM_TILE + (uint64_t)j * BLK + threadIdx.x;
            const bool h = i < n && ext_head<NW>(in, i);
            fl |= (h ? 1u : 0u) << j;
            bal[j] = __ballot(h);
            if (lane == 0) s_w[j * (BLK / 64) + wave] = (uint32_t)__popcll(bal[j]);
        }
        __syncthreads();
        if (threadIdx.x == 0) {  // exclusive scan of the XM_ITEMS * 4 wave counts (tile order: item-major, then wave)
            uint32_t run = 0;
            for (int t = 0; t < XM_ITEMS * (BLK / 64); ++t) {
                const uint32_t v = s_w[t];
                s_w[t] = run;
                run += v;
            }
        }
        __syncthreads();
        const uint64_t obase = toff[tile];
#pragma unroll
        for (int j = 0; j < XM_ITEMS; ++j) {
            if (!((fl >> j) & 1)) continue;
            uint64_t i = tile * XM_TILE + (uint64_t)j * BLK + threadIdx.x;
            const Rec<NW> raw = in[i];
            const Rec<NW> x = rec_pure<NW>(raw);
            unsigned m = (unsigned)(raw.w[NW - 1] & 0xFFu);
            for (++i; i < n; ++i) {  // the copies of this k-mer (possibly into the next tile)
                const Rec<NW> y = in[i];
                if (!rec_eq<NW>(rec_pure<NW>(y), x)) break;
                m |= (unsigned)(y.w[NW - 1] & 0xFFu);
            }
            const uint64_t o = obase + s_w[j * (BLK / 64) + wave] + __popcll(bal[j] & ((1ull << lane) - 1));
            if constexpr (!SPLIT) {
                Rec<NW> y = raw;
                y.w[NW - 1] = (raw.w[NW - 1] & ~0xFFull) | m;
                kmers[o] = y;
                continue;
            }
            kmers[o] = x;
            mask[o] = (uint8_t)m;
            bits += __popc(m);
            {  // x + c is its own reverse complement iff c = complement of base 0 and bases 1..k-1 are; likewise for rc(x) + c'
                const unsigned x0 = rec_nucl<NW>(x, 0), xl = rec_nucl<NW>(x, k - 1);
                if (((m >> (3 - x0)) & 1) && range_is_rc_palindrome<NW>(x, 1, k - 1)) ++pals;
                if (((m >> (7 - xl)) & 1) && range_is_rc_palindrome<NW>(x, 0, k - 1)) ++pals;
            }
        }
        __syncthreads();
    }
    if constexpr (SPLIT) {
        for (int o = 32; o > 0; o >>= 1) {
            bits += __shfl_down(bits, o, 64);
            pals += __shfl_down(pals, o, 64);
        }
        if (lane == 0) {
            if (bits) atomicAdd(&stats[0], bits);
            if (pals) atomicAdd(&stats[1], pals);
        }
    }
}
// The usual case: every k-mer occurs once (cut partitions were merged before the sort) — one streaming pass splits the records;
// stats[2] counts records that repeat the k-mer before them (then the general merge above runs instead).
template <int NW>
__global__ void __launch_bounds__(BLK) k_ext_split(const void *in_, uint64_t n, unsigned k, void *kmers_, uint8_t *mask, unsigned long long *stats) {
    const Rec<NW> *in = (const Rec<NW> *)in_;
    Rec<NW> *kmers = (Rec<NW> *)kmers_;
    const unsigned lane = threadIdx.x & 63;
    unsigned long long bits = 0, pals = 0, rep = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLK) {
        const Rec<NW> raw = in[i];
        const Rec<NW> x = rec_pure<NW>(raw);
        const unsigned m = (unsigned)(raw.w[NW - 1] & 0xFFu);
        if (i && rec_eq<NW>(rec_pure<NW>(in[i - 1]), x)) ++rep;
        kmers[i] = x;
        mask[i] = (uint8_t)m;
        bits += __popc(m);
        const unsigned x0 = rec_nucl<NW>(x, 0), xl = rec_nucl<NW>(x, k - 1);
        if (((m >> (3 - x0)) & 1) && range_is_rc_palindrome<NW>(x, 1, k - 1)) ++pals;
        if (((m >> (7 - xl)) & 1) && range_is_rc_palindrome<NW>(x, 0, k - 1)) ++pals;
    }
    for (int o = 32; o > 0; o >>= 1) {
        bits += __shfl_down(bits, o, 64);
        pals += __shfl_down(pals, o, 64);
        rep += __shfl_down(rep, o, 64);
    }
    if (lane == 0) {
        if (bits) atomicAdd(&stats[0], bits);
        if (pals) atomicAdd(&stats[1], pals);
        if (rep) atomicAdd(&stats[2], rep);
    }
}
// bucket offsets of the merged file: heads before the old offset
template <int NW>
__global__ void k_ext_boff(const void *in_, uint64_t n, const unsigned long long *toff, const unsigned long long *old_off, uint32_t nb1,
                           unsigned long long *new_off) {
    const Rec<NW> *in = (const Rec<NW> *)in_;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb1) return;
    const uint64_t pos = old_off[b];
    const uint64_t tile = pos / XM_TILE;
    unsigned long long c = pos < n ? toff[tile] : 0;
    if (pos >= n) {
        const uint64_t ntiles = (n + XM_TILE - 1) / XM_TILE;
        new_off[b] = toff[ntiles];  // total
        return;
    }
    // tile order is item-major: position p of the tile sits at index (p % BLK) of item (p / BLK) — the same order as the file
    for (uint64_t i = tile * XM_TILE; i < pos; ++i) c += ext_head<NW>(in, i) ? 1u : 0u;
    new_off[b] = c;
}

// Where the early clippers find a k-mer: they need nothing of the k-mer file but "the k-mer with this index" and "the index of this canonical
// k-mer, if the graph has it" — the order of the indices never reaches their result (marks per k-mer, bits per mask). FileFind: the sorted
// k-mer file and its rank directory (routes 1 and 2); PmFind (smx_pm.hip): the partition-major records of route 0, looked up by minimizer
// partition and chunk hash table (round 6: spades-core's default configuration — early_tip_clipper on — runs the route the bench measures).
template <int NW>
struct FileFind {
    const Rec<NW> *kmers;
    RankDir dir;
    __device__ __forceinline__ Rec<NW> kmer(uint64_t r) const { return kmers[r]; }
    __device__ __forceinline__ node_t find(const Rec<NW> &y) const { return kmer_rank<NW>(kmers, dir, y); }
    __device__ __forceinline__ node_t find_from(const Rec<NW> &y, uint64_t, int &byte0) const {  // (PmFind knows a shorter way from the k-mer it comes from)
        byte0 = -1;
        return find(y);
    }
    // successor of a node of a NON-junction k-mer in the table the clippers walk (k_succ's format here)
    __device__ __forceinline__ node_t next(const node_t *__restrict__ succ, node_t nd) const { return succ_node(succ[nd]); }
    // FindForward (early_simplification.hpp:102-112): from nd along non-junction k-mers, at most until cnt == bound; returns the node it stops at
    // (NODE_NONE: inconsistent index)
    __device__ __forceinline__ node_t advance(const node_t *__restrict__ succ, const uint8_t *__restrict__ mask, node_t nd, uint32_t &cnt, uint32_t bound, int = -1) const {
        while (cnt < bound && !mask_junction(mask[nd >> 1])) {
            ++cnt;
            nd = succ_node(succ[nd]);
            if (nd == NODE_NONE) break;  // (reported by k_succ: never walk off the array)
        }
        return nd;
    }
    // IsolateVertex on the len k-mers of a removed tip, from its first node on
    __device__ __forceinline__ void isolate_tip(const node_t *__restrict__ succ, node_t nd, uint32_t len, uint8_t *isolate, uint8_t *) const {
        for (uint32_t t = 0; t + 1 < len; ++t) {
            isolate[nd >> 1] = 1;
            nd = succ_node(succ[nd]);
        }
        isolate[nd >> 1] = 1;
    }
};

// the node table of clipped masks (spades-core variants): extensions from the masks, successors by lookup
// (PRESENT: the masks come from the reads themselves, every extension leads to a k-mer of the file)
template <int NW, bool PRESENT = false>
__global__ void __launch_bounds__(BLK) k_tab_from_masks(const void *kmers_, const uint8_t *mask, uint64_t D0, unsigned k, RankDir ix, node_t *tab,
                                                        uint32_t *err) {
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    for (uint64_t node = (uint64_t)blockIdx.x * BLK + threadIdx.x; node < 2 * D0; node += (uint64_t)gridDim.x * BLK) {
        const unsigned m = mask[node >> 1];
        const unsigned mo = ((node & 1) ? brev8(m) : m) & 15u;
        node_t e = (node_t)mo << TAB_OUT_SHIFT;
        if (uniq4(mo)) {
            unsigned yo;
            const Rec<NW> y = rec_canon<NW>(rec_shl<NW>(node_kmer<NW>(kmers, node, k), k, __ffs(mo) - 1), k, yo);
            const node_t ry = kmer_rank<NW, PRESENT>(kmers, ix, y);
            if (ry == NODE_NONE) atomicAdd(err, 1u);
            else e |= (ry << 1) | yo;
        }
        tab[node] = e;
    }
}

// successor of every non-junction node by lookup: GetOutgoing(kwh, GetUniqueOutgoing), debruijn_graph_constructor.hpp:228-235.
// Used after the early clippers changed the masks (the table k_fill_masks left behind describes the unclipped index).
template <int NW, class IX>
__global__ void __launch_bounds__(BLK) k_succ(IX ix, const uint8_t *mask, uint64_t D0, unsigned k, node_t *succ, uint32_t *err) {
    for (uint64_t node = (uint64_t)blockIdx.x * BLK + threadIdx.x; node < 2 * D0; node += (uint64_t)gridDim.x * BLK) {
        const uint64_t r = node >> 1;
        const unsigned o = (unsigned)(node & 1);
        const unsigned m = mask[r];
        if (mask_junction(m)) {
            succ[node] = NODE_NONE;
            continue;
        }
        const unsigned mo = o ? brev8(m) : m;
        Rec<NW> x = ix.kmer(r);
        if (o) x = rec_rc<NW>(x, k);
        unsigned yo;
        const unsigned c = __ffs(mo & 15) - 1;
        Rec<NW> y = rec_canon<NW>(rec_shl<NW>(x, k, c), k, yo);
        const node_t ry = ix.find(y);
        if (ry == NODE_NONE) atomicAdd(err, 1u);
        succ[node] = ry == NODE_NONE ? NODE_NONE : ((ry << 1) | yo | ((node_t)c << 62));
    }
}

// ---- early tip clipper (spades-core Construction stage; EarlyTipClipperProcessor, construction/early_simplification.hpp:38-162) ----
// One thread per oriented k-mer with >= 2 outgoing extensions (RemoveForward, :133-146). A branch is a tip if, from its first
// k-mer, the chain of non-junction k-mers (succ[], computed on the unclipped masks) reaches a dead end with a unique incoming
// extension within `bound` k-mers (FindForward, :102-112); tips shorter than the longest branch are removed (RemoveTips, :122-131).
// The reference updates the masks while it iterates; the decisions do not depend on that: a tip's k-mers have unique
// predecessors all the way back to ONE junction orientation, so no other junction ever walks them, and roots keep their phantom
// extension bits until the clean-up pass (checked against the reference run with 1-4 threads: tests/golden/etc_*). Hence three
// data-parallel passes: mark (reads the original masks), apply (IsolateVertex), fix (RemoveInconsistentForwardLinks, :21-36).
// Round 6: one lane per BRANCH, not per oriented k-mer. 4 % of the k-mers branch, so a kernel over all nodes has two or three working lanes per wave,
// each with a chain of ~25-95 dependent reads in front of it: latency times (nodes / 64) wave rounds — 1.75 s of a 2.9 s step at BASELINE config 3,
// measured (profiles/r06/bench_config3_sorted_route_early_tip_clipper.json). The branches are listed densely first (cand: the start de-edges of the
// junction k-mers as k_cand_expand lists them, (k-mer << 3 | orientation << 2 | nucleotide), the branches of one oriented k-mer side by side);
// k_tip_branch measures every branch of an orientation with >= 2 of them (0: none there), k_tip_decide compares a branch with its <= 3 siblings
// and isolates the tips that are shorter than the longest one.
constexpr uint32_t TIP_INF = 0xFFFFFFFFu;
template <int NW, class IX>
__global__ void __launch_bounds__(BLK) k_tip_branch(IX ix, const uint8_t *mask, const node_t *succ, const unsigned long long *__restrict__ cand, uint64_t C, unsigned k,
                                                    uint32_t bound, uint32_t *blen, node_t *bfirst, uint32_t *err) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < C; i += (uint64_t)gridDim.x * BLK) {
        const unsigned long long cd = cand[i];
        const uint64_t r = cd >> 3;
        const unsigned o = (unsigned)(cd >> 2) & 1u, c = (unsigned)(cd & 3);
        const unsigned m = mask[r];
        const unsigned mo = o ? brev8(m) : m;
        uint32_t len = 0;  // 0 = no branch to judge
        node_t first = NODE_NONE;
        if (__popc(mo & 15) >= 2) {
            Rec<NW> x = ix.kmer(r);
            if (o) x = rec_rc<NW>(x, k);
            unsigned yo;
            const Rec<NW> y = rec_canon<NW>(rec_shl<NW>(x, k, c), k, yo);
            int byte0;
            const node_t ry = ix.find_from(y, r, byte0);
            if (ry == NODE_NONE) {
                atomicAdd(err, 1u);
            } else {
                node_t nd = (ry << 1) | yo;
                uint32_t cnt = 0;
                first = nd;
                nd = ix.advance(succ, mask, nd, cnt, bound, byte0);
                if (nd == NODE_NONE) {
                    len = TIP_INF;
                } else {
                    ++cnt;
                    const unsigned ml = (nd & 1) ? brev8(mask[nd >> 1]) : mask[nd >> 1];
                    const bool tip = uniq4((ml >> 4) & 15) && (ml & 15) == 0;
                    len = tip ? cnt : TIP_INF;  // branching or too long: never removed, longer than any tip
                }
            }
        }
        blen[i] = len;
        bfirst[i] = first;
    }
}
// (The removed branch's own extension bit goes here as well — RemoveInconsistentForwardLinks, :21-36, deletes exactly the forward links of a tipped k-mer whose
// target lost its backward link, i.e. whose first k-mer was isolated: the removed branches, and no others — a tip's k-mers have ONE way in. A pass over all oriented
// k-mers looking for the few tipped ones (k_tip_fix, still what the A/T remover uses) was the longest kernel of the clipper: two or three working lanes per wave.)
template <class IX>
__global__ void __launch_bounds__(BLK) k_tip_decide(IX ix, const node_t *succ, const unsigned long long *__restrict__ cand, uint64_t C, const uint32_t *__restrict__ blen,
                                                    const node_t *__restrict__ bfirst, uint8_t *isolate, uint8_t *hmark, uint32_t *mask32, unsigned long long *stats) {
    __shared__ unsigned long long scratch[BLK / 64 + 2];
    unsigned long long n_kmers = 0, n_tips = 0;  // (summed per workgroup: one atomic per wave on ONE address was this kernel's whole time — 2.3 M of them at ~88 / us)
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < C; i += (uint64_t)gridDim.x * BLK) {
        const uint32_t len = blen[i];
        if (len == 0 || len == TIP_INF) continue;
        const unsigned long long key = cand[i] >> 2;  // the oriented k-mer the branch leaves
        uint32_t mx = len;
        for (int d = -3; d <= 3; ++d) {
            const int64_t j = (int64_t)i + d;
            if (d == 0 || j < 0 || j >= (int64_t)C) continue;
            if ((cand[j] >> 2) == key) mx = max(mx, blen[j]);
        }
        if (len >= mx) continue;
        ix.isolate_tip(succ, bfirst[i], len, isolate, hmark);
        {
            const uint64_t r = key >> 1;
            const unsigned o = (unsigned)(key & 1), c = (unsigned)(cand[i] & 3);
            atomicAnd(&mask32[r >> 2], ~((1u << (o ? 7 - c : c)) << ((r & 3) * 8)));  // DeleteOutgoing(kh, c), inout_mask.hpp:133-139
        }
        n_kmers += len;
        ++n_tips;
    }
    unsigned long long tot;
    block_excl_scan<unsigned long long>(n_kmers, scratch, &tot);
    if (threadIdx.x == 0 && tot) atomicAdd(&stats[0], tot);
    block_excl_scan<unsigned long long>(n_tips, scratch, &tot);
    if (threadIdx.x == 0 && tot) atomicAdd(&stats[1], tot);
}
__global__ void k_tip_apply(uint8_t *mask, const uint8_t *isolate, uint64_t D0) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < D0; r += (uint64_t)gridDim.x * blockDim.x)
        if (isolate[r]) mask[r] = 0;
}
template <int NW, class IX>
__global__ void __launch_bounds__(BLK) k_tip_fix(IX ix, uint32_t *mask32, const uint8_t *tipped, uint64_t D0, unsigned k, uint32_t *err) {
    const uint8_t *mask = (const uint8_t *)mask32;
    for (uint64_t node = (uint64_t)blockIdx.x * BLK + threadIdx.x; node < 2 * D0; node += (uint64_t)gridDim.x * BLK) {
        if (!tipped[node]) continue;
        const uint64_t r = node >> 1;
        const unsigned o = (unsigned)(node & 1);
        const unsigned m = mask[r];
        const unsigned mo = o ? brev8(m) : m;
        Rec<NW> x = ix.kmer(r);
        if (o) x = rec_rc<NW>(x, k);
        const unsigned firstn = rec_nucl<NW>(x, 0);
        for (unsigned c = 0; c < 4; ++c) {
            if (!(mo & (1u << c))) continue;
            unsigned yo;
            const Rec<NW> y = rec_canon<NW>(rec_shl<NW>(x, k, c), k, yo);
            const node_t ry = ix.find(y);
            if (ry == NODE_NONE) {
                atomicAdd(err, 1u);
                continue;
            }
            const unsigned mn = yo ? brev8(mask[ry]) : mask[ry];
            if (!(mn & (1u << (4 + firstn))))  // DeleteOutgoing(kh, c), inout_mask.hpp:133-139
                atomicAnd(&mask32[r >> 2], ~((1u << (o ? 7 - c : c)) << ((r & 3) * 8)));
        }
    }
}

// ---- early A/T remover (RNA pipelines; EarlyLowComplexityClipperProcessor(index, 0.8, 10, 200), early_simplification.hpp:164-347) ----
// nucleotide counts of a k-mer
template <int NW>
__device__ __forceinline__ void rec_counts(const Rec<NW> &x, unsigned k, unsigned (&cnt)[4]) {
    cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const unsigned nb = (unsigned)min((int)k - 32 * i, 32);  // nucleotides in this word
        if ((int)nb <= 0) break;
        const uint64_t valid = nb == 32 ? 0x5555555555555555ull : ((1ull << (2 * nb)) - 1) & 0x5555555555555555ull;
        const uint64_t lo = x.w[i] & 0x5555555555555555ull, hi = (x.w[i] >> 1) & 0x5555555555555555ull;
        cnt[1] += __popcll(lo & ~hi & valid);
        cnt[2] += __popcll(hi & ~lo & valid);
        cnt[3] += __popcll(hi & lo & valid);
        cnt[0] += __popcll(~hi & ~lo & valid);
    }
}
template <int NW>
__device__ __forceinline__ Rec<NW> rec_shr(const Rec<NW> &x, unsigned K, unsigned c) {  // operator>>: c + kmer[0..K-2]
    Rec<NW> r;
#pragma unroll
    for (int i = NW - 1; i > 0; --i) r.w[i] = (x.w[i] << 2) | (x.w[i - 1] >> 62);
    r.w[0] = (x.w[0] << 2) | (uint64_t)c;
    const unsigned tail = (K & 31) << 1;
    if (tail) r.w[NW - 1] &= (1ull << tail) - 1;
    return r;
}
// RemoveATEdges, :176-259, pass 1: edges of length 1 (the next k-mer is a junction or a dead end) leaving a low-complexity junction
// k-mer (some nucleotide occurs >= thr_edge times; thr_edge = smallest count that is not math::ls than 0.8 k, computed on the host)
template <int NW, class IX>
__global__ void __launch_bounds__(BLK) k_at_edges_mark(IX ix, const uint8_t *mask, uint64_t D0, unsigned k,
                                                       uint32_t thr_edge, uint8_t *atflag, uint32_t *err) {
    for (uint64_t node = (uint64_t)blockIdx.x * BLK + threadIdx.x; node < 2 * D0; node += (uint64_t)gridDim.x * BLK) {
        const uint64_t r = node >> 1;
        const unsigned o = (unsigned)(node & 1);
        const unsigned m = mask[r];
        if (!mask_junction(m)) continue;
        const unsigned mo = o ? brev8(m) : m;
        if ((mo & 15) == 0) continue;
        Rec<NW> x = ix.kmer(r);
        if (o) x = rec_rc<NW>(x, k);
        unsigned cnt[4];
        rec_counts<NW>(x, k, cnt);
        if (max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3])) < thr_edge) continue;
        unsigned fl = 0;
        for (unsigned c = 0; c < 4; ++c) {
            if (!(mo & (1u << c))) continue;
            unsigned yo;
            const Rec<NW> y = rec_canon<NW>(rec_shl<NW>(x, k, c), k, yo);
            const node_t ry = ix.find(y);
            if (ry == NODE_NONE) {
                atomicAdd(err, 1u);
                continue;
            }
            const unsigned mn = yo ? brev8(mask[ry]) : mask[ry];
            if (!mask_junction(mn) && (mn & 15) != 0) continue;
            fl |= 1u << c;
        }
        if (fl) atflag[node] = (uint8_t)fl;
    }
}
// pass 2: DeleteOutgoing(kh, c) + DeleteIncoming(next, kh[0]) for the marked edges (an edge marked from both of its ends clears
// the same two bits twice)
template <int NW, class IX>
__global__ void __launch_bounds__(BLK) k_at_edges_apply(IX ix, uint32_t *mask32, uint64_t D0, unsigned k,
                                                        const uint8_t *atflag, unsigned long long *stats, uint32_t *err) {
    for (uint64_t node = (uint64_t)blockIdx.x * BLK + threadIdx.x; node < 2 * D0; node += (uint64_t)gridDim.x * BLK) {
        const unsigned fl = atflag[node];
        if (!fl) continue;
        const uint64_t r = node >> 1;
        const unsigned o = (unsigned)(node & 1);
        Rec<NW> x = ix.kmer(r);
        if (o) x = rec_rc<NW>(x, k);
        const unsigned firstn = rec_nucl<NW>(x, 0);
        for (unsigned c = 0; c < 4; ++c) {
            if (!(fl & (1u << c))) continue;
            unsigned yo;
            const Rec<NW> y = rec_canon<NW>(rec_shl<NW>(x, k, c), k, yo);
            const node_t ry = ix.find(y);
            if (ry == NODE_NONE) {
                atomicAdd(err, 1u);
                continue;
            }
            atomicAnd(&mask32[r >> 2], ~((1u << (o ? 7 - c : c)) << ((r & 3) * 8)));
            const unsigned p = 4 + firstn;
            atomicAnd(&mask32[ry >> 2], ~((1u << (yo ? 7 - p : p)) << ((ry & 3) * 8)));
            atomicAdd(&stats[2], 1ull);
        }
    }
}
// RemoveATTips, :262-338: from every dead end with a unique incoming extension walk back to the junction the tip hangs on (at most
// max_len k-mers; succ[] of the opposite strand is the predecessor), count the last nucleotides of the tip k-mers (+ the root's up to
// min_len), remove the tip if one nucleotide makes up >= thr_tip[max(n, min_len)] of them.
template <int NW, class IX>
__global__ void __launch_bounds__(BLK) k_at_tips_mark(IX ix, const uint8_t *mask, const node_t *succ, uint64_t D0, unsigned k,
                                                      uint32_t min_len, uint32_t max_len, const uint16_t *thr_tip,
                                                      uint8_t *isolate, uint8_t *tipped, unsigned long long *stats, uint32_t *err) {
    for (uint64_t node = (uint64_t)blockIdx.x * BLK + threadIdx.x; node < 2 * D0; node += (uint64_t)gridDim.x * BLK) {
        const uint64_t r = node >> 1;
        const unsigned o = (unsigned)(node & 1);
        const unsigned m0 = mask[r];
        const unsigned mo0 = o ? brev8(m0) : m0;
        if ((mo0 & 15) != 0 || !uniq4((mo0 >> 4) & 15)) continue;  // start from tip ends
        Rec<NW> x = ix.kmer(r);
        if (o) x = rec_rc<NW>(x, k);
        unsigned cnt[4] = {0, 0, 0, 0};
        uint32_t n = 0;
        node_t nd = node;
        unsigned mo = mo0;
        bool bad = false;
        do {
            ++n;
            cnt[rec_nucl<NW>(x, k - 1)]++;
            const unsigned cin = __ffs((mo >> 4) & 15) - 1;
            x = rec_shr<NW>(x, k, cin);
            if (n == 1) {  // the dead end is a junction k-mer: its predecessor comes from a lookup, the rest by pointer chasing
                unsigned yo;
                const Rec<NW> y = rec_canon<NW>(x, k, yo);
                const node_t ry = ix.find(y);
                if (ry == NODE_NONE) {
                    bad = true;
                    break;
                }
                nd = (ry << 1) | yo;
            } else {
                const node_t sp = ix.next(succ, nd ^ 1);
                if (sp == NODE_NONE) {
                    bad = true;
                    break;
                }
                nd = sp ^ 1;
            }
            const unsigned mm = mask[nd >> 1];
            mo = (nd & 1) ? brev8(mm) : mm;
        } while (n < max_len && !mask_junction(mo));
        if (bad) {
            atomicAdd(err, 1u);
            continue;
        }
        if (((mo >> 4) & 15) == 0 || !mask_junction(mo)) continue;  // dead start, or the tip is too long
        for (uint32_t i = n - 1; i < min_len; ++i) cnt[rec_nucl<NW>(x, k - 1 - i)]++;
        const uint32_t curm = max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3]));
        if (curm < thr_tip[max(n, min_len)]) continue;
        // second walk: IsolateVertex on the n tip k-mers
        {
            node_t q = node;
            Rec<NW> z = ix.kmer(r);
            if (o) z = rec_rc<NW>(z, k);
            unsigned mq = mo0;
            for (uint32_t i = 0; i < n; ++i) {
                isolate[q >> 1] = 1;
                if (i + 1 == n) break;
                const unsigned cin = __ffs((mq >> 4) & 15) - 1;
                z = rec_shr<NW>(z, k, cin);
                if (i == 0) {
                    unsigned yo;
                    const Rec<NW> y = rec_canon<NW>(z, k, yo);
                    q = (ix.find(y) << 1) | yo;
                } else {
                    q = ix.next(succ, q ^ 1) ^ 1;
                }
                const unsigned mm = mask[q >> 1];
                mq = (q & 1) ? brev8(mm) : mm;
            }
        }
        tipped[nd] = 1;
        atomicAdd(&stats[0], (unsigned long long)n);
        atomicAdd(&stats[1], 1ull);
    }
}

// start de-edges per junction k-mer: out bits of kh, then out bits of !kh (AddStartDeEdges, :203-226), in k-mer-file order.
// Tiles of CAND_TILE ranks (8 consecutive ranks per thread): per-tile totals, a scan over the tiles, then every tile places its own.
constexpr int CAND_NJ = 256;  // partial junction counters
constexpr int CAND_PER = 8;
constexpr int CAND_TILE = BLK * CAND_PER;
__device__ __forceinline__ unsigned cand_of_mask(unsigned m) { return mask_junction(m) ? (unsigned)(__popc(m & 15) + __popc(m >> 4)) : 0u; }
__global__ void __launch_bounds__(BLK) k_cand_tiles(const uint8_t *mask, uint64_t D0, unsigned long long *tcnt, unsigned long long *njunction,
                                                    unsigned long long *tjcnt /* nullable: junction k-mers of every tile */) {
    __shared__ unsigned long long scratch[BLK / 64 + 2];
    const uint64_t r0 = (uint64_t)blockIdx.x * CAND_TILE + (uint64_t)threadIdx.x * CAND_PER;
    uint32_t c = 0, j_ = 0, jz = 0;
    static_assert(CAND_PER == 8, "a thread's masks are one 8-byte word");
    // (the thread's 8 bytes as ONE load where all of them exist, and the three counts through ONE block scan — 21 bits each: a tile has 2048 k-mers of at most 8 de-edges)
    uint64_t mm = 0;
    if (r0 + CAND_PER <= D0) mm = *reinterpret_cast<const uint64_t *>(mask + r0);
    else
        for (int j = 0; j < CAND_PER; ++j)
            if (r0 + j < D0) mm |= (uint64_t)mask[r0 + j] << (8 * j);
    for (int j = 0; j < CAND_PER; ++j)
        if (r0 + j < D0) {
            const unsigned m = (unsigned)(mm >> (8 * j)) & 0xFFu;
            c += cand_of_mask(m);
            j_ += mask_junction(m) ? 1u : 0u;
            jz += m ? 0u : 1u;  // isolated k-mers (an early clipper removed their tip): junction k-mers by the mask rule, without a single de-edge
        }
    unsigned long long all;
    block_excl_scan<unsigned long long>((unsigned long long)c | ((unsigned long long)j_ << 21) | ((unsigned long long)jz << 42), scratch, &all);
    const uint32_t tot = (uint32_t)(all & 0x1FFFFFu), jt = (uint32_t)((all >> 21) & 0x1FFFFFu), zt = (uint32_t)(all >> 42);
    if (threadIdx.x == 0) {
        tcnt[blockIdx.x] = tot;
        if (tjcnt) tjcnt[blockIdx.x] = jt - zt;  // (route 0 numbers the de-edges through the junction k-mers that HAVE some: smx_pm.hip, k_pm_junc_write)
        if (jt) atomicAdd(&njunction[blockIdx.x & (CAND_NJ - 1)], (unsigned long long)jt);  // spread: one address takes ~88 atomics/us
    }
}
__global__ void __launch_bounds__(BLK) k_cand_expand(const uint8_t *mask, const unsigned long long *toff, uint64_t D0, unsigned long long *cand) {
    __shared__ uint32_t scratch[BLK / 64 + 2];
    const uint64_t r0 = (uint64_t)blockIdx.x * CAND_TILE + (uint64_t)threadIdx.x * CAND_PER;
    uint32_t c = 0;
    uint64_t mm = 0;  // (the thread's 8 bytes as one load where all of them exist — k_cand_tiles does the same; bytes past D0 stay 0 = not a junction with de-edges)
    if (r0 + CAND_PER <= D0) mm = *reinterpret_cast<const uint64_t *>(mask + r0);
    else
        for (int j = 0; j < CAND_PER; ++j)
            if (r0 + j < D0) mm |= (uint64_t)mask[r0 + j] << (8 * j);
    for (int j = 0; j < CAND_PER; ++j)
        if (r0 + j < D0) c += cand_of_mask((unsigned)(mm >> (8 * j)) & 0xFFu);
    uint32_t tot;
    unsigned long long o = toff[blockIdx.x] + block_excl_scan<uint32_t>(c, scratch, &tot);
    for (int j = 0; j < CAND_PER; ++j) {
        const uint64_t r = r0 + j;
        if (r >= D0) break;
        const unsigned m = (unsigned)(mm >> (8 * j)) & 0xFFu;
        if (!mask_junction(m)) continue;
        for (unsigned cc = 0; cc < 4; ++cc)
            if (m & (1u << cc)) cand[o++] = (r << 3) | cc;
        const unsigned mi = brev8(m);
        for (unsigned cc = 0; cc < 4; ++cc)
            if (mi & (1u << cc)) cand[o++] = (r << 3) | 4u | cc;
    }
}

// ConstructSequenceWithEdge (:264-273), pass 1: length and end node of every start de-edge. One node-table read per step.
template <int NW>
__global__ void __launch_bounds__(BLK) k_walk_len(const unsigned long long *cand, uint64_t C, const void *kmers_,
                                                  const node_t *succ, unsigned k, RankDir ix,
                                                  uint64_t n_nodes, unsigned long long *len, node_t *first, node_t *last,
                                                  uint32_t *err) {
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < C; i += (uint64_t)gridDim.x * BLK) {
        const unsigned long long cd = cand[i];
        const unsigned c = (unsigned)(cd & 3);
        const Rec<NW> x = node_kmer<NW>(kmers, cd >> 2, k);  // cd >> 2 = 2 * rank + side
        unsigned yo;
        Rec<NW> y = rec_canon<NW>(rec_shl<NW>(x, k, c), k, yo);
        const node_t ry = kmer_rank<NW>(kmers, ix, y);
        if (ry == NODE_NONE) {
            atomicAdd(err, 1u);
            len[i] = 0;
            first[i] = last[i] = NODE_NONE;
            continue;
        }
        node_t node = (ry << 1) | yo;
        first[i] = node;
        uint64_t steps = 0;
        node_t nx;
        unsigned nuc;
        while (tab_step(succ, node, nx, nuc)) {  // a junction ends the path
            node = nx;
            if (++steps > n_nodes || node >= n_nodes) {  // cannot happen on a consistent index; never hang the GPU or leave the arrays
                atomicAdd(err, 1u);
                node = NODE_NONE;
                break;
            }
        }
        last[i] = node;
        len[i] = node == NODE_NONE ? 0 : k + 1 + steps;
        if (node == NODE_NONE) first[i] = NODE_NONE;
    }
}

// keep iff !(s < !s) (:305-306), decided before anything is written. RC(s) begins with the reverse complement of the last k-mer,
// i.e. with the oriented k-mer of node last^1, and s begins with the start k-mer: unless those two k-mers are equal they decide.
// If they are equal the path leads from the start node A to A^1 (a hairpin): RC(s) is then itself a path leaving A, and the two are
// compared nucleotide by nucleotide with two forward walkers. flags: bit 0 keep, bit 1 s == RC(s) (self-conjugate edge).
// kw[i] = 64-bit words of the packed sequence (0 if dropped), one[i] = keep. *interior += the non-junction k-mers (ranks) the kept
// paths run through: when that equals the number of non-junction k-mers there is no perfect loop left to look for.
template <int NW>
__global__ void __launch_bounds__(BLK) k_keep(const unsigned long long *cand, uint64_t C, const void *kmers_,
                                              const node_t *succ, unsigned k, const unsigned long long *len, const node_t *first,
                                              const node_t *last, uint8_t *flags, unsigned long long *kw, unsigned long long *one,
                                              unsigned long long *interior) {
    __shared__ unsigned long long scratch[BLK / 64 + 2];
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    unsigned long long inner = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < C; i += (uint64_t)gridDim.x * BLK) {
        const unsigned long long n = len[i];
        int cmp = -1;
        if (n) {
            const unsigned long long cd = cand[i];
            const node_t A = cd >> 2, L = last[i];
            const Rec<NW> x0 = node_kmer<NW>(kmers, A, k);
            cmp = rec_lex_cmp<NW>(x0, node_kmer<NW>(kmers, L ^ 1, k));
            if (cmp == 0) {
                // nodes after A: n_1 = first .. n_m = L = A^1, m = n - k. RC(s) = A, n_{m-1}^1, .., n_1^1, A^1.
                const unsigned long long m = n - k;
                node_t a = first[i], prev = A;
                for (unsigned long long t = 1; t < m; ++t) {
                    prev = a;
                    a = succ[a] & TAB_NODE_MASK;
                }
                // prev = n_{m-1} (A itself when m == 1); RC(s)[k] = complement of the first nucleotide of its k-mer
                const unsigned c2 = 3u - rec_nucl<NW>(node_kmer<NW>(kmers, prev, k), 0);
                const unsigned c1 = (unsigned)(cd & 3);
                cmp = c1 < c2 ? -1 : (c1 > c2 ? 1 : 0);
                a = first[i];
                node_t b = prev ^ 1;
                for (unsigned long long t = 1; t < m && cmp == 0; ++t) {  // interior nodes: exactly one outgoing bit each
                    const node_t ea = succ[a], eb = succ[b];
                    const unsigned na = __ffs(tab_out4(ea)) - 1, nb = __ffs(tab_out4(eb)) - 1;
                    cmp = na < nb ? -1 : (na > nb ? 1 : 0);
                    a = ea & TAB_NODE_MASK;
                    b = eb & TAB_NODE_MASK;
                }
            }
        }
        const bool keep = n > 0 && cmp >= 0;
        flags[i] = (uint8_t)((keep ? 1 : 0) | ((n > 0 && cmp == 0) ? 2 : 0));
        kw[i] = keep ? (n + 31) / 32 : 0;
        one[i] = keep ? 1 : 0;
        if (keep) inner += cmp == 0 ? (n - k - 1) / 2 : (n - k - 1);  // a self-conjugate path meets every rank twice
    }
    unsigned long long tot;
    block_excl_scan<unsigned long long>(inner, scratch, &tot);
    if (threadIdx.x == 0 && tot) atomicAdd(interior, tot);
}

// pass 2, kept paths only: the packed nucleotides (the start (k+1)-mer is the first NW words, then 2 bits per step, taken from the
// node table) at their final place, and the dense edge arrays
template <int NW>
__global__ void __launch_bounds__(BLK) k_walk_write(const unsigned long long *cand, uint64_t C, const void *kmers_,
                                                    const node_t *succ, unsigned k, const unsigned long long *len, const node_t *first,
                                                    const node_t *last, const uint8_t *flags, const unsigned long long *woff,
                                                    const unsigned long long *eidx, uint64_t *words, unsigned long long *eoffw,
                                                    unsigned long long *elen, node_t *estart, node_t *eend, uint8_t *eself) {
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < C; i += (uint64_t)gridDim.x * BLK) {
        if (!(flags[i] & 1)) continue;
        const unsigned long long cd = cand[i], n = len[i], e = eidx[i], wo = woff[i];
        const unsigned c = (unsigned)(cd & 3);
        const Rec<NW> x = node_kmer<NW>(kmers, cd >> 2, k);
        uint64_t *dst = words + wo;
#pragma unroll
        for (int w = 0; w < NW - 1; ++w) dst[w] = x.w[w];
        uint64_t cur = x.w[NW - 1] | ((uint64_t)c << ((k & 31) << 1));
        node_t node = first[i];
        for (unsigned long long p = k + 1; p < n; ++p) {
            if ((p & 31) == 0) {
                dst[(p >> 5) - 1] = cur;
                cur = 0;
            }
            const node_t en = succ[node];
            cur |= (uint64_t)(__ffs(tab_out4(en)) - 1) << ((p & 31) << 1);
            node = en & TAB_NODE_MASK;
        }
        dst[(n - 1) >> 5] = cur;
        eoffw[e] = wo;
        elen[e] = n;
        estart[e] = cd >> 2;
        eend[e] = last[i];
        eself[e] = (flags[i] >> 1) & 1;
    }
}
// only when the count of k_keep says that some non-junction k-mers are on no path: mark the ones that are
__global__ void __launch_bounds__(BLK) k_walk_mark(const unsigned long long *cand, uint64_t C, const node_t *succ, unsigned k,
                                                   const unsigned long long *len, const node_t *first, const uint8_t *flags, uint8_t *visited) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < C; i += (uint64_t)gridDim.x * BLK) {
        if (!(flags[i] & 1)) continue;
        const unsigned long long n = len[i];
        node_t node = first[i];
        for (unsigned long long p = k + 1; p < n; ++p) {
            visited[node >> 1] = 1;
            node = succ[node] & TAB_NODE_MASK;
        }
    }
}

// k-mers left on perfect loops: non-junction and not on any extracted path (CollectLoops pass 1, :362-374)
__global__ void k_loop_count(const uint8_t *mask, const uint8_t *visited, uint64_t D0, unsigned long long *count) {
    __shared__ unsigned long long scratch[BLK / 64 + 2];
    unsigned long long c = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < D0; r += (uint64_t)gridDim.x * blockDim.x)
        if (!mask_junction(mask[r]) && !visited[r]) ++c;
    unsigned long long tot;
    block_excl_scan<unsigned long long>(c, scratch, &tot);
    if (threadIdx.x == 0 && tot) atomicAdd(count, tot);
}
__global__ void k_loop_list(const uint8_t *mask, const uint8_t *visited, uint64_t D0, unsigned long long *count, unsigned long long *list,
                            unsigned long long cap) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < D0; r += (uint64_t)gridDim.x * blockDim.x) {
        if (!mask_junction(mask[r]) && !visited[r]) {
            const unsigned long long p = atomicAdd(count, 1ull);
            if (p < cap) list[p] = r;
        }
    }
}
template <int NW>
__global__ void k_gather_kmers(const void *kmers_, const uint8_t *mask, const unsigned long long *list, uint64_t n, void *out_, uint8_t *omask) {
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    Rec<NW> *out = (Rec<NW> *)out_;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        out[i] = kmers[list[i]];
        omask[i] = mask[list[i]];
    }
}

// ---- link records + vertices on the device (FastGraphFromSequencesConstructor::ConstructGraph, debruijn_graph_constructor.hpp:506-567) ----
// Record = two words: w0 = rank of the vertex k-mer (left-shifted by sh so that the MSD digits of the sorting pipeline see a
// spread-out fraction), w1 = EdgeAndMask (edge id << 2 | rc << 1 | is_start); the pipeline orders by (w0, w1) = CompareByVertexKMer-
// EdgeIdAndMask. The end record of a self-conjugate edge does not exist (LinkRecord() is invalid): it is emitted as a copy of the
// start record, which the sort + unique pipeline drops.
__global__ void k_link_keys(const node_t *estart, const node_t *eend, const uint8_t *eself, uint64_t ne, unsigned sh, Rec<2> *keys) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ne; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t edge = 3 + 2 * i;
        const node_t s = estart[i], e = eend[i];
        Rec<2> ks, ke;
        ks.w[0] = (s >> 1) << sh;
        ks.w[1] = (edge << 2) | ((s & 1) << 1) | 1ull;
        ke.w[0] = (e >> 1) << sh;
        ke.w[1] = (edge << 2) | ((e & 1) << 1);
        keys[2 * i] = ks;
        keys[2 * i + 1] = eself[i] ? ks : ke;
    }
}
// sorted records: one[i] = 1 where a new rank (vertex) starts
__global__ void k_vertex_flags(const Rec<2> *keys, uint64_t n, unsigned long long *one) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        one[i] = (i == 0 || keys[i].w[0] != keys[i - 1].w[0]) ? 1ull : 0ull;
}
// vertex v = the run of records starting at position i: vpos[v] = i, vkey[v] = (smallest EdgeAndMask of the vertex (shifted), v)
__global__ void k_vertex_collect(const Rec<2> *keys, const unsigned long long *one, const unsigned long long *vidx, uint64_t n,
                                 unsigned sh2, unsigned long long *vpos, Rec<2> *vkeys) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (!one[i]) continue;
        const unsigned long long v = vidx[i];
        vpos[v] = i;
        Rec<2> r;
        r.w[0] = keys[i].w[1] << sh2;
        r.w[1] = v;
        vkeys[v] = r;
    }
}
__global__ void k_vertex_permute(const Rec<2> *vkeys_sorted, const unsigned long long *vpos, uint64_t nv, unsigned long long *vstart) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nv; j += (uint64_t)gridDim.x * blockDim.x)
        vstart[j] = vpos[vkeys_sorted[j].w[1]];
}

// ---- sharded construction: owner-side mask fill -----------------------------------------------------------------
// Every rank holds a shard of the (k+1)-mer file. A (k+1)-mer sets one InOutMask bit in each of its two k-mers, and those live on
// the ranks that own THEIR buckets: the updates travel there in one all-to-all (precedent: hpcSPAdes counts per node and OR-reduces
// the masks, construction_mpi.cpp:343-354, partask_mpi.hpp:293). Update record = canonical k-mer (NW words) + one word = the bit.
// Rank r owns buckets [ceil(r*B/world), ceil((r+1)*B/world)), i.e. owner(b) = floor(b*world/B).
template <int NW>
struct Upd {
    uint64_t w[NW + 1];
};
template <int NW>
__device__ __forceinline__ void upd_of_kpo(const Rec<NW> &x, unsigned k, Upd<NW> &up, Upd<NW> &us) {
    const unsigned pn = rec_nucl<NW>(x, 0), nn = rec_nucl<NW>(x, k);
    unsigned prc, src;
    const Rec<NW> p = rec_canon<NW>(rec_prefix<NW>(x, k), k, prc), s = rec_canon<NW>(rec_suffix<NW>(x), k, src);
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        up.w[i] = p.w[i];
        us.w[i] = s.w[i];
    }
    up.w[NW] = prc ? 7 - nn : nn;       // out[prefix] |= bit(x_k), mirrored for a non-minimal key (inout_mask.hpp:92-94)
    us.w[NW] = src ? 3 - pn : pn + 4;   // in[suffix] |= bit(x_0)
}
template <int NW>
__device__ __forceinline__ uint32_t upd_owner(const Upd<NW> &u, uint32_t B, uint32_t world) {
    Rec<NW> r;
#pragma unroll
    for (int i = 0; i < NW; ++i) r.w[i] = u.w[i];
    return (uint32_t)(((uint64_t)bucket_of(xxh3_rec<NW>(r), B) * world) / B);
}
// pass 0: updates per owner; pass 1: place them (LDS counters per owner and tile, one reservation per owner and tile)
template <int NW, int PASS>
__global__ void __launch_bounds__(BLK) k_upd_partition(const void *kpo_, uint64_t n, unsigned k, uint32_t B, uint32_t world,
                                                       unsigned long long *hist_or_cursor, void *out_) {
    extern __shared__ unsigned long long lds_u[];  // [world] counts, then [world] bases
    unsigned long long *lcnt = lds_u, *lbase = lds_u + world;
    const Rec<NW> *kpo = (const Rec<NW> *)kpo_;
    Upd<NW> *out = (Upd<NW> *)out_;
    for (uint64_t base = (uint64_t)blockIdx.x * BLK; base < n; base += (uint64_t)gridDim.x * BLK) {
        for (uint32_t t = threadIdx.x; t < world; t += BLK) lcnt[t] = 0;
        __syncthreads();
        const uint64_t i = base + threadIdx.x;
        Upd<NW> up, us;
        uint32_t op = 0, os = 0;
        unsigned long long ip = 0, is = 0;
        if (i < n) {
            upd_of_kpo<NW>(kpo[i], k, up, us);
            op = upd_owner<NW>(up, B, world);
            os = upd_owner<NW>(us, B, world);
            ip = atomicAdd(&lcnt[op], 1ull);
            is = atomicAdd(&lcnt[os], 1ull);
        }
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < world; t += BLK)
            if (lcnt[t]) lbase[t] = atomicAdd(&hist_or_cursor[t], lcnt[t]);
        __syncthreads();
        if (PASS == 1 && i < n) {
            out[lbase[op] + ip] = up;
            out[lbase[os] + is] = us;
        }
        __syncthreads();
    }
}
template <int NW>
__global__ void k_upd_strip(const void *upd_, uint64_t n, void *out_) {
    const Upd<NW> *upd = (const Upd<NW> *)upd_;
    Rec<NW> *out = (Rec<NW> *)out_;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        Rec<NW> r;
#pragma unroll
        for (int w = 0; w < NW; ++w) r.w[w] = upd[i].w[w];
        out[i] = r;
    }
}
template <int NW>
__global__ void __launch_bounds__(BLK) k_upd_apply(const void *upd_, uint64_t n, const void *kmers_, RankDir ix, uint32_t *mask32, uint32_t *err) {
    const Upd<NW> *upd = (const Upd<NW> *)upd_;
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLK) {
        Rec<NW> r;
#pragma unroll
        for (int w = 0; w < NW; ++w) r.w[w] = upd[i].w[w];
        const node_t rk = kmer_rank<NW, true>(kmers, ix, r);
        if (rk == NODE_NONE) {
            atomicAdd(err, 1u);
            continue;
        }
        atomicOr(&mask32[rk >> 2], (1u << (upd[i].w[NW] & 7)) << ((rk & 3) * 8));
    }
}

// ---- packed unitigs <-> other forms -----------------------------------------------------------------------------------------
// (start, len) of every unitig as a read batch: the coverage pass marks its (k+1)-mer windows like those of reads
__global__ void k_edge_as_reads(const unsigned long long *eoffw, const unsigned long long *elen, uint64_t ne, uint64_t *start, uint32_t *len) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ne; i += (uint64_t)gridDim.x * blockDim.x) {
        start[i] = eoffw[i] * 32;
        len[i] = (uint32_t)min(elen[i], 0xFFFFFFFFull);
    }
}

// ---- coverage (-c) -----------------------------------------------------------------------------------------------
// CoverageHashMapBuilder::FillCoverageFromStream (kmer_index/ph_map/coverage_hash_map_builder.hpp:18-39): every
// (k+1)-mer instance of the read+RC stream whose orientation is minimal increments the counter of its canonical
// record. One window position contributes its canonical form once — twice when the (k+1)-mer is its own reverse
// complement (both strand instances are minimal then).
template <int NW>
__global__ void __launch_bounds__(BLK) k_kpo_coverage(const uint64_t *seq, const uint64_t *mask, uint64_t G, unsigned K1,
                                                      const void *kpo_, RankDir ix, uint32_t *cnt) {
    const Rec<NW> *kpo = (const Rec<NW> *)kpo_;
    for (uint64_t g = (uint64_t)blockIdx.x * BLK + threadIdx.x; g < G; g += (uint64_t)gridDim.x * BLK) {
        if (!((mask[g >> 6] >> (g & 63)) & 1)) continue;
        Rec<NW> x = load_window<NW>(seq, g, K1);
        Rec<NW> y = rec_rc<NW>(x, K1);
        const bool pal = rec_eq<NW>(x, y);
        const Rec<NW> c = rc_ge<NW>(y, x) ? x : y;
        const node_t r = kmer_rank<NW>(kpo, ix, c);
        if (r != NODE_NONE) atomicAdd(&cnt[r], pal ? 2u : 1u);
    }
}

// multiplicity histogram of the canonical (k+1)-mers (PHMCoverageFiller hands it to GenomicInfo::set_cov_histogram,
// stages/construction.cpp:414-431): hist[c] for c < COVH_N, larger multiplicities (repeats; rare) appended to a list
constexpr uint32_t COVH_LDS = 1024, COVH_N = 1u << 16;
__global__ void __launch_bounds__(BLK) k_cov_hist(const uint32_t *cnt, uint64_t n, unsigned long long *hist, unsigned long long *nbig, uint32_t *big,
                                                  uint32_t bigcap) {
    __shared__ uint32_t lh[COVH_LDS];
    for (uint32_t t = threadIdx.x; t < COVH_LDS; t += BLK) lh[t] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLK) {
        const uint32_t c = cnt[i];
        if (c < COVH_LDS) atomicAdd(&lh[c], 1u);
        else if (c < COVH_N) atomicAdd(&hist[c], 1ull);
        else {
            const unsigned long long p = atomicAdd(nbig, 1ull);
            if (p < bigcap) big[p] = c;
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < COVH_LDS; t += BLK)
        if (lh[t]) atomicAdd(&hist[t], (unsigned long long)lh[t]);
}

// edge raw coverage = sum of the counters of the edge's (k+1)-mers (graph_support/coverage_filling.hpp:46-62), uint32; flanking raw
// coverage = the same sum over the first `flank` (k+1)-mers (inc_coverage: offset < averaging_range, :40-44) of the edge (fl_s) and
// of its conjugate, i.e. the last `flank` ones (fl_e). The unitigs are a packed stream with word-aligned starts; wmask marks the
// positions where a (k+1)-mer of some unitig starts.
template <int NW>
__global__ void __launch_bounds__(BLK) k_edge_coverage(const uint64_t *words, const uint64_t *wmask, const unsigned long long *eoffw,
                                                       const unsigned long long *elen, uint64_t n_edges, uint64_t G,
                                                       unsigned K1, const void *kpo_, RankDir ix,
                                                       const uint32_t *cnt, uint32_t *ecov, uint32_t flank, uint32_t *fl_s, uint32_t *fl_e) {
    const Rec<NW> *kpo = (const Rec<NW> *)kpo_;
    for (uint64_t p = (uint64_t)blockIdx.x * BLK + threadIdx.x; p < G; p += (uint64_t)gridDim.x * BLK) {
        if (!((wmask[p >> 6] >> (p & 63)) & 1)) continue;
        const uint64_t pw = p >> 5;
        uint64_t lo = 0, hi = n_edges;  // edge e with eoffw[e] <= pw < eoffw[e+1]
        while (hi - lo > 1) {
            uint64_t mid = (lo + hi) >> 1;
            if (eoffw[mid] <= pw) lo = mid; else hi = mid;
        }
        const Rec<NW> x = load_window<NW>(words, p, K1);
        unsigned f;
        const Rec<NW> c = rec_canon<NW>(x, K1, f);
        const node_t r = kmer_rank<NW>(kpo, ix, c);
        if (r != NODE_NONE) {
            const uint32_t v = cnt[r];
            atomicAdd(&ecov[lo], v);
            const uint64_t j = p - eoffw[lo] * 32, nk = elen[lo] - K1 + 1;
            if (j < flank) atomicAdd(&fl_s[lo], v);
            if (j + flank >= nk) atomicAdd(&fl_e[lo], v);
        }
    }
}

}  // namespace smx
