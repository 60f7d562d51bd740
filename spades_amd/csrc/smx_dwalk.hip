// spades_amd/csrc/smx_dwalk.hip — distributed unitig walks: the k-mer-specific steps (included by smx_api.hip).
//
// SURVEY.md §8 row e2. The k-mer file stays sharded by bucket range (smx_graph_shard_from_ext / smx_graph_shard_build: this rank's
// sorted k-mers + InOutMask bytes), so a walk (UnbranchingPathExtractor::ConstructSequenceWithEdge,
// assembly_graph/construction/debruijn_graph_constructor.hpp:264-273) leaves its rank at every step. The walks are therefore not
// walked: every oriented non-junction k-mer asks the owner of its successor for that k-mer's place ONCE (k_dw_requests ->
// all-to-all -> k_dw_lookup), the chains of non-junction k-mers are ranked by pointer doubling over the exchange (spades_amd/dist.py,
// integer arrays only), every chain k-mer sends its outgoing nucleotide to the head of its chain, and the owner of the junction
// k-mer a unitig starts at assembles start (k+1)-mer + chain nucleotides, decides keep / drop exactly as the single-GPU route
// (k_keep: keep iff !(s < !s), :305-306) and hands out its kept unitigs in k-mer-file order (k_dw_keep, k_dw_write).
// Global node id = 2 * (rank of the k-mer in the whole file) + orientation, as in smx_graph.hip; a shard knows its k-mers by their
// LOCAL rank, the caller adds the shard's first global rank.
#pragma once
#include "smx_graph.hip"

namespace smx {

// tag of a request, kept by the asking rank in send order: item << 4 | is_cand << 3 | (successor is the RC of its canonical form) << 2
// | nucleotide. item = local oriented node (chain requests) or index of the start de-edge in this shard's k-mer-file order.
constexpr unsigned DW_TAG_SHIFT = 4;

template <int NW>
__device__ __forceinline__ uint32_t dw_owner(const Rec<NW> &canon, uint32_t B, uint32_t world) {
    return (uint32_t)(((uint64_t)bucket_of(xxh3_rec<NW>(canon), B) * world) / B);
}

// CAND = false: items are the 2 * D0 oriented nodes of the shard, those of junction k-mers ask nothing;
// CAND = true: items are the start de-edges (k_cand_expand). PASS 0 counts per owner, PASS 1 places record + tag.
template <int NW, bool CAND, int PASS>
__global__ void __launch_bounds__(BLK) k_dw_requests(const void *kmers_, const uint8_t *__restrict__ mask, const unsigned long long *__restrict__ cand,
                                                     uint64_t item0, uint64_t n_items, unsigned k, uint32_t B, uint32_t world,
                                                     unsigned long long *hist_or_cursor, void *out_, unsigned long long *tags) {
    extern __shared__ unsigned long long lds_dw[];  // [world] counts, then [world] bases
    unsigned long long *lcnt = lds_dw, *lbase = lds_dw + world;
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    Rec<NW> *out = (Rec<NW> *)out_;
    for (uint64_t base = (uint64_t)blockIdx.x * BLK; base < n_items; base += (uint64_t)gridDim.x * BLK) {
        for (uint32_t t = threadIdx.x; t < world; t += BLK) lcnt[t] = 0;
        __syncthreads();
        const uint64_t i = item0 + base + threadIdx.x;  // (the items [item0, item0 + n_items): a caller may ask range by range)
        bool asks = false;
        Rec<NW> y;
        uint32_t ow = 0;
        unsigned long long at = 0, tag = 0;
        if (base + threadIdx.x < n_items) {
            node_t node;
            unsigned c = 0;
            if (CAND) {
                const unsigned long long cd = cand[i];
                node = cd >> 2;
                c = (unsigned)(cd & 3);
                asks = true;
            } else {
                node = i;
                const unsigned m = mask[i >> 1];
                if (!mask_junction(m)) {
                    const unsigned out4 = (i & 1) ? (brev8(m) & 15u) : (m & 15u);
                    c = (unsigned)__ffs(out4) - 1;
                    asks = true;
                }
            }
            if (asks) {
                unsigned yo;
                y = rec_canon<NW>(rec_shl<NW>(node_kmer<NW>(kmers, node, k), k, c), k, yo);
                ow = dw_owner<NW>(y, B, world);
                at = atomicAdd(&lcnt[ow], 1ull);
                tag = ((unsigned long long)i << DW_TAG_SHIFT) | (CAND ? 8u : 0u) | (yo << 2) | c;
            }
        }
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < world; t += BLK)
            if (lcnt[t]) lbase[t] = atomicAdd(&hist_or_cursor[t], lcnt[t]);
        __syncthreads();
        if (PASS == 1 && asks) {
            out[lbase[ow] + at] = y;
            tags[lbase[ow] + at] = tag;
        }
        __syncthreads();
    }
}

// owner side: canonical k-mer -> local rank << 1 | (its mask makes it a junction k-mer); all ones = not in this shard
template <int NW>
__global__ void __launch_bounds__(BLK) k_dw_lookup(const void *recs_, uint64_t n, const void *kmers_, const uint8_t *__restrict__ mask, RankDir ix,
                                                   unsigned long long *reply) {
    const Rec<NW> *recs = (const Rec<NW> *)recs_, *kmers = (const Rec<NW> *)kmers_;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLK) {
        const node_t r = kmer_rank<NW>(kmers, ix, recs[i]);
        reply[i] = r == NODE_NONE ? ~0ull : ((r << 1) | (mask_junction(mask[r]) ? 1ull : 0ull));
    }
}

// nucleotide j of the unitig of start de-edge i: start k-mer, the de-edge's nucleotide, then one code per chain k-mer
template <int NW>
struct DwSeq {
    Rec<NW> x;
    unsigned k, c;
    const uint8_t *b;  // steps codes
    __device__ __forceinline__ unsigned at(unsigned long long j) const { return j < k ? rec_nucl<NW>(x, (unsigned)j) : (j == k ? c : (unsigned)b[j - k - 1] & 3u); }
};

// keep iff !(s < !s) (:305-306): flags bit 0 keep, bit 1 s == !s; kw = words of the kept unitig, one = keep
template <int NW>
__global__ void __launch_bounds__(BLK) k_dw_keep(const unsigned long long *cand, uint64_t C, const void *kmers_, unsigned k, const unsigned long long *steps,
                                                 const unsigned long long *boff, const uint8_t *bases, uint8_t *flags, unsigned long long *kw,
                                                 unsigned long long *one) {
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < C; i += (uint64_t)gridDim.x * BLK) {
        const unsigned long long cd = cand[i], n = k + 1 + steps[i];
        DwSeq<NW> s{node_kmer<NW>(kmers, cd >> 2, k), k, (unsigned)(cd & 3), bases + boff[i]};
        int cmp = 0;
        for (unsigned long long j = 0; j < n && cmp == 0; ++j) {
            const unsigned a = s.at(j), r = 3u - s.at(n - 1 - j);
            cmp = a < r ? -1 : (a > r ? 1 : 0);
        }
        const bool keep = cmp >= 0;
        flags[i] = (uint8_t)((keep ? 1 : 0) | (cmp == 0 ? 2 : 0));
        kw[i] = keep ? (n + 31) / 32 : 0;
        one[i] = keep ? 1 : 0;
    }
}

// the kept unitigs, packed (every unitig starts on a word boundary), and their edge arrays; node_base = 2 * first global rank of the shard
template <int NW>
__global__ void __launch_bounds__(BLK) k_dw_write(const unsigned long long *cand, uint64_t C, const void *kmers_, unsigned k, const unsigned long long *steps,
                                                  const unsigned long long *last, const unsigned long long *boff, const uint8_t *bases, const uint8_t *flags,
                                                  const unsigned long long *woff, const unsigned long long *eidx, unsigned long long node_base, uint64_t *words,
                                                  unsigned long long *eoffw, unsigned long long *elen, node_t *estart, node_t *eend, uint8_t *eself) {
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < C; i += (uint64_t)gridDim.x * BLK) {
        if (!(flags[i] & 1)) continue;
        const unsigned long long cd = cand[i], n = k + 1 + steps[i], e = eidx[i], wo = woff[i];
        DwSeq<NW> s{node_kmer<NW>(kmers, cd >> 2, k), k, (unsigned)(cd & 3), bases + boff[i]};
        uint64_t *dst = words + wo;
        uint64_t cur = 0;
        for (unsigned long long p = 0; p < n; ++p) {
            cur |= (uint64_t)s.at(p) << ((p & 31) << 1);
            if ((p & 31) == 31) {
                dst[p >> 5] = cur;
                cur = 0;
            }
        }
        if (n & 31) dst[(n - 1) >> 5] = cur;
        eoffw[e] = wo;
        elen[e] = n;
        estart[e] = node_base + (cd >> 2);
        eend[e] = last[i];
        eself[e] = (flags[i] >> 1) & 1;
    }
}

// word offsets of gathered unitigs: words of every edge (scanned by the caller)
__global__ void k_dw_words_of(const unsigned long long *elen, uint64_t ne, unsigned long long *w) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ne; i += (uint64_t)gridDim.x * blockDim.x) w[i] = (elen[i] + 31) / 32;
}

}  // namespace smx
