// spades_amd/csrc/smx_pm.hip — construction without the sort of the k-mers: nodes numbered by minimizer partition ("pm" route).
// Included by smx_api.hip after smx_graph.hip (one translation unit).
//
// The sorted k-mer file of the reference (KMerDiskCounter, kmer_index/kmer_mph/kmer_index_builder.hpp:306-332) exists to give every
// k-mer an index (KMerIndex::seq_idx, kmer_index.hpp:88-100) — and those indices never reach the output
// (assembly_graph/construction/debruijn_graph_constructor.hpp:540-547: link records are only GROUPED by them). What the output does
// depend on is the ORDER in which the junction k-mers are visited (AddStartDeEdges in k-mer-file order, :203-226). So here:
//   * a node is (index of the k-mer in the dedupe stage's output) * 2 + orientation. That output is partition-major: the distinct
//     k-mers of ~14 minimizer partitions per chunk, consecutive k-mers of a genomic path share their minimizer for ~17 steps, so
//     successor lookups and walk steps stay inside a few KB instead of hitting a random bucket of a hash-ordered file;
//   * only the junction k-mers (4 % of the k-mers) are sorted into the reference's file order, to number the start de-edges.
// Lookup of a k-mer: canonical minimizer -> partition -> chunk (pinfo), then the chunk's LDS hash table is probed again in HBM
// through its occupancy words (smx_superkmer.hip, PmOut). k-mers of partitions that were cut by the chunk capacity sit in a sorted
// tail behind the chunks ("dirty region") with a rank directory of their own.
#pragma once
#include "smx_graph.hip"
#include "smx_superkmer.hip"

namespace smx {

struct PmIndex {
    const void *recs;                 // EXT records: clean chunks [0, nclean), then the dirty region (sorted)
    const unsigned long long *pinfo;  // [partitions]
    const uint32_t *meta;             // [chunks * ngroups] group words
    uint32_t T, ngroups;              // table slots per chunk, T / 16
    unsigned K, m, w, pshift;
    uint64_t nclean;
    const void *dk;                   // dirty region without the bytes
    RankDir ddir;                     // its rank directory (one bucket)
};

struct PmWalk {  // what graph_from_masks (smx_construct.hpp) needs beyond the node table on this route
    PmIndex ix;
    const uint32_t *jmp;
};

// minimizer partition of a k-mer (either orientation: the m-mer keys are those of the canonical m-mers) — the same function the
// super-k-mer scan applies to the windows of the reads (k_skm_scan: skm_key of every m-mer, minimum, skm_part)
template <int NW>
__device__ __forceinline__ uint32_t pm_partition(const Rec<NW> &y, const PmIndex &ix) {
    const uint32_t mmask = ix.m >= 16 ? 0xFFFFFFFFu : ((1u << (2 * ix.m)) - 1);
    uint32_t best = 0xFFFFFFFFu;
#pragma unroll
    for (int wi = 0; wi < NW; ++wi) {
        const uint64_t lo = y.w[wi], hi = wi + 1 < NW ? y.w[wi + 1] : 0ull;
        const unsigned q0 = 32u * wi;
        if (q0 >= ix.w) break;
        const unsigned qn = min(32u, ix.w - q0);
        for (unsigned qq = 0; qq < qn; ++qq) {
            const unsigned sh = qq << 1;
            const uint64_t v = (lo >> sh) | ((hi << (63 - sh)) << 1);
            best = min(best, skm_key((uint32_t)v & mmask, ix.m));
        }
    }
    return skm_part(best, ix.pshift);
}

// probe of one chunk's hash table through its group words (gm: ngroups words, in LDS or HBM); y = canonical k-mer without byte
template <int NW>
__device__ __forceinline__ node_t pm_probe(const Rec<NW> *__restrict__ recs, const uint32_t *gm, uint32_t T, uint64_t base, const Rec<NW> &y, uint32_t h32) {
    uint32_t h = h32 & (T - 1);
    for (uint32_t it = 0; it < T; ++it) {
        const uint32_t g = gm[h >> 4];
        const uint32_t occ = g >> 16, bit = h & 15u;
        if (!((occ >> bit) & 1u)) return NODE_NONE;
        const uint64_t idx = base + (g & 0xFFFFu) + __popc(occ & ((1u << bit) - 1u));
        if (rec_eq<NW>(rec_pure<NW>(recs[idx]), y)) return idx;
        h = (h + 1) & (T - 1);
    }
    return NODE_NONE;
}
// index (rank in the partition-major numbering) of a canonical k-mer, NODE_NONE if the graph does not have it
template <int NW>
__device__ __forceinline__ node_t pm_find(const PmIndex &ix, const Rec<NW> &y) {
    const unsigned long long pi = ix.pinfo[pm_partition<NW>(y, ix)];
    if (pi == PM_EMPTY) return NODE_NONE;
    if (pi == PM_DIRTY) {
        const node_t r = kmer_rank<NW, false>((const Rec<NW> *)ix.dk, ix.ddir, y);
        return r == NODE_NONE ? NODE_NONE : ix.nclean + r;
    }
    return pm_probe<NW>((const Rec<NW> *)ix.recs, ix.meta + (size_t)(pi >> PM_BASE_BITS) * ix.ngroups, ix.T, pi & PM_BASE_MASK, y, rec_hash32<NW>(y));
}
template <int NW>
__device__ __forceinline__ Rec<NW> pm_node_kmer(const Rec<NW> *__restrict__ recs, node_t node, unsigned k) {  // oriented k-mer of a node
    const Rec<NW> x = rec_pure<NW>(recs[node >> 1]);
    return (node & 1) ? rec_rc<NW>(x, k) : x;
}

// Jump words: jmp[node] = (delta to the last node of the chain that stays inside the node's chunk, 16 bits signed) | steps << 16.
// A walk that enters a chunk reads ONE word to cross it (smx_pm_walk_len) instead of one node-table entry per k-mer.
constexpr uint16_t PM_ADV_NONE = 0xFFFFu;

// Node table of the clean chunks: one workgroup per chunk. Per node: outgoing extensions from the mask byte; where there is exactly
// one, the successor k-mer is looked up — first in the chunk itself (its group words are staged in LDS; consecutive k-mers of a path
// share their minimizer, ~9 in 10 are found here), else through the partition table. Then the chains inside the chunk are followed
// in LDS to write the jump words. stats: [0] extension bits, [1] palindromic (k+1)-mers among them (k_ext_split's figures).
// LDS (dynamic): gm[ngroups] u32 | adv[2 * maxn] u16
template <int NW>
__global__ void __launch_bounds__(BLK) k_pm_tab(PmIndex ix, const unsigned long long *__restrict__ cinfo, uint32_t nchunks, uint32_t maxn, unsigned k, node_t *tab,
                                                uint32_t *jmp, unsigned long long *stats, uint32_t *err) {
    extern __shared__ __attribute__((aligned(16))) uint32_t pm_lds[];
    uint32_t *gm = pm_lds;
    uint16_t *adv = (uint16_t *)(gm + ix.ngroups);
    const Rec<NW> *recs = (const Rec<NW> *)ix.recs;
    unsigned long long bits = 0, pals = 0;
    for (uint32_t cid = blockIdx.x; cid < nchunks; cid += gridDim.x) {
        const unsigned long long ci = cinfo[cid];
        const uint64_t base = ci & PM_BASE_MASK;
        const uint32_t n = (uint32_t)(ci >> PM_BASE_BITS);
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < ix.ngroups; t += BLK) gm[t] = ix.meta[(size_t)cid * ix.ngroups + t];
        __syncthreads();
        if (n > maxn) {  // cannot happen (a chunk never has more winners than its capacity): never leave the LDS arrays
            if (threadIdx.x == 0) atomicAdd(err, 1u);
            continue;
        }
        for (uint32_t r = threadIdx.x; r < n; r += BLK) {
            const Rec<NW> raw = recs[base + r];
            const Rec<NW> x = rec_pure<NW>(raw);
            const unsigned m = (unsigned)(raw.w[NW - 1] & 0xFFu);
            bits += __popc(m);
            {
                const unsigned x0 = rec_nucl<NW>(x, 0), xl = rec_nucl<NW>(x, k - 1);
                if (((m >> (3 - x0)) & 1) && range_is_rc_palindrome<NW>(x, 1, k - 1)) ++pals;
                if (((m >> (7 - xl)) & 1) && range_is_rc_palindrome<NW>(x, 0, k - 1)) ++pals;
            }
            const bool junction = mask_junction(m);
#pragma unroll
            for (unsigned o = 0; o < 2; ++o) {
                const unsigned mo = (o ? brev8(m) : m) & 15u;
                node_t e = (node_t)mo << TAB_OUT_SHIFT;
                uint16_t a = PM_ADV_NONE;
                if (uniq4(mo)) {
                    unsigned yo;
                    const Rec<NW> xo = o ? rec_rc<NW>(x, k) : x;
                    const Rec<NW> y = rec_canon<NW>(rec_shl<NW>(xo, k, __ffs(mo) - 1), k, yo);
                    node_t ry = pm_probe<NW>(recs, gm, ix.T, base, y, rec_hash32<NW>(y));
                    if (ry != NODE_NONE) {
                        if (!junction) a = (uint16_t)(((uint32_t)(ry - base) << 1) | yo);
                    } else {
                        ry = pm_find<NW>(ix, y);
                    }
                    if (ry == NODE_NONE) atomicAdd(err, 1u);
                    else e |= (ry << 1) | yo;
                }
                tab[2 * (base + r) + o] = e;
                adv[2 * r + o] = a;
            }
        }
        __syncthreads();
        const uint32_t nn = 2 * n;
        for (uint32_t nd = threadIdx.x; nd < nn; nd += BLK) {
            uint32_t cur = nd, s = 0;
            for (uint32_t a; (a = adv[cur]) != PM_ADV_NONE && s < nn; ++s) cur = a;
            jmp[2 * base + nd] = ((uint32_t)(cur - nd) & 0xFFFFu) | (s << 16);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        bits += __shfl_down(bits, o, 64);
        pals += __shfl_down(pals, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (bits) atomicAdd(&stats[0], bits);
        if (pals) atomicAdd(&stats[1], pals);
    }
}
// ... and of the dirty region: every successor through the partition table; no jumps (delta 0, 0 steps)
template <int NW>
__global__ void __launch_bounds__(BLK) k_pm_tab_dirty(PmIndex ix, uint64_t nd, unsigned k, node_t *tab, uint32_t *jmp, unsigned long long *stats, uint32_t *err) {
    const Rec<NW> *recs = (const Rec<NW> *)ix.recs;
    unsigned long long bits = 0, pals = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < nd; i += (uint64_t)gridDim.x * BLK) {
        const uint64_t r = ix.nclean + i;
        const Rec<NW> raw = recs[r];
        const Rec<NW> x = rec_pure<NW>(raw);
        const unsigned m = (unsigned)(raw.w[NW - 1] & 0xFFu);
        bits += __popc(m);
        const unsigned x0 = rec_nucl<NW>(x, 0), xl = rec_nucl<NW>(x, k - 1);
        if (((m >> (3 - x0)) & 1) && range_is_rc_palindrome<NW>(x, 1, k - 1)) ++pals;
        if (((m >> (7 - xl)) & 1) && range_is_rc_palindrome<NW>(x, 0, k - 1)) ++pals;
#pragma unroll
        for (unsigned o = 0; o < 2; ++o) {
            const unsigned mo = (o ? brev8(m) : m) & 15u;
            node_t e = (node_t)mo << TAB_OUT_SHIFT;
            if (uniq4(mo)) {
                unsigned yo;
                const Rec<NW> xo = o ? rec_rc<NW>(x, k) : x;
                const Rec<NW> y = rec_canon<NW>(rec_shl<NW>(xo, k, __ffs(mo) - 1), k, yo);
                const node_t ry = pm_find<NW>(ix, y);
                if (ry == NODE_NONE) atomicAdd(err, 1u);
                else e |= (ry << 1) | yo;
            }
            tab[2 * r + o] = e;
            jmp[2 * r + o] = 0;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        bits += __shfl_down(bits, o, 64);
        pals += __shfl_down(pals, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (bits) atomicAdd(&stats[0], bits);
        if (pals) atomicAdd(&stats[1], pals);
    }
}
// dirty region: k-mers without their bytes (what the rank directory indexes) + the bytes into the mask array
template <int NW>
__global__ void k_pm_dirty_split(const void *recs_, uint64_t nclean, uint64_t nd, void *dk_, uint8_t *mask) {
    const Rec<NW> *recs = (const Rec<NW> *)recs_;
    Rec<NW> *dk = (Rec<NW> *)dk_;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nd; i += (uint64_t)gridDim.x * blockDim.x) {
        const Rec<NW> raw = recs[nclean + i];
        dk[i] = rec_pure<NW>(raw);
        mask[nclean + i] = (uint8_t)(raw.w[NW - 1] & 0xFFu);
    }
}

// ---- the reference's order of the start de-edges ------------------------------------------------------------------------------
// Junction k-mers (EXT records, byte included) compacted in node order; tiles as k_cand_tiles (which also counts them per tile).
template <int NW>
__global__ void __launch_bounds__(BLK) k_pm_junc_write(const uint8_t *mask, const void *recs_, const unsigned long long *tjoff, uint64_t D0, void *out_) {
    __shared__ uint32_t scratch[BLK / 64 + 2];
    const Rec<NW> *recs = (const Rec<NW> *)recs_;
    Rec<NW> *out = (Rec<NW> *)out_;
    const uint64_t r0 = (uint64_t)blockIdx.x * CAND_TILE + (uint64_t)threadIdx.x * CAND_PER;
    uint32_t c = 0, fl = 0;
    for (int j = 0; j < CAND_PER; ++j)
        if (r0 + j < D0 && mask_junction(mask[r0 + j])) {
            fl |= 1u << j;
            ++c;
        }
    uint32_t tot;
    unsigned long long o = tjoff[blockIdx.x] + block_excl_scan<uint32_t>(c, scratch, &tot);
    for (int j = 0; j < CAND_PER; ++j)
        if (fl & (1u << j)) out[o++] = recs[r0 + j];
}
// start de-edges of every sorted junction k-mer (the reference's enumeration: its position in the file, then the de-edge)
__global__ void k_pm_cand_counts(const uint8_t *jm, uint64_t nj, unsigned long long *cnt) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nj; i += (uint64_t)gridDim.x * blockDim.x) cnt[i] = cand_of_mask(jm[i]);
}
// k_cand_expand in node order + q[o] = number of the de-edge in the reference's order: first de-edge of the junction k-mer in the
// sorted junction file (rank lookup, candoff) + its place among the k-mer's de-edges
template <int NW>
__global__ void __launch_bounds__(BLK) k_pm_cand_expand(const uint8_t *mask, const void *recs_, const unsigned long long *toff, uint64_t D0, const void *jk_,
                                                        RankDir jix, const unsigned long long *candoff, unsigned long long *cand, unsigned long long *q,
                                                        uint32_t *err) {
    __shared__ uint32_t scratch[BLK / 64 + 2];
    const Rec<NW> *recs = (const Rec<NW> *)recs_;
    const Rec<NW> *jk = (const Rec<NW> *)jk_;
    const uint64_t r0 = (uint64_t)blockIdx.x * CAND_TILE + (uint64_t)threadIdx.x * CAND_PER;
    uint32_t c = 0;
    for (int j = 0; j < CAND_PER; ++j)
        if (r0 + j < D0) c += cand_of_mask(mask[r0 + j]);
    uint32_t tot;
    unsigned long long o = toff[blockIdx.x] + block_excl_scan<uint32_t>(c, scratch, &tot);
    for (int j = 0; j < CAND_PER; ++j) {
        const uint64_t r = r0 + j;
        if (r >= D0) break;
        const unsigned m = mask[r];
        if (!mask_junction(m)) continue;
        const node_t jr = kmer_rank<NW, false>(jk, jix, rec_pure<NW>(recs[r]));
        unsigned long long qq = 0;
        if (jr == NODE_NONE) atomicAdd(err, 1u);
        else qq = candoff[jr];
        for (unsigned cc = 0; cc < 4; ++cc)
            if (m & (1u << cc)) {
                cand[o] = (r << 3) | cc;
                q[o++] = qq++;
            }
        const unsigned mi = brev8(m);
        for (unsigned cc = 0; cc < 4; ++cc)
            if (mi & (1u << cc)) {
                cand[o] = (r << 3) | 4u | cc;
                q[o++] = qq++;
            }
    }
}

// ---- walks ----------------------------------------------------------------------------------------------------------------------
// k_walk_len on the partition-major numbering: the first node by pm_find, chunks crossed by their jump words
template <int NW>
__global__ void __launch_bounds__(BLK) k_pm_walk_len(const unsigned long long *cand, uint64_t C, PmIndex ix, const node_t *tab, const uint32_t *jmp, unsigned k,
                                                     uint64_t n_nodes, unsigned long long *len, node_t *first, node_t *last, uint32_t *err) {
    const Rec<NW> *recs = (const Rec<NW> *)ix.recs;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < C; i += (uint64_t)gridDim.x * BLK) {
        const unsigned long long cd = cand[i];
        const Rec<NW> x = pm_node_kmer<NW>(recs, cd >> 2, k);
        unsigned yo;
        const Rec<NW> y = rec_canon<NW>(rec_shl<NW>(x, k, (unsigned)(cd & 3)), k, yo);
        const node_t ry = pm_find<NW>(ix, y);
        if (ry == NODE_NONE) {
            atomicAdd(err, 1u);
            len[i] = 0;
            first[i] = last[i] = NODE_NONE;
            continue;
        }
        node_t node = (ry << 1) | yo;
        first[i] = node;
        uint64_t steps = 0;
        for (;;) {
            const uint32_t j = jmp[node];  // to the end of the chain inside this chunk: non-junction k-mers all the way, the last one may be a junction
            node = (node_t)((long long)node + (long long)(int16_t)(j & 0xFFFFu));
            steps += j >> 16;
            node_t nx;
            unsigned nuc;
            if (node >= n_nodes || !tab_step(tab, node, nx, nuc)) break;  // a junction ends the path
            node = nx;
            if (++steps > n_nodes || node >= n_nodes) {  // cannot happen on a consistent index; never hang the GPU or leave the arrays
                node = n_nodes;
                break;
            }
        }
        if (node >= n_nodes) {
            atomicAdd(err, 1u);
            node = NODE_NONE;
        }
        last[i] = node;
        len[i] = node == NODE_NONE ? 0 : k + 1 + steps;
        if (node == NODE_NONE) first[i] = NODE_NONE;
    }
}
// k_keep with EXT records; the flags / word counts leave in node order, vq[q[i]] = words << 1 | keep in the reference's order
template <int NW>
__global__ void __launch_bounds__(BLK) k_pm_keep(const unsigned long long *cand, const unsigned long long *q, uint64_t C, const void *recs_, const node_t *succ,
                                                 unsigned k, const unsigned long long *len, const node_t *first, const node_t *last, uint8_t *flags,
                                                 unsigned long long *vq, unsigned long long *interior) {
    __shared__ unsigned long long scratch[BLK / 64 + 2];
    const Rec<NW> *recs = (const Rec<NW> *)recs_;
    unsigned long long inner = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < C; i += (uint64_t)gridDim.x * BLK) {
        const unsigned long long n = len[i];
        int cmp = -1;
        if (n) {
            const unsigned long long cd = cand[i];
            const node_t A = cd >> 2, L = last[i];
            const Rec<NW> x0 = pm_node_kmer<NW>(recs, A, k);
            cmp = rec_lex_cmp<NW>(x0, pm_node_kmer<NW>(recs, L ^ 1, k));
            if (cmp == 0) {  // hairpin: see k_keep
                const unsigned long long m = n - k;
                node_t a = first[i], prev = A;
                for (unsigned long long t = 1; t < m; ++t) {
                    prev = a;
                    a = succ[a] & TAB_NODE_MASK;
                }
                const unsigned c2 = 3u - rec_nucl<NW>(pm_node_kmer<NW>(recs, prev, k), 0);
                const unsigned c1 = (unsigned)(cd & 3);
                cmp = c1 < c2 ? -1 : (c1 > c2 ? 1 : 0);
                a = first[i];
                node_t b = prev ^ 1;
                for (unsigned long long t = 1; t < m && cmp == 0; ++t) {
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
        vq[q[i]] = keep ? ((((n + 31) / 32) << 1) | 1ull) : 0ull;
        if (keep) inner += cmp == 0 ? (n - k - 1) / 2 : (n - k - 1);
    }
    unsigned long long tot;
    block_excl_scan<unsigned long long>(inner, scratch, &tot);
    if (threadIdx.x == 0 && tot) atomicAdd(interior, tot);
}
__global__ void k_pm_unpack(const unsigned long long *vq, uint64_t C, unsigned long long *kw, unsigned long long *one) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < C; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long v = vq[i];
        kw[i] = v >> 1;
        one[i] = v & 1ull;
    }
}
// k_walk_write: the kept paths, walked in node order, written at their place in the reference's order (woffq / eidxq are indexed by q)
template <int NW>
__global__ void __launch_bounds__(BLK) k_pm_walk_write(const unsigned long long *cand, const unsigned long long *q, uint64_t C, const void *recs_, const node_t *succ,
                                                       unsigned k, const unsigned long long *len, const node_t *first, const node_t *last, const uint8_t *flags,
                                                       const unsigned long long *woffq, const unsigned long long *eidxq, uint64_t *words,
                                                       unsigned long long *eoffw, unsigned long long *elen, node_t *estart, node_t *eend, uint8_t *eself) {
    const Rec<NW> *recs = (const Rec<NW> *)recs_;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < C; i += (uint64_t)gridDim.x * BLK) {
        if (!(flags[i] & 1)) continue;
        const unsigned long long qi = q[i];
        const unsigned long long cd = cand[i], n = len[i], e = eidxq[qi], wo = woffq[qi];
        const unsigned c = (unsigned)(cd & 3);
        const Rec<NW> x = pm_node_kmer<NW>(recs, cd >> 2, k);
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

// Numbering-independent fingerprint of the link structure (smx_graph_fingerprint_portable): vertices in id order, the EdgeAndMask
// words of each in record order. out[0] += sum of words, out[1] += sum of word * (2 * (vertex * 64 + place) + 1) (mod 2^64).
__global__ void __launch_bounds__(BLK) k_pm_link_fingerprint(const Rec<2> *lrecs, uint64_t nrec, const unsigned long long *vstart, uint64_t nv, unsigned long long *out) {
    unsigned long long s0 = 0, s1 = 0;
    for (uint64_t v = (uint64_t)blockIdx.x * BLK + threadIdx.x; v < nv; v += (uint64_t)gridDim.x * BLK) {
        uint64_t i = vstart[v];
        const uint64_t key = lrecs[i].w[0];
        for (uint64_t j = 0; i < nrec && lrecs[i].w[0] == key; ++i, ++j) {
            const unsigned long long w = lrecs[i].w[1];
            s0 += w;
            s1 += w * (2ull * (v * 64 + j) + 1ull);
        }
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

}  // namespace smx
