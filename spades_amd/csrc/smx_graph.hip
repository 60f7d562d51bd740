// spades_amd/csrc/smx_graph.hip — de Bruijn construction on the device results of the counting path.
// Included by smx_api.hip (one translation unit).
//
// Reference rows (paths relative to /root/reference/src/common):
//   a13 DeBruijnKMerKMerSplitter        kmer_index/kmer_mph/kmer_splitters.hpp:138-207      -> k_derive_kmers + count
//   a15 FillExtensionsFromIndex/InOutMask  extension_index/kmer_extension_index_builder.hpp:45-60,
//                                           extension_index/inout_mask.hpp:92-131           -> k_fill_masks
//   a16 UnbranchingPathExtractor        assembly_graph/construction/debruijn_graph_constructor.hpp:184-410
//                                        -> k_succ, k_cand_*, k_walk_len, k_walk_write, k_keep, k_gather (+ host loops)
//   a17 FastGraphFromSequencesConstructor  same file :412-568                               -> host link records
//   a19 gfa::GFAWriter                  io/graph/gfa_writer.cpp:19-47,73-87,113-116         -> host writer
//
// Device layout: k-mers = the sorted-unique k-mer "file" (bucket-major, B = 10*threads), rank r = position in it
// (stands in for the MPHF index, whose values never reach the output: debruijn_graph_constructor.hpp:540-547).
// One byte mask per rank. A *node* is an oriented k-mer: node = 2*rank + o, o=1 meaning RC of the stored k-mer.
#pragma once
#include "smx_device.hpp"

namespace smx {

constexpr uint32_t NODE_NONE = 0xFFFFFFFFu;

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
// Where to look for a k-mer in a sorted k-mer file: the offsets of the fine bins the counting pipeline sorted it by (bucket, then
// the mixed-radix digits of the key fraction) — a few hundred records per bin — or, without them, just the bucket offsets.
struct RankIndex {
    const unsigned long long *off;  // [bins + 1]
    uint32_t B, S1, nf, f[6];
    unsigned K;
};
// rank of a canonical k-mer (KMerIndex::seq_idx stand-in, kmer_index.hpp:88-100): bucket by hash, bin by key digits, binary search
template <int NW>
__device__ __forceinline__ uint32_t kmer_rank(const Rec<NW> *__restrict__ kmers, const RankIndex &ix, const Rec<NW> &canon) {
    uint64_t bin = bucket_of(xxh3_rec<NW>(canon), ix.B);
    uint64_t fr = key_top64<NW>(canon, ix.K);
    if (ix.S1 > 1) {
        bin = bin * ix.S1 + __umul64hi(fr, (uint64_t)ix.S1);
        fr *= ix.S1;
    }
    for (uint32_t i = 0; i < ix.nf; ++i) {
        bin = bin * ix.f[i] + __umul64hi(fr, (uint64_t)ix.f[i]);
        fr *= ix.f[i];
    }
    uint64_t lo = ix.off[bin], hi = ix.off[bin + 1];
    const uint64_t end = hi;
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if (rec_less<NW>(kmers[mid], canon)) lo = mid + 1; else hi = mid;
    }
    return (lo < end && rec_eq<NW>(kmers[lo], canon)) ? (uint32_t)lo : NODE_NONE;
}
__device__ __forceinline__ unsigned brev8(unsigned m) { return __brev(m) >> 24; }  // InOutMask::conjugate, inout_mask.hpp:18-39,112-115
__device__ __forceinline__ bool uniq4(unsigned m) { return m && !(m & (m - 1)); }
__device__ __forceinline__ bool mask_junction(unsigned m) { return !uniq4(m & 15) || !uniq4((m >> 4) & 15); }  // inout_mask.hpp:157-159

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

// a15: out[prefix] |= bit(x_k), in[suffix] |= bit(x_0), positions mirrored (7-p) for non-minimal keys
template <int NW>
__global__ void __launch_bounds__(BLK) k_fill_masks(const void *kpo_, uint64_t n, unsigned k, const void *kmers_,
                                                    RankIndex ix, uint32_t *mask32, uint32_t *err) {
    const Rec<NW> *kpo = (const Rec<NW> *)kpo_;
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLK) {
        Rec<NW> x = kpo[i];
        const unsigned pn = rec_nucl<NW>(x, 0), nn = rec_nucl<NW>(x, k);
        unsigned prc, src;
        Rec<NW> p = rec_canon<NW>(rec_prefix<NW>(x, k), k, prc);
        Rec<NW> s = rec_canon<NW>(rec_suffix<NW>(x), k, src);
        uint32_t rp = kmer_rank<NW>(kmers, ix, p), rs = kmer_rank<NW>(kmers, ix, s);
        if (rp == NODE_NONE || rs == NODE_NONE) {
            atomicAdd(err, 1u);
            continue;
        }
        const unsigned bp = prc ? 7 - nn : nn;
        const unsigned bs = src ? 3 - pn : pn + 4;
        atomicOr(&mask32[rp >> 2], (1u << bp) << ((rp & 3) * 8));
        atomicOr(&mask32[rs >> 2], (1u << bs) << ((rs & 3) * 8));
    }
}

// successor of every non-junction node: GetOutgoing(kwh, GetUniqueOutgoing), debruijn_graph_constructor.hpp:228-235
template <int NW>
__global__ void __launch_bounds__(BLK) k_succ(const void *kmers_, const uint8_t *mask, uint64_t D0, unsigned k,
                                              RankIndex ix, uint32_t *succ, uint32_t *err) {
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    for (uint64_t node = (uint64_t)blockIdx.x * BLK + threadIdx.x; node < 2 * D0; node += (uint64_t)gridDim.x * BLK) {
        const uint32_t r = (uint32_t)(node >> 1), o = (uint32_t)(node & 1);
        const unsigned m = mask[r];
        if (mask_junction(m)) {
            succ[node] = NODE_NONE;
            continue;
        }
        const unsigned mo = o ? brev8(m) : m;
        Rec<NW> x = kmers[r];
        if (o) x = rec_rc<NW>(x, k);
        unsigned yo;
        Rec<NW> y = rec_canon<NW>(rec_shl<NW>(x, k, __ffs(mo & 15) - 1), k, yo);
        uint32_t ry = kmer_rank<NW>(kmers, ix, y);
        if (ry == NODE_NONE) atomicAdd(err, 1u);
        succ[node] = ry == NODE_NONE ? NODE_NONE : (ry << 1) | yo;
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
template <int NW>
__global__ void __launch_bounds__(BLK) k_tip_mark(const void *kmers_, const uint8_t *mask, const uint32_t *succ, uint64_t D0, unsigned k,
                                                  RankIndex ix, uint32_t bound, uint8_t *isolate, uint8_t *tipped, unsigned long long *stats,
                                                  uint32_t *err) {
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    for (uint64_t node = (uint64_t)blockIdx.x * BLK + threadIdx.x; node < 2 * D0; node += (uint64_t)gridDim.x * BLK) {
        const uint32_t r = (uint32_t)(node >> 1), o = (uint32_t)(node & 1);
        const unsigned m = mask[r];
        const unsigned mo = o ? brev8(m) : m;
        if (__popc(mo & 15) < 2) continue;
        Rec<NW> x = kmers[r];
        if (o) x = rec_rc<NW>(x, k);
        uint32_t first[4], len[4];
        uint32_t mx = 0;
#pragma unroll
        for (unsigned c = 0; c < 4; ++c) {
            first[c] = NODE_NONE;
            len[c] = 0;  // 0 = no branch
            if (!(mo & (1u << c))) continue;
            unsigned yo;
            const Rec<NW> y = rec_canon<NW>(rec_shl<NW>(x, k, c), k, yo);
            const uint32_t ry = kmer_rank<NW>(kmers, ix, y);
            if (ry == NODE_NONE) {
                atomicAdd(err, 1u);
                continue;
            }
            uint32_t nd = (ry << 1) | yo, cnt = 0;
            first[c] = nd;
            while (cnt < bound && !mask_junction(mask[nd >> 1])) {
                ++cnt;
                nd = succ[nd];
                if (nd == NODE_NONE) break;  // inconsistent index (reported by k_succ): never walk off the array
            }
            if (nd == NODE_NONE) {
                len[c] = 0xFFFFFFFFu;
                mx = 0xFFFFFFFFu;
                continue;
            }
            ++cnt;
            const unsigned ml = (nd & 1) ? brev8(mask[nd >> 1]) : mask[nd >> 1];
            const bool tip = uniq4((ml >> 4) & 15) && (ml & 15) == 0;
            len[c] = tip ? cnt : 0xFFFFFFFFu;  // branching or too long: never removed, longer than any tip
            mx = max(mx, len[c]);
        }
        bool any = false;
#pragma unroll
        for (unsigned c = 0; c < 4; ++c) {
            if (len[c] == 0 || len[c] == 0xFFFFFFFFu || len[c] >= mx) continue;
            uint32_t nd = first[c];
            for (uint32_t i = 0; i + 1 < len[c]; ++i) {
                isolate[nd >> 1] = 1;
                nd = succ[nd];
            }
            isolate[nd >> 1] = 1;
            any = true;
            atomicAdd(&stats[0], (unsigned long long)len[c]);
            atomicAdd(&stats[1], 1ull);
        }
        if (any) tipped[node] = 1;
    }
}
__global__ void k_tip_apply(uint8_t *mask, const uint8_t *isolate, uint64_t D0) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < D0; r += (uint64_t)gridDim.x * blockDim.x)
        if (isolate[r]) mask[r] = 0;
}
template <int NW>
__global__ void __launch_bounds__(BLK) k_tip_fix(const void *kmers_, uint32_t *mask32, const uint8_t *tipped, uint64_t D0, unsigned k, RankIndex ix,
                                                 uint32_t *err) {
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    const uint8_t *mask = (const uint8_t *)mask32;
    for (uint64_t node = (uint64_t)blockIdx.x * BLK + threadIdx.x; node < 2 * D0; node += (uint64_t)gridDim.x * BLK) {
        if (!tipped[node]) continue;
        const uint32_t r = (uint32_t)(node >> 1), o = (uint32_t)(node & 1);
        const unsigned m = mask[r];
        const unsigned mo = o ? brev8(m) : m;
        Rec<NW> x = kmers[r];
        if (o) x = rec_rc<NW>(x, k);
        const unsigned firstn = rec_nucl<NW>(x, 0);
        for (unsigned c = 0; c < 4; ++c) {
            if (!(mo & (1u << c))) continue;
            unsigned yo;
            const Rec<NW> y = rec_canon<NW>(rec_shl<NW>(x, k, c), k, yo);
            const uint32_t ry = kmer_rank<NW>(kmers, ix, y);
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
template <int NW>
__global__ void __launch_bounds__(BLK) k_at_edges_mark(const void *kmers_, const uint8_t *mask, uint64_t D0, unsigned k, RankIndex ix,
                                                       uint32_t thr_edge, uint8_t *atflag, uint32_t *err) {
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    for (uint64_t node = (uint64_t)blockIdx.x * BLK + threadIdx.x; node < 2 * D0; node += (uint64_t)gridDim.x * BLK) {
        const uint32_t r = (uint32_t)(node >> 1), o = (uint32_t)(node & 1);
        const unsigned m = mask[r];
        if (!mask_junction(m)) continue;
        const unsigned mo = o ? brev8(m) : m;
        if ((mo & 15) == 0) continue;
        Rec<NW> x = kmers[r];
        if (o) x = rec_rc<NW>(x, k);
        unsigned cnt[4];
        rec_counts<NW>(x, k, cnt);
        if (max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3])) < thr_edge) continue;
        unsigned fl = 0;
        for (unsigned c = 0; c < 4; ++c) {
            if (!(mo & (1u << c))) continue;
            unsigned yo;
            const Rec<NW> y = rec_canon<NW>(rec_shl<NW>(x, k, c), k, yo);
            const uint32_t ry = kmer_rank<NW>(kmers, ix, y);
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
template <int NW>
__global__ void __launch_bounds__(BLK) k_at_edges_apply(const void *kmers_, uint32_t *mask32, uint64_t D0, unsigned k, RankIndex ix,
                                                        const uint8_t *atflag, unsigned long long *stats, uint32_t *err) {
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    for (uint64_t node = (uint64_t)blockIdx.x * BLK + threadIdx.x; node < 2 * D0; node += (uint64_t)gridDim.x * BLK) {
        const unsigned fl = atflag[node];
        if (!fl) continue;
        const uint32_t r = (uint32_t)(node >> 1), o = (uint32_t)(node & 1);
        Rec<NW> x = kmers[r];
        if (o) x = rec_rc<NW>(x, k);
        const unsigned firstn = rec_nucl<NW>(x, 0);
        for (unsigned c = 0; c < 4; ++c) {
            if (!(fl & (1u << c))) continue;
            unsigned yo;
            const Rec<NW> y = rec_canon<NW>(rec_shl<NW>(x, k, c), k, yo);
            const uint32_t ry = kmer_rank<NW>(kmers, ix, y);
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
template <int NW>
__global__ void __launch_bounds__(BLK) k_at_tips_mark(const void *kmers_, const uint8_t *mask, const uint32_t *succ, uint64_t D0, unsigned k,
                                                      RankIndex ix, uint32_t min_len, uint32_t max_len, const uint16_t *thr_tip,
                                                      uint8_t *isolate, uint8_t *tipped, unsigned long long *stats, uint32_t *err) {
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    for (uint64_t node = (uint64_t)blockIdx.x * BLK + threadIdx.x; node < 2 * D0; node += (uint64_t)gridDim.x * BLK) {
        const uint32_t r = (uint32_t)(node >> 1), o = (uint32_t)(node & 1);
        const unsigned m0 = mask[r];
        const unsigned mo0 = o ? brev8(m0) : m0;
        if ((mo0 & 15) != 0 || !uniq4((mo0 >> 4) & 15)) continue;  // start from tip ends
        Rec<NW> x = kmers[r];
        if (o) x = rec_rc<NW>(x, k);
        unsigned cnt[4] = {0, 0, 0, 0};
        uint32_t n = 0, nd = (uint32_t)node;
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
                const uint32_t ry = kmer_rank<NW>(kmers, ix, y);
                if (ry == NODE_NONE) {
                    bad = true;
                    break;
                }
                nd = (ry << 1) | yo;
            } else {
                const uint32_t sp = succ[nd ^ 1];
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
            uint32_t q = (uint32_t)node;
            Rec<NW> z = kmers[r];
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
                    q = (kmer_rank<NW>(kmers, ix, y) << 1) | yo;
                } else {
                    q = succ[q ^ 1] ^ 1;
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

// start de-edges per junction k-mer: out bits of kh, then out bits of !kh (AddStartDeEdges, :203-226)
__global__ void k_cand_count(const uint8_t *mask, uint64_t D0, unsigned long long *cnt) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < D0; r += (uint64_t)gridDim.x * blockDim.x) {
        unsigned m = mask[r];
        cnt[r] = mask_junction(m) ? (unsigned long long)(__popc(m & 15) + __popc(m >> 4)) : 0ull;
    }
}
__global__ void k_cand_expand(const uint8_t *mask, const unsigned long long *cand_off, uint64_t D0, unsigned long long *cand) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < D0; r += (uint64_t)gridDim.x * blockDim.x) {
        unsigned m = mask[r];
        if (!mask_junction(m)) continue;
        unsigned long long o = cand_off[r];
        for (unsigned c = 0; c < 4; ++c)
            if (m & (1u << c)) cand[o++] = (r << 3) | c;
        const unsigned mi = brev8(m);
        for (unsigned c = 0; c < 4; ++c)
            if (mi & (1u << c)) cand[o++] = (r << 3) | 4u | c;
    }
}

// ConstructSequenceWithEdge (:264-273), pass 1: length and end node of every start de-edge
template <int NW>
__global__ void __launch_bounds__(BLK) k_walk_len(const unsigned long long *cand, uint64_t C, const void *kmers_, const uint8_t *mask,
                                                  const uint32_t *succ, unsigned k, RankIndex ix,
                                                  uint64_t max_steps, unsigned long long *len, uint32_t *first, uint32_t *last,
                                                  uint32_t *err) {
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < C; i += (uint64_t)gridDim.x * BLK) {
        const unsigned long long cd = cand[i];
        const uint32_t r = (uint32_t)(cd >> 3), side = (uint32_t)((cd >> 2) & 1), c = (uint32_t)(cd & 3);
        Rec<NW> x = kmers[r];
        if (side) x = rec_rc<NW>(x, k);
        unsigned yo;
        Rec<NW> y = rec_canon<NW>(rec_shl<NW>(x, k, c), k, yo);
        uint32_t ry = kmer_rank<NW>(kmers, ix, y);
        if (ry == NODE_NONE) {
            atomicAdd(err, 1u);
            len[i] = 0;
            first[i] = last[i] = NODE_NONE;
            continue;
        }
        uint32_t node = (ry << 1) | yo;
        first[i] = node;
        uint64_t steps = 0;
        while (!mask_junction(mask[node >> 1])) {
            node = succ[node];
            if (++steps > max_steps || node == NODE_NONE) {  // cannot happen on a consistent index; never hang the GPU
                atomicAdd(err, 1u);
                break;
            }
        }
        last[i] = node;
        len[i] = k + 1 + steps;
    }
}

// pass 2: write the nucleotides (ASCII) and mark every k-mer on the path (both strands share the rank) as visited
template <int NW>
__global__ void __launch_bounds__(BLK) k_walk_write(const unsigned long long *cand, uint64_t C, const void *kmers_, const uint8_t *mask,
                                                    const uint32_t *succ, unsigned k, const uint32_t *first, const unsigned long long *soff,
                                                    char *seq, uint8_t *visited) {
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < C; i += (uint64_t)gridDim.x * BLK) {
        const unsigned long long cd = cand[i];
        const uint32_t r = (uint32_t)(cd >> 3), side = (uint32_t)((cd >> 2) & 1), c = (uint32_t)(cd & 3);
        uint32_t node = first[i];
        if (node == NODE_NONE) continue;
        Rec<NW> x = kmers[r];
        if (side) x = rec_rc<NW>(x, k);
        char *s = seq + soff[i];
        const unsigned long long n = soff[i + 1] - soff[i];
        for (unsigned j = 0; j < k; ++j) s[j] = "ACGT"[rec_nucl<NW>(x, j)];
        s[k] = "ACGT"[c];
        visited[r] = 1;
        unsigned long long p = k + 1;
        while (p < n) {
            const unsigned m = mask[node >> 1];
            const unsigned mo = (node & 1) ? brev8(m) : m;
            visited[node >> 1] = 1;
            s[p++] = "ACGT"[__ffs(mo & 15) - 1];
            node = succ[node];
        }
        visited[node >> 1] = 1;
    }
}

// keep iff !(s < !s) (:305-306); also self-conjugate flag (s == !s)
__global__ void k_keep(const char *seq, const unsigned long long *soff, uint64_t C, unsigned long long *keeplen, uint8_t *flags) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < C; i += (uint64_t)gridDim.x * blockDim.x) {
        const char *s = seq + soff[i];
        const unsigned long long n = soff[i + 1] - soff[i];
        int cmp = 0;  // sign of s - rc(s)
        for (unsigned long long j = 0; j < n; ++j) {
            const char a = s[j], b = s[n - 1 - j];
            const char cb = b == 'A' ? 'T' : b == 'C' ? 'G' : b == 'G' ? 'C' : 'A';
            if (a != cb) {
                cmp = a < cb ? -1 : 1;
                break;
            }
        }
        const bool keep = n > 0 && cmp >= 0;
        flags[i] = (uint8_t)((keep ? 1 : 0) | (cmp == 0 ? 2 : 0));
        keeplen[i] = keep ? n : 0;
    }
}
// kept candidates -> dense edge arrays
__global__ void k_gather(const char *seq, const unsigned long long *soff, const unsigned long long *koff, const uint8_t *flags,
                         const unsigned long long *eidx, const unsigned long long *cand, const uint32_t *last, uint64_t C,
                         char *kseq, unsigned long long *eoff, uint32_t *estart, uint32_t *eend, uint8_t *eself) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < C; i += (uint64_t)gridDim.x * blockDim.x) {
        if (!(flags[i] & 1)) continue;
        const unsigned long long e = eidx[i], n = soff[i + 1] - soff[i], o = koff[i];
        const char *s = seq + soff[i];
        for (unsigned long long j = 0; j < n; ++j) kseq[o + j] = s[j];
        eoff[e] = o;
        estart[e] = (uint32_t)(((cand[i] >> 3) << 1) | ((cand[i] >> 2) & 1));
        eend[e] = last[i];
        eself[e] = (flags[i] >> 1) & 1;
    }
}
__global__ void k_keep_flag(const uint8_t *flags, uint64_t C, unsigned long long *one) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < C; i += (uint64_t)gridDim.x * blockDim.x) one[i] = flags[i] & 1;
}
// k-mers left on perfect loops: non-junction and not on any extracted path (CollectLoops pass 1, :362-374)
__global__ void k_loop_nodes(const uint8_t *mask, const uint8_t *visited, uint64_t D0, uint32_t *count, uint32_t *list, uint32_t cap) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < D0; r += (uint64_t)gridDim.x * blockDim.x) {
        if (!mask_junction(mask[r]) && !visited[r]) {
            uint32_t p = atomicAdd(count, 1u);
            if (p < cap) list[p] = (uint32_t)r;
        }
    }
}
template <int NW>
__global__ void k_gather_kmers(const void *kmers_, const uint32_t *list, uint32_t n, void *out_) {
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    Rec<NW> *out = (Rec<NW> *)out_;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = kmers[list[i]];
}

// ---- link records + vertices on the device (FastGraphFromSequencesConstructor::ConstructGraph, debruijn_graph_constructor.hpp:506-567) ----
// Record key = rank << 33 | EdgeAndMask (edge id << 2 | rc << 1 | is_start), left-shifted by sh so that the MSD digits of the
// sorting pipeline see a spread-out fraction. The end record of a self-conjugate edge does not exist (LinkRecord() is invalid): it is
// emitted as a copy of the start record, which the sort + unique pipeline drops.
__global__ void k_link_keys(const uint32_t *estart, const uint32_t *eend, const uint8_t *eself, uint64_t ne, unsigned sh,
                            unsigned long long *keys) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ne; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t edge = 3 + 2 * i;
        const uint32_t s = estart[i], e = eend[i];
        const unsigned long long ks = ((uint64_t)(s >> 1) << 33) | (edge << 2) | ((uint64_t)(s & 1) << 1) | 1ull;
        const unsigned long long ke = ((uint64_t)(e >> 1) << 33) | (edge << 2) | ((uint64_t)(e & 1) << 1);
        keys[2 * i] = ks << sh;
        keys[2 * i + 1] = (eself[i] ? ks : ke) << sh;
    }
}
// sorted keys (still shifted): un-shift in place; one[i] = 1 where a new rank (vertex) starts
__global__ void k_vertex_flags(unsigned long long *keys, uint64_t n, unsigned sh, unsigned long long *one) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long k = keys[i] >> sh;
        const unsigned long long p = i ? keys[i - 1] >> sh : ~0ull;
        one[i] = (i == 0 || (k >> 33) != (p >> 33)) ? 1ull : 0ull;
    }
}
__global__ void k_unshift(unsigned long long *keys, uint64_t n, unsigned sh) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) keys[i] >>= sh;
}
// vertex v = the run of records starting at position i: vpos[v] = i, vkey[v] = smallest EdgeAndMask of the vertex << 31 | v (shifted)
__global__ void k_vertex_collect(const unsigned long long *keys, const unsigned long long *one, const unsigned long long *vidx, uint64_t n,
                                 unsigned sh2, unsigned long long *vpos, unsigned long long *vkeys) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (!one[i]) continue;
        const unsigned long long v = vidx[i];
        vpos[v] = i;
        vkeys[v] = (((keys[i] & ((1ull << 33) - 1)) << 31) | v) << sh2;
    }
}
__global__ void k_vertex_permute(const unsigned long long *vkeys_sorted, const unsigned long long *vpos, uint64_t nv, unsigned sh2,
                                 unsigned long long *vstart) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nv; j += (uint64_t)gridDim.x * blockDim.x)
        vstart[j] = vpos[(vkeys_sorted[j] >> sh2) & ((1ull << 31) - 1)];
}

// ---- coverage (-c) -----------------------------------------------------------------------------------------------
// CoverageHashMapBuilder::FillCoverageFromStream (kmer_index/ph_map/coverage_hash_map_builder.hpp:18-39): every
// (k+1)-mer instance of the read+RC stream whose orientation is minimal increments the counter of its canonical
// record. One window position contributes its canonical form once — twice when the (k+1)-mer is its own reverse
// complement (both strand instances are minimal then).
template <int NW>
__global__ void __launch_bounds__(BLK) k_kpo_coverage(const uint64_t *seq, const uint64_t *mask, uint64_t G, unsigned K1,
                                                      const void *kpo_, RankIndex ix, uint32_t *cnt) {
    const Rec<NW> *kpo = (const Rec<NW> *)kpo_;
    for (uint64_t g = (uint64_t)blockIdx.x * BLK + threadIdx.x; g < G; g += (uint64_t)gridDim.x * BLK) {
        if (!((mask[g >> 6] >> (g & 63)) & 1)) continue;
        Rec<NW> x = load_window<NW>(seq, g, K1);
        Rec<NW> y = rec_rc<NW>(x, K1);
        const bool pal = rec_eq<NW>(x, y);
        const Rec<NW> c = rc_ge<NW>(y, x) ? x : y;
        const uint32_t r = kmer_rank<NW>(kpo, ix, c);
        if (r != NODE_NONE) atomicAdd(&cnt[r], pal ? 2u : 1u);
    }
}

// edge raw coverage = sum of the counters of the edge's (k+1)-mers (graph_support/coverage_filling.hpp:46-62), uint32; flanking raw
// coverage = the same sum over the first `flank` (k+1)-mers (inc_coverage: offset < averaging_range, :40-44) of the edge (fl_s) and
// of its conjugate, i.e. the last `flank` ones (fl_e)
template <int NW>
__global__ void __launch_bounds__(BLK) k_edge_coverage(const char *seq, const unsigned long long *eoff, uint64_t n_edges, uint64_t total,
                                                       unsigned K1, const void *kpo_, RankIndex ix,
                                                       const uint32_t *cnt, uint32_t *ecov, uint32_t flank, uint32_t *fl_s, uint32_t *fl_e) {
    const Rec<NW> *kpo = (const Rec<NW> *)kpo_;
    for (uint64_t p = (uint64_t)blockIdx.x * BLK + threadIdx.x; p < total; p += (uint64_t)gridDim.x * BLK) {
        uint64_t lo = 0, hi = n_edges;  // edge e with eoff[e] <= p < eoff[e+1]
        while (hi - lo > 1) {
            uint64_t mid = (lo + hi) >> 1;
            if (eoff[mid] <= p) lo = mid; else hi = mid;
        }
        if (p + K1 > eoff[lo + 1]) continue;
        Rec<NW> x;
#pragma unroll
        for (int w = 0; w < NW; ++w) x.w[w] = 0;
        for (unsigned j = 0; j < K1; ++j) {
            const char ch = seq[p + j];
            const uint64_t code = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3;
            x.w[j >> 5] |= code << ((j & 31) << 1);
        }
        unsigned f;
        const Rec<NW> c = rec_canon<NW>(x, K1, f);
        const uint32_t r = kmer_rank<NW>(kpo, ix, c);
        if (r != NODE_NONE) {
            const uint32_t v = cnt[r];
            atomicAdd(&ecov[lo], v);
            const uint64_t j = p - eoff[lo], nk = eoff[lo + 1] - eoff[lo] - K1 + 1;
            if (j < flank) atomicAdd(&fl_s[lo], v);
            if (j + flank >= nk) atomicAdd(&fl_e[lo], v);
        }
    }
}

}  // namespace smx
