// spades_amd/csrc/smx_loops.hip — perfect loops on the device (included by smx_api.hip; option "device_loops").
//
// Reference: CollectLoops / FindMinimalKMerInLoop / ConstructLoopFromVertex / SplitLoop,
// assembly_graph/construction/debruijn_graph_constructor.hpp:252-293,359-397 — the k-mers that no unbranching path holds lie on cycles
// without a junction. The reference meets them in index order; the first k-mer of a cycle (as stored) starts a loop, which is rotated to
// its minimal k-mer (over both strands, nucleotide order), written from there once around, split at its first palindromic (k+1)-mer and
// emitted as max(part, RC(part)).
//
// Here everything that is O(loop k-mers) runs on the device, in NODE space: the successor table the walks used (one entry per oriented
// k-mer) IS the cycle structure, so nothing is looked up by k-mer.
//   k_lp_leaders   one thread per left-over k-mer r: follow the successors of node 2r until a k-mer that comes EARLIER in the k-mer file
//                  shows up (then r is not the first of its cycle: expected after O(log length) steps in a hash-ordered file) or 2r comes
//                  back (then it is: the cycle's length is known). "Earlier" is the rank where the nodes are numbered in file order, and
//                  (bucket, record) of the k-mers on the partition-major numbering of route 0.
//   (host)         the leaders — one per loop, a handful — are put into file order and handed back;
//   k_lp_measure   one thread per loop: the smallest stored k-mer on the cycle (nucleotide order) and, walking from it, the first edge
//                  that is its own reverse complement (u -> v with v = u on the other strand);
//   (host)         edge bookkeeping per loop: one edge, or two where the loop is split (lengths, first and last node of every part);
//   k_lp_write     one thread per edge: part against RC(part) — two walks side by side, the reverse complement of a part is spelled by the
//                  other strand's nodes from RC(its last k-mer) — and the larger one written, 2 bits per nucleotide, into the graph's arrays.
// The host version (smx_loops_host.hpp) stays the default and the checker (tests: both must agree on every loop golden).
#pragma once
#include "smx_graph.hip"

namespace smx {

// nucleotide-lexicographic order of two k-mers (RtSeq operator<): the first nucleotide that differs decides
template <int NW>
__device__ __forceinline__ int lp_lex_cmp(const Rec<NW> &a, const Rec<NW> &b) {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const uint64_t x = a.w[i] ^ b.w[i];
        if (x) {
            const unsigned p = (unsigned)(__ffsll((unsigned long long)x) - 1) & ~1u;
            return ((a.w[i] >> p) & 3) < ((b.w[i] >> p) & 3) ? -1 : 1;
        }
    }
    return 0;
}
// the k-mer a node stands for (EXT: the records carry the mask byte)
template <int NW, bool EXT>
__device__ __forceinline__ Rec<NW> lp_stored(const Rec<NW> *kmers, uint64_t rank) {
    return EXT ? rec_pure<NW>(kmers[rank]) : kmers[rank];
}
template <int NW, bool EXT>
__device__ __forceinline__ Rec<NW> lp_oriented(const Rec<NW> *kmers, node_t v, unsigned k) {
    const Rec<NW> x = lp_stored<NW, EXT>(kmers, v >> 1);
    return (v & 1) ? rec_rc<NW>(x, k) : x;
}
// last nucleotide of the oriented k-mer of node v
template <int NW, bool EXT>
__device__ __forceinline__ unsigned lp_last_base(const Rec<NW> *kmers, node_t v, unsigned k) {
    const Rec<NW> x = lp_stored<NW, EXT>(kmers, v >> 1);
    if (v & 1) return 3u - (unsigned)(x.w[0] & 3u);
    return (unsigned)(x.w[(k - 1) >> 5] >> (((k - 1) & 31u) << 1)) & 3u;
}

// PMORDER: nodes are not numbered in k-mer-file order (route 0): file order = (bucket, record)
template <int NW, bool EXT, bool PMORDER>
__global__ void __launch_bounds__(BLK) k_lp_leaders(const unsigned long long *__restrict__ llist, uint64_t L, const node_t *__restrict__ succ, const void *kmers_,
                                                    uint32_t B, unsigned long long max_steps, unsigned long long *lead, unsigned long long *llen,
                                                    unsigned long long *counters /* [0] leaders, [1] errors */) {
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < L; i += (uint64_t)gridDim.x * BLK) {
        const uint64_t r = llist[i];
        Rec<NW> xr;
        uint32_t br = 0;
        if constexpr (PMORDER) {
            xr = lp_stored<NW, EXT>(kmers, r);
            br = bucket_of(xxh3_rec<NW>(xr), B);
        }
        node_t v = 2 * r;
        unsigned long long steps = 0;
        bool leader = true;
        for (;;) {
            const node_t e = succ[v];
            v = e & TAB_NODE_MASK;
            ++steps;
            if (v == 2 * r) break;
            const uint64_t a = v >> 1;
            if (a != r) {
                bool earlier;
                if constexpr (PMORDER) {
                    const Rec<NW> xa = lp_stored<NW, EXT>(kmers, a);
                    const uint32_t ba = bucket_of(xxh3_rec<NW>(xa), B);
                    earlier = ba != br ? ba < br : rec_less<NW>(xa, xr);
                } else {
                    earlier = a < r;
                }
                if (earlier) {
                    leader = false;
                    break;
                }
            }
            if (steps > max_steps) {  // never expected: the left-over k-mers do not close into cycles
                atomicAdd(&counters[1], 1ull);
                leader = false;
                break;
            }
        }
        if (leader) {
            const unsigned long long p = atomicAdd(&counters[0], 1ull);
            lead[p] = r;
            llen[p] = steps;
        }
    }
}

// per loop (leaders in file order): out[4 i + 0] = the node the loop is written from (its smallest stored k-mer, as stored), [1] = position of
// the first palindromic edge counted from there (~0: none), [2] / [3] = the two nodes of that edge
template <int NW, bool EXT>
__global__ void __launch_bounds__(BLK) k_lp_measure(const unsigned long long *__restrict__ lead, const unsigned long long *__restrict__ llen, uint64_t nl,
                                                    const node_t *__restrict__ succ, const void *kmers_, unsigned long long *out) {
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < nl; i += (uint64_t)gridDim.x * BLK) {
        const uint64_t r = lead[i], len = llen[i];
        uint64_t best = r;
        Rec<NW> xb = lp_stored<NW, EXT>(kmers, r);
        node_t v = 2 * r;
        for (uint64_t t = 1; t < len; ++t) {
            v = succ[v] & TAB_NODE_MASK;
            const uint64_t a = v >> 1;
            const Rec<NW> xa = lp_stored<NW, EXT>(kmers, a);
            if (lp_lex_cmp<NW>(xa, xb) < 0) {
                xb = xa;
                best = a;
            }
        }
        node_t w = 2 * best;
        unsigned long long pos = ~0ull;
        node_t pa = 0, pb = 0;
        for (uint64_t t = 0; t < len; ++t) {
            const node_t nx = succ[w] & TAB_NODE_MASK;
            if (nx == (w ^ 1ull)) {
                pos = t;
                pa = w;
                pb = nx;
                break;
            }
            w = nx;
        }
        out[4 * i] = 2 * best;
        out[4 * i + 1] = pos;
        out[4 * i + 2] = pa;
        out[4 * i + 3] = pb;
    }
}

// per edge e: the part that starts at node ea[e], ends at node eb[e] and appends nnt[e] nucleotides to its first k-mer; the larger of
// {part, RC(part)} goes to the graph's arrays at index first + e (uwords: its words start at offw[e], zeroed by the caller)
template <int NW, bool EXT>
__global__ void __launch_bounds__(BLK) k_lp_write(const node_t *__restrict__ ea, const node_t *__restrict__ eb, const unsigned long long *__restrict__ nnt,
                                                  const unsigned long long *__restrict__ offw, uint64_t ne, const node_t *__restrict__ succ, const void *kmers_,
                                                  unsigned k, uint64_t *uwords, node_t *estart, node_t *eend, uint8_t *eself) {
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    for (uint64_t e = (uint64_t)blockIdx.x * BLK + threadIdx.x; e < ne; e += (uint64_t)gridDim.x * BLK) {
        const node_t a = ea[e], b = eb[e];
        const uint64_t n = nnt[e];
        // part against RC(part): the latter starts with RC(last k-mer) = node b^1 and follows the successors of that strand
        int c3 = lp_lex_cmp<NW>(lp_oriented<NW, EXT>(kmers, a, k), lp_oriented<NW, EXT>(kmers, b ^ 1ull, k));
        {
            node_t x = a, y = b ^ 1ull;
            for (uint64_t t = 0; t < n && c3 == 0; ++t) {
                x = succ[x] & TAB_NODE_MASK;
                y = succ[y] & TAB_NODE_MASK;
                const unsigned bx = lp_last_base<NW, EXT>(kmers, x, k), by = lp_last_base<NW, EXT>(kmers, y, k);
                if (bx != by) c3 = bx < by ? -1 : 1;
            }
        }
        const node_t s0 = c3 < 0 ? (b ^ 1ull) : a, s1 = c3 < 0 ? (a ^ 1ull) : b;
        estart[e] = s0;
        eend[e] = s1;
        eself[e] = c3 == 0 ? 1 : 0;
        uint64_t *dst = uwords + offw[e];
        const Rec<NW> x0 = lp_oriented<NW, EXT>(kmers, s0, k);
        // the first k-mer, then one nucleotide per step; a 64-bit accumulator is flushed every 32 nucleotides
        uint64_t acc = 0, wi = 0;
        unsigned fill = 0;
        auto put = [&](unsigned base) {
            acc |= (uint64_t)base << (fill << 1);
            if (++fill == 32) {
                dst[wi++] = acc;
                acc = 0;
                fill = 0;
            }
        };
        for (unsigned j = 0; j < k; ++j) put((unsigned)(x0.w[j >> 5] >> ((j & 31u) << 1)) & 3u);
        node_t v = s0;
        for (uint64_t t = 0; t < n; ++t) {
            v = succ[v] & TAB_NODE_MASK;
            put(lp_last_base<NW, EXT>(kmers, v, k));
        }
        if (fill) dst[wi] = acc;
    }
}

}  // namespace smx
