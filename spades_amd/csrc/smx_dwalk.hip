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

__device__ __forceinline__ unsigned long long dw_wave_count(unsigned long long *lcnt, bool has, uint32_t ow);
// CAND = false: items are the 2 * D0 oriented nodes of the shard, those of junction k-mers ask nothing;
// CAND = true: items are the start de-edges (k_cand_expand). PASS 0 counts per owner, PASS 1 places record + tag.
template <int NW, bool CAND, int PASS>
__global__ void __launch_bounds__(BLK) k_dw_requests(const void *kmers_, const uint8_t *__restrict__ mask, const unsigned long long *__restrict__ cand,
                                                     uint64_t item0, uint64_t n_items, unsigned k, uint32_t B, uint32_t world,
                                                     unsigned long long *bc_or_off, void *out_, unsigned long long *tags) {
    // Grouping by owner without one global atomic per workgroup round (21 M atomics on ONE address per pass of a 5.4 G-node shard: 240 ms each at the
    // ~88 atomics / us one address takes — what the first measurement of these kernels at size spent its time on): PASS 0 counts in LDS over ALL the
    // rounds of the workgroup and leaves bc[owner * gridDim + block]; the caller scans that array (owner-major: the scan IS the place of every
    // (owner, block) in the send order); PASS 1 — same grid, same rounds — starts its LDS cursors there. No barrier inside the loop.
    extern __shared__ unsigned long long lds_dw[];  // [world] counters (PASS 0) / cursors (PASS 1)
    const Rec<NW> *kmers = (const Rec<NW> *)kmers_;
    Rec<NW> *out = (Rec<NW> *)out_;
    for (uint32_t t = threadIdx.x; t < world; t += BLK) lds_dw[t] = PASS == 0 ? 0ull : bc_or_off[(size_t)t * gridDim.x + blockIdx.x];
    __syncthreads();
    for (uint64_t base = (uint64_t)blockIdx.x * BLK; base < n_items; base += (uint64_t)gridDim.x * BLK) {
        const uint64_t i = item0 + base + threadIdx.x;  // (the items [item0, item0 + n_items): a caller may ask range by range)
        bool asks = false;
        Rec<NW> y;
        uint32_t ow = 0;
        unsigned long long tag = 0;
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
                tag = ((unsigned long long)i << DW_TAG_SHIFT) | (CAND ? 8u : 0u) | (yo << 2) | c;
            }
        }
        const unsigned long long at = dw_wave_count(lds_dw, asks, ow);  // (one LDS atomic per wave and owner)
        if (PASS == 1 && asks) {
            out[at] = y;
            tags[at] = tag;
        }
    }
    __syncthreads();
    if (PASS == 0)
        for (uint32_t t = threadIdx.x; t < world; t += BLK) bc_or_off[(size_t)t * gridDim.x + blockIdx.x] = lds_dw[t];
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


// ---- the plumbing between the k-mer steps, on the device (round 6) -------------------------------------------------------------------
// Until round 5 the routing by owner, the pointer doubling and the ragged fetch of the chains were torch tensor code in spades_amd/dist.py
// (at::native sort / searchsorted / index kernels: 19.8 s for one rank's share of BASELINE config 4). They are these kernels now, driven by
// dw_walks (smx_dwalk.hpp) with the caller's collectives between them — from C++ (tools/gbuilder_mgpu.hpp: grouped ncclSend / ncclRecv) or
// Python (dist.py: torch.distributed). Role model in the reference tree: hpcspades/mpi/stages/construction_mpi.cpp:303-412 (construction
// phases spread over processes); the algorithm is SURVEY.md §8(e)'s "bulk-synchronous pointer-jumping rounds with an all-to-all per round".
//
// State of a local oriented node: ONE packed word + one byte (as dist.py had it since round 4):
//   word  F << 63 | T << 62 | id << HB | hops     F: the end of the chain is known; T: this node IS the end (its successor is a junction
//         open:      id = pointer, hops = steps to it      k-mer); id: 62 - HB bits; hops: HB bits (24: a chain of 16.7 M k-mers or more is refused)
//         tail (T):  id = the junction node behind it
//         finished:  id = the tail of its chain, hops = steps to the tail
//   byte  bit 0 chain k-mer (non-junction), bit 1 its successor is a junction k-mer, bits 2-3 its outgoing nucleotide
struct DwBits {
    unsigned hb;
    unsigned long long idm, hm;
};
constexpr unsigned long long DW_F = 1ull << 63, DW_T = 1ull << 62;
constexpr unsigned DW_MAX_WORLD = 64;
struct DwOwners {  // bound[p] = first global node of rank p + 1: owner_of(node) = the first p with node < bound[p]
    unsigned long long bound[DW_MAX_WORLD];
    unsigned world;
};
struct DwSegs {  // send-order segments of an exchange: segment p = [off[p], off[p + 1]) went to / came from rank p; add[p]: what its answers are relative to
    unsigned long long off[DW_MAX_WORLD + 1];
    unsigned long long add[DW_MAX_WORLD];
    unsigned world;
};
__device__ __forceinline__ uint32_t dw_owner_of(const DwOwners &o, unsigned long long node) {
    uint32_t p = 0;
    while (p + 1 < o.world && node >= o.bound[p]) ++p;
    return p;
}
__device__ __forceinline__ bool dw_open(unsigned long long w, uint8_t f) { return (f & 1) && !(w & DW_F); }
__device__ __forceinline__ bool dw_done(unsigned long long w, uint8_t f) { return (f & 1) && (w & DW_F); }

// A wave's lanes with a message take their places in the workgroup's counters owner by owner: ONE LDS atomic per (wave, owner) instead of one per
// lane — with a few ranks nearly every lane of a workgroup hits the same handful of counters (world = 1: all 256 the same one; the first GPU run of
// these kernels was measured on exactly that). Every lane of the wave calls it (ballots / shuffles outside divergent code); returns the lane's
// place among the workgroup's messages to `ow`.
__device__ __forceinline__ unsigned long long dw_wave_count(unsigned long long *lcnt, bool has, uint32_t ow) {
    unsigned long long todo = __ballot(has), at = 0;
    const unsigned lane = threadIdx.x & 63u;
    while (todo) {
        const int lead = __ffsll(todo) - 1;
        const uint32_t o = (uint32_t)__shfl((int)ow, lead, 64);
        const unsigned long long same = __ballot(has && ow == o) & todo;
        unsigned long long base = 0;
        if ((int)lane == lead) base = atomicAdd(&lcnt[o], (unsigned long long)__popcll(same));
        base = __shfl(base, lead, 64);
        if ((same >> lane) & 1ull) at = base + (unsigned long long)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    return at;
}
// The grouping kernels below share k_dw_requests' scheme: PASS 0 leaves bc[owner * gridDim + block] (counts over all rounds of the workgroup), the
// caller scans it, PASS 1 starts its LDS cursors at off[owner * gridDim + block] — one LDS atomic per wave and owner, no global atomic, no barrier in
// the loop. dw_pass_begin / dw_pass_end are the two ends of that.
template <int PASS>
__device__ __forceinline__ void dw_pass_begin(unsigned long long *lctr, uint32_t world, const unsigned long long *bc_or_off) {
    for (uint32_t t = threadIdx.x; t < world; t += BLK) lctr[t] = PASS == 0 ? 0ull : bc_or_off[(size_t)t * gridDim.x + blockIdx.x];
    __syncthreads();
}
template <int PASS>
__device__ __forceinline__ void dw_pass_end(const unsigned long long *lctr, uint32_t world, unsigned long long *bc_or_off) {
    __syncthreads();
    if (PASS == 0)
        for (uint32_t t = threadIdx.x; t < world; t += BLK) bc_or_off[(size_t)t * gridDim.x + blockIdx.x] = lctr[t];
}

// answers of the owners to the successor requests (k_dw_requests -> all-to-all -> k_dw_lookup -> all-to-all back), in send order:
// chain requests set the node's word and byte, start requests the first node of the de-edge and whether that is a junction k-mer
template <bool CAND>
__global__ void __launch_bounds__(BLK) k_dw_apply_succ(const unsigned long long *__restrict__ tags, const unsigned long long *__restrict__ back, uint64_t n, DwSegs sg,
                                                       DwBits b, unsigned long long *word, uint8_t *flag, unsigned long long *c_first, uint8_t *c_fj, uint32_t *err) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLK) {
        uint32_t p = 0;
        while (p + 1 < sg.world && i >= sg.off[p + 1]) ++p;
        const unsigned long long r = back[i], tag = tags[i];
        if (r == ~0ull) {  // the successor k-mer is in no shard: the k-mer file and the masks disagree
            atomicAdd(err, 1u);
            continue;
        }
        const unsigned long long junc = r & 1ull, node = ((((r >> 1) + sg.add[p]) << 1) | ((tag >> 2) & 1ull)), item = tag >> DW_TAG_SHIFT;
        if (CAND) {
            c_first[item] = node;
            c_fj[item] = (uint8_t)junc;
        } else {
            word[item] = junc ? ((node << b.hb) | DW_F | DW_T) : ((node << b.hb) | 1ull);
            flag[item] = (uint8_t)(1u | (unsigned)(junc << 1) | ((unsigned)(tag & 3) << 2));
        }
    }
}
__global__ void __launch_bounds__(BLK) k_dw_count_open(const unsigned long long *__restrict__ word, const uint8_t *__restrict__ flag, uint64_t n2, unsigned long long *out) {
    __shared__ unsigned long long scratch[BLK / 64 + 2];
    unsigned long long c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n2; i += (uint64_t)gridDim.x * BLK) c += dw_open(word[i], flag[i]) ? 1 : 0;
    unsigned long long tot;
    block_excl_scan<unsigned long long>(c, scratch, &tot);
    if (threadIdx.x == 0 && tot) atomicAdd(out, tot);
}
// doubling: the open nodes of [a, a + n) ask the owners of their pointers for those nodes' words; q[slot] = pointer, tag[slot] = asking node
template <int PASS>
__global__ void __launch_bounds__(BLK) k_dw_open_req(const unsigned long long *__restrict__ word, const uint8_t *__restrict__ flag, uint64_t a, uint64_t n, DwBits b,
                                                     DwOwners ow, unsigned long long *bc_or_off, unsigned long long *q, unsigned long long *tag) {
    extern __shared__ unsigned long long lds_dw[];
    dw_pass_begin<PASS>(lds_dw, ow.world, bc_or_off);
    for (uint64_t base = (uint64_t)blockIdx.x * BLK; base < n; base += (uint64_t)gridDim.x * BLK) {
        const uint64_t i = a + base + threadIdx.x;
        bool has = false;
        uint32_t o = 0;
        unsigned long long tg = 0;
        if (base + threadIdx.x < n) {
            const unsigned long long w = word[i];
            if (dw_open(w, flag[i])) {
                has = true;
                tg = (w >> b.hb) & b.idm;
                o = dw_owner_of(ow, tg);
            }
        }
        const unsigned long long at = dw_wave_count(lds_dw, has, o);
        if (PASS == 1 && has) {
            q[at] = tg;
            tag[at] = i;
        }
    }
    dw_pass_end<PASS>(lds_dw, ow.world, bc_or_off);
}
// owner side: the words of the nodes that were asked for (global ids -> local)
__global__ void __launch_bounds__(BLK) k_dw_gather_words(const unsigned long long *__restrict__ q, uint64_t n, unsigned long long my_base, uint64_t n2,
                                                         const unsigned long long *__restrict__ word, unsigned long long *rows, uint32_t *err) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLK) {
        const unsigned long long loc = q[i] - my_base;
        if (loc >= n2) {
            atomicAdd(err, 1u);
            rows[i] = DW_F | DW_T;
        } else {
            rows[i] = word[loc];
        }
    }
}
// one doubling step of the asking nodes: wp = word of the pointer. A pointer only ever moves ahead along its chain, so a word of this round
// or of the one before is equally good. Only a node that FINISHES can have a chain length to overflow (a node on a perfect loop never
// finishes and doubles its count every round): open nodes saturate at HM — sticky — and a node that finishes with HM or more is refused.
__global__ void __launch_bounds__(BLK) k_dw_double_apply(const unsigned long long *__restrict__ tag, const unsigned long long *__restrict__ wp_, uint64_t n, DwBits b,
                                                         unsigned long long *word, uint32_t *too_long) {
    for (uint64_t s = (uint64_t)blockIdx.x * BLK + threadIdx.x; s < n; s += (uint64_t)gridDim.x * BLK) {
        const unsigned long long node = tag[s], mw = word[node], wp = wp_[s];
        const unsigned long long tg = (mw >> b.hb) & b.idm;
        const bool p_tail = (wp & DW_T) != 0, p_fin = (wp & DW_F) != 0;
        unsigned long long hops = (mw & b.hm) + (p_tail ? 0ull : (wp & b.hm));
        if (p_fin && hops >= b.hm) atomicAdd(too_long, 1u);
        if (hops > b.hm) hops = b.hm;
        const unsigned long long nid = p_tail ? tg : ((wp >> b.hb) & b.idm);
        word[node] = (p_fin ? DW_F : 0ull) | (nid << b.hb) | hops;
    }
}
// k-mers that never finished (perfect loops): tiles of CAND_TILE local ranks — counts, a scan over the tiles, ascending list
template <int PASS>
__global__ void __launch_bounds__(BLK) k_dw_loops(const unsigned long long *__restrict__ word, const uint8_t *__restrict__ flag, uint64_t D0, unsigned long long *tcnt_or_off,
                                                  unsigned long long *list) {
    __shared__ uint32_t scratch[BLK / 64 + 2];
    const uint64_t r0 = (uint64_t)blockIdx.x * CAND_TILE + (uint64_t)threadIdx.x * CAND_PER;
    uint32_t c = 0, fl = 0;
    for (int j = 0; j < CAND_PER; ++j) {
        const uint64_t r = r0 + j;
        if (r < D0 && (dw_open(word[2 * r], flag[2 * r]) || dw_open(word[2 * r + 1], flag[2 * r + 1]))) {
            fl |= 1u << j;
            ++c;
        }
    }
    uint32_t tot;
    const uint32_t ex = block_excl_scan<uint32_t>(c, scratch, &tot);
    if (PASS == 0) {
        if (threadIdx.x == 0) tcnt_or_off[blockIdx.x] = tot;
    } else {
        unsigned long long o = tcnt_or_off[blockIdx.x] + ex;
        for (int j = 0; j < CAND_PER; ++j)
            if (fl & (1u << j)) list[o++] = r0 + j;
    }
}
// heads of the chains: a finished chain k-mer whose reverse-strand node is a tail. One bit per node and the number of heads before every
// 64 nodes (hpre, after the caller's scan): slot of a head = hpre[node / 64] + popcount of the lower bits — no search, 16 B per 64 nodes.
__global__ void __launch_bounds__(BLK) k_dw_head_bits(const unsigned long long *__restrict__ word, const uint8_t *__restrict__ flag, uint64_t n2, uint64_t nwords,
                                                      unsigned long long *hbits, unsigned long long *hcnt) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t w = (uint64_t)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6); w < nwords; w += (uint64_t)gridDim.x * (BLK / 64)) {
        const uint64_t nd = w * 64 + lane;
        const bool h = nd < n2 && dw_done(word[nd], flag[nd]) && (word[nd ^ 1] & DW_T);
        const unsigned long long bal = __ballot(h);
        if (lane == 0) {
            hbits[w] = bal;
            hcnt[w] = (unsigned long long)__popcll(bal);
        }
    }
}
__device__ __forceinline__ bool dw_head_slot(const unsigned long long *__restrict__ hbits, const unsigned long long *__restrict__ hpre, unsigned long long loc, uint64_t n2,
                                             unsigned long long &slot) {
    if (loc >= n2) return false;
    const unsigned long long bal = hbits[loc >> 6], bit = loc & 63ull;
    if (!((bal >> bit) & 1ull)) return false;
    slot = hpre[loc >> 6] + (unsigned long long)__popcll(bal & ((1ull << bit) - 1ull));
    return true;
}
// k-mers of every chain (a head that is its own tail: 1), in head order; the caller scans them into the chains' places
__global__ void __launch_bounds__(BLK) k_dw_head_len(const unsigned long long *__restrict__ word, const unsigned long long *__restrict__ hbits,
                                                     const unsigned long long *__restrict__ hpre, uint64_t n2, DwBits b, unsigned long long *hlen) {
    for (uint64_t nd = (uint64_t)blockIdx.x * BLK + threadIdx.x; nd < n2; nd += (uint64_t)gridDim.x * BLK) {
        unsigned long long slot;
        if (!dw_head_slot(hbits, hpre, nd, n2, slot)) continue;
        const unsigned long long hw = word[nd];
        hlen[slot] = ((hw & DW_T) ? 0ull : (hw & b.hm)) + 1ull;
    }
}
// every finished chain k-mer x of [a, a + n) to the head of its chain: x is hops(x^1) steps behind the head tail(x^1)^1 (the reverse strand
// went through the same doubling) — message (head, steps << 2 | outgoing nucleotide); a tail also tells which junction node ends the chain:
// (head, -(end node) - 1)
template <int PASS>
__global__ void __launch_bounds__(BLK) k_dw_head_msgs(const unsigned long long *__restrict__ word, const uint8_t *__restrict__ flag, uint64_t a, uint64_t n,
                                                      unsigned long long my_base, DwBits b, DwOwners ow, unsigned long long *bc_or_off, ulonglong2 *msg) {
    extern __shared__ unsigned long long lds_dw[];
    dw_pass_begin<PASS>(lds_dw, ow.world, bc_or_off);
    for (uint64_t base = (uint64_t)blockIdx.x * BLK; base < n; base += (uint64_t)gridDim.x * BLK) {
        const uint64_t xs = a + base + threadIdx.x;
        bool has0 = false, has1 = false;
        uint32_t o = 0;
        unsigned long long head = 0, pay0 = 0, pay1 = 0;
        if (base + threadIdx.x < n) {
            const unsigned long long ws = word[xs];
            const uint8_t f = flag[xs];
            if (dw_done(ws, f)) {
                const unsigned long long xr = xs ^ 1ull, wr = word[xr];
                head = ((wr & DW_T) ? (xr + my_base) : ((wr >> b.hb) & b.idm)) ^ 1ull;
                const unsigned long long back = (wr & DW_T) ? 0ull : (wr & b.hm);
                pay0 = (back << 2) | ((unsigned long long)(f >> 2) & 3ull);
                has0 = true;
                o = dw_owner_of(ow, head);
                if (ws & DW_T) {
                    has1 = true;
                    pay1 = ~((ws >> b.hb) & b.idm);  // = -(end) - 1
                }
            }
        }
        const unsigned long long at0 = dw_wave_count(lds_dw, has0, o), at1 = dw_wave_count(lds_dw, has1, o);
        if (PASS == 1) {
            if (has0) msg[at0] = make_ulonglong2(head, pay0);
            if (has1) msg[at1] = make_ulonglong2(head, pay1);
        }
    }
    dw_pass_end<PASS>(lds_dw, ow.world, bc_or_off);
}
// owner of the heads: every message to its place. stats[0] += nucleotides placed, [1] += messages for a node that heads no chain or beyond its chain
__global__ void __launch_bounds__(BLK) k_dw_place(const ulonglong2 *__restrict__ msg, uint64_t n, unsigned long long my_base, uint64_t n2,
                                                  const unsigned long long *__restrict__ hbits, const unsigned long long *__restrict__ hpre,
                                                  const unsigned long long *__restrict__ hoff, unsigned long long *hend, uint8_t *bases, unsigned long long *stats) {
    __shared__ unsigned long long scratch[BLK / 64 + 2];
    unsigned long long placed = 0, bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLK) {
        const ulonglong2 m = msg[i];
        unsigned long long slot;
        if (!dw_head_slot(hbits, hpre, m.x - my_base, n2, slot)) {
            ++bad;
            continue;
        }
        if ((long long)m.y < 0) {
            hend[slot] = ~m.y;
        } else {
            const unsigned long long pos = hoff[slot] + (m.y >> 2);
            if (pos >= hoff[slot + 1]) {
                ++bad;
                continue;
            }
            bases[pos] = (uint8_t)(m.y & 3ull);
            ++placed;
        }
    }
    unsigned long long tot;
    block_excl_scan<unsigned long long>(placed, scratch, &tot);
    if (threadIdx.x == 0 && tot) atomicAdd(&stats[0], tot);
    block_excl_scan<unsigned long long>(bad, scratch, &tot);
    if (threadIdx.x == 0 && tot) atomicAdd(&stats[1], tot);
}
__global__ void __launch_bounds__(BLK) k_dw_count_unset(const unsigned long long *__restrict__ hend, uint64_t n, unsigned long long *out) {
    __shared__ unsigned long long scratch[BLK / 64 + 2];
    unsigned long long c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLK) c += hend[i] == ~0ull ? 1 : 0;
    unsigned long long tot;
    block_excl_scan<unsigned long long>(c, scratch, &tot);
    if (threadIdx.x == 0 && tot) atomicAdd(out, tot);
}
// the start de-edges of [a, a + n) whose first node is a chain k-mer ask the owner of that node — the head of a chain — for the chain
template <int PASS>
__global__ void __launch_bounds__(BLK) k_dw_start_asks(const unsigned long long *__restrict__ c_first, const uint8_t *__restrict__ c_fj, uint64_t a, uint64_t n,
                                                       DwOwners ow, unsigned long long *bc_or_off, unsigned long long *q, unsigned long long *tag) {
    extern __shared__ unsigned long long lds_dw[];
    dw_pass_begin<PASS>(lds_dw, ow.world, bc_or_off);
    for (uint64_t base = (uint64_t)blockIdx.x * BLK; base < n; base += (uint64_t)gridDim.x * BLK) {
        const uint64_t i = a + base + threadIdx.x;
        bool has = false;
        uint32_t o = 0;
        unsigned long long tg = 0;
        if (base + threadIdx.x < n && !c_fj[i]) {
            has = true;
            tg = c_first[i];
            o = dw_owner_of(ow, tg);
        }
        const unsigned long long at = dw_wave_count(lds_dw, has, o);
        if (PASS == 1 && has) {
            q[at] = tg;
            tag[at] = i;
        }
    }
    dw_pass_end<PASS>(lds_dw, ow.world, bc_or_off);
}
// owner of the heads: (length, end node) of the chain behind every asked node, its slot and its length once more for the scan of the bases
__global__ void __launch_bounds__(BLK) k_dw_answer(const unsigned long long *__restrict__ asks, uint64_t n, unsigned long long my_base, uint64_t n2,
                                                   const unsigned long long *__restrict__ hbits, const unsigned long long *__restrict__ hpre,
                                                   const unsigned long long *__restrict__ hoff, const unsigned long long *__restrict__ hend, ulonglong2 *rows,
                                                   unsigned long long *slots, unsigned long long *lens, uint32_t *err) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLK) {
        unsigned long long slot = 0, len = 0, end = ~0ull;
        if (dw_head_slot(hbits, hpre, asks[i] - my_base, n2, slot)) {
            len = hoff[slot + 1] - hoff[slot];
            end = hend[slot];
        } else {
            atomicAdd(err, 1u);  // a start de-edge leads to a k-mer that heads no chain
        }
        if (rows) rows[i] = make_ulonglong2(len, end);
        if (slots) slots[i] = slot;
        if (lens) lens[i] = len;
    }
}
__global__ void __launch_bounds__(BLK) k_dw_place_rows(const unsigned long long *__restrict__ tag, const ulonglong2 *__restrict__ rows, uint64_t n, unsigned long long *steps,
                                                       unsigned long long *last, uint32_t *err) {
    for (uint64_t s = (uint64_t)blockIdx.x * BLK + threadIdx.x; s < n; s += (uint64_t)gridDim.x * BLK) {
        const ulonglong2 r = rows[s];
        if (r.x == 0) atomicAdd(err, 1u);
        steps[tag[s]] = r.x;
        last[tag[s]] = r.y;
    }
}
// the nucleotides of the asked chains, one after the other in the order of the asks (off = scan of their lengths)
__global__ void __launch_bounds__(BLK) k_dw_ragged_copy(const unsigned long long *__restrict__ slots, const unsigned long long *__restrict__ off, uint64_t n,
                                                        const unsigned long long *__restrict__ hoff, const uint8_t *__restrict__ bases, uint8_t *flat) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLK) {
        const unsigned long long o = off[i], len = off[i + 1] - o, src = hoff[slots[i]];
        for (unsigned long long j = 0; j < len; ++j) flat[o + j] = bases[src + j];
    }
}
__global__ void __launch_bounds__(BLK) k_dw_lens_of(const unsigned long long *__restrict__ tag, uint64_t n, const unsigned long long *__restrict__ steps, unsigned long long *lens) {
    for (uint64_t s = (uint64_t)blockIdx.x * BLK + threadIdx.x; s < n; s += (uint64_t)gridDim.x * BLK) lens[s] = steps[tag[s]];
}
// ... and on the asking side from the order of the asks to the places of the start de-edges (boff: scan of steps in de-edge order)
__global__ void __launch_bounds__(BLK) k_dw_scatter_bases(const unsigned long long *__restrict__ tag, const unsigned long long *__restrict__ off, uint64_t n,
                                                          const unsigned long long *__restrict__ boff, const uint8_t *__restrict__ mine, uint8_t *my_bases) {
    for (uint64_t s = (uint64_t)blockIdx.x * BLK + threadIdx.x; s < n; s += (uint64_t)gridDim.x * BLK) {
        const unsigned long long o = off[s], len = off[s + 1] - o, dst = boff[tag[s]];
        for (unsigned long long j = 0; j < len; ++j) my_bases[dst + j] = mine[o + j];
    }
}

}  // namespace smx
