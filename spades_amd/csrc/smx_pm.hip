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
    unsigned xs;                      // EXT_BITS: the records carry their InOutMask byte (EXT layout); 0: plain k-mers, the byte lives in `mask` alone ("nx", k without 8 spare record bits)
    const uint8_t *mask;              // [records] InOutMask bytes (what an nx graph reads instead of the record's low byte)
    unsigned bym;                     // 1: the bytes are read from `mask` whatever the records carry — plain records, or EXT records whose bytes an early
                                      // clipper has left behind (it edits the mask array: the record bytes are the UNCLIPPED ones from then on)
};
template <int NW>
__device__ __forceinline__ unsigned pm_byte(const PmIndex &ix, const Rec<NW> &raw, uint64_t r) {
    return ix.bym ? (unsigned)ix.mask[r] : (unsigned)(raw.w[NW - 1] & 0xFFu);
}

struct PmWalk {  // what graph_from_masks (smx_construct.hpp) needs beyond the node table on this route
    PmIndex ix;
    const uint32_t *jmp;
    const unsigned long long *cinfo;  // [nchunks] base | winners << 40
    const uint32_t *cob;              // chunk of every 256th record (k_pm_cob)
    uint32_t nchunks;
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
__device__ __forceinline__ node_t pm_probe(const Rec<NW> *__restrict__ recs, const uint32_t *gm, uint32_t T, uint64_t base, const Rec<NW> &y, uint32_t h32, unsigned xs = EXT_BITS,
                                           Rec<NW> *raw_out = nullptr /* the record as stored (EXT: with its byte) when found */) {
    uint32_t h = h32 & (T - 1);
    for (uint32_t it = 0; it < T; ++it) {
        const uint32_t g = gm[h >> 4];
        const uint32_t occ = g >> 16, bit = h & 15u;
        if (!((occ >> bit) & 1u)) return NODE_NONE;
        const uint64_t idx = base + (g & 0xFFFFu) + __popc(occ & ((1u << bit) - 1u));
        const Rec<NW> raw = recs[idx];
        if (rec_eq<NW>(rec_pure_xs<NW>(raw, xs), y)) {
            if (raw_out) *raw_out = raw;
            return idx;
        }
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
    return pm_probe<NW>((const Rec<NW> *)ix.recs, ix.meta + (size_t)(pi >> PM_BASE_BITS) * ix.ngroups, ix.T, pi & PM_BASE_MASK, y, rec_hash32<NW>(y), ix.xs);
}
// the same for a successor of a k-mer of the sorted tail: its partition was cut, and a path seldom leaves a cut partition at once — the
// tail's own directory first (hash + directory word + record, no minimizer scan), the partition table only when it is not there
template <int NW>
__device__ __forceinline__ node_t pm_find_from_tail(const PmIndex &ix, const Rec<NW> &y) {
    const node_t r = kmer_rank<NW, false>((const Rec<NW> *)ix.dk, ix.ddir, y);
    return r != NODE_NONE ? ix.nclean + r : pm_find<NW>(ix, y);
}
template <int NW>
__device__ __forceinline__ Rec<NW> pm_node_kmer(const Rec<NW> *__restrict__ recs, node_t node, unsigned k, unsigned xs = EXT_BITS) {  // oriented k-mer of a node
    const Rec<NW> x = rec_pure_xs<NW>(recs[node >> 1], xs);
    return (node & 1) ? rec_rc<NW>(x, k) : x;
}

// chunk (id, base) of clean record r
__device__ __forceinline__ uint32_t pm_chunk_of(const unsigned long long *__restrict__ cinfo, const uint32_t *__restrict__ cob, uint32_t nchunks, uint64_t r,
                                                uint64_t &base) {
    uint32_t c = cob[r >> 8];  // the chunk of record 256 * (r >> 8); r lies in it or in one of the next few
    unsigned long long ci = cinfo[c];
    while (c + 1 < nchunks) {
        const unsigned long long nx = cinfo[c + 1];
        if ((nx & PM_BASE_MASK) > r) break;
        ci = nx;
        ++c;
    }
    base = ci & PM_BASE_MASK;
    return c;
}

__device__ __forceinline__ uint32_t pm_jump_steps(uint32_t j) { return (j >> 16) & 0x7FFFu; }  // (bit 31 of a jump word: its far end is a junction, see k_pm_tab)
// the early clippers' view of the graph on this route (FileFind in smx_graph.hip is the other one): index = place in the partition-major records
template <int NW>
struct PmFind {
    PmIndex ix;
    __device__ __forceinline__ Rec<NW> kmer(uint64_t r) const { return rec_pure_xs<NW>(((const Rec<NW> *)ix.recs)[r], ix.xs); }
    __device__ __forceinline__ node_t find(const Rec<NW> &y) const { return pm_find<NW>(ix, y); }
    // the successor y of the k-mer at index r: in r's own chunk 19 times in 20 (no minimizer scan, no partition word: one line of 128 B less per branch), and the
    // byte of the found record where the record bytes are still the graph's masks (bytes_ok: no clipper has edited the mask array yet) — else -1
    const unsigned long long *cinfo = nullptr;
    const uint32_t *cob = nullptr;
    uint32_t nchunks = 0;
    unsigned bytes_ok = 0;
    __device__ __forceinline__ node_t find_from(const Rec<NW> &y, uint64_t r, int &byte0) const {
        byte0 = -1;
        if (cinfo && r < ix.nclean) {
            uint64_t cbase;
            const uint32_t cid = pm_chunk_of(cinfo, cob, nchunks, r, cbase);
            Rec<NW> raw;
            const node_t ry = pm_probe<NW>((const Rec<NW> *)ix.recs, ix.meta + (size_t)cid * ix.ngroups, ix.T, cbase, y, rec_hash32<NW>(y), ix.xs, &raw);
            if (ry != NODE_NONE) {
                if (bytes_ok && ix.xs) byte0 = (int)(raw.w[NW - 1] & 0xFFu);
                return ry;
            }
        }
        return pm_find<NW>(ix, y);
    }
    // (the table the clippers walk IS the route's node table here: no second array of 16 B per k-mer next to it)
    __device__ __forceinline__ node_t next(const node_t *__restrict__ tab, node_t nd) const { return tab[nd] & TAB_NODE_MASK; }
    const uint32_t *jmp;  // the jump words of the node table (chains inside a chunk)
    // FindForward with the route's jump words: a chain of s non-junction k-mers inside a chunk is crossed with one word, as the unitig walks
    // cross it — exactly: the reference's loop takes a step while cnt < bound, so it reaches the far end of the chain iff cnt + s <= bound, and
    // otherwise stops INSIDE it, on a non-junction k-mer (which is no dead end: the branch is "too long" whatever that node is). 95 steps are
    // ~6 chunks: a dozen sectors instead of 95 dependent ones per branch.
    __device__ __forceinline__ node_t advance(const node_t *__restrict__ tab, const uint8_t *__restrict__ mask, node_t nd, uint32_t &cnt, uint32_t bound, int byte0 = -1) const {
        while (cnt < bound) {
            const unsigned mb = byte0 >= 0 ? (unsigned)byte0 : (unsigned)mask[nd >> 1];  // (the first node's byte may come with its record: find_from)
            byte0 = -1;
            if (mask_junction(mb)) break;
            const uint32_t j = jmp[nd], s = pm_jump_steps(j);
            if (s) {
                if (cnt + s > bound) {  // the loop would stop on an interior k-mer of this chain: any of them tells the caller "not a tip"
                    cnt = bound;
                    return nd;
                }
                cnt += s;
                nd = (node_t)((long long)nd + (long long)(int16_t)(j & 0xFFFFu));
                if (bytes_ok) {
                    // Bit 31 of the jump word still tells the truth (no clipper has edited the masks): set — the chain's last node is a junction, the loop would
                    // read its byte and stop (the caller reads that byte anyway); clear — it is a non-junction k-mer whose successor lies in another chunk: the loop
                    // would read its byte (no junction), its jump word (0: no chain starts at a chain's end) and step through its entry. One line instead of three.
                    if (j >> 31) break;
                    if (cnt < bound) {
                        ++cnt;
                        nd = tab[nd] & TAB_NODE_MASK;
                    }
                }
                continue;
            }
            ++cnt;
            nd = tab[nd] & TAB_NODE_MASK;
        }
        return nd;
    }

    // The k-mers of a removed tip. A lane that steps through them one by one has a dependent read per k-mer in front of it (150 ms of the clipper at
    // config 3, measured); here it crosses a chunk with the jump word and only MARKS the head of the chain (hmark[head]: "isolate the s + 1 k-mers of this
    // chain"): a chain of non-junction k-mers lies inside the tip as a whole (a tip ends at a dead end, which ends its chain). k_pm_isolate_chains then follows
    // the marked chains chunk by chunk, in LDS. K-mers outside chains (no jump word: the sorted tail, single k-mers) are isolated here.
    __device__ __forceinline__ void isolate_tip(const node_t *__restrict__ tab, node_t nd, uint32_t len, uint8_t *isolate, uint8_t *hmark) const {
        uint32_t left = len;
        while (left) {
            const uint32_t j = jmp[nd], s = pm_jump_steps(j);
            if (s && s + 1 <= left) {
                hmark[nd] = 1;
                left -= s + 1;
                nd = (node_t)((long long)nd + (long long)(int16_t)(j & 0xFFFFu));  // the far end: isolated with its chain
            } else {
                isolate[nd >> 1] = 1;
                --left;
            }
            if (left) nd = tab[nd] & TAB_NODE_MASK;
        }
    }
};

// the marked chains of a chunk (PmFind::isolate_tip): the chunk's node-table entries are loaded into LDS once, every marked head is followed for the steps its jump
// word counts. One workgroup per chunk; chunks without a mark cost their 2 bytes per k-mer of marks.
__global__ void __launch_bounds__(BLK) k_pm_isolate_chains(const unsigned long long *__restrict__ cinfo, uint32_t nchunks, uint32_t maxn, const node_t *__restrict__ tab,
                                                           const uint32_t *__restrict__ jmp, const uint8_t *__restrict__ hmark, uint8_t *isolate,
                                                           const uint32_t *__restrict__ llink /* the dedupe stage's local links where they were kept (4 B per k-mer instead of the
                                                                                                 16 B of its two node entries: a chain IS a run of these links), else nullptr */) {
    extern __shared__ __attribute__((aligned(16))) uint32_t pm_lds[];
    uint32_t *ls = pm_lds;                     // [2 * maxn] local successor of every node (0xFFFFFFFF: leaves the chunk)
    uint16_t *list = (uint16_t *)(ls + 2 * maxn);  // marked heads
    __shared__ uint32_t s_n;
    for (uint32_t cid = blockIdx.x; cid < nchunks; cid += gridDim.x) {
        const unsigned long long ci = cinfo[cid];
        const uint64_t base = ci & PM_BASE_MASK;
        const uint32_t nn = 2 * (uint32_t)(ci >> PM_BASE_BITS);
        __syncthreads();
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        if (nn > 2 * maxn) continue;
        for (uint32_t nd = threadIdx.x; nd < nn; nd += BLK)
            if (hmark[2 * base + nd]) list[atomicAdd(&s_n, 1u)] = (uint16_t)nd;
        __syncthreads();
        const uint32_t n = s_n;
        if (!n) continue;
        if (llink) {
            for (uint32_t r = threadIdx.x; r < nn / 2; r += BLK) {
                const uint32_t ll = llink[base + r], l0 = ll & 0xFFFFu, l1 = ll >> 16;
                ls[2 * r] = l0 < nn ? l0 : 0xFFFFFFFFu;
                ls[2 * r + 1] = l1 < nn ? l1 : 0xFFFFFFFFu;
            }
        } else {
            for (uint32_t nd = threadIdx.x; nd < nn; nd += BLK) {
                const node_t t = tab[2 * base + nd] & TAB_NODE_MASK;
                ls[nd] = (t >= 2 * base && t < 2 * base + nn) ? (uint32_t)(t - 2 * base) : 0xFFFFFFFFu;
            }
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += BLK) {
            uint32_t cur = list[i];
            const uint32_t s = pm_jump_steps(jmp[2 * base + cur]);
            isolate[base + (cur >> 1)] = 1;
            for (uint32_t t = 0; t < s && cur != 0xFFFFFFFFu; ++t) {
                cur = ls[cur];
                if (cur != 0xFFFFFFFFu) isolate[base + (cur >> 1)] = 1;
            }
        }
    }
}

// Jump words: jmp[node] = (delta to the last node of the chain that stays inside the node's chunk, 16 bits signed) | steps << 16 (15 bits) | bit 31: that last node
// is a junction BY THE MASKS THE TABLE WAS MADE ON — a walk that reads the word need not read the far end's node entry to learn that its path ends there
// (k_pm_walk_len; ignored where an early clipper has edited the masks since, PmIndex::bym).
// A walk that enters a chunk reads ONE word to cross it (smx_pm_walk_len) instead of one node-table entry per k-mer.
constexpr uint16_t PM_ADV_NONE = 0xFFFFu;

// Node table of the clean chunks: one workgroup per chunk. The dedupe stage already knows which k-mers of a chunk follow each other:
// two k-mers side by side in one super-k-mer of some read are a de Bruijn edge between two entries of its LDS hash table, and it
// hands those local links out with the records (PmOut::llink, 16 bits per node; ~95 % of the successors — consecutive k-mers of a
// path share their minimizer for ~17 steps). So this kernel computes nothing about k-mers: per node, outgoing extensions from the
// mask byte; where there is exactly one, the successor is the local link, or, without one, TAB_NODE_MASK for k_pm_remote (the
// successor lies across a super-k-mer boundary: another partition, looked up there). Then the chains inside the chunk are followed
// in LDS, once each, from their heads (the nodes no local link leads to — the only places where a walk can enter a chain) by a
// dense loop over a list; only heads get a jump word, the others 0 (= step by the node table).
// (Round-3 history: with the successor k-mer hashed and probed here — canonical form, hash, group word, record compare, ~900
// instructions per k-mer — this kernel took 124 ms at config 3, ALU-bound.)
// stats[0] += extension bits. LDS (dynamic): jw[2 * maxn] u32 | list[2 * maxn] u16 (chain heads) | hp[maxn / 16] u32
// Canonical successor k-mers of both orientations of x from ONE reverse complement: with y = RC(x), the successor of orientation 0
// by nucleotide c is z = x[1..] + c and RC(z) = (3 - c) + y[..k-2]; of orientation 1, z = y[1..] + c and RC(z) = (3 - c) + x[..k-2] —
// shifts of x and y. yo = 1 where the successor is the non-minimal strand.
template <int NW>
__device__ __forceinline__ Rec<NW> pm_succ_from(const Rec<NW> &fw, const Rec<NW> &bw, unsigned k, unsigned mo, unsigned &yo) {
    const unsigned c = __ffs(mo) - 1;
    const Rec<NW> z = rec_shl<NW>(fw, k, c), zr = rec_shr<NW>(bw, k, 3u - c);
    const bool minimal = rc_ge<NW>(zr, z);
    yo = minimal ? 0u : 1u;
    return minimal ? z : zr;
}
template <int NW>
__device__ __forceinline__ Rec<NW> pm_succ_kmer(const Rec<NW> &x, unsigned k, unsigned o, unsigned mo, unsigned &yo) {  // canonical successor of orientation o
    const Rec<NW> y = rec_rc<NW>(x, k);
    return o ? pm_succ_from<NW>(y, x, k, mo, yo) : pm_succ_from<NW>(x, y, k, mo, yo);
}
// palindromic (k+1)-mers among the extensions of x (k_ext_split's second figure), registers only. x + c is its own reverse complement
// iff c = complement of x[0] and x[1..k-1] equals its reverse complement, which is y[0..k-2] (y = RC(x)); likewise b + x on the other
// side with x[0..k-2] against y[1..k-1].
template <int NW>
__device__ __forceinline__ unsigned pm_palindromes(const Rec<NW> &x, unsigned m, unsigned k) {
    const unsigned x0 = rec_nucl<NW>(x, 0), xl = (unsigned)(x.w[NW - 1] >> (((k - 1) & 31u) << 1)) & 3u;
    const Rec<NW> y = rec_rc<NW>(x, k);
    unsigned n = 0;
    if ((m >> (3 - x0)) & 1) n += rec_eq<NW>(rec_suffix<NW>(x), rec_prefix<NW>(y, k - 1)) ? 1u : 0u;
    if ((m >> (7 - xl)) & 1) n += rec_eq<NW>(rec_prefix<NW>(x, k - 1), rec_suffix<NW>(y)) ? 1u : 0u;
    return n;
}
__global__ void __launch_bounds__(BLK) k_pm_tab(const unsigned long long *__restrict__ cinfo, uint32_t nchunks, uint32_t maxn, const uint8_t *__restrict__ mask,
                                                const uint32_t *__restrict__ llink, node_t *tab, uint32_t *jmp, unsigned long long *stats, uint32_t *err,
                                                unsigned long long *prof, uint32_t *rbits /* [nchunks * (maxn >> 4)]: bit nd of a chunk's words = node nd asks k_pm_remote */,
                                                const uint8_t *__restrict__ mask_orig /* the masks the local links were made on, when an early clipper has edited `mask` since; else nullptr */) {
    extern __shared__ __attribute__((aligned(16))) uint32_t pm_lds[];
    uint32_t *jw = pm_lds;
    uint16_t *list = (uint16_t *)(jw + 2 * maxn);
    uint32_t *hp = (uint32_t *)(list + 2 * maxn);  // bit nd: some local link leads to node nd
    uint32_t *rm = hp + (maxn >> 4);               // bit nd: the successor of node nd is not in this chunk (for k_pm_remote)
    __shared__ uint32_t s_nhead;
    unsigned long long bits = 0;
    // SMX_DEBUG: prof[0..3] = 100 MHz ticks of thread 0 (stage, node table, chain heads, jumps), [4] chunks, [5] successors outside their chunk, [6] chain heads
    unsigned long long pt[4] = {0, 0, 0, 0}, t0 = 0, pc[3] = {0, 0, 0};
#define PM_T(i)                                  \
    if (prof && threadIdx.x == 0) {              \
        unsigned long long t1 = wall_clock64();  \
        pt[i] += t1 - t0;                        \
        t0 = t1;                                 \
    }
    if (prof && threadIdx.x == 0) t0 = wall_clock64();
    unsigned long long ci = blockIdx.x < nchunks ? cinfo[blockIdx.x] : 0ull;
    for (uint32_t cid = blockIdx.x; cid < nchunks; cid += gridDim.x) {
        const uint64_t base = ci & PM_BASE_MASK;
        const uint32_t n = (uint32_t)(ci >> PM_BASE_BITS), nn = 2 * n;
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < (maxn >> 4); t += BLK) {
            hp[t] = 0;
            rm[t] = 0;
        }
        if (threadIdx.x == 0) s_nhead = 0;
        if (cid + gridDim.x < nchunks) ci = cinfo[cid + gridDim.x];  // the next chunk's descriptor flies while this one is worked on
        __syncthreads();
        PM_T(0)
        if (n > maxn) {  // cannot happen (a chunk never has more winners than its capacity): never leave the LDS arrays
            if (threadIdx.x == 0) atomicAdd(err, 1u);
            continue;
        }
        // (the byte and the link word of the thread's NEXT record are asked for before this one is worked on: a chunk gives a thread ~4
        // records, and with one load -> use -> store chain after the other the phase was latency, round-5 ticks: 574 of a chunk's 1100)
        unsigned m_nx = 0, mo_nx = 0;
        uint32_t ll_nx = 0;
        if (threadIdx.x < n) {
            m_nx = mask[base + threadIdx.x];
            ll_nx = llink[base + threadIdx.x];
            if (mask_orig) mo_nx = mask_orig[base + threadIdx.x];
        }
        for (uint32_t r = threadIdx.x; r < n; r += BLK) {
            const unsigned m = m_nx, m_was = mask_orig ? mo_nx : m_nx;
            const uint32_t ll = ll_nx;
            if (r + BLK < n) {
                m_nx = mask[base + r + BLK];
                ll_nx = llink[base + r + BLK];
                if (mask_orig) mo_nx = mask_orig[base + r + BLK];
            }
            bits += __popc(m);
            const bool junction = mask_junction(m);
            node_t e[2];
#pragma unroll
            for (unsigned o = 0; o < 2; ++o) {
                const unsigned mo = (o ? brev8(m) : m) & 15u;
                const uint32_t l = o ? (ll >> 16) : (ll & 0xFFFFu);
                uint32_t w = 0xFFFFu;  // no local successor
                e[o] = (node_t)mo << TAB_OUT_SHIFT;
                if (uniq4(mo)) {
                    // (a local link is the neighbour in SOME read: it is the successor only where the k-mer had this one extension when the links were
                    // made — an orientation that lost branches to a clipper since looks its remaining successor up like a remote one)
                    const bool link_ok = uniq4((o ? brev8(m_was) : m_was) & 15u);
                    if (l != 0xFFFFu && l < nn && link_ok) {
                        e[o] |= 2 * base + l;
                        if (!junction) {
                            w = l;
                            atomicOr(&hp[w >> 5], 1u << (w & 31u));
                        }
                    } else {  // not next to it in any super-k-mer: k_pm_remote looks it up through the partition table
                        e[o] |= TAB_NODE_MASK;
                        atomicOr(&rm[(2 * r + o) >> 5], 1u << ((2 * r + o) & 31u));
                        if (prof) ++pc[0];
                    }
                }
                jw[2 * r + o] = w;
            }
            __builtin_nontemporal_store(smx_ull2{e[0], e[1]}, reinterpret_cast<smx_ull2 *>(tab + 2 * (base + r)));  // both orientations: one 16-byte store, streamed (nt: 37.0 -> 33.2 ms)
        }
        __syncthreads();
        PM_T(1)
        for (uint32_t t = threadIdx.x; t < (maxn >> 4); t += BLK) rbits[(size_t)cid * (maxn >> 4) + t] = rm[t];
        for (uint32_t nd = threadIdx.x; nd < nn; nd += BLK)  // chain heads: a local successor, no local predecessor
            if (jw[nd] != 0xFFFFu && !((hp[nd >> 5] >> (nd & 31u)) & 1u)) list[atomicAdd(&s_nhead, 1u)] = (uint16_t)nd;
        __syncthreads();
        PM_T(2)
        const uint32_t nhead = s_nhead;
        for (uint32_t i = threadIdx.x; i < nhead; i += BLK) {  // every chain once, from its head
            const uint32_t h = list[i];
            uint32_t cur = h, st = 0;
            for (uint32_t a; (a = jw[cur] & 0xFFFFu) != 0xFFFFu && st < nn; ++st) cur = a;
            // (bit 31 marks the word as a result; st <= nn < 2^15; bit 30 here = the far end is a junction, moved to bit 31 of the word that is stored)
            jw[h] = (((cur - h) & 0xFFFFu) | (st << 16)) | 0x80000000u | (mask_junction(mask[base + (cur >> 1)]) ? 0x40000000u : 0u);
        }
        __syncthreads();
        for (uint32_t nd = threadIdx.x; nd < nn; nd += BLK) {
            const uint32_t v = jw[nd];
            __builtin_nontemporal_store((v & 0x80000000u) ? ((v & 0x3FFFFFFFu) | ((v & 0x40000000u) << 1)) : 0u, jmp + 2 * base + nd);
        }
        if (prof && threadIdx.x == 0) {
            pc[1] += nhead;
            pc[2] += 1;
        }
        PM_T(3)
    }
    if (prof) {
        if (threadIdx.x == 0) {
            for (int i = 0; i < 4; ++i) atomicAdd(&prof[i], pt[i]);
            atomicAdd(&prof[4], pc[2]);
            atomicAdd(&prof[6], pc[1]);
        }
        if (pc[0]) atomicAdd(&prof[5], pc[0]);
    }
#undef PM_T
    for (int o = 32; o > 0; o >>= 1) bits += __shfl_down(bits, o, 64);
    if ((threadIdx.x & 63) == 0 && bits) atomicAdd(&stats[0], bits);
}
// The successors k_pm_tab did not find in the node's own chunk (~5 % of the nodes; their entries carry TAB_NODE_MASK in the node field and
// their bit in the chunk's words of rbits): a wave takes a chunk, expands its bits (64 words at a time) into an LDS list of its own, then
// a dense loop in which every lane has a lookup — minimizer scan of the successor k-mer, partition table, the other
// chunk's group word, the record. (Left inside k_pm_tab's per-node loop, nearly every wave paid that path for its few misses: 62 of 80 us
// per chunk. Round 3 found the marked nodes by scanning the whole node table: 69 GB read for 3 GB of answers.)
// Round 6, `mode`: a de Bruijn edge u -> v that leaves its chunk is asked for from BOTH ends — u's successor v, and v^1's successor u^1 (the same (k+1)-mer read
// the other way) — and each answer costs ~5 lines of 128 B (record, partition word, group word, found record, the entry's line): the kernel runs at the fabric's
// request rate (tools/ubench_random_access.hip: 48 G lines/s whatever the bytes used). So the two ends share ONE lookup: mode 1 answers only for the end that reads
// the (k+1)-mer in its smaller orientation (its k-mer <= the reverse complement of the successor, as integers — the other end sees the two swapped; equal: a
// palindrome, both ends are one node) and also writes the entry of v^1 where that one waits for exactly this answer (one outgoing extension, successor pending: v has no
// other predecessor, so it IS u^1); mode 2 then visits the listed nodes again and looks up what is still pending (the other end was a junction and never asked):
// N/2 x 6 + N/2 x 1 lines instead of N x 5 on paper; measured 41.5-41.8 against 43.8-44.3 ms once the lookups are worked off a queue (below; without it the first pass ran
// half-empty waves and the second scanned full lists: no gain). mode 0: every listed node, no mirror (the whole table again after an early clipper; option pm_remote_mirror = 0).
constexpr int PMR_CH = BLK / 64;
// One lookup (lane `lane` of a full batch): the successor k-mer of `node` through the partition table, its entry written; mode 1: the other end's entry too.
template <int NW>
__device__ __forceinline__ void pm_remote_one(const PmIndex &ix, const Rec<NW> *__restrict__ recs, node_t node, unsigned k, node_t *tab, uint32_t *err, unsigned mode) {
    const Rec<NW> raw = recs[node >> 1];
    const unsigned m = pm_byte<NW>(ix, raw, node >> 1), o = (unsigned)(node & 1);
    const unsigned mo = (o ? brev8(m) : m) & 15u;
    unsigned yo;
    const Rec<NW> y = pm_succ_kmer<NW>(rec_pure_xs<NW>(raw, ix.xs), k, o, mo, yo);
    const node_t ry = pm_find<NW>(ix, y);
    node_t e = (node_t)mo << TAB_OUT_SHIFT;
    if (ry == NODE_NONE) atomicAdd(err, 1u);
    else e |= (ry << 1) | yo;
    st_pol<4>(tab + node, e);
    if (mode == 1 && ry != NODE_NONE) {
        const node_t mn = ((ry << 1) | yo) ^ 1;  // v^1
        if (mn != node) {
            const node_t em = tab[mn];
            if (uniq4(tab_out4(em)) && (em & TAB_NODE_MASK) == TAB_NODE_MASK) tab[mn] = (em & ~TAB_NODE_MASK) | (node ^ 1);
        }
    }
}
// Round 6: the marked nodes of a chunk's window are ~46 (half of them in mode 1): handled window by window they left a quarter to two thirds of a wave's lanes
// idle through five dependent memory round trips. The wave now QUEUES the nodes that need a lookup (mode 0: all; mode 1: the owning ends, their bit cleared in
// rbits so that mode 2 lists the others only; mode 2: the listed nodes whose entry is still pending) and works the queue off 64 at a time, across windows and chunks.
template <int NW>
__global__ void __launch_bounds__(BLK) k_pm_remote(PmIndex ix, const unsigned long long *__restrict__ cinfo, uint32_t nchunks, uint32_t wpc /* words of rbits per chunk */,
                                                   uint32_t *rbits, unsigned k, node_t *tab, uint32_t *err, unsigned mode) {
    __shared__ uint16_t lst[PMR_CH][64 * 32];  // per wave: the marked local nodes of 64 words of its chunk
    __shared__ unsigned long long qs[PMR_CH][128];  // per wave: nodes waiting for their lookup
    __shared__ uint32_t wbs[PMR_CH][64];            // per wave (mode 1): the window's words without the owning ends
    const Rec<NW> *recs = (const Rec<NW> *)ix.recs;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint16_t *my = lst[wave];
    unsigned long long *q = qs[wave];
    uint32_t *wb = wbs[wave];
    uint32_t qn = 0;  // (the same in every lane)
    for (uint32_t cid = blockIdx.x * PMR_CH + wave; cid < nchunks; cid += gridDim.x * PMR_CH) {  // (a wave per chunk: no workgroup barrier in here)
        const uint64_t base = cinfo[cid] & PM_BASE_MASK;
        for (uint32_t w0 = 0; w0 < wpc; w0 += 64) {
            const uint32_t w = w0 + lane;
            const uint32_t bits = w < wpc ? rbits[(size_t)cid * wpc + w] : 0u;
            const uint32_t cnt = __popc(bits);
            uint32_t inc = cnt;  // inclusive scan over the wave
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(inc, d, 64);
                if ((int)lane >= d) inc += o;
            }
            const uint32_t n = __shfl(inc, 63, 64);
            uint32_t at = inc - cnt;
            for (uint32_t b = bits; b; b &= b - 1) my[at++] = (uint16_t)(w * 32u + (uint32_t)__ffs(b) - 1u);
            if (mode == 1) wb[lane] = bits;
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (uint32_t i0 = 0; i0 < n; i0 += 64) {
                const uint32_t i = i0 + lane;
                const uint32_t id = i < n ? my[i] : 0u;
                const node_t node = 2 * base + id;
                bool want = i < n;
                if (want && mode == 2) want = (tab[node] & TAB_NODE_MASK) == TAB_NODE_MASK;  // (else: answered from the other end)
                if (want && mode == 1) {
                    // the end that reads the (k+1)-mer in its smaller orientation answers for both (see above)
                    const Rec<NW> raw = recs[node >> 1];
                    const unsigned m = pm_byte<NW>(ix, raw, node >> 1), o = (unsigned)(node & 1);
                    const unsigned c = __ffs((o ? brev8(m) : m) & 15u) - 1;
                    const Rec<NW> x0 = rec_pure_xs<NW>(raw, ix.xs), x1 = rec_rc<NW>(x0, k);
                    want = rc_ge<NW>(rec_shr<NW>(o ? x0 : x1, k, 3u - c), o ? x1 : x0);
                    if (want) atomicAnd(&wb[(id >> 5) - w0], ~(1u << (id & 31u)));
                }
                const unsigned long long wm = __ballot(want);
                if (want) q[qn + __popcll(wm & ((1ull << lane) - 1ull))] = node;
                qn += (uint32_t)__popcll(wm);
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (qn >= 64) {  // a full wave of lookups
                    pm_remote_one<NW>(ix, recs, q[lane], k, tab, err, mode);
                    const unsigned long long rest = lane + 64 < qn ? q[lane + 64] : 0ull;
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane + 64 < qn) q[lane] = rest;
                    qn -= 64;
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
            if (mode == 1 && w < wpc && wb[lane] != bits) rbits[(size_t)cid * wpc + w] = wb[lane];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (lane < qn) pm_remote_one<NW>(ix, recs, q[lane], k, tab, err, mode);  // what is left in the queue
}
// ... and of the dirty region: every successor through the partition table; no jumps (delta 0, 0 steps)
template <int NW>
__global__ void __launch_bounds__(BLK) k_pm_tab_dirty(PmIndex ix, uint64_t nd, unsigned k, node_t *tab, uint32_t *jmp, unsigned long long *stats, uint32_t *err) {
    const Rec<NW> *recs = (const Rec<NW> *)ix.recs;
    unsigned long long bits = 0, pals = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < nd; i += (uint64_t)gridDim.x * BLK) {
        const uint64_t r = ix.nclean + i;
        const Rec<NW> raw = recs[r];
        const Rec<NW> x = rec_pure_xs<NW>(raw, ix.xs);
        const unsigned m = pm_byte<NW>(ix, raw, r);
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
                const node_t ry = pm_find_from_tail<NW>(ix, y);
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
// The node table after an early clipper has edited some masks — incrementally (round 6). A clipper only ever touches k-mers that END chains: the junction k-mer a
// tip or an A/T edge hangs on, the dead end or junction at its other side, and the k-mers of removed tips (isolated: mask 0, no walk reaches them any more). Every jump
// word therefore still describes a chain of unchanged non-junction k-mers, and only the entries of the EDITED k-mers are made again: the extension bits from the new
// mask; the successor kept where the orientation had one extension before (it still has that one, or none), looked up through the partition table where it had several
// and has one now. mask_was: the masks as of the last time the table was right; brought up to date here. (The full pass — k_pm_tab with the local links + k_pm_remote
// over all chunks, 80 ms at config 3 — is what this replaces; option pm_full_retab = 1 still takes it.)
// (The few orientations that need a lookup — 4 % of the edited k-mers: the roots of removed tips — are listed for a dense second kernel, k_pm_retab_lookups: left in
// this one they were two or three working lanes per wave, each with the route's four dependent reads in front of it. A full list falls back to the lookup in place.)
template <int NW>
__device__ __forceinline__ node_t pm_retab_lookup(const PmIndex &ix, uint64_t r, unsigned o, unsigned on, unsigned k, uint32_t *err) {
    const Rec<NW> *recs = (const Rec<NW> *)ix.recs;
    unsigned yo;
    const Rec<NW> x = rec_pure_xs<NW>(recs[r], ix.xs);
    const Rec<NW> y = pm_succ_kmer<NW>(x, k, o, on, yo);
    const node_t ry = r < ix.nclean ? pm_find<NW>(ix, y) : pm_find_from_tail<NW>(ix, y);
    if (ry == NODE_NONE) {
        atomicAdd(err, 1u);
        return 0;
    }
    return (ry << 1) | yo;
}
constexpr unsigned PM_RL_SUB = 256;  // sub-lists, each with a counter of its own (64 B apart): ONE address takes ~88 atomics per microsecond
template <int NW>
__global__ void __launch_bounds__(BLK) k_pm_retab_lookups(PmIndex ix, const unsigned long long *__restrict__ list, const unsigned long long *__restrict__ nlist, uint64_t subcap,
                                                          unsigned k, node_t *tab, uint32_t *err) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < (uint64_t)PM_RL_SUB * subcap; i += (uint64_t)gridDim.x * BLK) {
        const uint64_t sub = i / subcap, j = i % subcap;
        if (j >= nlist[sub * 8]) continue;
        const node_t node = list[i];
        const node_t e = tab[node];  // (the extension bits are in place; the successor is what is missing)
        tab[node] = (e & ~TAB_NODE_MASK) | pm_retab_lookup<NW>(ix, node >> 1, (unsigned)(node & 1), tab_out4(e), k, err);
    }
}
template <int NW>
__global__ void __launch_bounds__(BLK) k_pm_retab_changed(PmIndex ix, uint8_t *mask_was, uint64_t D0, unsigned k, node_t *tab, uint32_t *err,
                                                          unsigned long long *list, unsigned long long *nlist /* [PM_RL_SUB * 8]: entries asked for, per sub-list */, uint64_t subcap) {
    for (uint64_t r = (uint64_t)blockIdx.x * BLK + threadIdx.x; r < D0; r += (uint64_t)gridDim.x * BLK) {
        const unsigned mn = ix.mask[r], mw = mask_was[r];
        if (mn == mw) continue;
        mask_was[r] = (uint8_t)mn;
        // isolated (most edited k-mers are: the k-mers of removed tips, 2.5 G of 4.3 G at bench scale): nothing leads to such a k-mer any more — the junction lost
        // the branch bit, the rest of the tip is isolated with it — so its two entries are never read again and are LEFT as they were (writing 16 B of zeros for each
        // was 40 GB of the 36 ms this pass took)
        if (mn == 0) continue;
#pragma unroll
        for (unsigned o = 0; o < 2; ++o) {
            const unsigned on = (o ? brev8(mn) : mn) & 15u, ow = (o ? brev8(mw) : mw) & 15u;
            node_t e = (node_t)on << TAB_OUT_SHIFT;
            if (uniq4(on)) {
                if (uniq4(ow)) {
                    e |= tab[2 * r + o] & TAB_NODE_MASK;
                } else {
                    const uint64_t sub = blockIdx.x & (PM_RL_SUB - 1);
                    const unsigned long long at = atomicAdd(&nlist[sub * 8], 1ull);
                    if (at < subcap) list[sub * subcap + at] = 2 * r + o;
                    else e |= pm_retab_lookup<NW>(ix, r, o, on, k, err);  // (that sub-list is full: in place)
                }
            }
            tab[2 * r + o] = e;
        }
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

// EXT records of a graph whose masks an early clipper has edited: the byte in the record follows the mask array again (before the records go
// through the sort that makes the k-mer file: the file's masks are split off the records)
template <int NW>
__global__ void k_pm_sync_bytes(void *recs_, const uint8_t *__restrict__ mask, uint64_t n) {
    Rec<NW> *recs = (Rec<NW> *)recs_;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t w = recs[i].w[NW - 1];
        recs[i].w[NW - 1] = (w & ~0xFFull) | (uint64_t)mask[i];
    }
}

// nx graph -> sorted k-mer file (pm_materialize_file): the plain records were sorted into the file; every k-mer of the old (partition-major) array
// takes its byte to its place there
template <int NW>
__global__ void __launch_bounds__(BLK) k_nx_file_masks(const void *old_, const uint8_t *old_mask, uint64_t n, const void *file_, RankDir dir, uint8_t *mask, uint32_t *err) {
    const Rec<NW> *old = (const Rec<NW> *)old_, *file = (const Rec<NW> *)file_;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLK) {
        const node_t r = kmer_rank<NW, false>(file, dir, old[i]);
        if (r == NODE_NONE) atomicAdd(err, 1u);
        else mask[r] = old_mask[i];
    }
}

// ---- the reference's order of the start de-edges ------------------------------------------------------------------------------
// Junction k-mers (EXT records, byte included) compacted in node order; tiles as k_cand_tiles (which also counts them per tile).
template <int NW>
__global__ void __launch_bounds__(BLK) k_pm_junc_write(const uint8_t *mask, const void *recs_, const unsigned long long *tjoff, uint64_t D0, void *out_,
                                                       unsigned long long *jrank_of /* [junctions] node-order rank of every junction k-mer */) {
    __shared__ uint32_t scratch[BLK / 64 + 2];
    const Rec<NW> *recs = (const Rec<NW> *)recs_;
    Rec<NW> *out = (Rec<NW> *)out_;
    const uint64_t r0 = (uint64_t)blockIdx.x * CAND_TILE + (uint64_t)threadIdx.x * CAND_PER;
    uint32_t c = 0, fl = 0;
    uint64_t mm = 0;  // (the thread's 8 bytes as one load where all of them exist)
    if (r0 + CAND_PER <= D0) mm = *reinterpret_cast<const uint64_t *>(mask + r0);
    else
        for (int j = 0; j < CAND_PER; ++j)
            if (r0 + j < D0) mm |= (uint64_t)mask[r0 + j] << (8 * j);
    for (int j = 0; j < CAND_PER; ++j) {
        const unsigned m = (unsigned)(mm >> (8 * j)) & 0xFFu;
        if (r0 + j < D0 && m && mask_junction(m)) {  // (a k-mer with mask 0 — isolated by an early clipper — starts nothing: not listed, as k_cand_tiles counts)
            fl |= 1u << j;
            ++c;
        }
    }
    uint32_t tot;
    unsigned long long o = tjoff[blockIdx.x] + block_excl_scan<uint32_t>(c, scratch, &tot);
    for (int j = 0; j < CAND_PER; ++j)
        if (fl & (1u << j)) {
            jrank_of[o] = r0 + j;
            out[o++] = recs[r0 + j];
        }
}
// start de-edges of the junction k-mers in node order (the byte of the EXT record is the mask)
template <int NW>
__global__ void k_pm_cand_counts_node(const void *jn_, uint64_t nj, unsigned long long *cnt, const uint8_t *mask /* nx: the graph's mask array, else nullptr */,
                                      const unsigned long long *jrank_of) {
    const Rec<NW> *jn = (const Rec<NW> *)jn_;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nj; i += (uint64_t)gridDim.x * blockDim.x)
        cnt[i] = cand_of_mask(mask ? (unsigned)mask[jrank_of[i]] : (unsigned)(jn[i].w[NW - 1] & 0xFFu));
}
// start de-edges of every sorted junction k-mer (the reference's enumeration: its position in the file, then the de-edge)
__global__ void k_pm_cand_counts(const uint8_t *jm, uint64_t nj, unsigned long long *cnt) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nj; i += (uint64_t)gridDim.x * blockDim.x) cnt[i] = cand_of_mask(jm[i]);
}
// first start de-edge of every junction k-mer in the reference's order: its place in the sorted junction file (rank lookup) -> candoff.
// Dense over the junction k-mers in node order (jn: what k_pm_junc_write left), every lane does a lookup.
template <int NW>
__global__ void __launch_bounds__(BLK) k_pm_jrank(const void *jn_, uint64_t nj, const void *jk_, RankDir jix, const unsigned long long *candoff, unsigned long long *qbase,
                                                  uint32_t *err) {
    const Rec<NW> *jn = (const Rec<NW> *)jn_;
    const Rec<NW> *jk = (const Rec<NW> *)jk_;
    for (uint64_t j = (uint64_t)blockIdx.x * BLK + threadIdx.x; j < nj; j += (uint64_t)gridDim.x * BLK) {
        const node_t jr = kmer_rank<NW, false>(jk, jix, rec_pure<NW>(jn[j]));
        if (jr == NODE_NONE) atomicAdd(err, 1u);
        qbase[j] = jr == NODE_NONE ? 0ull : candoff[jr];
    }
}
// The same for an nx graph (plain k-mer records; the bytes are in the graph's mask array), in two steps, because the sorted junction file carries no
// bytes from which the de-edges of the reference's order could be counted first: (1) every junction k-mer (node order) finds its place jr in the
// sorted file, leaves it in qbase and puts ITS byte there (jm[jr]); then the caller counts and scans jm into candoff; (2) qbase[j] = candoff[jr].
template <int NW>
__global__ void __launch_bounds__(BLK) k_pm_jrank_nx1(const void *jn_, uint64_t nj, const void *jk_, RankDir jix, const uint8_t *mask, const unsigned long long *jrank_of,
                                                      unsigned long long *qbase, uint8_t *jm, uint32_t *err, unsigned xs /* 0: jn holds plain k-mers; EXT_BITS: EXT records */) {
    const Rec<NW> *jn = (const Rec<NW> *)jn_;
    const Rec<NW> *jk = (const Rec<NW> *)jk_;
    for (uint64_t j = (uint64_t)blockIdx.x * BLK + threadIdx.x; j < nj; j += (uint64_t)gridDim.x * BLK) {
        const node_t jr = kmer_rank<NW, false>(jk, jix, rec_pure_xs<NW>(jn[j], xs));
        if (jr == NODE_NONE) atomicAdd(err, 1u);
        else jm[jr] = mask[jrank_of[j]];
        qbase[j] = jr;
    }
}
__global__ void k_pm_jrank_nx2(uint64_t nj, const unsigned long long *candoff, unsigned long long *qbase) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nj; j += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long jr = qbase[j];
        qbase[j] = jr == NODE_NONE ? 0ull : candoff[jr];
    }
}
// The start de-edges in node order (what k_cand_expand lists from the masks) with q[o] = their number in the reference's order: dense
// over the junction k-mers — coff: first de-edge of the junction in node order, qbase: in the reference's order; then out bits of the
// k-mer, out bits of its reverse complement (AddStartDeEdges, debruijn_graph_constructor.hpp:203-226).
template <int NW>
__global__ void __launch_bounds__(BLK) k_pm_cand_expand(const void *jn_, const unsigned long long *jrank_of, uint64_t nj, const unsigned long long *coff,
                                                        const unsigned long long *qbase, unsigned long long *cand, unsigned long long *q,
                                                        const uint8_t *mask /* nx: the graph's mask array, else nullptr */) {
    const Rec<NW> *jn = (const Rec<NW> *)jn_;
    for (uint64_t j = (uint64_t)blockIdx.x * BLK + threadIdx.x; j < nj; j += (uint64_t)gridDim.x * BLK) {
        const unsigned long long r = jrank_of[j];
        const unsigned m = mask ? (unsigned)mask[r] : (unsigned)(jn[j].w[NW - 1] & 0xFFu);
        unsigned long long o = coff[j], qq = qbase[j];
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

// cob[b] = chunk that holds record 256 * b (one entry per 256 records): from a record index to its chunk without a search
__global__ void k_pm_cob(const unsigned long long *__restrict__ cinfo, uint32_t nchunks, uint32_t *cob) {
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += gridDim.x * blockDim.x) {
        const unsigned long long ci = cinfo[c];
        const uint64_t base = ci & PM_BASE_MASK, n = ci >> PM_BASE_BITS;
        for (uint64_t b = (base + 255) >> 8; (b << 8) < base + n; ++b) cob[b] = c;
    }
}
// ---- walks ----------------------------------------------------------------------------------------------------------------------
// k_walk_len on the partition-major numbering: the first node by pm_find, chunks crossed by their jump words
template <int NW>
__global__ void __launch_bounds__(BLK) k_pm_walk_len(const unsigned long long *cand, uint64_t C, PmIndex ix, const unsigned long long *cinfo, const uint32_t *cob,
                                                     uint32_t nchunks, const node_t *tab, const uint32_t *jmp, unsigned k, uint64_t n_nodes, unsigned long long *len,
                                                     node_t *first, node_t *last, uint8_t *flags, uint32_t *err) {
    const Rec<NW> *recs = (const Rec<NW> *)ix.recs;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < C; i += (uint64_t)gridDim.x * BLK) {
        const unsigned long long cd = ld_pol<3>(cand + i);
        const Rec<NW> jraw = recs[cd >> 3];
        const Rec<NW> jk = rec_pure_xs<NW>(jraw, ix.xs);
        const Rec<NW> x = ((cd >> 2) & 1) ? rec_rc<NW>(jk, k) : jk;
        unsigned yo = 0;
        const Rec<NW> y = rec_canon<NW>(rec_shl<NW>(x, k, (unsigned)(cd & 3)), k, yo);
        // the first k-mer of the path: in the junction's own chunk 19 times in 20 (no minimizer scan, group word and record next to
        // what the neighbouring lanes read), else through the partition table
        node_t ry = NODE_NONE;
        const uint64_t rj = cd >> 3;
        Rec<NW> yraw;
        bool have_raw = false;
        // ... unless this orientation of the junction k-mer has ONE outgoing extension (a join, a tip's start: a third of the start de-edges): its node entry
        // holds that successor like any other node's (one line, shared with the entry of the other orientation, instead of a group word and a record)
        node_t from_tab = NODE_NONE;
        if (ix.xs && !ix.bym) {
            const unsigned mj = (unsigned)(jraw.w[NW - 1] & 0xFFu);
            if (uniq4((((cd >> 2) & 1) ? brev8(mj) : mj) & 15u)) {
                const node_t e = tab[cd >> 2] & TAB_NODE_MASK;
                if (e != TAB_NODE_MASK && e < n_nodes) from_tab = e;
            }
        }
        if (from_tab != NODE_NONE) {
            ry = from_tab >> 1;
            yo = (unsigned)(from_tab & 1);
        } else if (rj < ix.nclean) {
            uint64_t cbase;
            const uint32_t cid = pm_chunk_of(cinfo, cob, nchunks, rj, cbase);
            ry = pm_probe<NW>(recs, ix.meta + (size_t)cid * ix.ngroups, ix.T, cbase, y, rec_hash32<NW>(y), ix.xs, &yraw);
            have_raw = ry != NODE_NONE;
        }
        if (ry == NODE_NONE) ry = rj < ix.nclean ? pm_find<NW>(ix, y) : pm_find_from_tail<NW>(ix, y);
        if (ry == NODE_NONE) {
            atomicAdd(err, 1u);
            len[i] = 0;
            first[i] = last[i] = NODE_NONE;
            flags[i] = 0;
            continue;
        }
        node_t node = (ry << 1) | yo;
        const node_t first_node = node;
        uint64_t steps = 0;
        // The first k-mer is a junction itself (an edge between two junctions: a third of the start de-edges where every few bases of the genome carry a
        // branch): the byte of the record the probe just compared says so (EXT records, no clipper since), the path ends here and its last k-mer is at
        // hand — no jump word, no node entry, no second record read (3 lines of 128 B instead of 6).
        if (have_raw && ix.xs && !ix.bym && mask_junction((unsigned)(yraw.w[NW - 1] & 0xFFu))) {
            st_pol<3>(last + i, node);
            st_pol<3>(len + i, (unsigned long long)(k + 1));
            st_pol<3>(first + i, node);
            const Rec<NW> yk = rec_pure_xs<NW>(yraw, ix.xs);
            const int cmp = rec_lex_cmp<NW>(x, ((node ^ 1) & 1) ? rec_rc<NW>(yk, k) : yk);
            st_pol<3>(flags + i, (uint8_t)(cmp > 0 ? 1 : (cmp == 0 ? 4 : 0)));
            continue;
        }
        for (;;) {
            const uint32_t j = jmp[node];  // to the end of the chain inside this chunk: non-junction k-mers all the way, the last one may be a junction
            node = (node_t)((long long)node + (long long)(int16_t)(j & 0xFFFFu));
            steps += pm_jump_steps(j);
            node_t nx;
            unsigned nuc;
            if (node >= n_nodes) break;
            if ((j >> 31) && !ix.bym) break;  // the word says so: the chain's last node is a junction — the path ends there, its node entry is not read
            if (!tab_step(tab, node, nx, nuc)) break;  // a junction ends the path
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
        st_pol<3>(last + i, node);
        st_pol<3>(len + i, (unsigned long long)(node == NODE_NONE ? 0 : k + 1 + steps));
        st_pol<3>(first + i, node == NODE_NONE ? NODE_NONE : first_node);
        // keep iff !(s < RC(s)): the start k-mer (in registers) against the reverse complement of the last one decides unless they are
        // equal (a hairpin: k_pm_keep walks it); flags: bit 0 keep, bit 2 undecided
        uint8_t fl = 0;
        if (node != NODE_NONE) {
            const int cmp = rec_lex_cmp<NW>(x, pm_node_kmer<NW>(recs, node ^ 1, k, ix.xs));
            fl = cmp > 0 ? 1 : (cmp == 0 ? 4 : 0);
        }
        st_pol<3>(flags + i, fl);
    }
}
// The rest of k_keep: the hairpins k_pm_walk_len left undecided (flag bit 2) are compared nucleotide by nucleotide; then, for every
// de-edge, vq[q[i]] = words << 1 | keep in the reference's order and the count of non-junction k-mers on kept paths.
template <int NW>
__global__ void __launch_bounds__(BLK) k_pm_keep(const unsigned long long *cand, const unsigned long long *q, uint64_t C, const void *recs_, const node_t *succ,
                                                 unsigned k, const unsigned long long *len, const node_t *first, uint8_t *flags, unsigned long long *vq,
                                                 unsigned long long *interior, unsigned xs,
                                                 unsigned pack_shift /* 0: vq = words << 1 | keep (k_pm_unpack + two scans follow); else vq = words | keep << pack_shift: ONE scan
                                                                        gives a kept path its word offset and its edge index in one word (one line per path in k_pm_walk_write) */) {
    __shared__ unsigned long long scratch[BLK / 64 + 2];
    const Rec<NW> *recs = (const Rec<NW> *)recs_;
    unsigned long long inner = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < C; i += (uint64_t)gridDim.x * BLK) {
        const unsigned long long n = len[i];
        uint8_t fl = flags[i];
        if (fl & 4) {  // the path leads from the start node A to A^1: RC(s) is itself a path leaving A (see k_keep)
            const unsigned long long cd = cand[i];
            const node_t A = cd >> 2;
            const unsigned long long m = n - k;
            node_t a = first[i], prev = A;
            for (unsigned long long t = 1; t < m; ++t) {
                prev = a;
                a = succ[a] & TAB_NODE_MASK;
            }
            const unsigned c2 = 3u - rec_nucl<NW>(pm_node_kmer<NW>(recs, prev, k, xs), 0);
            const unsigned c1 = (unsigned)(cd & 3);
            int cmp = c1 < c2 ? -1 : (c1 > c2 ? 1 : 0);
            a = first[i];
            node_t b = prev ^ 1;
            for (unsigned long long t = 1; t < m && cmp == 0; ++t) {
                const node_t ea = succ[a], eb = succ[b];
                const unsigned na = __ffs(tab_out4(ea)) - 1, nb = __ffs(tab_out4(eb)) - 1;
                cmp = na < nb ? -1 : (na > nb ? 1 : 0);
                a = ea & TAB_NODE_MASK;
                b = eb & TAB_NODE_MASK;
            }
            fl = (uint8_t)((cmp >= 0 ? 1 : 0) | (cmp == 0 ? 2 : 0));
            flags[i] = fl;
        }
        const bool keep = fl & 1;
        // (vq was cleared: only the kept half of the de-edges touches its line — a scattered 8-byte store is a line of 128 B read and written)
        if (keep) vq[q[i]] = pack_shift ? (((n + 31) / 32) | (1ull << pack_shift)) : ((((n + 31) / 32) << 1) | 1ull);
        if (keep) inner += (fl & 2) ? (n - k - 1) / 2 : (n - k - 1);  // a self-conjugate path meets every rank twice
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
// 2-bit stream written word by word
struct PmBitOut {
    uint64_t *dst;
    uint64_t cur, pos;  // bits of word (pos >> 6) so far; position in bits
};
__device__ __forceinline__ void pm_put(PmBitOut &b, uint64_t v, unsigned nbits) {  // append the low nbits (1..64) of v
    if (nbits < 64) v &= (1ull << nbits) - 1;
    const unsigned off = (unsigned)(b.pos & 63);
    b.cur |= v << off;
    if (off + nbits >= 64) {
        st_pol<5>(b.dst + (b.pos >> 6), b.cur);
        b.cur = off ? (v >> (64 - off)) : 0ull;
    }
    b.pos += nbits;
}
template <int NW>
__device__ __forceinline__ Rec<NW> rec_shr_bits(const Rec<NW> &x, unsigned b) {  // x >> b over all words, b < 64 * NW
    const unsigned sw = b >> 6, sb = b & 63u;
    Rec<NW> r;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        uint64_t lo = 0, hi = 0;
#pragma unroll
        for (int t = 0; t < NW; ++t) {
            if ((unsigned)t == i + sw) lo = x.w[t];
            if ((unsigned)t == i + sw + 1) hi = x.w[t];
        }
        r.w[i] = sb ? ((lo >> sb) | (hi << (64 - sb))) : lo;
    }
    return r;
}
// k_walk_write: the kept paths, walked in node order, written at their place in the reference's order (woffq / eidxq are indexed by
// q). The path is followed like k_pm_walk_len followed it — one jump word per chunk — and the nucleotides of the s <= k steps of a
// jump are the last s bases of the k-mer at its far end (every step appends one base): one record read per chunk instead of one
// node-table read per k-mer.
template <int NW>
__global__ void __launch_bounds__(BLK) k_pm_walk_write(const unsigned long long *cand, const unsigned long long *q, uint64_t C, const void *recs_, const node_t *succ,
                                                       const uint32_t *jmp, unsigned k, const unsigned long long *len, const node_t *first, const node_t *last,
                                                       const uint8_t *flags, const unsigned long long *woffq, const unsigned long long *eidxq, uint64_t *words,
                                                       ulonglong4 *erec /* [edges]: word offset, length, start node, end node | self << 63 */, unsigned xs,
                                                       unsigned pack_shift /* != 0: woffq[q] = word offset | edge index << pack_shift (k_pm_keep), eidxq unused */) {
    const Rec<NW> *recs = (const Rec<NW> *)recs_;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < C; i += (uint64_t)gridDim.x * BLK) {
        const uint8_t fl = ld_pol<3>(flags + i);
        if (!(fl & 1)) continue;
        const unsigned long long qi = ld_pol<3>(q + i);
        const unsigned long long cd = ld_pol<3>(cand + i), n = ld_pol<3>(len + i);
        unsigned long long e, wo;
        if (pack_shift) {
            const unsigned long long v = woffq[qi];
            wo = v & ((1ull << pack_shift) - 1);
            e = v >> pack_shift;
        } else {
            e = eidxq[qi];
            wo = woffq[qi];
        }
        const unsigned c = (unsigned)(cd & 3);
        const Rec<NW> x = pm_node_kmer<NW>(recs, cd >> 2, k, xs);
        PmBitOut bo;
        bo.dst = words + wo;
#pragma unroll
        for (int w = 0; w < NW - 1; ++w) bo.dst[w] = x.w[w];
        bo.cur = x.w[NW - 1] | ((uint64_t)c << ((k & 31) << 1));
        bo.pos = 2ull * (k + 1);
        if (((k + 1) & 31u) == 0) {  // the start (k+1)-mer fills its last word
            bo.dst[NW - 1] = bo.cur;
            bo.cur = 0;
        }
        node_t node = ld_pol<3>(first + i);
        const node_t last_node = ld_pol<3>(last + i);
        unsigned long long p = k + 1;
        // A path of at most 2k + 1 bases: what follows the start (k+1)-mer are the last n - k - 1 <= k bases of its LAST k-mer, whatever chunks the path crossed —
        // one record instead of a jump word, a far record and a node entry per chunk (where every few bases of the genome carry a branch and an error's bubble is
        // k + 1 edges long, that is nearly every path: 7.3 -> ~5 lines of 128 B per kept path).
        if (n > p && n - p <= k) {
            const unsigned s = (unsigned)(n - p);
            const Rec<NW> t = rec_shr_bits<NW>(pm_node_kmer<NW>(recs, last_node, k, xs), 2 * (k - s));
            unsigned rem = 2 * s;
#pragma unroll
            for (int w = 0; w < NW; ++w)
                if (rem) {
                    const unsigned nb = rem < 64 ? rem : 64;
                    pm_put(bo, t.w[w], nb);
                    rem -= nb;
                }
            p = n;
        }
        while (p < n) {
            const uint32_t j = jmp[node];
            const uint32_t s = pm_jump_steps(j);
            const node_t far = (node_t)((long long)node + (long long)(int16_t)(j & 0xFFFFu));
            if (s) {
                if (s <= k && p + s <= n) {
                    const Rec<NW> t = rec_shr_bits<NW>(pm_node_kmer<NW>(recs, far, k, xs), 2 * (k - s));
                    unsigned rem = 2 * s;
#pragma unroll
                    for (int w = 0; w < NW; ++w)
                        if (rem) {
                            const unsigned nb = rem < 64 ? rem : 64;
                            pm_put(bo, t.w[w], nb);
                            rem -= nb;
                        }
                    p += s;
                    node = far;
                } else {  // a chain longer than a k-mer (or an inconsistent length): step by step
                    for (uint32_t t = 0; t < s && p < n; ++t, ++p) {
                        const node_t en = succ[node];
                        pm_put(bo, (uint64_t)(__ffs(tab_out4(en)) - 1), 2);
                        node = en & TAB_NODE_MASK;
                    }
                }
            }
            if (p >= n) break;
            const node_t en = succ[node];  // the step into the next chunk
            pm_put(bo, (uint64_t)(__ffs(tab_out4(en)) - 1), 2);
            node = en & TAB_NODE_MASK;
            ++p;
        }
        if (bo.pos & 63) st_pol<5>(bo.dst + (bo.pos >> 6), bo.cur);
        {  // one 32-byte record; k_pm_edges spreads it
            smx_ull2 *er = reinterpret_cast<smx_ull2 *>(erec + e);
            st_pol<5>(er, smx_ull2{wo, n});
            st_pol<5>(er + 1, smx_ull2{cd >> 2, last_node | ((unsigned long long)((fl >> 1) & 1) << 63)});
        }
    }
}
__global__ void k_pm_edges(const ulonglong4 *erec, uint64_t ne, unsigned long long *eoffw, unsigned long long *elen, node_t *estart, node_t *eend, uint8_t *eself) {
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += (uint64_t)gridDim.x * blockDim.x) {
        const ulonglong4 r = erec[e];
        eoffw[e] = r.x;
        elen[e] = r.y;
        estart[e] = r.z;
        eend[e] = r.w & ~(1ull << 63);
        eself[e] = (uint8_t)(r.w >> 63);
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
