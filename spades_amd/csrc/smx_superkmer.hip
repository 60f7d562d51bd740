// spades_amd/csrc/smx_superkmer.hip — on-chip pre-deduplication of k-mer instances (included by smx_api.hip).
//
// The reference materialises every k-mer instance before it sorts and uniques them (KMerSortingSplitter buffers,
// kmer_index/kmer_mph/kmer_splitter.hpp:55-179): I records of W bytes. On the GPU that stream is the HBM traffic of
// every MSD level. This stage removes most duplicates BEFORE anything of size I*W exists:
//   1. k_skm_scan   cut the valid windows of the 2-bit read stream into super-k-mers (maximal runs of consecutive
//                   windows sharing their canonical minimizer, m = 16) and counting-sort them by minimizer key into
//                   fixed slots of 2*NW words (<= 2K-m nucleotides + a count byte): ~1.3 bytes per instance;
//   2. k_skm_plan / k_skm_dedupe2 (smx_skm_dedupe.hip)  identical slots are folded, the slots of consecutive minimizer keys are cut
//                   into chunks, and a workgroup expands a chunk of a few thousand instances inside LDS, keeps one canonical copy
//                   of each k-mer (exact hash set keyed by (slot, offset) references) and appends the survivors to a record array.
// All copies of a k-mer (either strand) share their canonical minimizer, hence their key, hence — unless the key is
// cut by the LDS capacity — their workgroup. The stage is only a FILTER: the sort/unique pipeline that follows is
// exact on any multiset, so duplicates that survive a cut cost time, never correctness, and the survivors' order
// (atomics) does not reach the output. set(survivors) == set(canonical k-mers of the selected windows) is the only
// contract (tests: goldens with the stage forced on).
#pragma once
#include "smx_device.hpp"
#include "smx_graph.hip"  // node entries, mask_junction (the dedupe stage writes the node table of its chunks)

namespace smx {

constexpr uint32_t SKM_NKEY_MIN = 1u << 16;           // partitions ("keys") the minimizers are hashed into: 2^16 .. 2^28 by the number of windows, so
constexpr uint32_t SKM_NKEY_MAX = 1u << 28;           // that a partition holds ~one genomic locus whatever the input size (run_prededupe; the floor
                                                      // was 2^24 until round 5: 0.5 GB of tables and 65 536 planning workgroups for any input)
constexpr int SKM_WMAX = 128 - 16 + 1;                // windows per super-k-mer <= K - m + 1
// Minimizer length: long enough that an m-mer is (nearly) unique in a genome — with m = 12 every 12-mer recurs ~6 times
// per strand of a 50 Mbp genome, the few 12-mers that win the minimizer order collect thousands of instances each and
// most keys had to be cut (measured: +21 % surviving duplicates) — but short enough to leave >= 9 windows per super-k-mer.
__host__ __device__ inline unsigned skm_m(unsigned K) { return K >= 25 ? 16u : K - 8u; }
constexpr int SKM_TC = BLK * 8;                       // windows examined per tile, 8 consecutive per thread
constexpr int SKM_TP = SKM_TC - 128;                  // windows a tile emits starts for; the rest is look-ahead (>= WMAX)
constexpr uint32_t SKM_SCAP = 512;                    // most slots staged per dedupe chunk (two per thread in the prefix scan)
constexpr uint32_t SKM_KEYS_PER_ITEM = 256;
constexpr unsigned SKM_DIRTY_BUCKETS = 1024;          // hash buckets of the sort of the survivors of cut partitions
// Round-4 dedupe kernel (smx_skm_dedupe.hip): a thread works through a SEGMENT of up to SEG consecutive instances of one super-k-mer.
// The counting pass of the scan also counts the segments of every partition (upper 24 bits of its 64-bit counter: the chunk plan
// needs them and would otherwise read every slot once more).
#ifndef SMX_SEG_OVERRIDE
#define SMX_SEG_OVERRIDE 0
#endif
template <int NW>
struct SkmSeg {
    static constexpr int value = SMX_SEG_OVERRIDE ? SMX_SEG_OVERRIDE : (NW <= 2 ? 8 : 4);  // (registers: SEG k-mers of NW words in flight per thread)
};
constexpr unsigned SKM_CNT_BITS = 40;  // cnt[key]: slots in the low 40 bits, segments above
constexpr unsigned long long SKM_CNT_MASK = (1ull << SKM_CNT_BITS) - 1;

struct SkmArgs {
    const uint64_t *seq;
    uint64_t nwords;  // words of seq that may be read
    const uint64_t *mask;
    uint64_t g0, G;   // windows starting outside [g0, G) are not part of this run
    uint64_t vG;      // 0: windows outside [g0, G) count as invalid (a run ends at the range). Else the range is "open": validity is the
                      // chunk's own ([0, vG)), a run may reach past G and only its START must lie in [g0, G) — the ranges of one batch
                      // that is scanned piece by piece (no (K+1)-mer across a boundary loses its extension bits)
    unsigned K, m, w;  // w = K - m + 1
    unsigned pshift;   // partition = mixed key >> pshift (32 - log2 of the partition count)
    unsigned long long *cnt;             // [NKEY] super-k-mers per key (phase 0)
    unsigned long long *cursor;          // [NKEY] next free slot of every key, starts at its offset (phase 1)
    uint64_t *slots;
    unsigned long long *prof;            // SMX_DEBUG: ticks per phase
    // phase 0 with staging: the super-k-mers are also written out in scan order (dense, coalesced) together with their partition,
    // so that placing them (k_skm_permute) does not have to scan the reads a second time
    uint64_t *stage_slots;               // [stage_cap * SW] or nullptr
    unsigned long long *stage_part;      // [stage_cap], preset to all ones (= unused entry): partition | rank inside the partition << 32 (what the
                                         // counting atomic returned: placing the entry needs no second atomic, only the partition's offset)
    unsigned long long stage_cap;
    unsigned long long *stage_alloc;     // [2]: next free staging entry (handed out in blocks), overflow flag
};

// canonical m-mer -> order key: a bijection of 32 bits, so equal keys mean equal minimizers and the order is a
// pseudo-random order of the m-mers (low-complexity ones are not favoured).
__device__ __forceinline__ uint32_t skm_key(uint32_t v, unsigned m) {
    const uint32_t M = m >= 16 ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1);
    uint32_t r = (uint32_t)(rev2_64((uint64_t)v) >> (64 - 2 * m)) ^ M;
    uint32_t c = v < r ? v : r;
    c *= 0x2C9277B5u;
    c ^= c >> 15;
    c *= 0x1B873593u;
    c ^= c >> 13;
    return c;
}
// the same from the m-mer and its reverse complement (both masked to 2m bits) — the scan rolls the two along the stream
__device__ __forceinline__ uint32_t skm_key_vr(uint32_t v, uint32_t r) {
    uint32_t c = v < r ? v : r;
    c *= 0x2C9277B5u;
    c ^= c >> 15;
    c *= 0x1B873593u;
    c ^= c >> 13;
    return c;
}
// order key -> partition: the order keys of the chosen minimizers crowd the low end of the key space by construction
// (a minimizer IS the smallest key of its window), so the partition id is an independent mix of the same value.
__device__ __forceinline__ uint32_t skm_part(uint32_t key, unsigned pshift) {
    key *= 0x9E3779B1u;
    key ^= key >> 16;
    key *= 0x85EBCA6Bu;
    return key >> pshift;
}

#ifndef SMX_SCAN_WPE
#define SMX_SCAN_WPE 6  // waves per SIMD the scan is compiled for: it waits on the counting atomics (measured 65.9 / 60.5 / 56.4 / 55.4 ms at 4 / 5 / 6 / 8)
#endif
template <int PHASE, int NW>
__global__ void __launch_bounds__(BLK, SMX_SCAN_WPE) k_skm_scan(SkmArgs a) {
    constexpr int SW = 2 * NW;
    constexpr int NSW = (SKM_TC + SKM_WMAX + 64) / 32 + 2 * 4 + 4;  // staged stream words
    constexpr int NKQ = SKM_TC + SKM_WMAX + 16;
    constexpr int NFW = SKM_TC / 64 + 3;
    __shared__ uint64_t sw[NSW];
    __shared__ __attribute__((aligned(16))) uint32_t keys[NKQ];
    __shared__ uint64_t bmin[NKQ / 8 + 2];  // leftmost minimum of every aligned block of 8 keys: key << 32 | position
    __shared__ uint64_t fw[NFW];  // break flags, bit i <-> window position p0 + i
    __shared__ uint64_t mw[NFW];  // window-valid bits, bit i <-> position 64*mq0 + i
    __shared__ uint32_t slist[SKM_TP];  // starts of the tile: (window position << 12) | minimizer position
    __shared__ uint32_t s_nstart;
    __shared__ unsigned long long s_stbase;  // staging entries of this workgroup: [s_stbase, s_stbase + s_stleft)
    __shared__ uint32_t s_stleft;
    uint8_t *fb = (uint8_t *)fw;
    if (threadIdx.x == 0) {
        s_stbase = 0;
        s_stleft = 0;
    }
    const unsigned w = a.w, K = a.K, m = a.m;
    const uint32_t mmask = m >= 16 ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1);
    const uint64_t ntiles = (a.G - a.g0 + SKM_TP - 1) / SKM_TP;
    const int64_t nbases = (int64_t)(a.nwords * 32);
    unsigned long long pt[5] = {0, 0, 0, 0, 0}, t0 = 0;
#define SKM_T(i)                                 \
    if (a.prof && threadIdx.x == 0) {            \
        unsigned long long t1 = wall_clock64();  \
        pt[i] += t1 - t0;                        \
        t0 = t1;                                 \
    }
    if (a.prof && threadIdx.x == 0) t0 = wall_clock64();
    // stream / mask words of a tile, one per thread, fetched one tile ahead (the loads fly while the previous tile is scanned)
    static_assert(NSW <= BLK && NFW <= BLK, "one staged word per thread");
    const int64_t mwords = (int64_t)(((a.vG ? a.vG : a.G) + 63) >> 6);
    auto fetch = [&](uint64_t tile, uint64_t &rs, uint64_t &rm) {
        const int64_t o = (int64_t)(a.g0 + tile * SKM_TP) - 1;
        const int64_t wq0 = o < 0 ? 0 : (o >> 5), mq0 = o < 0 ? 0 : (o >> 6);
        const int i = threadIdx.x;
        rs = (i < NSW && (uint64_t)(wq0 + i) < a.nwords) ? a.seq[wq0 + i] : 0ull;
        uint64_t v = 0;
        if (i < NFW) {  // valid = mask bit, restricted to [g0, G) (g0 is a multiple of 64; G is cut inside its word) or, open range, to [0, vG)
            v = (mq0 + i < mwords) ? a.mask[mq0 + i] : 0ull;
            const int64_t first = (mq0 + i) << 6;
            const int64_t lo = a.vG ? 0 : (int64_t)a.g0, hi = a.vG ? (int64_t)a.vG : (int64_t)a.G;
            if (first < lo) v = 0;
            if (first + 64 > hi) v = first >= hi ? 0ull : (v & ((1ull << (hi - first)) - 1));
        }
        rm = v;
    };
    uint64_t pf_s = 0, pf_m = 0;
    if (blockIdx.x < ntiles) fetch(blockIdx.x, pf_s, pf_m);
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t p0 = (int64_t)(a.g0 + tile * SKM_TP);
        const int64_t o = p0 - 1;  // origin of the local indices: qi = q - o, pi = p - o
        const int64_t wq0 = o < 0 ? 0 : (o >> 5);
        const int64_t mq0 = o < 0 ? 0 : (o >> 6);
        if (threadIdx.x < NSW) sw[threadIdx.x] = pf_s;
        if (threadIdx.x < NFW) {
            fw[threadIdx.x] = ~0ull;
            mw[threadIdx.x] = pf_m;
        }
        if (threadIdx.x == 0) s_nstart = 0;
        __syncthreads();
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x, pf_s, pf_m);
        // Keys of the m-mers at the local positions 0 .. nq-1, a block of 8 consecutive positions per thread: the m-mer and its reverse
        // complement are taken from the stream once and ROLLED over the block (two shifts per position instead of an extraction and a
        // 64-bit bit reversal). The block's leftmost minimum goes to bmin; the thread keeps the suffix minima of ITS block (8t..8t+7).
        const int nq = SKM_TC + (int)w + 8;
        const int b8 = threadIdx.x * 8;
        uint32_t sk[8], sp[8];  // suffix minima of the keys b8 .. b8+7 (leftmost wins ties)
        for (int blk = threadIdx.x; blk * 8 < nq; blk += BLK) {
            const int q0i = blk * 8;
            const int64_t q0 = o + q0i;
            uint32_t kk[8];
            if (q0 >= 0 && q0 + 8 + (int64_t)m <= nbases) {
                const int wi = (int)((q0 >> 5) - wq0);
                const unsigned sh = (unsigned)(q0 & 31) << 1;
                uint32_t v = (uint32_t)((sw[wi] >> sh) | ((sw[wi + 1] << (63 - sh)) << 1)) & mmask;
                uint32_t r = __brev(v);
                r = ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
                r = (r >> (32 - 2 * m)) ^ mmask;
                const int64_t qn = q0 + (int64_t)m;  // the 8 bases behind the first m-mer
                const int wn = (int)((qn >> 5) - wq0);
                const unsigned shn = (unsigned)(qn & 31) << 1;
                uint32_t nxt = (uint32_t)((sw[wn] >> shn) | ((sw[wn + 1] << (63 - shn)) << 1)) & 0xFFFFu;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    kk[j] = skm_key_vr(v, r);
                    const uint32_t b = nxt & 3u;
                    nxt >>= 2;
                    v = (v >> 2) | (b << (2 * m - 2));
                    r = ((r << 2) | (3u - b)) & mmask;
                }
            } else {  // at either end of the stream: position by position
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int64_t q = q0 + j;
                    uint32_t key = 0xFFFFFFFFu;
                    if (q >= 0 && q + (int64_t)m <= nbases) {
                        const int wi = (int)((q >> 5) - wq0);
                        const unsigned sh = (unsigned)(q & 31) << 1;
                        const uint64_t v = (sw[wi] >> sh) | ((sw[wi + 1] << (63 - sh)) << 1);
                        key = skm_key((uint32_t)v & mmask, m);
                    }
                    kk[j] = key;
                }
            }
            *(uint4 *)&keys[q0i] = make_uint4(kk[0], kk[1], kk[2], kk[3]);
            *(uint4 *)&keys[q0i + 4] = make_uint4(kk[4], kk[5], kk[6], kk[7]);
            uint32_t tk[8], tp[8];
            tk[7] = kk[7];
            tp[7] = (uint32_t)(q0i + 7);
#pragma unroll
            for (int j = 6; j >= 0; --j) {
                const bool le = kk[j] <= tk[j + 1];
                tk[j] = le ? kk[j] : tk[j + 1];
                tp[j] = le ? (uint32_t)(q0i + j) : tp[j + 1];
            }
            bmin[blk] = ((uint64_t)tk[0] << 32) | tp[0];
            if (blk == (int)threadIdx.x) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sk[j] = tk[j];
                    sp[j] = tp[j];
                }
            }
        }
        __syncthreads();
        SKM_T(0)
        // Minimizer position (qi) of the windows pi = b8 .. b8+8: pi = b8 is the predecessor of the thread's 8 windows
        // p0 + b8 .. p0 + b8 + 7. Window pi spans qi in [pi, pi+w-1] = L (b8+j..b8+7) + C (b8+8..b8+w-1, shared by all nine: whole
        // blocks of 8 from bmin, the rest key by key) + R (b8+w..b8+w+j-1); leftmost position wins ties.
        uint32_t m9[9];
        {
            const int cend = b8 + (int)w - 1;
            uint32_t ck = 0xFFFFFFFFu, cp = (uint32_t)(b8 + 8);
            int q4 = b8 + 8;
            for (; q4 + 7 <= cend; q4 += 8) {
                const uint64_t e = bmin[q4 >> 3];
                if ((uint32_t)(e >> 32) < ck) {
                    ck = (uint32_t)(e >> 32);
                    cp = (uint32_t)e;
                }
            }
            for (; q4 <= cend; ++q4) {
                const uint32_t e = keys[q4];
                if (e < ck) {
                    ck = e;
                    cp = (uint32_t)q4;
                }
            }
            uint32_t rk[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) rk[j] = keys[b8 + (int)w + j];
            uint32_t pk = 0xFFFFFFFFu, pp = 0;  // prefix minimum of R (nothing yet: never smaller)
            bool phave = false;
#pragma unroll
            for (int j = 0; j <= 8; ++j) {
                uint32_t bk = ck, bp = cp;
                if (j < 8 && sk[j] <= ck) {
                    bk = sk[j];
                    bp = sp[j];
                }
                if (phave && pk < bk) {
                    bk = pk;
                    bp = pp;
                }
                m9[j] = bp;
                if (j < 8) {
                    if (!phave || rk[j] < pk) {
                        pk = rk[j];
                        pp = (uint32_t)(b8 + (int)w + j);
                        phave = true;
                    }
                }
            }
        }
        // validity of the 9 windows, break flags of the 8 owned ones
        uint32_t vb = 0;
#pragma unroll
        for (int j = 0; j <= 8; ++j) {
            const int64_t p = o + b8 + j;
            bool ok = p >= 0 && ((mw[(p >> 6) - mq0] >> (p & 63)) & 1);
            vb |= (ok ? 1u : 0u) << j;
        }
        uint32_t brk = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bool b = !((vb >> (j + 1)) & 1) || !((vb >> j) & 1) || m9[j + 1] != m9[j];
            brk |= (b ? 1u : 0u) << j;
        }
        fb[b8 >> 3] = (uint8_t)brk;
        __syncthreads();
        SKM_T(1)
        // one super-k-mer per owned start; the last SKM_TC - SKM_TP windows are only the look-ahead of the run lengths.
        // The starts are first gathered into a dense list so that every lane of the emitting loop has one.
        {
            uint32_t starts = b8 < SKM_TP ? ((vb >> 1) & brk & 0xFFu) : 0u;
            if (a.vG && p0 + b8 + 8 > (int64_t)a.G)  // open range: starts at positions >= G belong to the next range
                starts &= p0 + b8 >= (int64_t)a.G ? 0u : ((1u << (unsigned)((int64_t)a.G - p0 - b8)) - 1u);
            if (starts) {
                uint32_t at = atomicAdd(&s_nstart, (uint32_t)__popc(starts));
#pragma unroll
                for (int j = 0; j < 8; ++j)  // bit 31: the window before the start is valid (same read, no N): its first base precedes the run
                    if (starts & (1u << j)) slist[at++] = (((vb >> j) & 1u) << 31) | ((uint32_t)(b8 + j) << 12) | m9[j + 1];
            }
        }
        __syncthreads();
        const uint32_t nstart = s_nstart;
        unsigned long long stage0 = ~0ull;  // first staging entry of this tile (PHASE 0 with staging)
        if (PHASE == 0 && a.stage_slots) {
            if (threadIdx.x == 0 && nstart) {
                if (s_stleft < nstart) {  // next block of staging entries (the unused rest of the old one stays marked unused)
                    constexpr uint32_t CH = 4096;  // >= SKM_TP, the most starts a tile can have
                    const unsigned long long b = atomicAdd(&a.stage_alloc[0], (unsigned long long)CH);
                    if (b + CH > a.stage_cap) {
                        a.stage_alloc[1] = 1;  // overflow: the host falls back to the second scan
                        s_stleft = 0;
                        s_stbase = ~0ull;
                    } else {
                        s_stbase = b;
                        s_stleft = CH;
                    }
                }
                if (s_stbase != ~0ull) {
                    s_stleft -= nstart;
                    s_stbase += nstart;
                }
            }
            __syncthreads();
            stage0 = s_stbase == ~0ull ? ~0ull : s_stbase - nstart;
        }
        for (uint32_t si = threadIdx.x; si < nstart; si += BLK) {
            const uint32_t e = slist[si];
            const int pr = (int)((e >> 12) & 0x7FFFFu);  // window position relative to p0
            uint32_t c;                     // windows until the next break (within w: the minimizer leaves the window)
            {
                const int nb1 = pr + 1;
                const int wi = nb1 >> 6, bit = nb1 & 63;
                uint64_t x = fw[wi] >> bit;
                int acc = 0;
                if (x == 0) {
                    acc = 64 - bit;
                    x = fw[wi + 1];
                    if (x == 0) {
                        acc += 64;
                        x = fw[wi + 2];
                    }
                }
                c = (uint32_t)(acc + (x ? __ffsll((unsigned long long)x) - 1 : 64)) + 1;
            }
            if (c > w) c = w;
            const uint32_t key = skm_part(keys[e & 0xFFFu], a.pshift);
            uint64_t *dst = nullptr;
            unsigned long long rank = 0;  // PHASE 0: what the counting atomic returned (raw: masked where it is used, after the slot words)
            if constexpr (PHASE == 0) {
                // One atomic per super-k-mer — unless the neighbouring lanes hold the same partition: every window of a homopolymer or
                // short-period run is a super-k-mer of its own, millions of them share a handful of minimizers, and ONE address takes
                // ~88 atomics per microsecond (measured: +150 ms at 1 % low-complexity sequence). Up to three peels: the lanes that hold
                // the key of the first remaining lane add up in one atomic and number themselves.
                const uint32_t nsg = (c + SkmSeg<NW>::value - 1) / SkmSeg<NW>::value;  // segments of this super-k-mer (<= 29: 5 bits)
                {
                    unsigned long long todo = __ballot(1);
                    bool mine = true;
                    // (only where the wave shows repeats at all: neighbouring lanes with one key — ordinary sequence never does)
                    const bool rep = __popcll(__ballot(key == (uint32_t)__shfl_down((int)key, 1, 64))) >= 8;
                    for (int peel = 0; rep && peel < 3 && todo; ++peel) {
                        const int lead = __ffsll(todo) - 1;
                        const uint32_t k0 = (uint32_t)__shfl((int)key, lead, 64);
                        const bool hit = mine && key == k0;
                        const unsigned long long same = __ballot(hit) & todo;
                        unsigned long long segs = 0;  // segments of the lanes in `same` (ballots: only the active lanes count)
#pragma unroll
                        for (int b = 0; b < 5; ++b) segs += (unsigned long long)__popcll(__ballot(hit && ((nsg >> b) & 1u)) & same) << b;
                        unsigned long long base = 0;
                        if ((int)(threadIdx.x & 63) == lead) base = atomicAdd(&a.cnt[k0], (unsigned long long)__popcll(same) | (segs << SKM_CNT_BITS));
                        base = __shfl(base, lead, 64);  // (every active lane takes part: no shuffle inside a divergent branch)
                        if (hit) {
                            rank = (base & SKM_CNT_MASK) + __popcll(same & ((1ull << (threadIdx.x & 63)) - 1));
                            mine = false;
                        }
                        todo &= ~same;
                    }
                    // (the plain case: what this atomic returns is needed for ONE thing, the (partition, rank) word of the staged entry — that
                    // store comes AFTER the slot words have been put together below, so the ~150 instructions of that work run while the
                    // atomic is on its way; SQ counters of round 4: the waves of this kernel wait 65 % of their cycles)
                    if (mine) rank = atomicAdd(&a.cnt[key], 1ull | ((unsigned long long)nsg << SKM_CNT_BITS));
                }
                if (stage0 != ~0ull) dst = a.stage_slots + (stage0 + si) * SW;
            } else {
                dst = a.slots + atomicAdd(&a.cursor[key], 1ull) * SW;
            }
            if (dst) {
                const int64_t p = p0 + pr;
                const int wi = (int)((p >> 5) - wq0);
                const unsigned sh = (unsigned)(p & 31) << 1;
                const unsigned nbits = 2 * (c + K - 1);
                // the bases next to the run, where the neighbouring window is valid — i.e. where the (K+1)-mer across the end of the run
                // exists — as two sets of one bit per base: bits 0-3 the base before the run, bits 4-7 the base behind it (slot bits 48..55
                // of the last word; sets, because the dedupe stage folds identical slots and ORs their neighbours together)
                uint32_t nb = 0;
                auto base_at = [&](int64_t q) -> uint32_t {
                    const int bw = (int)((q >> 5) - wq0);
                    return (uint32_t)(sw[bw] >> ((unsigned)(q & 31) << 1)) & 3u;
                };
                if (e >> 31) nb |= 1u << base_at(p - 1);
                {
                    const int64_t q = p + (int64_t)c;  // the window behind the run
                    if ((mw[(q >> 6) - mq0] >> (q & 63)) & 1) nb |= 16u << base_at(q + (int64_t)K - 1);
                }
                // Orientation: the slot is stored on the strand on which its minimizer reads as the canonical m-mer, so that copies of one
                // stretch of the genome from reads of either strand are the SAME slot, bit for bit, and the dedupe stage can fold them before
                // it expands them (smx_skm_dedupe.hip). A slot is only a bag of k-mer instances with their neighbour bases: either strand
                // yields the same canonical k-mers and extension bits. (A minimizer that is its own reverse complement stays as read.)
                bool flip;
                {
                    const int64_t q = o + (int64_t)(e & 0xFFFu);
                    const int mi = (int)((q >> 5) - wq0);
                    const unsigned msh = (unsigned)(q & 31) << 1;
                    const uint32_t mv = (uint32_t)((sw[mi] >> msh) | ((sw[mi + 1] << (63 - msh)) << 1)) & mmask;
                    const uint32_t mr = (uint32_t)(rev2_64((uint64_t)mv) >> (64 - 2 * m)) ^ mmask;
                    flip = mr < mv;
                }
                uint64_t vals[SW];
                if (!flip) {
#pragma unroll
                    for (int i = 0; i < SW; ++i) {
                        uint64_t v = sw[wi + i] >> sh;
                        if (sh) v |= sw[wi + i + 1] << (64 - sh);
                        if (nbits <= 64u * i) v = 0;
                        else if (nbits < 64u * (i + 1)) v &= (1ull << (nbits - 64u * i)) - 1;
                        if (i == SW - 1) v |= ((uint64_t)c << 56) | ((uint64_t)nb << 48);
                        vals[i] = v;
                    }
                } else {
                    // reverse complement of the run: word i holds the complements of the bases pl + len - 32 i - 1 down to pl + len - 32 i - 32
                    // (pl = the run's first base, local to the staged words); what lies before the run is masked out with the tail
                    const int pl = (int)(p - (wq0 << 5)), len = (int)(c + K - 1);
                    nb = (__brev(nb & 15u) >> 24) | (__brev(nb >> 4) >> 28);  // the sets change sides, every base to its complement (bit b -> 3 - b)
#pragma unroll
                    for (int i = 0; i < SW; ++i) {
                        const int s0 = pl + len - 32 * (i + 1);
                        const int wj = s0 >> 5;  // (arithmetic shift: may be negative)
                        const unsigned sj = (unsigned)(s0 & 31) << 1;
                        const uint64_t lo = wj >= 0 ? sw[wj] : 0ull, hi = wj + 1 >= 0 ? sw[wj + 1] : 0ull;
                        uint64_t v = lo >> sj;
                        if (sj) v |= hi << (64 - sj);
                        v = rev2_64(~v);
                        if (nbits <= 64u * i) v = 0;
                        else if (nbits < 64u * (i + 1)) v &= (1ull << (nbits - 64u * i)) - 1;
                        if (i == SW - 1) v |= ((uint64_t)c << 56) | ((uint64_t)nb << 48);
                        vals[i] = v;
                    }
                }
                if constexpr (PHASE == 0) st_pol<8>(a.stage_part + stage0 + si, (unsigned long long)key | ((rank & SKM_CNT_MASK) << 32));  // (dst != nullptr in phase 0 <=> the tile has staging entries)
#pragma unroll
                for (int i = 0; i < SW; ++i) st_pol<8>((uint64_t *)dst + i, (uint64_t)vals[i]);
            }
        }
        __syncthreads();
        SKM_T(2)
        if (a.prof && threadIdx.x == 0) pt[3] += 1;
    }
    if (a.prof && threadIdx.x == 0)
        for (int i = 0; i < 4; ++i) atomicAdd(&a.prof[8 * PHASE + i], pt[i]);
#undef SKM_T
}

// staged super-k-mers (scan order) -> their place in the partition-sorted slot array
template <int NW>
__global__ void __launch_bounds__(BLK) k_skm_permute(const uint64_t *__restrict__ stage_slots, const unsigned long long *__restrict__ stage_part,
                                                     uint64_t n_stage, const unsigned long long *__restrict__ soff, uint64_t *slots) {
    constexpr int SW = 2 * NW;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n_stage; i += (uint64_t)gridDim.x * BLK) {
        const unsigned long long pr = ld_pol<1>(stage_part + i);
        if (pr == ~0ull) continue;
        uint64_t v[SW];
#pragma unroll
        for (int t = 0; t < SW; ++t) v[t] = ld_pol<1>(stage_slots + i * SW + t);
        uint64_t *dst = slots + (soff[(uint32_t)pr] + (pr >> 32)) * SW;  // the place the counting pass reserved
#pragma unroll
        for (int t = 0; t < SW; ++t) st_pol<0>(dst + t, v[t]);
    }
}

// counters of the counting pass -> slots per partition (in place) and segments per partition
__global__ void __launch_bounds__(BLK) k_skm_split(unsigned long long *cnt, uint32_t *kseg, uint32_t n) {
    for (uint32_t i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK) {
        const unsigned long long v = cnt[i];
        kseg[i] = (uint32_t)(v >> SKM_CNT_BITS);
        cnt[i] = v & SKM_CNT_MASK;
    }
}

// sum and sum of squares of the per-partition slot counts: sum2 / sum = the partition size a random super-k-mer lives in
// ... and sums[2..4] = slots in partitions of more than thr0 / thr1 / thr2 slots (the partitions a chunk capacity would cut)
__global__ void __launch_bounds__(BLK) k_skm_moments(const unsigned long long *__restrict__ cnt, uint32_t n, unsigned long long *sums, unsigned long long thr0,
                                                     unsigned long long thr1, unsigned long long thr2) {
    __shared__ unsigned long long scratch[BLK / 64 + 2];
    unsigned long long s1 = 0, s2 = 0, b0 = 0, b1 = 0, b2 = 0;
    for (uint32_t i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK) {
        const unsigned long long c = cnt[i];
        s1 += c;
        s2 += c * c;
        if (c > thr0) b0 += c;
        if (c > thr1) b1 += c;
        if (c > thr2) b2 += c;
    }
    unsigned long long t1, t2, u0, u1, u2;
    block_excl_scan<unsigned long long>(s1, scratch, &t1);
    block_excl_scan<unsigned long long>(s2, scratch, &t2);
    block_excl_scan<unsigned long long>(b0, scratch, &u0);
    block_excl_scan<unsigned long long>(b1, scratch, &u1);
    block_excl_scan<unsigned long long>(b2, scratch, &u2);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[0], t1);
        atomicAdd(&sums[1], t2);
        if (u0) atomicAdd(&sums[2], u0);
        if (u1) atomicAdd(&sums[3], u1);
        if (u2) atomicAdd(&sums[4], u2);
    }
}

// keys (partitions) of more than thr slots: (key, first slot, slots) triples appended to list (count in list[0]; capacity cap triples)
__global__ void __launch_bounds__(BLK) k_skm_bigkeys(const unsigned long long *__restrict__ soff, uint32_t n, unsigned long long thr, unsigned long long *list,
                                                     uint32_t cap) {
    for (uint32_t i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK) {
        const unsigned long long a = soff[i], c = soff[i + 1] - a;
        if (c > thr) {
            const unsigned long long p = atomicAdd(&list[0], 1ull);
            if (p < cap) {
                list[1 + 3 * p] = i;
                list[2 + 3 * p] = a;
                list[3 + 3 * p] = c;
            }
        }
    }
}

// k-mer number j of a staged slot
template <int NW>
__device__ __forceinline__ Rec<NW> skm_extract(const uint64_t *s, unsigned j, unsigned K) {
    Rec<NW> x;
    const unsigned wi = j >> 5, sh = (j & 31) << 1;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        uint64_t v = s[wi + i] >> sh;
        if (sh) v |= s[wi + i + 1] << (64 - sh);
        x.w[i] = v;
    }
    const unsigned tail = (K & 31) << 1;
    if (tail) x.w[NW - 1] &= (1ull << tail) - 1;
    return x;
}

// Partition-major output of the dedupe stage (MODE 2; construction route "pm", smx_pm.hip): the winners of a clean chunk leave in the
// order of their LDS hash-table slots, and the occupancy of the table travels with them — 16 slots per group word (low half: winners
// before the group, high half: occupancy bits) — so that the table can be probed again in HBM: slot h of chunk c holds record
// base(c) + pre(h >> 4) + popc(occ & below(h & 15)). pinfo[partition] = base | chunk << 40 tells where the k-mers of a minimizer
// partition went (PM_DIRTY: the partition was cut, its k-mers are in the sorted tail of the array; PM_EMPTY: it has none).
struct PmOut {
    unsigned long long *pinfo;  // [partitions], preset to PM_EMPTY
    uint32_t *meta;             // [max_chunks * (T / 16)] group words
    unsigned long long *cinfo;  // [max_chunks] base | winners << 40
    uint8_t *mask;              // [out_cap] InOutMask byte of every clean winner, at its record's index
    uint32_t *llink;            // [out_cap] preset to ~0: per winner two 16-bit local links (low half: successor of the k-mer as stored,
                                // high half: of its reverse complement) = 2 * (index in the chunk) + orientation of the successor node where
                                // some read holds the two k-mers side by side inside one super-k-mer; 0xFFFF: none seen (k_pm_remote asks)
    unsigned long long *pals;   // palindromic (k+1)-mers among the extensions of the clean winners (k_ext_split's second figure)
    uint32_t max_chunks;
    uint32_t *overflow;         // set when a chunk got no room in meta / cinfo
    // The node table of the clean chunks made by the dedupe stage itself (round 6; nullptr: k_pm_tab makes it from mask + llink afterwards): the
    // chunk's bytes and local links are in LDS when its records leave, so the entries, the jump words and the remote bits of smx_pm.hip's k_pm_tab
    // are written from there and the links never travel (4 B per k-mer written, cleared and read again otherwise).
    unsigned long long *tab;    // [2 * out_cap + 2] node entries (smx_graph.hip: successor | outgoing extensions << TAB_OUT_SHIFT)
    uint32_t *jmp;              // [2 * out_cap + 2] jump words of the chain heads, 0 elsewhere
    uint32_t *rbits;            // [max_chunks * T / 32] per chunk: bit nd = node nd's successor lies outside the chunk (k_pm_remote looks it up)
    unsigned long long *tab_stats;  // [0] += extension bits of the clean winners
};
constexpr unsigned PM_BASE_BITS = 40;
constexpr unsigned long long PM_BASE_MASK = (1ull << PM_BASE_BITS) - 1;
constexpr unsigned long long PM_EMPTY = ~0ull, PM_DIRTY = ~0ull - 1;

}  // namespace smx

#include "smx_skm_dedupe.hip"
