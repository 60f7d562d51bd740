// spades_amd/csrc/smx_superkmer.hip — on-chip pre-deduplication of k-mer instances (included by smx_api.hip).
//
// The reference materialises every k-mer instance before it sorts and uniques them (KMerSortingSplitter buffers,
// kmer_index/kmer_mph/kmer_splitter.hpp:55-179): I records of W bytes. On the GPU that stream is the HBM traffic of
// every MSD level. This stage removes most duplicates BEFORE anything of size I*W exists:
//   1. k_skm_scan   cut the valid windows of the 2-bit read stream into super-k-mers (maximal runs of consecutive
//                   windows sharing their canonical minimizer, m = 11) and counting-sort them by minimizer key into
//                   fixed slots of 2*NW words (<= 2K-m nucleotides + a count byte): ~1.3 bytes per instance;
//   2. k_skm_dedupe a workgroup expands a few thousand instances of consecutive minimizer keys inside LDS, keeps one
//                   canonical copy of each k-mer (exact hash set keyed by (slot, offset) references) and appends the
//                   survivors to a record array in HBM.
// All copies of a k-mer (either strand) share their canonical minimizer, hence their key, hence — unless the key is
// cut by the LDS capacity — their workgroup. The stage is only a FILTER: the sort/unique pipeline that follows is
// exact on any multiset, so duplicates that survive a cut cost time, never correctness, and the survivors' order
// (atomics) does not reach the output. set(survivors) == set(canonical k-mers of the selected windows) is the only
// contract (tests: goldens with the stage forced on).
#pragma once
#include "smx_device.hpp"

namespace smx {

constexpr uint32_t SKM_NKEY_MIN = 1u << 24;           // partitions ("keys") the minimizers are hashed into: 2^24 .. 2^28, so that a
constexpr uint32_t SKM_NKEY_MAX = 1u << 28;           // partition holds ~one genomic locus whatever the input size (run_prededupe)
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
// order key -> partition: the order keys of the chosen minimizers crowd the low end of the key space by construction
// (a minimizer IS the smallest key of its window), so the partition id is an independent mix of the same value.
__device__ __forceinline__ uint32_t skm_part(uint32_t key, unsigned pshift) {
    key *= 0x9E3779B1u;
    key ^= key >> 16;
    key *= 0x85EBCA6Bu;
    return key >> pshift;
}

template <int PHASE, int NW>
__global__ void __launch_bounds__(BLK) k_skm_scan(SkmArgs a) {
    constexpr int SW = 2 * NW;
    constexpr int NSW = (SKM_TC + SKM_WMAX + 64) / 32 + 2 * 4 + 4;  // staged stream words
    constexpr int NKQ = SKM_TC + SKM_WMAX + 16;
    constexpr int NFW = SKM_TC / 64 + 3;
    __shared__ uint64_t sw[NSW];
    __shared__ __attribute__((aligned(16))) uint32_t keys[NKQ];
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
        const int nq = SKM_TC + (int)w + 8;
        for (int qi = threadIdx.x; qi < nq; qi += BLK) {
            const int64_t q = o + qi;
            uint32_t key = 0xFFFFFFFFu;
            if (q >= 0 && q + (int64_t)m <= nbases) {
                const int wi = (int)((q >> 5) - wq0);
                const unsigned sh = (unsigned)(q & 31) << 1;
                const uint64_t v = (sw[wi] >> sh) | ((sw[wi + 1] << (63 - sh)) << 1);
                key = skm_key((uint32_t)v & mmask, m);
            }
            keys[qi] = key;
        }
        __syncthreads();
        SKM_T(0)
        // Minimizer position (qi) of the windows pi = b8 .. b8+8: pi = b8 is the predecessor of the thread's 8 windows
        // p0 + b8 .. p0 + b8 + 7. Window pi spans qi in [pi, pi+w-1] = L (b8+j..b8+7) + C (b8+8..b8+w-1, shared by all nine)
        // + R (b8+w..b8+w+j-1); leftmost position wins ties.
        const int b8 = threadIdx.x * 8;
        uint32_t m9[9];
        {
            const int cend = b8 + (int)w - 1;
            uint32_t ck = keys[b8 + 8], cp = (uint32_t)(b8 + 8);
            for (int q4 = b8 + 8; q4 <= cend; q4 += 4) {
                const uint4 k4 = *(const uint4 *)&keys[q4];
                const uint32_t e0 = k4.x, e1 = q4 + 1 <= cend ? k4.y : 0xFFFFFFFFu, e2 = q4 + 2 <= cend ? k4.z : 0xFFFFFFFFu,
                               e3 = q4 + 3 <= cend ? k4.w : 0xFFFFFFFFu;
                if (e0 < ck) { ck = e0; cp = (uint32_t)q4; }
                if (e1 < ck) { ck = e1; cp = (uint32_t)q4 + 1; }
                if (e2 < ck) { ck = e2; cp = (uint32_t)q4 + 2; }
                if (e3 < ck) { ck = e3; cp = (uint32_t)q4 + 3; }
            }
            uint32_t lk[8], rk[8];
            {
                const uint4 l0 = *(const uint4 *)&keys[b8], l1 = *(const uint4 *)&keys[b8 + 4];
                lk[0] = l0.x; lk[1] = l0.y; lk[2] = l0.z; lk[3] = l0.w;
                lk[4] = l1.x; lk[5] = l1.y; lk[6] = l1.z; lk[7] = l1.w;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) rk[j] = keys[b8 + (int)w + j];
            uint32_t sk[8], sp[8];  // suffix minima of L
            sk[7] = lk[7];
            sp[7] = (uint32_t)(b8 + 7);
#pragma unroll
            for (int j = 6; j >= 0; --j) {
                const bool t = lk[j] <= sk[j + 1];
                sk[j] = t ? lk[j] : sk[j + 1];
                sp[j] = t ? (uint32_t)(b8 + j) : sp[j + 1];
            }
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
            if constexpr (PHASE == 0) {
                // One atomic per super-k-mer — unless the neighbouring lanes hold the same partition: every window of a homopolymer or
                // short-period run is a super-k-mer of its own, millions of them share a handful of minimizers, and ONE address takes
                // ~88 atomics per microsecond (measured: +150 ms at 1 % low-complexity sequence). Up to three peels: the lanes that hold
                // the key of the first remaining lane add up in one atomic and number themselves.
                unsigned long long rank = 0;
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
                        unsigned long long base = 0;
                        if ((int)(threadIdx.x & 63) == lead) base = atomicAdd(&a.cnt[k0], (unsigned long long)__popcll(same));
                        base = __shfl(base, lead, 64);  // (every active lane takes part: no shuffle inside a divergent branch)
                        if (hit) {
                            rank = base + __popcll(same & ((1ull << (threadIdx.x & 63)) - 1));
                            mine = false;
                        }
                        todo &= ~same;
                    }
                    if (mine) rank = atomicAdd(&a.cnt[key], 1ull);
                }
                if (stage0 != ~0ull) {
                    dst = a.stage_slots + (stage0 + si) * SW;
                    a.stage_part[stage0 + si] = (unsigned long long)key | (rank << 32);
                }
            } else {
                dst = a.slots + atomicAdd(&a.cursor[key], 1ull) * SW;
            }
            if (dst) {
                const int64_t p = p0 + pr;
                const int wi = (int)((p >> 5) - wq0);
                const unsigned sh = (unsigned)(p & 31) << 1;
                const unsigned nbits = 2 * (c + K - 1);
                // the bases next to the run, where the neighbouring window is valid — i.e. where the (K+1)-mer across the end of the run
                // exists: bit 0 left valid, bits 1-2 left base, bit 3 right valid, bits 4-5 right base (slot bits 48..53 of the last word)
                uint32_t nb = 0;
                auto base_at = [&](int64_t q) -> uint32_t {
                    const int bw = (int)((q >> 5) - wq0);
                    return (uint32_t)(sw[bw] >> ((unsigned)(q & 31) << 1)) & 3u;
                };
                if (e >> 31) nb |= 1u | (base_at(p - 1) << 1);
                {
                    const int64_t q = p + (int64_t)c;  // the window behind the run
                    if ((mw[(q >> 6) - mq0] >> (q & 63)) & 1) nb |= 8u | (base_at(q + (int64_t)K - 1) << 4);
                }
#pragma unroll
                for (int i = 0; i < SW; ++i) {
                    uint64_t v = sw[wi + i] >> sh;
                    if (sh) v |= sw[wi + i + 1] << (64 - sh);
                    if (nbits <= 64u * i) v = 0;
                    else if (nbits < 64u * (i + 1)) v &= (1ull << (nbits - 64u * i)) - 1;
                    if (i == SW - 1) v |= ((uint64_t)c << 56) | ((uint64_t)nb << 48);
                    dst[i] = v;
                }
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
        const unsigned long long pr = stage_part[i];
        if (pr == ~0ull) continue;
        uint64_t v[SW];
#pragma unroll
        for (int t = 0; t < SW; ++t) v[t] = stage_slots[i * SW + t];
        uint64_t *dst = slots + (soff[(uint32_t)pr] + (pr >> 32)) * SW;  // the place the counting pass reserved
#pragma unroll
        for (int t = 0; t < SW; ++t) dst[t] = v[t];
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
};
constexpr unsigned PM_BASE_BITS = 40;
constexpr unsigned long long PM_BASE_MASK = (1ull << PM_BASE_BITS) - 1;
constexpr unsigned long long PM_EMPTY = ~0ull, PM_DIRTY = ~0ull - 1;

// One workgroup per item of SKM_KEYS_PER_ITEM consecutive minimizer keys; the item's slots are consumed in chunks of
// whole keys holding <= cap instances (a key larger than that is cut). Per chunk: stage the slots in LDS, expand the
// instance list (slot, offset), insert every instance into an exact hash set whose entries are 16-bit fingerprint |
// 16-bit (slot, offset) reference (the k-mers themselves stay in the staged slots), append the winners to HBM.
// LDS (dynamic): sl[scap*SW] u64 | tab[T] u32 | cpre[514] u32 | imap[cap] u16 | wl[cap] u16 | cl[512] u8 | nbv[512] u8
// EXT: every instance also knows the bases next to it inside its read (the slot, or the slot's neighbour bases at its two ends) —
// the extensions the (K+1)-mers around it give its k-mer (InOutMask bits in the k-mer's canonical frame: out bits 0-3 by next
// base, in bits 4-7 by previous base; kmer_extension_index_builder.hpp:45-60, inout_mask.hpp:92-131). The table entries then are
// 8-bit fingerprint | 8 extension bits (OR of all copies) | reference, and the survivors leave in the EXT layout (smx_device.hpp).
// MODE: 0 plain, 1 EXT, 2 EXT + partition-major output (out_count then packs chunks << 40 | clean records)
template <int NW, int MODE>
__global__ void __launch_bounds__(BLK) k_skm_dedupe(const uint64_t *__restrict__ slots, const unsigned long long *__restrict__ slot_off,
                                                    unsigned K, uint32_t nitems, uint32_t cap, uint32_t T, uint32_t scap, void *out_,
                                                    unsigned long long out_cap, unsigned long long clean_cap, unsigned long long dirty_cap, unsigned long long *out_count,
                                                    unsigned long long *dirty_count, unsigned long long *prof, PmOut pm, unsigned long long skip_slots,
                                                    uint32_t force_dirty, uint32_t item_stride /* entries of slot_off per item: 256 (items share their
                                                    boundary entries) or 257 (rows of their own) */) {
    // skip_slots: a key (partition) of more slots than this is left out here — one workgroup would chew on a homopolymer partition of
    // 10^7 instances for 100 ms while the others idle; its slot range is cut into pieces that a second launch deals out as items of their
    // own (slot_off = one pseudo-key per piece, force_dirty = 1: every chunk of such a piece is a dirty one, its partition is cut by
    // construction, and the partition table is not touched).
    constexpr int SW = 2 * NW;
    constexpr bool EXT = MODE >= 1, PM = MODE == 2;
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    uint64_t *sl = lds64;
    uint32_t *tab = (uint32_t *)(sl + (size_t)scap * SW);
    uint32_t *cpre = tab + T;
    uint16_t *imap = (uint16_t *)(cpre + 514);
    uint16_t *wl = imap + cap;
    uint8_t *cl = (uint8_t *)(wl + cap);
    uint8_t *nbv = cl + 512;
    __shared__ unsigned long long koff[SKM_KEYS_PER_ITEM + 1];
    __shared__ uint32_t scr[BLK / 64 + 2];
    __shared__ uint32_t s_take, s_nfit, s_ninst, s_wcount, s_endb, s_cid;
    __shared__ unsigned long long s_gbase, s_kend;
    __shared__ uint32_t s_skip;  // the output buffer is full: the counters keep counting (the host sees the overflow), nothing is written
    Rec<NW> *out = (Rec<NW> *)out_;
    const unsigned lane = threadIdx.x & 63;
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, t0 = 0;  // SMX_DEBUG: 100 MHz ticks per phase seen by thread 0
    unsigned npal = 0;
#define SKM_T(i)                                 \
    if (prof && threadIdx.x == 0) {              \
        unsigned long long t1 = wall_clock64();  \
        pt[i] += t1 - t0;                        \
        t0 = t1;                                 \
    }
    if (prof && threadIdx.x == 0) t0 = wall_clock64();
    for (uint32_t item = blockIdx.x; item < nitems; item += gridDim.x) {
        __syncthreads();
        for (uint32_t t = threadIdx.x; t <= SKM_KEYS_PER_ITEM; t += BLK) koff[t] = slot_off[(uint64_t)item * item_stride + t];
        __syncthreads();
        uint64_t s_cur = koff[0];
        const uint64_t s_end = koff[SKM_KEYS_PER_ITEM];
        bool on_boundary = true;  // the chunk starts with the first slot of a key
        while (s_cur < s_end) {
            if (skip_slots != ~0ull && on_boundary) {  // an oversized key starts here: its pieces are items of the second launch
                if (threadIdx.x == 0) s_kend = 0;
                __syncthreads();
                for (uint32_t t = threadIdx.x; t < SKM_KEYS_PER_ITEM; t += BLK)
                    if (koff[t] == s_cur && koff[t + 1] > koff[t] && koff[t + 1] - koff[t] > skip_slots) {
                        s_kend = koff[t + 1];
                        if constexpr (PM) pm.pinfo[(uint64_t)item * SKM_KEYS_PER_ITEM + t] = PM_DIRTY;
                    }
                __syncthreads();
                const unsigned long long ke = s_kend;
                __syncthreads();
                if (ke) {
                    s_cur = ke;
                    continue;
                }
            }
            const uint32_t nst = (uint32_t)((s_end - s_cur < scap) ? s_end - s_cur : scap);
            for (uint32_t i = threadIdx.x; i < nst * SW; i += BLK) {
                uint64_t v = slots[s_cur * SW + i];
                if (i % SW == SW - 1) {
                    cl[i / SW] = (uint8_t)(v >> 56);
                    nbv[i / SW] = (uint8_t)((v >> 48) & 0x3Fu);
                    v &= ~(0xFFFFull << 48);
                }
                sl[i] = v;
            }
            for (uint32_t i = threadIdx.x; i < T; i += BLK) tab[i] = 0xFFFFFFFFu;
            if (threadIdx.x == 0) s_wcount = 0;
            __syncthreads();
            SKM_T(0)
            {  // exclusive prefix of the window counts, two consecutive slots per thread
                const uint32_t i0 = 2 * threadIdx.x, i1 = i0 + 1;
                const uint32_t c0 = i0 < nst ? cl[i0] : 0, c1 = i1 < nst ? cl[i1] : 0;
                uint32_t tot;
                uint32_t ex = block_excl_scan<uint32_t>(c0 + c1, scr, &tot);
                cpre[i0] = ex;
                cpre[i1] = ex + c0;
                if (threadIdx.x == BLK - 1) cpre[2 * BLK] = tot;
            }
            __syncthreads();
            {  // chunk = the staged slots whose instances fit the table, cut back to the last key boundary inside if there
               // is one (every thread tests its own candidates; exactly one matches each condition)
                if (threadIdx.x == 0) {
                    s_take = 0;
                    s_endb = 0;
                }
                __syncthreads();
                for (uint32_t n = threadIdx.x + 1; n <= nst; n += BLK)  // largest n in [1, nst] with cpre[n] <= cap
                    if (cpre[n] <= cap && (n == nst || cpre[n + 1] > cap)) s_nfit = n;
                __syncthreads();
                const unsigned long long x = s_cur + s_nfit;
                if (on_boundary) {
                    if (x < s_end)
                        for (uint32_t t = threadIdx.x; t < SKM_KEYS_PER_ITEM; t += BLK)  // koff[t] <= x < koff[t+1], koff[t] > s_cur
                            if (koff[t] <= x && koff[t + 1] > x && koff[t] > s_cur) s_take = (uint32_t)(koff[t] - s_cur);
                } else {  // the chunk continues a cut key: it takes the rest of THAT key only (whole keys behind it get clean chunks of their own)
                    for (uint32_t t = threadIdx.x; t < SKM_KEYS_PER_ITEM; t += BLK)
                        if (koff[t] <= s_cur && koff[t + 1] > s_cur) s_kend = koff[t + 1];
                }
                __syncthreads();
                if (threadIdx.x == 0) {
                    if (on_boundary) {
                        s_endb = (x >= s_end || s_take != 0) ? 1u : 0u;  // the chunk ends with the last slot of a key
                        if (s_take == 0) s_take = s_nfit;
                    } else if (x >= s_kend) {
                        s_take = (uint32_t)(s_kend - s_cur);
                        s_endb = 1;
                    } else {
                        s_take = s_nfit;
                        s_endb = 0;
                    }
                    s_ninst = cpre[s_take];
                }
            }
            __syncthreads();
            const uint32_t ntake = s_take, ninst = s_ninst;
            for (uint32_t s = threadIdx.x; s < ntake; s += BLK) {
                const uint32_t base = cpre[s], c = cl[s];
                for (uint32_t j = 0; j < c; ++j) imap[base + j] = (uint16_t)((s << 7) | j);
            }
            __syncthreads();
            SKM_T(1)
            for (uint32_t i0 = 0; i0 < ninst; i0 += BLK) {
                const uint32_t i = i0 + threadIdx.x;
                bool won = false;
                uint32_t code = 0;
                if (i < ninst) {
                    code = imap[i];
                    const Rec<NW> x = skm_extract<NW>(sl + (size_t)(code >> 7) * SW, code & 127u, K);
                    const Rec<NW> y = rec_rc<NW>(x, K);
                    const bool fwd = rc_ge<NW>(y, x);
                    Rec<NW> cx;
#pragma unroll
                    for (int t = 0; t < NW; ++t) cx.w[t] = fwd ? x.w[t] : y.w[t];
                    const uint32_t hh = rec_hash32<NW>(cx);
                    uint32_t h = hh & (T - 1);
                    // (EXT: fingerprint 0xFF is not used, so that no live entry can ever read as the empty marker)
                    const uint32_t fp = EXT ? (((hh >> 24) == 0xFFu) ? 0xFEu : (hh >> 24)) : (hh >> 16);
                    const uint32_t entry = EXT ? ((fp << 24) | code) : ((fp << 16) | code);
                    for (;;) {
                        const uint32_t old = atomicCAS(&tab[h], 0xFFFFFFFFu, entry);
                        if (old == 0xFFFFFFFFu) {
                            won = true;
                            break;
                        }
                        if ((EXT ? (old >> 24) : (old >> 16)) == fp) {
                            const Rec<NW> xo = skm_extract<NW>(sl + (size_t)((old & 0xFFFFu) >> 7) * SW, old & 127u, K);
                            if (rec_eq<NW>(xo, x) || rec_eq<NW>(xo, y)) break;  // same canonical k-mer already present
                        }
                        h = (h + 1) & (T - 1);
                    }
                    if constexpr (EXT) {
                        const uint32_t s = code >> 7, j = code & 127u, c = cl[s], nb = nbv[s];
                        const uint64_t *ss = sl + (size_t)s * SW;
                        auto base = [&](uint32_t t) -> uint32_t { return (uint32_t)(ss[t >> 5] >> ((t & 31u) << 1)) & 3u; };
                        const bool hasl = j > 0 || (nb & 1u), hasr = j + 1 < c || (nb & 8u);
                        const uint32_t L = j > 0 ? base(j - 1) : ((nb >> 1) & 3u), R = j + 1 < c ? base(j + K) : ((nb >> 4) & 3u);
                        uint32_t eb;
                        if (fwd) eb = (hasr ? (1u << R) : 0u) | (hasl ? (16u << L) : 0u);
                        else eb = (hasl ? (1u << (3u - L)) : 0u) | (hasr ? (16u << (3u - R)) : 0u);
                        if (eb) atomicOr(&tab[h], eb << 16);
                        code = h;  // the winners are listed by their table entry
                        if constexpr (PM) imap[i] = (uint16_t)((h << 2) | (fwd ? 2u : 0u) | (j + 1 == c ? 1u : 0u));  // for the link pass below
                    }
                }
                if constexpr (!PM) {
                    const unsigned long long m = __ballot(won);
                    if (m) {
                        const int leader = __ffsll((unsigned long long)m) - 1;
                        uint32_t base = 0;
                        if ((int)lane == leader) base = atomicAdd(&s_wcount, (uint32_t)__popcll(m));
                        base = __shfl(base, leader, 64);
                        if (won) wl[base + __popcll(m & ((1ull << lane) - 1))] = (uint16_t)code;
                    }
                }
            }
            __syncthreads();
            // PM: the winners are listed in table-slot order; every thread owns gpt consecutive groups of 16 slots
            const uint32_t ngroups = T >> 4, gpt = ngroups >= (uint32_t)BLK ? ngroups / BLK : 1u, g0 = threadIdx.x * gpt;
            uint32_t occ[4] = {0, 0, 0, 0}, pre = 0;
            if constexpr (PM) {
                uint32_t cnt = 0;
                if (g0 < ngroups)
                    for (uint32_t g = 0; g < gpt; ++g) {
                        uint32_t o = 0;
                        for (uint32_t j = 0; j < 16; ++j)
                            if (tab[(g0 + g) * 16 + j] != 0xFFFFFFFFu) o |= 1u << j;
                        occ[g] = o;
                        cnt += __popc(o);
                    }
                uint32_t tot;
                pre = block_excl_scan<uint32_t>(cnt, scr, &tot);
                if (g0 < ngroups) {
                    uint32_t p = pre;
                    for (uint32_t g = 0; g < gpt; ++g)
                        for (uint32_t o = occ[g]; o; o &= o - 1) wl[p++] = (uint16_t)((g0 + g) * 16 + __ffs(o) - 1);
                }
                if (threadIdx.x == 0) s_wcount = tot;
                if (ngroups * 4u <= 2056u && g0 < ngroups) {  // the group words in LDS too (over cpre, dead by now): rank of any table slot
                    uint32_t p = pre;
                    for (uint32_t g = 0; g < gpt; ++g) {
                        cpre[g0 + g] = p | (occ[g] << 16);
                        p += __popc(occ[g]);
                    }
                }
                __syncthreads();
            }
            SKM_T(2)
            const uint32_t wcount = s_wcount;
            // A chunk of whole keys holds every copy of its k-mers: its winners are exactly distinct ("clean", front of out).
            // Winners of a key that had to be cut may recur in its other chunks ("dirty", stacked from the back of out; the
            // host uniques that part on its own before the two are joined).
            const bool dirty = !on_boundary || !s_endb || force_dirty;
            on_boundary = s_endb != 0;
            if (threadIdx.x == 0) {
                s_skip = 0;
                if (!wcount) s_gbase = 0;
                else if (!dirty) {
                    if constexpr (PM) {
                        const unsigned long long v = atomicAdd(out_count, (unsigned long long)wcount | (1ull << PM_BASE_BITS));
                        s_gbase = v & PM_BASE_MASK;
                        s_cid = (uint32_t)(v >> PM_BASE_BITS);
                        s_skip = s_gbase + wcount > clean_cap;
                        if (s_cid >= pm.max_chunks) {
                            s_skip = 1;
                            *pm.overflow = 1;
                        }
                    } else {
                        s_gbase = atomicAdd(out_count, (unsigned long long)wcount);
                        s_skip = s_gbase + wcount > clean_cap;
                    }
                } else {
                    const unsigned long long d = atomicAdd(dirty_count, (unsigned long long)wcount);
                    s_skip = d + wcount > dirty_cap;
                    s_gbase = s_skip ? 0 : out_cap - d - wcount;
                }
            }
            __syncthreads();
            if constexpr (PM) {
                if (dirty) {  // the cut partition: its k-mers go to the sorted tail
                    if (!force_dirty)
                        for (uint32_t t = threadIdx.x; t < SKM_KEYS_PER_ITEM; t += BLK)
                            if (koff[t] <= s_cur && koff[t + 1] > s_cur) pm.pinfo[(uint64_t)item * SKM_KEYS_PER_ITEM + t] = PM_DIRTY;
                } else if (!s_skip && wcount) {
                    const unsigned long long gb = s_gbase;
                    const uint32_t cid = s_cid;
                    if (g0 < ngroups) {
                        uint32_t p = pre;
                        for (uint32_t g = 0; g < gpt; ++g) {
                            pm.meta[(size_t)cid * ngroups + g0 + g] = p | (occ[g] << 16);
                            p += __popc(occ[g]);
                        }
                    }
                    if (threadIdx.x == 0) pm.cinfo[cid] = gb | ((unsigned long long)wcount << PM_BASE_BITS);
                    for (uint32_t t = threadIdx.x; t < SKM_KEYS_PER_ITEM; t += BLK)  // the partitions that lie in this chunk
                        if (koff[t + 1] > koff[t] && koff[t] >= s_cur && koff[t + 1] <= s_cur + ntake)
                            pm.pinfo[(uint64_t)item * SKM_KEYS_PER_ITEM + t] = gb | ((unsigned long long)cid << PM_BASE_BITS);
                }
            }
            if (!s_skip) {
                Rec<NW> *dst = out + s_gbase;
                for (uint32_t i = threadIdx.x; i < wcount; i += BLK) {
                    uint32_t code = wl[i], eb = 0;
                    if constexpr (EXT) {
                        const uint32_t ent = tab[code];
                        code = ent & 0xFFFFu;
                        eb = (ent >> 16) & 0xFFu;
                    }
                    const Rec<NW> x = skm_extract<NW>(sl + (size_t)(code >> 7) * SW, code & 127u, K);
                    const Rec<NW> y = rec_rc<NW>(x, K);
                    const bool fwd = rc_ge<NW>(y, x);
                    Rec<NW> cx;
#pragma unroll
                    for (int t = 0; t < NW; ++t) cx.w[t] = fwd ? x.w[t] : y.w[t];
                    if constexpr (PM) {
                        if (!dirty) {
                            pm.mask[s_gbase + i] = (uint8_t)eb;
                            // palindromic (k+1)-mers among the extensions (registers only): cx + c is its own reverse complement iff c is the
                            // complement of cx[0] and cx[1..K-1] equals RC(cx)[0..K-2]; likewise b + cx with cx[0..K-2] against RC(cx)[1..K-1]
                            const Rec<NW> &rx = fwd ? y : x;  // reverse complement of the canonical k-mer
                            const unsigned c0 = (unsigned)cx.w[0] & 3u, cl_ = (unsigned)(cx.w[NW - 1] >> (((K - 1) & 31u) << 1)) & 3u;
                            // (first the matching extension bit AND complementary outermost bases of the inner (K-1)-mer: 1 winner in 16)
                            const uint64_t wsel = ((K - 2) >> 5) == (unsigned)(NW - 1) ? cx.w[NW - 1] : cx.w[NW > 1 ? NW - 2 : 0];
                            const unsigned c1 = (unsigned)(cx.w[0] >> 2) & 3u, cm = (unsigned)(wsel >> (((K - 2) & 31u) << 1)) & 3u;
                            if ((((eb >> (3 - c0)) & 1) && c1 + cl_ == 3) || (((eb >> (7 - cl_)) & 1) && c0 + cm == 3)) {
                                Rec<NW> xs, rs, xp = cx, rp = rx;  // suffixes (drop base 0) and prefixes (drop base K-1)
#pragma unroll
                                for (int t = 0; t < NW; ++t) {
                                    xs.w[t] = (cx.w[t] >> 2) | (t + 1 < NW ? cx.w[t + 1] << 62 : 0ull);
                                    rs.w[t] = (rx.w[t] >> 2) | (t + 1 < NW ? rx.w[t + 1] << 62 : 0ull);
                                }
                                const uint64_t topm = ~(3ull << (((K - 1) & 31u) << 1));
                                xp.w[NW - 1] &= topm;
                                rp.w[NW - 1] &= topm;
                                if (((eb >> (3 - c0)) & 1) && rec_eq<NW>(xs, rp)) ++npal;
                                if (((eb >> (7 - cl_)) & 1) && rec_eq<NW>(xp, rs)) ++npal;
                            }
                        }
                    }
                    if constexpr (EXT) cx.w[NW - 1] = (cx.w[NW - 1] << EXT_BITS) | eb;
                    dst[i] = cx;
                }
            }
            if constexpr (PM) {
                // Local links. imap[i] = (table slot, strand, last of its slot) of instance i: instances i, i+1 of one super-k-mer are k-mers
                // X_i -> X_{i+1} side by side in a read, i.e. a de Bruijn edge between the nodes of their table entries (and the reverse
                // one between the other strands). A node with ONE outgoing extension has one successor, whichever read shows it; where
                // several reads disagree the node has several extensions and nobody reads the link. The link table (16 bits per node,
                // indexed by winner rank) lies over the slot staging area, dead after the output above.
                __syncthreads();
                uint16_t *lnk = (uint16_t *)sl;
                const bool fits = !dirty && !s_skip && wcount && (size_t)wcount * 4 <= (size_t)scap * SW * 8 && ngroups * 4u <= 2056u;
                if (fits) {
                    for (uint32_t t = threadIdx.x; t < 2 * wcount; t += BLK) lnk[t] = 0xFFFFu;
                    __syncthreads();
                    auto rank_of = [&](uint32_t h) -> uint32_t {
                        const uint32_t g = cpre[h >> 4];
                        return (g & 0xFFFFu) + __popc((g >> 16) & ((1u << (h & 15u)) - 1u));
                    };
                    for (uint32_t i = threadIdx.x; i + 1 < ninst; i += BLK) {
                        const uint32_t v = imap[i];
                        if (v & 1u) continue;  // the last window of its super-k-mer: its neighbour lives in another slot
                        const uint32_t w2 = imap[i + 1];
                        const uint32_t na = 2 * rank_of(v >> 2) + ((v & 2u) ? 0u : 1u), nb = 2 * rank_of(w2 >> 2) + ((w2 & 2u) ? 0u : 1u);
                        lnk[na] = (uint16_t)nb;
                        lnk[nb ^ 1u] = (uint16_t)(na ^ 1u);
                    }
                    __syncthreads();
                    uint16_t *gl = (uint16_t *)(pm.llink + s_gbase);
                    for (uint32_t t = threadIdx.x; t < 2 * wcount; t += BLK) gl[t] = lnk[t];
                }
            }
            s_cur += ntake;
            __syncthreads();
            SKM_T(3)
            if (prof && threadIdx.x == 0) {
                pt[4] += 1;
                pt[5] += ntake;
            }
        }
    }
    if (prof && threadIdx.x == 0)
        for (int i = 0; i < 6; ++i) atomicAdd(&prof[i], pt[i]);
    if constexpr (PM)
        if (npal) atomicAdd(pm.pals, (unsigned long long)npal);
#undef SKM_T
}

}  // namespace smx
