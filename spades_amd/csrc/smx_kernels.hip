// spades_amd/csrc/smx_kernels.hip — gfx950 kernels of the k-mer counting path.
//
// Pipeline (one resident batch; all integer work, HBM-bound — no MFMA by design):
//   mark     per read: set one bit per valid K-mer start in a position bitmask (reads shorter than K
//            contribute nothing: kmer_splitters.hpp:30 / kmercount.cpp:71)
//   L1 hist  every valid window -> record(s) -> XXH3 -> bucket -> level-1 bin; LDS histograms
//   L1 scat  same extraction, LDS-staged multisplit so that each bin receives contiguous runs
//   L2 hist / L2 scat   MSD refinement of every level-1 bin on the next key bits
//   sort     one workgroup per fine bin: LDS bitonic sort by (w0,w1,..) + adjacent-unique
//            (= pdqsort_pod + std::unique of kmer_splitter.hpp:140-141); oversized bins take the
//            global-memory merge path (= the loser-tree merge of kmer_index_builder.hpp:357-415)
//   compact  fine bins are already in (bucket, key) order -> concatenate uniques
//
// Bin order == output order: the key (top 64 bits, left aligned) is read as a fraction x in [0,1); level digits are the
// mixed-radix expansion d1 = floor(x*S1), x' = frac(x*S1), d2 = floor(x'*F2), ... (monotone in the key, any fan-out, so the
// average leaf size can be chosen freely), level-1 bin = bucket*S1 + d1. Concatenating sorted fine bins yields exactly "buckets 0..B-1, each strictly
// increasing" (kmer_index_builder.hpp:190-203).
#include "smx_device.hpp"

namespace smx {

constexpr int BLK = 256;

enum { SRC_READS_ALL = 0, SRC_READS_CANON = 1, SRC_RECS = 2 };
enum { BIN_L1 = 0, BIN_LK = 1, BIN_OWNER = 2 };

struct PassArgs {
    // reads source
    const uint64_t *seq;
    const uint64_t *mask;
    uint64_t G;   // end of the position range handled by this launch (nucleotides)
    uint64_t g0;  // first position of the range (multiple of 64)
    // records source
    const void *recs;
    const unsigned long long *seg_off;  // [nseg+1] record offsets of the input segments
    const unsigned long long *tile_start;  // [nseg+1] prefix of tiles per segment
    uint32_t nseg;
    uint32_t tile_recs;  // records per tile for this launch
    // binning
    unsigned K;
    uint32_t num_buckets;
    uint32_t bucket0;   // BIN_L1: first bucket that receives records (a bucket range of a file): level-1 bins start there
    uint32_t S1;        // BIN_L1: bin = (bucket - bucket0) * S1 + floor(keyfrac * S1)
    uint32_t nprev;     // BIN_LK: key fan-outs already applied (S1, F2, ...), mixed radix on the key fraction
    uint32_t fprev[6];
    uint32_t world;
    uint32_t expand;  // records source, level 1 only: instance i = record i/2, odd i = its reverse complement
    uint32_t ext;     // records carry an extension byte below the k-mer bits of their last word (smx_device.hpp, EXT layout)
    uint32_t F;  // bins per segment
    unsigned long long *hist;    // [nseg*F]
    unsigned long long *cursor;  // [nseg*F]
    void *out;
};

template <int NW, int BINF>
__device__ __forceinline__ uint32_t bin_of(const Rec<NW> &x, const PassArgs &a) {
    if constexpr (BINF == BIN_L1) {
        uint32_t b = bucket_of(xxh3_rec<NW>(a.ext ? rec_pure<NW>(x) : x), a.num_buckets) - a.bucket0;
        return a.S1 > 1 ? b * a.S1 + (uint32_t)__umul64hi(key_top64<NW>(x, a.K), (uint64_t)a.S1) : b;
    } else if constexpr (BINF == BIN_LK) {
        uint64_t f = key_top64<NW>(x, a.K);  // key as a fraction in [0,1) scaled by 2^64
        for (uint32_t i = 0; i < a.nprev; ++i) f *= a.fprev[i];  // fractional part after each earlier digit
        return (uint32_t)__umul64hi(f, (uint64_t)a.F);
    } else {
        uint32_t b = bucket_of(xxh3_rec<NW>(a.ext ? rec_pure<NW>(x) : x), a.num_buckets);
        return (uint32_t)(((uint64_t)b * a.world) / a.num_buckets);
    }
}

// ---------------------------------------------------------------------------------------- ingest
// ASCII reads -> 2-bit stream + (start, len) of the longest ACGTacgt run of every read, first one on ties
// (io::LongestValid, common/io/reads/longest_valid_wrapper.hpp:16-53; dignucl, common/sequence/nucl.hpp:132-142).
// The stream keeps every input position (non-nucleotides become 'A'); (start, len) select the valid run.
__device__ __forceinline__ bool ascii_is_nucl(char c) {
    return c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'a' || c == 'c' || c == 'g' || c == 't';
}
__global__ void k_longest_valid(const char *__restrict__ bases, const unsigned long long *__restrict__ off, uint64_t n,
                                uint64_t *start, uint32_t *len) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint64_t b = off[r], e = off[r + 1];
    uint64_t best_len = 0, best_pos = b, run = 0;
    for (uint64_t i = b; i <= e; ++i) {
        if (i < e && ascii_is_nucl(bases[i])) {
            ++run;
        } else {
            if (run > best_len) {
                best_len = run;
                best_pos = i - run;
            }
            run = 0;
        }
    }
    start[r] = best_pos;
    len[r] = (uint32_t)best_len;
}
__global__ void k_pack_ascii(const char *__restrict__ bases, uint64_t nbases, uint64_t *words, uint64_t nwords) {
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t v = 0;
        const uint64_t p0 = w * 32;
#pragma unroll 8
        for (unsigned j = 0; j < 32; ++j) {
            const uint64_t p = p0 + j;
            if (p < nbases) {
                const char c = bases[p];
                const uint64_t code = (c == 'C' || c == 'c') ? 1 : (c == 'G' || c == 'g') ? 2 : (c == 'T' || c == 't') ? 3 : 0;
                v |= code << (j << 1);
            }
        }
        words[w] = v;
    }
}

// submitted (start, len) pairs: furthest nucleotide any read reaches, and how many reads leave the stream (checked on the device:
// a host loop over 10^8 reads would sit in the timed path of every submission)
__global__ void k_reads_extent(const uint64_t *__restrict__ start, const uint32_t *__restrict__ len, uint64_t n, uint64_t limit,
                               unsigned long long *out /* [2]: max end, reads beyond limit */) {
    unsigned long long mx = 0, bad = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long e = start[r] + len[r];
        if (e > limit || e < start[r]) ++bad;
        else mx = max(mx, e);
    }
    for (int d = 32; d > 0; d >>= 1) {
        mx = max(mx, (unsigned long long)__shfl_xor(mx, d, 64));
        bad += __shfl_xor(bad, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (mx) atomicMax(&out[0], mx);
        if (bad) atomicAdd(&out[1], bad);
    }
}

// ------------------------------------------------------------------------------------------ mark
// min_len: only reads (valid runs) of at least this many nucleotides contribute windows (K-mers of the runs that hold a (K+1)-mer:
// the k-mer file of the construction, kmer_splitters.hpp:138-207, taken from the reads instead of from the (k+1)-mer file)
// limit: positions the mask covers — a read that leaves them is skipped (an asynchronous submission is checked by k_reads_extent on
// the copy stream; its verdict reaches the host only after this kernel was launched)
__global__ void k_mark_windows(const uint64_t *__restrict__ start, const uint32_t *__restrict__ len, uint64_t n,
                               unsigned K, unsigned long long *mask, unsigned long long *total, unsigned min_len = 0, uint64_t limit = ~0ull) {
    __shared__ unsigned long long scratch[BLK / 64 + 2];
    uint64_t r = (uint64_t)blockIdx.x * BLK + threadIdx.x;
    unsigned long long nwin = 0;
    if (r < n) {
        uint32_t l = len[r];
        if (l >= K && l >= min_len && start[r] <= limit && (uint64_t)l <= limit - start[r]) {
            nwin = l - K + 1;
            uint64_t s = start[r], e = s + nwin - 1;
            uint64_t fw = s >> 6, lw = e >> 6;
            unsigned long long fm = ~0ull << (s & 63);
            unsigned long long lm = ~0ull >> (63 - (e & 63));
            if (fw == lw) {
                atomicOr(&mask[fw], fm & lm);
            } else {
                atomicOr(&mask[fw], fm);
                for (uint64_t w = fw + 1; w < lw; ++w) mask[w] = ~0ull;  // interior words belong to this read only
                atomicOr(&mask[lw], lm);
            }
        }
    }
    unsigned long long tot;
    block_excl_scan<unsigned long long>(nwin, scratch, &tot);
    if (threadIdx.x == 0 && tot) atomicAdd(total, tot);
}

// valid windows in the position range [g0, g1) of a chunk (g0, g1 multiples of 64)
__global__ void k_count_range(const unsigned long long *mask, uint64_t w0, uint64_t w1, unsigned long long *total) {
    __shared__ unsigned long long scratch[BLK / 64 + 2];
    unsigned long long c = 0;
    for (uint64_t w = w0 + (uint64_t)blockIdx.x * BLK + threadIdx.x; w < w1; w += (uint64_t)gridDim.x * BLK) c += __popcll(mask[w]);
    unsigned long long tot;
    block_excl_scan<unsigned long long>(c, scratch, &tot);
    if (threadIdx.x == 0 && tot) atomicAdd(total, tot);
}

// -------------------------------------------------------------------------------- record sources
// Fills r[]/valid for this thread of tile `tile`. READS: RPT/RPP positions per thread, position =
// tile*TP + j*BLK + tid (a wave covers 64 consecutive positions = one mask word).
template <int NW, int SRC, int RPT>
__device__ __forceinline__ void fetch_records(const PassArgs &a, uint64_t tile, uint64_t seg_base, uint32_t seg_n,
                                              Rec<NW> (&r)[RPT], uint32_t &validmask) {
    validmask = 0;
    if constexpr (SRC == SRC_RECS) {
        const Rec<NW> *in = (const Rec<NW> *)a.recs;
        const uint64_t base = tile * (uint64_t)(RPT * BLK);
        if (a.expand) {
            // instance pair (2r, 2r+1) = record r and its reverse complement; seg_base and seg_n are even (tiles hold an even
            // number of instances), so every thread loads RPT/2 records and emits both strands of each
            const uint64_t rbase = (seg_base + base) >> 1;
            const uint64_t rn = seg_n > base ? (seg_n - base) >> 1 : 0;  // records left in this tile's part of the segment
#pragma unroll
            for (int j = 0; j < RPT / 2; ++j) {
                const uint64_t ri = (uint64_t)j * BLK + threadIdx.x;
                if (ri < rn) {
                    const Rec<NW> x = in[rbase + ri];
                    r[2 * j] = x;
                    r[2 * j + 1] = rec_rc<NW>(x, a.K);
                    validmask |= 3u << (2 * j);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                uint64_t i = base + (uint64_t)j * BLK + threadIdx.x;
                if (i < seg_n) {
                    r[j] = in[seg_base + i];
                    validmask |= 1u << j;
                }
            }
        }
    } else {
        constexpr int RPP = (SRC == SRC_READS_ALL) ? 2 : 1;
        constexpr int PPT = RPT / RPP;
        const uint64_t base = a.g0 + tile * (uint64_t)(PPT * BLK);
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            uint64_t g = base + (uint64_t)j * BLK + threadIdx.x;
            bool ok = g < a.G && ((a.mask[g >> 6] >> (g & 63)) & 1);
            if (ok) {
                Rec<NW> x = load_window<NW>(a.seq, g, a.K);
                Rec<NW> y = rec_rc<NW>(x, a.K);
                if constexpr (SRC == SRC_READS_ALL) {
                    r[2 * j] = x;
                    r[2 * j + 1] = y;
                    validmask |= 3u << (2 * j);
                } else {
                    r[j] = rc_ge<NW>(y, x) ? x : y;  // canonical representative (IsMinimal filter on read+RC stream)
                    validmask |= 1u << j;
                }
            }
        }
    }
}

// tile -> (segment, tile index inside segment) for the records source
__device__ __forceinline__ bool locate_tile(const PassArgs &a, uint64_t blk, uint32_t *sh, uint32_t &seg, uint32_t &tin) {
    if (threadIdx.x == 0) {
        uint32_t lo = 0, hi = a.nseg;  // find seg with tile_start[seg] <= blk < tile_start[seg+1]
        if (blk >= a.tile_start[a.nseg]) {
            sh[0] = 0xFFFFFFFFu;
        } else {
            while (hi - lo > 1) {
                uint32_t mid = (lo + hi) >> 1;
                if (a.tile_start[mid] <= blk) lo = mid; else hi = mid;
            }
            sh[0] = lo;
            sh[1] = (uint32_t)(blk - a.tile_start[lo]);
        }
    }
    __syncthreads();
    seg = sh[0];
    tin = sh[1];
    __syncthreads();
    return seg != 0xFFFFFFFFu;
}

// ------------------------------------------------------------------------------------------ hist
// READS: persistent grid-stride over position tiles, one LDS histogram per workgroup, flushed once.
// RECS:  one workgroup per (segment, tile of a.tile_recs records), flushed per tile.
template <int NW, int SRC, int BINF, int RPT>
__global__ void __launch_bounds__(BLK) k_hist(PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lh[];
    __shared__ uint32_t sh[2];
    for (uint32_t i = threadIdx.x; i < a.F; i += BLK) lh[i] = 0;
    __syncthreads();
    if constexpr (SRC == SRC_RECS) {
        uint32_t seg, tin;
        if (!locate_tile(a, blockIdx.x, sh, seg, tin)) return;
        const uint64_t sb = a.seg_off[seg];
        const uint64_t sn = a.seg_off[seg + 1] - sb;
        const uint64_t t0 = (uint64_t)tin * a.tile_recs;
        const uint64_t t1 = (sn < t0 + a.tile_recs) ? sn : t0 + a.tile_recs;
        const Rec<NW> *in = (const Rec<NW> *)a.recs;
        if (a.expand) {  // instances (2r, 2r+1) = record r and its reverse complement; sb, t0, t1 are even
            for (uint64_t ri = ((sb + t0) >> 1) + threadIdx.x; ri < ((sb + t1) >> 1); ri += BLK) {
                const Rec<NW> x = in[ri];
                atomicAdd(&lh[bin_of<NW, BINF>(x, a)], 1u);
                atomicAdd(&lh[bin_of<NW, BINF>(rec_rc<NW>(x, a.K), a)], 1u);
            }
        } else {
            for (uint64_t i = t0 + threadIdx.x; i < t1; i += BLK) atomicAdd(&lh[bin_of<NW, BINF>(in[sb + i], a)], 1u);
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < a.F; i += BLK)
            if (lh[i]) atomicAdd(&a.hist[(uint64_t)seg * a.F + i], (unsigned long long)lh[i]);
    } else {
        constexpr int RPP = (SRC == SRC_READS_ALL) ? 2 : 1;
        constexpr int PPT = RPT / RPP;
        const uint64_t ntiles = (a.G - a.g0 + (uint64_t)PPT * BLK - 1) / ((uint64_t)PPT * BLK);
        for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            Rec<NW> r[RPT];
            uint32_t vm;
            fetch_records<NW, SRC, RPT>(a, tile, 0, 0, r, vm);
#pragma unroll
            for (int j = 0; j < RPT; ++j)
                if (vm & (1u << j)) atomicAdd(&lh[bin_of<NW, BINF>(r[j], a)], 1u);
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < a.F; i += BLK)
            if (lh[i]) atomicAdd(&a.hist[i], (unsigned long long)lh[i]);
    }
}

// Level-1 histogram from records, fused with the level-2 histogram: one LDS table of F1*F2 counters per persistent workgroup
// (bin1 * F2 + second mixed-radix digit), flushed once; the level-2 hist pass over the scattered records (one full re-read) is
// not needed any more. (Caching the level-1 bins as bytes for the scatter was measured too: the byte loads cost more than re-hashing.)
// a.F = F1, a.hist = joint histogram [F1*F2]; level-1 totals are its row sums (k_rowsum).
template <int NW>
__global__ void __launch_bounds__(1024) k_hist_l1_joint(PassArgs a, uint32_t F2, uint64_t n_in) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lh[];
    const uint32_t nj = a.F * F2;
    for (uint32_t i = threadIdx.x; i < nj; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const Rec<NW> *in = (const Rec<NW> *)a.recs;
    const uint64_t S1 = a.S1;
    for (uint64_t ri = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; ri < n_in; ri += (uint64_t)gridDim.x * blockDim.x) {
        const Rec<NW> x = in[ri];
        {
            const uint32_t b1 = bin_of<NW, BIN_L1>(x, a);
            uint64_t f = key_top64<NW>(x, a.K);
            if (S1 > 1) f *= S1;
            atomicAdd(&lh[b1 * F2 + (uint32_t)__umul64hi(f, (uint64_t)F2)], 1u);
        }
        if (a.expand) {
            const Rec<NW> y = rec_rc<NW>(x, a.K);
            const uint32_t b1 = bin_of<NW, BIN_L1>(y, a);
            uint64_t f = key_top64<NW>(y, a.K);
            if (S1 > 1) f *= S1;
            atomicAdd(&lh[b1 * F2 + (uint32_t)__umul64hi(f, (uint64_t)F2)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nj; i += blockDim.x)
        if (lh[i]) atomicAdd(&a.hist[i], (unsigned long long)lh[i]);
}
__global__ void k_rowsum(const unsigned long long *joint, uint32_t F1, uint32_t F2, unsigned long long *rows) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= F1) return;
    unsigned long long s = 0;
    for (uint32_t d = 0; d < F2; ++d) s += joint[(uint64_t)b * F2 + d];
    rows[b] = s;
}

// --------------------------------------------------------------------------------------- scatter
// LDS-staged multisplit of one tile (<= RPT*BLK records) into a.F bins of its segment.
// LDS carve (dynamic): stage[TR*NW] u64 | ldelta[F] u64 | lhist[F] u32 | sbin[TR] u16
template <int NW, int SRC, int BINF, int RPT>
__global__ void __launch_bounds__(BLK) k_scatter(PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    constexpr int TR = RPT * BLK;
    uint64_t *stage = lds64;
    uint64_t *ldelta = stage + (size_t)TR * NW;
    uint32_t *lhist = (uint32_t *)(ldelta + a.F);
    uint16_t *sbin = (uint16_t *)(lhist + a.F);
    __shared__ uint32_t sh[2];
    __shared__ uint32_t scr[BLK / 64 + 2];

    uint32_t seg = 0, tin = blockIdx.x;
    uint64_t sb = 0;
    uint32_t sn = 0;
    if constexpr (SRC == SRC_RECS) {
        if (!locate_tile(a, blockIdx.x, sh, seg, tin)) return;
        sb = a.seg_off[seg];
        uint64_t full = a.seg_off[seg + 1] - sb;
        // fetch_records indexes inside the segment with a 64-bit base; seg_n is only compared, clamp to u32 range per tile
        uint64_t t0 = (uint64_t)tin * TR;
        sb += t0;
        sn = (uint32_t)((full - t0 < (uint64_t)TR) ? full - t0 : (uint64_t)TR);
        tin = 0;
    }
    for (uint32_t i = threadIdx.x; i < a.F; i += BLK) lhist[i] = 0;
    __syncthreads();

    Rec<NW> r[RPT];
    uint32_t vm;
    fetch_records<NW, SRC, RPT>(a, tin, sb, sn, r, vm);
    uint32_t packed[RPT];  // bin << 16 | slot... slot can reach TR-1 (<= 4095): 13+ bits each -> use two arrays
    uint32_t slot[RPT];
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        if (vm & (1u << j)) {
            packed[j] = bin_of<NW, BINF>(r[j], a);
            slot[j] = atomicAdd(&lhist[packed[j]], 1u);
        }
    }
    __syncthreads();
    // exclusive scan of lhist over F bins + global reservation
    const uint32_t per = (a.F + BLK - 1) / BLK;
    const uint32_t d0 = threadIdx.x * per, d1 = min(a.F, d0 + per);
    uint32_t sum = 0;
    for (uint32_t d = d0; d < d1; ++d) sum += lhist[d];
    uint32_t total;
    uint32_t run = block_excl_scan<uint32_t>(sum, scr, &total);
    unsigned long long *cur = a.cursor + (uint64_t)seg * a.F;
    for (uint32_t d = d0; d < d1; ++d) {
        uint32_t c = lhist[d];
        lhist[d] = run;
        if (c) ldelta[d] = (uint64_t)atomicAdd(&cur[d], (unsigned long long)c) - run;
        run += c;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        if (vm & (1u << j)) {
            uint32_t idx = lhist[packed[j]] + slot[j];
#pragma unroll
            for (int w = 0; w < NW; ++w) stage[(size_t)idx * NW + w] = r[j].w[w];
            sbin[idx] = (uint16_t)packed[j];
        }
    }
    __syncthreads();
    Rec<NW> *out = (Rec<NW> *)a.out;
    const Rec<NW> *st = (const Rec<NW> *)stage;
    for (uint32_t idx = threadIdx.x; idx < total; idx += BLK) out[ldelta[sbin[idx]] + idx] = st[idx];
}

// Level-1 scatter straight from the reads. Same multisplit as k_scatter, but the LDS stage holds only a 16-bit
// (local position, strand) tag per record instead of the record: 19 KB of LDS instead of 76 KB (8 instead of 2
// workgroups per CU) and no record registers live across the barriers. The copy-out phase re-extracts the window
// (L1/L2 hits) and writes the record to its reserved slot. LDS: spos[TR] u16 | sbin[TR] u16 | ldelta[F] u64 | lhist[F] u32
template <int NW, int SRC, int BINF, int RPT>
__global__ void __launch_bounds__(BLK) k_scatter_reads(PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    constexpr int TR = RPT * BLK;
    constexpr int RPP = (SRC == SRC_READS_ALL) ? 2 : 1;
    constexpr int PPT = RPT / RPP;
    uint64_t *ldelta = lds64;
    uint32_t *lhist = (uint32_t *)(ldelta + a.F);
    uint16_t *spos = (uint16_t *)(lhist + a.F);
    uint16_t *sbin = spos + TR;
    __shared__ uint32_t scr[BLK / 64 + 2];
    for (uint32_t i = threadIdx.x; i < a.F; i += BLK) lhist[i] = 0;
    __syncthreads();
    const uint64_t base = a.g0 + (uint64_t)blockIdx.x * (uint64_t)(PPT * BLK);
    uint32_t packed[RPT];  // bin << 16 | slot
    uint32_t vm = 0;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const uint64_t g = base + (uint64_t)j * BLK + threadIdx.x;
        const bool ok = g < a.G && ((a.mask[g >> 6] >> (g & 63)) & 1);
        if (ok) {
            Rec<NW> x = load_window<NW>(a.seq, g, a.K);
            Rec<NW> y = rec_rc<NW>(x, a.K);
            if constexpr (SRC == SRC_READS_ALL) {
                const uint32_t b0 = bin_of<NW, BINF>(x, a), b1 = bin_of<NW, BINF>(y, a);
                packed[2 * j] = (b0 << 16) | atomicAdd(&lhist[b0], 1u);
                packed[2 * j + 1] = (b1 << 16) | atomicAdd(&lhist[b1], 1u);
                vm |= 3u << (2 * j);
            } else {
                const Rec<NW> c = rc_ge<NW>(y, x) ? x : y;
                const uint32_t b0 = bin_of<NW, BINF>(c, a);
                packed[j] = (b0 << 16) | atomicAdd(&lhist[b0], 1u);
                vm |= 1u << j;
            }
        }
    }
    __syncthreads();
    const uint32_t per = (a.F + BLK - 1) / BLK;
    const uint32_t d0 = threadIdx.x * per, d1 = min(a.F, d0 + per);
    uint32_t sum = 0;
    for (uint32_t d = d0; d < d1; ++d) sum += lhist[d];
    uint32_t total;
    uint32_t run = block_excl_scan<uint32_t>(sum, scr, &total);
    for (uint32_t d = d0; d < d1; ++d) {
        uint32_t c = lhist[d];
        lhist[d] = run;
        if (c) ldelta[d] = (uint64_t)atomicAdd(&a.cursor[d], (unsigned long long)c) - run;
        run += c;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        if (vm & (1u << j)) {
            const uint32_t bin = packed[j] >> 16;
            const uint32_t idx = lhist[bin] + (packed[j] & 0xFFFFu);
            const uint32_t lpos = (uint32_t)(j / RPP) * BLK + threadIdx.x;
            spos[idx] = (uint16_t)((lpos << 1) | (RPP == 2 ? (j & 1) : 0));
            sbin[idx] = (uint16_t)bin;
        }
    }
    __syncthreads();
    Rec<NW> *out = (Rec<NW> *)a.out;
    for (uint32_t idx = threadIdx.x; idx < total; idx += BLK) {
        const uint32_t tag = spos[idx];
        Rec<NW> x = load_window<NW>(a.seq, base + (tag >> 1), a.K);
        if constexpr (SRC == SRC_READS_ALL) {
            if (tag & 1) x = rec_rc<NW>(x, a.K);
        } else {
            Rec<NW> y = rec_rc<NW>(x, a.K);
            if (!rc_ge<NW>(y, x)) x = y;
        }
        out[ldelta[sbin[idx]] + idx] = x;
    }
}

// ------------------------------------------------------------------------------------------ scans
// Single-workgroup exclusive scan (n small, e.g. level-1 histogram): out[i] = sum in[0..i), out[n] = total.
__global__ void k_scan_small(const unsigned long long *in, unsigned long long *out, uint32_t n) {
    __shared__ unsigned long long scr[BLK / 64 + 2];
    const uint32_t per = (n + BLK - 1) / BLK;
    const uint32_t d0 = min(n, threadIdx.x * per), d1 = min(n, d0 + per);
    unsigned long long sum = 0;
    for (uint32_t d = d0; d < d1; ++d) sum += in[d];
    unsigned long long tot;
    unsigned long long run = block_excl_scan<unsigned long long>(sum, scr, &tot);
    for (uint32_t d = d0; d < d1; ++d) {
        unsigned long long c = in[d];
        out[d] = run;
        run += c;
    }
    if (threadIdx.x == 0) out[n] = tot;
}

constexpr int SCAN_PER = 8;
constexpr int SCAN_TILE = BLK * SCAN_PER;
// large scan, phase 1: per-tile sums
__global__ void k_scan_reduce(const unsigned long long *in, uint64_t n, unsigned long long *partial) {
    __shared__ unsigned long long scr[BLK / 64 + 2];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_PER;
    unsigned long long s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_PER; ++j)
        if (base + j < n) s += in[base + j];
    unsigned long long tot;
    block_excl_scan<unsigned long long>(s, scr, &tot);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}
// phase 3: per-tile exclusive scan with tile offset; also writes out[n] = grand total from the last tile
__global__ void k_scan_apply(const unsigned long long *in, uint64_t n, const unsigned long long *partial_off,
                             unsigned long long *out) {
    __shared__ unsigned long long scr[BLK / 64 + 2];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_PER;
    unsigned long long v[SCAN_PER], s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_PER; ++j) {
        v[j] = (base + j < n) ? in[base + j] : 0;
        s += v[j];
    }
    unsigned long long tot;
    unsigned long long run = block_excl_scan<unsigned long long>(s, scr, &tot) + partial_off[blockIdx.x];
#pragma unroll
    for (int j = 0; j < SCAN_PER; ++j) {
        if (base + j < n) out[base + j] = run;
        run += v[j];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == BLK - 1) out[n] = run;
}

// tiles per segment (scanned afterwards into PassArgs::tile_start)
__global__ void k_tile_counts(const unsigned long long *seg_off, uint32_t nseg, uint32_t tile, unsigned long long *cnt) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < nseg) cnt[s] = (seg_off[s + 1] - seg_off[s] + tile - 1) / tile;
}

// ------------------------------------------------------------------------------------------ sort
template <int NW>
__device__ __forceinline__ Rec<NW> lds_get(const uint64_t *s, uint32_t i) {
    Rec<NW> r;
#pragma unroll
    for (int w = 0; w < NW; ++w) r.w[w] = s[(size_t)i * NW + w];
    return r;
}
template <int NW>
__device__ __forceinline__ void lds_put(uint64_t *s, uint32_t i, const Rec<NW> &r) {
#pragma unroll
    for (int w = 0; w < NW; ++w) s[(size_t)i * NW + w] = r.w[w];
}

// Normalised bitonic network (every comparator puts the smaller record at the lower index), so the
// virtual +inf padding at indices >= n never moves and n need not be a power of two.
template <int NW>
__device__ void lds_bitonic_sort(uint64_t *s, uint32_t n) {
    uint32_t N = 1;
    while (N < n) N <<= 1;
    for (uint32_t k = 2; k <= N; k <<= 1) {
        const uint32_t hk = k >> 1;
        for (uint32_t t = threadIdx.x; t < (N >> 1); t += BLK) {
            uint32_t i = (t / hk) * k + (t % hk);
            uint32_t j = i ^ (k - 1);
            if (j < n) {
                Rec<NW> x = lds_get<NW>(s, i), y = lds_get<NW>(s, j);
                if (rec_less<NW>(y, x)) {
                    lds_put<NW>(s, i, y);
                    lds_put<NW>(s, j, x);
                }
            }
        }
        __syncthreads();
        for (uint32_t jj = k >> 2; jj > 0; jj >>= 1) {
            for (uint32_t t = threadIdx.x; t < (N >> 1); t += BLK) {
                uint32_t i = (t / jj) * (jj << 1) + (t % jj);
                uint32_t j = i + jj;
                if (j < n) {
                    Rec<NW> x = lds_get<NW>(s, i), y = lds_get<NW>(s, j);
                    if (rec_less<NW>(y, x)) {
                        lds_put<NW>(s, i, y);
                        lds_put<NW>(s, j, x);
                    }
                }
            }
            __syncthreads();
        }
    }
}

// sorted LDS array -> unique records written to dst (global); returns unique count (all threads).
template <int NW>
__device__ uint32_t lds_unique_store(const uint64_t *s, uint32_t n, Rec<NW> *dst, uint32_t *scr) {
    const uint32_t per = (n + BLK - 1) / BLK;
    const uint32_t i0 = min(n, threadIdx.x * per), i1 = min(n, i0 + per);
    uint32_t cnt = 0;
    for (uint32_t i = i0; i < i1; ++i)
        cnt += (i == 0) || !rec_eq<NW>(lds_get<NW>(s, i), lds_get<NW>(s, i - 1));
    uint32_t tot;
    uint32_t run = block_excl_scan<uint32_t>(cnt, scr, &tot);
    for (uint32_t i = i0; i < i1; ++i) {
        Rec<NW> x = lds_get<NW>(s, i);
        if (i == 0 || !rec_eq<NW>(x, lds_get<NW>(s, i - 1))) dst[run++] = x;
    }
    return tot;
}

// Per-thread "insertion sort with dedup" of one small LDS range [b, e): keeps a sorted, duplicate-free
// prefix [b, b+d) and returns d. Equal keys (coverage duplicates) cost one compare and no move;
// the prefix never overtakes the read cursor (b+d <= i).
template <int NW>
__device__ __forceinline__ uint32_t lds_insertion_unique(uint64_t *s, uint32_t b, uint32_t e) {
    uint32_t d = 0;
    for (uint32_t i = b; i < e; ++i) {
        Rec<NW> x = lds_get<NW>(s, i);
        uint32_t j = b + d;
        bool dup = false;
        while (j > b) {
            Rec<NW> y = lds_get<NW>(s, j - 1);
            if (rec_less<NW>(x, y)) {
                --j;
            } else {
                dup = rec_eq<NW>(x, y);
                break;
            }
        }
        if (dup) continue;
        for (uint32_t t = b + d; t > j; --t) lds_put<NW>(s, t, lds_get<NW>(s, t - 1));
        lds_put<NW>(s, j, x);
        ++d;
    }
    return d;
}

// Leaf kernel for medium bins (129..cap records, all sharing their bucket and the key bits consumed by the MSD levels);
// persistent workgroups loop over the medium list. Duplicates (coverage) are removed FIRST, by an exact LDS hash table
// (open addressing on a 32-bit slot holding the index of the owning record; a probe that meets an occupied slot compares
// the full key, so equal keys meet their owner and different keys move on). Only the distinct records are then ordered:
// digit = next sub_bits key bits -> count per digit + per-digit linked list -> prefix scan -> rank = prefix[digit] +
// #smaller keys in the digit's list (lists hold ~1 record) -> store at the final position of the bin's region.
// Skewed leaves (a digit with > 128 distinct keys) fall back to the bitonic network + adjacent-unique.
// LDS: stage[cap*NW] u64 | tab[T] | head[S] | cntf[S+1] | next[cap]  (u32 each)
struct FracArgs {  // key fan-outs applied by the MSD levels (mixed radix), for the in-LDS digit of the leaf kernels
    uint32_t n;
    uint32_t f[6];
};
template <int NW>
__device__ __forceinline__ uint32_t frac_digit(const Rec<NW> &x, unsigned K, const FracArgs &fa, uint32_t S) {
    uint64_t f = key_top64<NW>(x, K);
    for (uint32_t i = 0; i < fa.n; ++i) f *= fa.f[i];
    return (uint32_t)__umul64hi(f, (uint64_t)S);
}

// 32-bit hash of a record for the on-chip hash tables (LDS dedupe sets, and the chunk tables of the "pm" route that are probed again
// in HBM: every user must see the same function). The 32-bit halves are folded by XOR under odd rotations, then murmur3's fmix32: two
// v_mul_lo_u32 per record — the round-3 form (three 64-bit multiplies = nine quarter-rate multiplies) was a quarter of the dedupe
// kernel's VALU time. A colliding pair only costs a probe: every table compares the records themselves.
template <int NW>
__device__ __forceinline__ uint32_t rec_hash32(const Rec<NW> &x) {
    uint32_t h = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const uint32_t lo = (uint32_t)x.w[w], hi = (uint32_t)(x.w[w] >> 32);
        h ^= __builtin_rotateleft32(lo, (7 * w) & 31) ^ __builtin_rotateleft32(hi, (7 * w + 13) & 31);
    }
    // (round 5: ONE multiply instead of fmix32's two — h *= 0x9E3779B1; h ^= h >> 15 — made the dedupe kernel slower, 85.0 -> 88.4 ms at
    // config 3: longer probe sequences cost more than the multiply saves; profiles/r05/kernel_variants_ab_and_phase_ticks.log)
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

template <int NW, int LPT>
__global__ void __launch_bounds__(BLK) k_sort_small(void *buf, const unsigned long long *off, uint32_t nbins, uint32_t cap,
                                                    unsigned K, FracArgs fa, unsigned sub_bits, uint32_t T,
                                                    unsigned long long *ucount, uint32_t *biglist, uint32_t *bigcount,
                                                    const uint32_t *list, const uint32_t *listcount) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    __shared__ uint32_t scr[BLK / 64 + 2];
    __shared__ uint32_t maxc;
    const uint32_t S = 1u << sub_bits;
    uint32_t *tab = (uint32_t *)(lds64 + (size_t)cap * NW);  // [T]
    uint32_t *head = tab + T;                                // [S]
    uint32_t *cntf = head + S;                               // [S+1]
    uint32_t *nxt = cntf + S + 1;                            // [cap]
    const uint32_t nwork = list ? *listcount : nbins;
    for (uint32_t bi = blockIdx.x; bi < nwork; bi += gridDim.x) {
        const uint32_t b = list ? list[bi] : bi;
        const uint64_t o = off[b];
        const uint64_t n64 = off[b + 1] - o;
        if (n64 == 0) {
            if (threadIdx.x == 0) ucount[b] = 0;
            continue;
        }
        if (n64 > cap) {
            if (threadIdx.x == 0) biglist[atomicAdd(bigcount, 1u)] = b;
            continue;
        }
        const uint32_t n = (uint32_t)n64;
        Rec<NW> *g = (Rec<NW> *)buf + o;
        bool bitonic = (sub_bits == 0 || n <= 64);
        if (bitonic) {
            for (uint32_t i = threadIdx.x; i < n; i += BLK) lds_put<NW>(lds64, i, g[i]);
            __syncthreads();
        } else {
            for (uint32_t i = threadIdx.x; i < T; i += BLK) tab[i] = 0xFFFFFFFFu;
            for (uint32_t i = threadIdx.x; i <= S; i += BLK) {
                cntf[i] = 0;
                if (i < S) head[i] = 0xFFFFFFFFu;
            }
            if (threadIdx.x == 0) maxc = 0;
            Rec<NW> r[LPT];
#pragma unroll
            for (int j = 0; j < LPT; ++j) {
                uint32_t i = threadIdx.x + j * BLK;
                if (i < n) {
                    r[j] = g[i];
                    lds_put<NW>(lds64, i, r[j]);
                }
            }
            __syncthreads();
            uint32_t fm = 0;
#pragma unroll
            for (int j = 0; j < LPT; ++j) {
                uint32_t i = threadIdx.x + j * BLK;
                if (i < n) {
                    uint32_t slot = rec_hash32<NW>(r[j]) & (T - 1);
                    bool first = false;
                    for (;;) {
                        uint32_t old = atomicCAS(&tab[slot], 0xFFFFFFFFu, i);
                        if (old == 0xFFFFFFFFu) {
                            first = true;
                            break;
                        }
                        if (rec_eq<NW>(lds_get<NW>(lds64, old), r[j])) break;  // duplicate of record `old`
                        slot = (slot + 1) & (T - 1);
                    }
                    if (first) {
                        fm |= 1u << j;
                        const uint32_t d = frac_digit<NW>(r[j], K, fa, S);
                        atomicAdd(&cntf[d], 1u);
                        nxt[i] = atomicExch(&head[d], i);
                    }
                }
            }
            __syncthreads();
            const uint32_t per = (S + BLK - 1) / BLK;
            const uint32_t d0 = threadIdx.x * per, d1 = min(S, d0 + per);
            uint32_t sum = 0, mx = 0;
            for (uint32_t d = d0; d < d1; ++d) {
                uint32_t c = cntf[d];
                sum += c;
                mx = max(mx, c);
            }
            if (mx > 128) atomicMax(&maxc, mx);
            uint32_t tot;
            uint32_t run = block_excl_scan<uint32_t>(sum, scr, &tot);
            for (uint32_t d = d0; d < d1; ++d) {
                uint32_t c = cntf[d];
                cntf[d] = run;
                run += c;
            }
            __syncthreads();
            bitonic = maxc > 128;
            if (!bitonic) {
#pragma unroll
                for (int j = 0; j < LPT; ++j) {
                    if (fm & (1u << j)) {
                        const uint32_t d = frac_digit<NW>(r[j], K, fa, S);
                        uint32_t smaller = 0;
                        for (uint32_t p = head[d]; p != 0xFFFFFFFFu; p = nxt[p])
                            smaller += rec_less<NW>(lds_get<NW>(lds64, p), r[j]) ? 1u : 0u;
                        g[cntf[d] + smaller] = r[j];
                    }
                }
                if (threadIdx.x == 0) ucount[b] = tot;
                __syncthreads();
                continue;
            }
        }
        lds_bitonic_sort<NW>(lds64, n);
        uint32_t u = lds_unique_store<NW>(lds64, n, g, scr);
        if (threadIdx.x == 0) ucount[b] = u;
        __syncthreads();
    }
}

// Fast path of the medium-leaf kernel: same algorithm as k_sort_small's hash branch, but (1) the records of the NEXT leaf are
// loaded into registers before the current one is processed and every in-leaf barrier orders LDS only (lds_barrier), so the
// global loads stay in flight across the whole leaf; (2) no fallback code in the kernel: a skewed leaf (a digit with > 128
// distinct keys) is left untouched and queued for k_sort_small. LDS as k_sort_small.
template <int NW, int LPT, bool NODUP>
__global__ void __launch_bounds__(BLK) k_sort_hash(void *buf, const unsigned long long *off, uint32_t cap, unsigned K, FracArgs fa,
                                                   unsigned sub_bits, uint32_t T, unsigned long long *ucount,
                                                   const uint32_t *list, const uint32_t *listcount, uint32_t *fblist, uint32_t *fbcount) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    __shared__ uint32_t scr[BLK / 64 + 2];
    __shared__ uint32_t maxc, dupflag;
    const uint32_t S = 1u << sub_bits;
    uint32_t *tab = (uint32_t *)(lds64 + (size_t)cap * NW);  // [T]
    uint32_t *head = tab + T;                                // [S]
    uint32_t *cntf = head + S;                               // [S+1]
    uint32_t *nxt = cntf + S + 1;                            // [cap]
    const uint32_t nwork = *listcount;
    uint32_t bi = blockIdx.x;
    if (bi >= nwork) return;
    // software pipeline: records of leaf bi are in r_n when its iteration starts
    uint32_t b_n = list[bi];
    uint64_t o_n = off[b_n];
    uint32_t n_n = (uint32_t)(off[b_n + 1] - o_n);
    Rec<NW> r_n[LPT];
    {
        const Rec<NW> *g = (const Rec<NW> *)buf + o_n;
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            uint32_t i = threadIdx.x + j * BLK;
            if (i < n_n) r_n[j] = g[i];
        }
    }
    for (; bi < nwork; bi += gridDim.x) {
        const uint32_t b = b_n, n = n_n;
        Rec<NW> *g = (Rec<NW> *)buf + o_n;
        Rec<NW> r[LPT];
#pragma unroll
        for (int j = 0; j < LPT; ++j) r[j] = r_n[j];
        if constexpr (!NODUP)
            for (uint32_t i = threadIdx.x; i < T; i += BLK) tab[i] = 0xFFFFFFFFu;
        for (uint32_t i = threadIdx.x; i <= S; i += BLK) {
            cntf[i] = 0;
            if (i < S) head[i] = 0xFFFFFFFFu;
        }
        if (threadIdx.x == 0) {
            maxc = 0;
            dupflag = 0;
        }
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            uint32_t i = threadIdx.x + j * BLK;
            if (i < n) lds_put<NW>(lds64, i, r[j]);
        }
        // prefetch the next leaf of this workgroup (loads complete while this leaf is hashed and ranked)
        const uint32_t bi2 = bi + gridDim.x;
        if (bi2 < nwork) {
            b_n = list[bi2];
            o_n = off[b_n];
            n_n = (uint32_t)(off[b_n + 1] - o_n);
            const Rec<NW> *g2 = (const Rec<NW> *)buf + o_n;
#pragma unroll
            for (int j = 0; j < LPT; ++j) {
                uint32_t i = threadIdx.x + j * BLK;
                if (i < n_n) r_n[j] = g2[i];
            }
        }
        lds_barrier();
        uint32_t fm = 0;
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            uint32_t i = threadIdx.x + j * BLK;
            if (i < n) {
                bool first = NODUP;  // NODUP: the caller expects distinct records; equal ones are detected while ranking
                if constexpr (!NODUP) {
                    uint32_t slot = rec_hash32<NW>(r[j]) & (T - 1);
                    for (;;) {
                        uint32_t old = atomicCAS(&tab[slot], 0xFFFFFFFFu, i);
                        if (old == 0xFFFFFFFFu) {
                            first = true;
                            break;
                        }
                        if (rec_eq<NW>(lds_get<NW>(lds64, old), r[j])) break;  // duplicate of record `old`
                        slot = (slot + 1) & (T - 1);
                    }
                }
                if (first) {
                    fm |= 1u << j;
                    const uint32_t d = frac_digit<NW>(r[j], K, fa, S);
                    atomicAdd(&cntf[d], 1u);
                    nxt[i] = atomicExch(&head[d], i);
                }
            }
        }
        lds_barrier();
        const uint32_t per = (S + BLK - 1) / BLK;
        const uint32_t d0 = threadIdx.x * per, d1 = min(S, d0 + per);
        uint32_t sum = 0, mx = 0;
        for (uint32_t d = d0; d < d1; ++d) {
            uint32_t c = cntf[d];
            sum += c;
            mx = max(mx, c);
        }
        if (mx > 128) atomicMax(&maxc, mx);
        uint32_t tot;
        uint32_t run = block_excl_scan_lds<uint32_t>(sum, scr, &tot);
        for (uint32_t d = d0; d < d1; ++d) {
            uint32_t c = cntf[d];
            cntf[d] = run;
            run += c;
        }
        lds_barrier();
        if (maxc > 128) {
            if (threadIdx.x == 0) fblist[atomicAdd(fbcount, 1u)] = b;  // skewed: global data untouched, k_sort_small finishes it
        } else if constexpr (!NODUP) {
#pragma unroll
            for (int j = 0; j < LPT; ++j) {
                if (fm & (1u << j)) {
                    const uint32_t d = frac_digit<NW>(r[j], K, fa, S);
                    uint32_t smaller = 0;
                    for (uint32_t p = head[d]; p != 0xFFFFFFFFu; p = nxt[p])
                        smaller += rec_less<NW>(lds_get<NW>(lds64, p), r[j]) ? 1u : 0u;
                    g[cntf[d] + smaller] = r[j];
                }
            }
            if (threadIdx.x == 0) ucount[b] = tot;
        } else {
            uint32_t pos[LPT];
            bool dup = false;
#pragma unroll
            for (int j = 0; j < LPT; ++j) {
                if (fm & (1u << j)) {
                    const uint32_t me = threadIdx.x + j * BLK;
                    const uint32_t d = frac_digit<NW>(r[j], K, fa, S);
                    uint32_t smaller = 0;
                    for (uint32_t p = head[d]; p != 0xFFFFFFFFu; p = nxt[p]) {
                        const Rec<NW> o = lds_get<NW>(lds64, p);
                        smaller += rec_less<NW>(o, r[j]) ? 1u : 0u;
                        dup |= p != me && rec_eq<NW>(o, r[j]);
                    }
                    pos[j] = cntf[d] + smaller;
                }
            }
            if (dup) dupflag = 1;
            lds_barrier();
            if (dupflag) {  // not distinct after all: nothing written yet, the general kernel sorts AND uniques this leaf
                if (threadIdx.x == 0) fblist[atomicAdd(fbcount, 1u)] = b;
            } else {
#pragma unroll
                for (int j = 0; j < LPT; ++j)
                    if (fm & (1u << j)) g[pos[j]] = r[j];
                if (threadIdx.x == 0) ucount[b] = tot;
            }
        }
        lds_barrier();  // LDS tables are re-initialised by the next iteration
    }
}

// ---------------------------------------------------------------------------------- wave-level leaf sort
template <int NW>
__device__ __forceinline__ Rec<NW> rec_shfl_xor(const Rec<NW> &x, int m) {
    Rec<NW> y;
#pragma unroll
    for (int w = 0; w < NW; ++w) y.w[w] = __shfl_xor((unsigned long long)x.w[w], m, 64);
    return y;
}
template <int NW>
__device__ __forceinline__ Rec<NW> rec_shfl_up1(const Rec<NW> &x) {
    Rec<NW> y;
#pragma unroll
    for (int w = 0; w < NW; ++w) y.w[w] = __shfl_up((unsigned long long)x.w[w], 1, 64);
    return y;
}
template <int NW>
__device__ __forceinline__ Rec<NW> rec_readlane63(const Rec<NW> &x) {
    Rec<NW> y;
#pragma unroll
    for (int w = 0; w < NW; ++w) y.w[w] = __shfl((unsigned long long)x.w[w], 63, 64);
    return y;
}

// Bitonic network over 64*R records held striped in registers (element e = r*64 + lane). Compare-exchange
// partners at distance < 64 live in another lane (cross-lane shuffle), at distance >= 64 in another register
// of the same lane. No LDS, no barriers.
template <int NW, int R>
__device__ __forceinline__ void wave_bitonic(Rec<NW> (&x)[R]) {
    const unsigned lane = threadIdx.x & 63;
#pragma unroll
    for (unsigned k = 2; k <= 64u * R; k <<= 1) {
#pragma unroll
        for (unsigned j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                const unsigned dj = j >> 6;
#pragma unroll
                for (unsigned r = 0; r < (unsigned)R; ++r) {
                    if ((r & dj) == 0) {
                        const unsigned r2 = r | dj;
                        const bool asc = (((r << 6) | lane) & k) == 0;
                        const bool sw = asc ? rec_less<NW>(x[r2], x[r]) : rec_less<NW>(x[r], x[r2]);
                        if (sw) {
                            Rec<NW> t = x[r];
                            x[r] = x[r2];
                            x[r2] = t;
                        }
                    }
                }
            } else {
#pragma unroll
                for (unsigned r = 0; r < (unsigned)R; ++r) {
                    Rec<NW> y = rec_shfl_xor<NW>(x[r], (int)j);
                    const bool lower = (lane & j) == 0;
                    const bool asc = (((r << 6) | lane) & k) == 0;
                    const bool keep_min = lower == asc;
                    const bool take = keep_min ? rec_less<NW>(y, x[r]) : rec_less<NW>(x[r], y);
                    if (take) x[r] = y;
                }
            }
        }
    }
}

// sort + unique + in-place store of one leaf of n <= 64*R records by one wave; returns the unique count.
// Pads are all-ones records: indistinguishable from a real all-ones record, which is harmless because the first n
// sorted positions then still hold exactly the multiset of the real records.
template <int NW, int R>
__device__ __forceinline__ uint32_t wave_leaf(Rec<NW> *g, uint32_t n) {
    const unsigned lane = threadIdx.x & 63;
    Rec<NW> x[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t e = (uint32_t)r * 64 + lane;
        if (e < n) x[r] = g[e];
        else {
#pragma unroll
            for (int w = 0; w < NW; ++w) x[r].w[w] = ~0ull;
        }
    }
    wave_bitonic<NW, R>(x);
    uint32_t base = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t e = (uint32_t)r * 64 + lane;
        Rec<NW> prev = rec_shfl_up1<NW>(x[r]);
        if (r > 0) {
            Rec<NW> p63 = rec_readlane63<NW>(x[r - 1 < 0 ? 0 : r - 1]);
            if (lane == 0) prev = p63;
        }
        const bool flag = e < n && (e == 0 || !rec_eq<NW>(x[r], prev));
        const unsigned long long bal = __ballot(flag);
        if (flag) g[base + __popcll(bal & ((1ull << lane) - 1))] = x[r];
        base += (uint32_t)__popcll(bal);
    }
    return base;
}

// size classes of the fine bins: empty | small (register sort, one wave) | medium (LDS kernel) | big (merge kernel).
// One thread per bin; the per-thread atomicAdd(.,1) is wave-aggregated by the compiler, so the three list counters
// see one atomic per wave instead of one per bin (a single word retires only ~88 atomics/us).
__global__ void k_classify(const unsigned long long *off, uint32_t nbins, uint32_t cap1, uint32_t cap2, unsigned long long *ucount,
                           uint32_t *smalllist, uint32_t *smallcount, uint32_t *medlist, uint32_t *medcount,
                           uint32_t *med2list, uint32_t *med2count, uint32_t *biglist, uint32_t *bigcount) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbins) return;
    const uint64_t n = off[b + 1] - off[b];
    if (n == 0) ucount[b] = 0;
    else if (n > 0xFFFFFFFEull) atomicOr(bigcount, 0x80000000u);  // the rank-merge path keeps 32-bit run lengths: the host reports it
    else if (n > cap2) biglist[atomicAdd(bigcount, 1u)] = b;
    else if (n > cap1) med2list[atomicAdd(med2count, 1u)] = b;  // second LDS class (4x larger leaves, 1 workgroup per CU)
    else if (n > 128) medlist[atomicAdd(medcount, 1u)] = b;     // R=4 register sorts cost more per record than the LDS kernel
    else smalllist[atomicAdd(smallcount, 1u)] = b;
}

// One WAVE per small fine bin (persistent waves over the small list): <= 128 records, finished in registers.
template <int NW>
__global__ void __launch_bounds__(BLK) k_sort_wave(void *buf, const unsigned long long *off, unsigned long long *ucount,
                                                   const uint32_t *list, const uint32_t *listcount) {
    const unsigned lane = threadIdx.x & 63;
    const uint32_t wid = (blockIdx.x * BLK + threadIdx.x) >> 6, nwaves = (gridDim.x * BLK) >> 6;
    const uint32_t nwork = *listcount;
    for (uint32_t i = wid; i < nwork; i += nwaves) {
        const uint32_t b = list[i];
        const uint64_t o = off[b];
        const uint32_t n = (uint32_t)(off[b + 1] - o);
        Rec<NW> *g = (Rec<NW> *)buf + o;
        uint32_t u;
        if (n <= 64) u = wave_leaf<NW, 1>(g, n);
        else u = wave_leaf<NW, 2>(g, n);
        if (lane == 0) ucount[b] = u;
    }
}

// fine bins -> final output, one wave per bin (bins are contiguous in output order)
template <int NW>
__global__ void __launch_bounds__(BLK) k_compact_wave(const void *buf, const unsigned long long *off, const unsigned long long *ucount,
                                                      const unsigned long long *uoff, uint32_t nbins, void *out) {
    const unsigned lane = threadIdx.x & 63;
    const uint32_t wid = (blockIdx.x * BLK + threadIdx.x) >> 6, nwaves = (gridDim.x * BLK) >> 6;
    for (uint32_t b = wid; b < nbins; b += nwaves) {
        const uint32_t u = (uint32_t)ucount[b];
        if (u == 0) continue;
        const Rec<NW> *src = (const Rec<NW> *)buf + off[b];
        Rec<NW> *dst = (Rec<NW> *)out + uoff[b];
        for (uint32_t i = lane; i < u; i += 64) dst[i] = src[i];
    }
}

template <int NW>
__device__ __forceinline__ uint32_t lower_bound_g(const Rec<NW> *a, uint32_t n, const Rec<NW> &x) {  // #elements < x
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (rec_less<NW>(a[mid], x)) lo = mid + 1; else hi = mid;
    }
    return lo;
}
template <int NW>
__device__ __forceinline__ uint32_t upper_bound_g(const Rec<NW> *a, uint32_t n, const Rec<NW> &x) {  // #elements <= x
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (!rec_less<NW>(x, a[mid])) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Oversized fine bins (skewed key prefixes, poly-A ...): chunked LDS sort+unique into runs, then
// log2(#runs) rank-merge levels through the (free) ping-pong region, then one unique pass.
// One workgroup per oversized bin; correctness path, not a throughput path.
template <int NW>
__global__ void __launch_bounds__(BLK) k_sort_big(void *buf, void *scratch, const unsigned long long *off, uint32_t cap,
                                                  unsigned long long *ucount, const uint32_t *biglist,
                                                  const uint32_t *bigcount, uint32_t *runlen) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    __shared__ uint32_t scr[BLK / 64 + 2];
    __shared__ uint64_t lastrec[4];
    for (uint32_t bi = blockIdx.x; bi < *bigcount; bi += gridDim.x) {
        const uint32_t b = biglist[bi];
        const uint64_t o = off[b];
        const uint64_t n = off[b + 1] - o;
        Rec<NW> *A = (Rec<NW> *)buf + o;
        Rec<NW> *B = (Rec<NW> *)scratch + o;
        uint32_t *rl = runlen + (o / cap) + b;
        const uint32_t nch = (uint32_t)((n + cap - 1) / cap);
        // runs: chunk c -> sorted unique at A + c*cap, length rl[c]
        for (uint32_t c = 0; c < nch; ++c) {
            const uint64_t c0 = (uint64_t)c * cap;
            const uint32_t cn = (uint32_t)((n - c0 < (uint64_t)cap) ? n - c0 : (uint64_t)cap);
            for (uint32_t i = threadIdx.x; i < cn; i += BLK) lds_put<NW>(lds64, i, A[c0 + i]);
            __syncthreads();
            lds_bitonic_sort<NW>(lds64, cn);
            uint32_t u = lds_unique_store<NW>(lds64, cn, A + c0, scr);
            if (threadIdx.x == 0) rl[c] = u;
            __syncthreads();
        }
        Rec<NW> *src = A, *dst = B;
        for (uint32_t w = 1; w < nch; w <<= 1) {
            for (uint32_t c = 0; c < nch; c += 2 * w) {
                const Rec<NW> *X = src + (uint64_t)c * cap;
                const uint32_t la = rl[c];
                const bool hasY = c + w < nch;
                const Rec<NW> *Y = src + (uint64_t)(c + w) * cap;
                const uint32_t lb = hasY ? rl[c + w] : 0;
                Rec<NW> *O = dst + (uint64_t)c * cap;
                for (uint32_t i = threadIdx.x; i < la; i += BLK) {
                    Rec<NW> x = X[i];
                    O[i + lower_bound_g<NW>(Y, lb, x)] = x;
                }
                for (uint32_t j = threadIdx.x; j < lb; j += BLK) {
                    Rec<NW> y = Y[j];
                    O[j + upper_bound_g<NW>(X, la, y)] = y;
                }
                __syncthreads();
                if (threadIdx.x == 0) rl[c] = la + lb;
                __syncthreads();
            }
            Rec<NW> *t = src;
            src = dst;
            dst = t;
        }
        // unique pass src[0..len) -> A[0..u)  (in place when src == A: writes never pass the read front)
        const uint32_t len = rl[0];
        uint32_t outpos = 0;
        for (uint32_t base = 0; base < len; base += BLK) {
            const uint32_t i = base + threadIdx.x;
            Rec<NW> x;
            bool keep = false;
            if (i < len) {
                x = src[i];
                if (i == 0) keep = true;
                else if (threadIdx.x == 0) keep = !rec_eq<NW>(x, *(const Rec<NW> *)lastrec);
                else keep = !rec_eq<NW>(x, src[i - 1]);
            }
            uint32_t tot;
            uint32_t rank = block_excl_scan<uint32_t>(keep ? 1u : 0u, scr, &tot);  // barriers: all reads done before writes
            if (i < len && (threadIdx.x == BLK - 1 || i == len - 1)) *(Rec<NW> *)lastrec = x;
            if (keep) A[outpos + rank] = x;
            outpos += tot;
            __syncthreads();
        }
        if (threadIdx.x == 0) ucount[b] = outpos;
        __syncthreads();
    }
}

// fine bins -> final output (bins are contiguous in output order)
template <int NW>
__global__ void __launch_bounds__(BLK) k_compact(const void *buf, const unsigned long long *off, const unsigned long long *ucount,
                                                 const unsigned long long *uoff, uint32_t nbins, void *out) {
    for (uint32_t b = blockIdx.x; b < nbins; b += gridDim.x) {
        const Rec<NW> *src = (const Rec<NW> *)buf + off[b];
        Rec<NW> *dst = (Rec<NW> *)out + uoff[b];
        const uint32_t u = (uint32_t)ucount[b];
        for (uint32_t i = threadIdx.x; i < u; i += BLK) dst[i] = src[i];
    }
}

// bucket_off[b] = uoff[b * bins_per_bucket], bucket_off[B] = total
__global__ void k_bucket_offsets(const unsigned long long *uoff, uint32_t num_buckets, uint32_t bins_per_bucket,
                                 unsigned long long *bucket_off) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b <= num_buckets) bucket_off[b] = uoff[(uint64_t)b * bins_per_bucket];
}


// ---- two-strand result of a both-strands count (spades-kmercount: every k-mer of read and reverse complement, kmercount.cpp:48-122) --
// The set is closed under reverse complement: it is the sorted canonical set C and the sorted set R = RC(C) (minus the k-mers that are
// their own reverse complement, which C holds), bucket by bucket merged. k_ts_rc makes R's records; k_ts_merge merges one bucket.
template <int NW>
__global__ void __launch_bounds__(BLK) k_ts_rc(const void *c_, uint64_t n, unsigned K, void *r_, unsigned long long *count /* records written (when compacting) */,
                                               uint32_t B, uint32_t b0, uint32_t b1 /* only the reverse complements that file under the buckets [b0, b1) */,
                                               uint64_t cap /* records r has room for: nothing is written beyond (the count still counts) */) {
    const Rec<NW> *c = (const Rec<NW> *)c_;
    Rec<NW> *r = (Rec<NW> *)r_;
    __shared__ uint32_t scratch[BLK / 64 + 2];
    __shared__ unsigned long long s_base;
    // no k-mer of odd length is its own reverse complement: with all buckets wanted R[i] = RC(C[i]), nothing to leave out
    const bool direct = (K & 1u) && b0 == 0 && b1 >= B;
    constexpr int PER = 16;
    for (uint64_t t0 = (uint64_t)blockIdx.x * BLK * PER; t0 < n; t0 += (uint64_t)gridDim.x * BLK * PER) {
        if (direct) {
#pragma unroll 4
            for (int j = 0; j < PER; ++j) {
                const uint64_t i = t0 + (uint64_t)j * BLK + threadIdx.x;
                if (i < n) r[i] = rec_rc<NW>(c[i], K);
            }
            continue;
        }
        uint32_t keep = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint64_t i = t0 + (uint64_t)j * BLK + threadIdx.x;
            if (i < n) {
                const Rec<NW> x = c[i], y = rec_rc<NW>(x, K);
                const uint32_t b = bucket_of(xxh3_rec<NW>(y), B);
                if (!rec_eq<NW>(x, y) && b >= b0 && b < b1) keep |= 1u << j;
            }
        }
        uint32_t tot;
        const uint32_t ex = block_excl_scan<uint32_t>(__popc(keep), scratch, &tot);
        if (threadIdx.x == 0) s_base = tot ? atomicAdd(count, (unsigned long long)tot) : 0ull;
        __syncthreads();
        uint64_t o = s_base + ex;
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if (keep & (1u << j)) {
                if (o < cap) r[o] = rec_rc<NW>(c[t0 + (uint64_t)j * BLK + threadIdx.x], K);  // (again: 16 records of NW words are not kept in registers)
                ++o;
            }
        __syncthreads();
    }
}

// merge of two sorted, disjoint record arrays (merge path): tiles of TS_TILE outputs; the two splits of a tile by binary search in
// HBM, its inputs staged in LDS, every thread merges TS_TILE / BLK consecutive outputs from its own split
constexpr int TS_TILE = BLK * 8;
template <int NW>
__global__ void __launch_bounds__(BLK) k_ts_merge(const void *a_, uint64_t na, const void *b_, uint64_t nb, void *out_) {
    const Rec<NW> *A = (const Rec<NW> *)a_, *Bv = (const Rec<NW> *)b_;
    Rec<NW> *out = (Rec<NW> *)out_;
    extern __shared__ __attribute__((aligned(16))) uint64_t ts_lds[];
    Rec<NW> *st = (Rec<NW> *)ts_lds;  // [TS_TILE]: the tile's part of A, then its part of B
    __shared__ uint64_t s_split[2];
    const uint64_t total = na + nb, ntiles = (total + TS_TILE - 1) / TS_TILE;
    auto split = [&](uint64_t d) -> uint64_t {  // elements of A among the first d outputs (A before B on ties: there are none)
        uint64_t lo = d > nb ? d - nb : 0, hi = d < na ? d : na;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (rec_less<NW>(A[mid], Bv[d - 1 - mid])) lo = mid + 1;
            else hi = mid;
        }
        return lo;
    };
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t d0 = tile * TS_TILE, d1 = d0 + TS_TILE < total ? d0 + TS_TILE : total;
        __syncthreads();
        if (threadIdx.x < 2) s_split[threadIdx.x] = split(threadIdx.x ? d1 : d0);
        __syncthreads();
        const uint64_t a0 = s_split[0], a1 = s_split[1], b0 = d0 - a0, b1 = d1 - a1;
        const uint32_t ca = (uint32_t)(a1 - a0), cb = (uint32_t)(b1 - b0);
        for (uint32_t i = threadIdx.x; i < ca; i += BLK) st[i] = A[a0 + i];
        for (uint32_t i = threadIdx.x; i < cb; i += BLK) st[ca + i] = Bv[b0 + i];
        __syncthreads();
        const uint32_t n = ca + cb, dd = min(threadIdx.x * 8u, n), de = min(dd + 8u, n);
        uint32_t lo = dd > cb ? dd - cb : 0, hi = dd < ca ? dd : ca;  // local merge path
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (rec_less<NW>(st[mid], st[ca + dd - 1 - mid])) lo = mid + 1;
            else hi = mid;
        }
        uint32_t ia = lo, ib = dd - lo;
        for (uint32_t o = dd; o < de; ++o) {
            const bool ta = ib >= cb || (ia < ca && rec_less<NW>(st[ia], st[ca + ib]));
            out[d0 + o] = ta ? st[ia] : st[ca + ib];
            ia += ta ? 1u : 0u;
            ib += ta ? 0u : 1u;
        }
    }
}

}  // namespace smx
