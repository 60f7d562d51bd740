// spades_amd/csrc/smx_device.hpp — device-side k-mer primitives for gfx950 (wave64).
//
// What each primitive reproduces (reference paths relative to /root/reference/src/common):
//   Rec<NW>            RtSeq payload, sequence/rtseq.hpp:131-151,379-382
//   load_window        seq.start<RtSeq>(K) >> 'A' ; kmer <<= seq[j]   (kmer_splitters.hpp:33-36) —
//                      computed directly as a funnel shift of the packed read stream
//   rec_rc             operator! / FastRC, rtseq.hpp:81-117,389-402 — via v_bfrev instead of the mask ladder
//   rec_is_minimal     IsMinimal, rtseq.hpp:409-417 — as one multiword integer compare (see proof below)
//   xxh3_rec           RtSeq::GetHash -> XXH3_64bits_withSeed(.,8*NW,0), rtseq.hpp:690-696;
//                      ext/include/xxh/xxhash.h:4537-4551 (8 B), :4553-4568 (16 B), :4640-4674 (24/32 B)
//   bucket_of          KMerSegmentPolicy, kmer_index/kmer_mph/kmer_buckets.hpp:47-52 (mulhi, lemiere_mod_reduce.hpp:18-36)
//   rec_less / rec_eq  adt::array_less / array_equal_to, adt/array_vector.hpp:332-350 (word 0 most significant)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smx {

template <int NW>
struct Rec {
    uint64_t w[NW];
};

// Cache-policy experiments (A/B builds: hipcc -DSMX_NT_MASK=<bits>): a load or store tagged with bit B is non-temporal ("nt": streamed past
// the L2 / Infinity Cache replacement) in a build whose mask has that bit, plain otherwise. Bits: 0 k_skm_permute slot stores, 1 its staged loads,
// 2 (was: k_pm_tab's table / jump stores — adopted: nt by default, 37.0 -> 33.2 ms), 3 the walks' streamed lists, 4 k_pm_remote's entry store, 5 k_pm_walk_write's path
// words and edge record, 6 the dedupe stage's record / byte copy-out, 7 its slot loads, 8 the scan's staging stores. Measured (profiles/r06/nontemporal_ab_*.log):
// nt on SCATTERED stores is a loss (k_skm_permute 35 -> 65 ms, k_pm_walk_write 44 -> 59 ms), on the walks' streamed lists and k_pm_remote's store nothing.
#ifndef SMX_NT_MASK
#define SMX_NT_MASK 0
#endif
typedef unsigned long long smx_ull2 __attribute__((ext_vector_type(2)));
template <int BIT, typename T>
__device__ __forceinline__ void st_pol(T *p, T v) {
    if constexpr ((SMX_NT_MASK >> BIT) & 1) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <int BIT, typename T>
__device__ __forceinline__ T ld_pol(const T *p) {
    if constexpr ((SMX_NT_MASK >> BIT) & 1) return __builtin_nontemporal_load(p);
    else return *p;
}

__device__ __forceinline__ uint64_t rev2_64(uint64_t v) {  // reverse the order of the 32 2-bit groups
    v = __brevll(v);
    return ((v >> 1) & 0x5555555555555555ull) | ((v & 0x5555555555555555ull) << 1);
}

// K-mer starting at nucleotide g of a packed stream (tail of the stream is padded by >= 8 words).
template <int NW>
__device__ __forceinline__ Rec<NW> load_window(const uint64_t *__restrict__ seq, uint64_t g, unsigned K) {
    Rec<NW> x;
    const uint64_t q = g >> 5;
    const unsigned sh = (unsigned)(g & 31) << 1;
    uint64_t lo = seq[q];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        uint64_t hi = seq[q + i + 1];
        x.w[i] = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
        lo = hi;
    }
    const unsigned tail = (K & 31) << 1;  // bits used in the last word (0 -> all 64)
    if (tail) x.w[NW - 1] &= (1ull << tail) - 1;
    return x;
}

// Reverse complement. rev2 of the whole NW*64-bit string puts nucleotide K-1 first but leaves the
// k-mer in the TOP 2K bits; complementing before the right shift makes the vacated high bits zero.
template <int NW>
__device__ __forceinline__ Rec<NW> rec_rc(const Rec<NW> &x, unsigned K) {
    uint64_t t[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) t[i] = ~rev2_64(x.w[NW - 1 - i]);
    const unsigned pad = 64u * NW - 2u * K;  // 0..62
    Rec<NW> r;
    if (pad == 0) {
#pragma unroll
        for (int i = 0; i < NW; ++i) r.w[i] = t[i];
    } else {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            uint64_t hi = (i + 1 < NW) ? (t[i + 1] << (64 - pad)) : 0ull;
            r.w[i] = (t[i] >> pad) | hi;
        }
    }
    return r;
}

// IsMinimal(x) <=> x <=_lex RC(x) with nucleotide 0 first. Let i be the first index with
// x[i] != 3-x[K-1-i]; the reference returns x[i] < 3-x[K-1-i]. The highest nucleotide index where
// RC(x) and x differ is j=K-1-i, where RC(x)[j]=3-x[i] and x[j]=x[K-1-i]; so INT(RC(x)) > INT(x)
// (word NW-1 most significant) <=> 3-x[i] > x[K-1-i] <=> x[i] < 3-x[K-1-i]. Palindromes: equal -> true.
// (As code: rc >= x as NW-word integers <=> rc - x leaves no borrow — ONE subtract-with-borrow chain, 2 NW full-rate VALU instructions
// and no branch. The word-by-word form "first difference decides" compiled to exec-mask juggling: 14 instructions per call at NW = 2 in
// the dedupe kernel's insert loop, which is VALU-issue bound, profiles/r04/dedupe2_sq_counters_20M.txt.)
template <int NW>
__device__ __forceinline__ bool rc_ge(const Rec<NW> &rc, const Rec<NW> &x) {
    bool borrow = false;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        uint64_t d;
        const bool b1 = __builtin_sub_overflow(rc.w[i], x.w[i], &d);
        const bool b2 = __builtin_sub_overflow(d, (uint64_t)borrow, &d);
        borrow = b1 | b2;
    }
    return !borrow;
}

template <int NW>
__device__ __forceinline__ bool rec_less(const Rec<NW> &a, const Rec<NW> &b) {
#pragma unroll
    for (int i = 0; i < NW; ++i)
        if (a.w[i] != b.w[i]) return a.w[i] < b.w[i];
    return false;
}
template <int NW>
__device__ __forceinline__ bool rec_eq(const Rec<NW> &a, const Rec<NW> &b) {
    bool e = true;
#pragma unroll
    for (int i = 0; i < NW; ++i) e &= (a.w[i] == b.w[i]);
    return e;
}

// ---- XXH3-64, seed 0, default secret, fixed lengths 8/16/24/32 -------------------------------
// constants = XXH_readLE64(kSecret+o) combinations (xxhash.h:4239-4252)
#define SMX_XXH_BITFLIP8 0xc73ab174c5ecd5a2ull   /* secret[8..16) ^ secret[16..24) */
#define SMX_XXH_BITFLIP16A 0x6782737bea4239b9ull /* secret[24..32) ^ secret[32..40) */
#define SMX_XXH_BITFLIP16B 0xaf56bc3b0996523aull /* secret[40..48) ^ secret[48..56) */
#define SMX_XXH_SEC0 0xbe4ba423396cfeb8ull
#define SMX_XXH_SEC8 0x1cad21f72c81017cull
#define SMX_XXH_SEC16 0xdb979083e96dd4deull
#define SMX_XXH_SEC24 0x1f67b3b7a4a44072ull
#define SMX_XXH_MX1 0x165667919E3779F9ull
#define SMX_XXH_MX2 0x9FB21C651E98DF25ull
#define SMX_XXH_P64_1 0x9E3779B185EBCA87ull

__device__ __forceinline__ uint64_t mul128_fold64(uint64_t a, uint64_t b) { return (a * b) ^ __umul64hi(a, b); }
__device__ __forceinline__ uint64_t rotl64(uint64_t v, int r) { return (v << r) | (v >> (64 - r)); }
__device__ __forceinline__ uint64_t bswap64(uint64_t v) { return __builtin_bswap64(v); }
__device__ __forceinline__ uint64_t xxh3_avalanche(uint64_t h) {
    h ^= h >> 37;
    h *= SMX_XXH_MX1;
    h ^= h >> 32;
    return h;
}

template <int NW>
__device__ __forceinline__ uint64_t xxh3_rec(const Rec<NW> &x) {
    if constexpr (NW == 1) {  // XXH3_len_4to8_64b, len 8: input64 = hi32(w) + (lo32(w) << 32)
        uint64_t h = rotl64(x.w[0], 32) ^ SMX_XXH_BITFLIP8;
        h ^= rotl64(h, 49) ^ rotl64(h, 24);  // XXH3_rrmxmx
        h *= SMX_XXH_MX2;
        h ^= (h >> 35) + 8;
        h *= SMX_XXH_MX2;
        return h ^ (h >> 28);
    } else if constexpr (NW == 2) {  // XXH3_len_9to16_64b, len 16
        uint64_t lo = x.w[0] ^ SMX_XXH_BITFLIP16A, hi = x.w[1] ^ SMX_XXH_BITFLIP16B;
        return xxh3_avalanche(16 + bswap64(lo) + hi + mul128_fold64(lo, hi));
    } else {  // XXH3_len_17to128_64b, len 24 / 32: mix16B(in, secret) + mix16B(in+len-16, secret+16)
        uint64_t acc = (uint64_t)(8 * NW) * SMX_XXH_P64_1;
        acc += mul128_fold64(x.w[0] ^ SMX_XXH_SEC0, x.w[1] ^ SMX_XXH_SEC8);
        acc += mul128_fold64(x.w[NW - 2] ^ SMX_XXH_SEC16, x.w[NW - 1] ^ SMX_XXH_SEC24);
        return xxh3_avalanche(acc);
    }
}

// EXT layout (construction route that takes the extensions straight from the reads): the last word of a record holds
// (k-mer bits << 8) | InOutMask byte. Word 0 is untouched when NW >= 2, so the MSD key digits are those of the k-mer (with one word
// the sort simply treats the record as a (K+4)-mer), and the raw word order of two records is the order of their k-mers, then of
// their bytes: copies of a k-mer with different bytes sort next to each other and are merged (OR) after the sort. Only the bucket
// hash has to see the k-mer alone.
constexpr unsigned EXT_BITS = 8;
template <int NW>
__host__ __device__ __forceinline__ Rec<NW> rec_pure(Rec<NW> x) {
    x.w[NW - 1] >>= EXT_BITS;
    return x;
}
__host__ __device__ inline bool ext_layout_fits(unsigned K, int nw) { return 2 * K + EXT_BITS <= 64u * (unsigned)nw; }
// ... and where the record has no 8 spare bits (k = 29, 31, 61, 63, 93, 95, 125, 127) the partition-major route keeps the byte in its mask
// array alone ("nx": records are the plain k-mers). Code that serves both takes the shift as a value: EXT_BITS, or 0.
template <int NW>
__host__ __device__ __forceinline__ Rec<NW> rec_pure_xs(Rec<NW> x, unsigned xs) {
    x.w[NW - 1] >>= xs;
    return x;
}

__device__ __forceinline__ uint32_t bucket_of(uint64_t hash, uint32_t num_buckets) {
    return num_buckets == 1 ? 0u : (uint32_t)__umul64hi(hash, (uint64_t)num_buckets);
}

// Top 64 bits of the sort key, left-aligned (word 0 is the most significant word of the order; inside
// it the HIGH bits, i.e. the LAST nucleotides it holds, are most significant — pdqsort_pod.h:725-734).
// For K < 32 the 2K key bits are shifted to the top; MSD digits are cut from this value.
template <int NW>
__device__ __forceinline__ uint64_t key_top64(const Rec<NW> &x, unsigned K) {
    return K >= 32 ? x.w[0] : (x.w[0] << (64 - 2 * K));
}

// ---- block-wide helpers (256 threads = 4 waves of 64) ------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_incl_scan(T v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// exclusive scan of one value per thread across the block; returns exclusive prefix, *total = block sum.
// scratch: >= blockDim.x/64 + 1 elements of T in LDS. Contains barriers: call from all threads.
template <typename T>
__device__ __forceinline__ T block_excl_scan(T v, T *scratch, T *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    T inc = wave_incl_scan(v);
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        T run = 0;
        for (int i = 0; i < nw; ++i) {
            T t = scratch[i];
            scratch[i] = run;
            run += t;
        }
        scratch[nw] = run;
    }
    __syncthreads();
    T res = inc - v + scratch[wave];
    *total = scratch[nw];
    __syncthreads();
    return res;
}

// Workgroup barrier that orders LDS only: waits for this wave's LDS/SMEM traffic and rendezvous, but leaves global
// loads in flight (a __syncthreads() also drains vmcnt, which would serialise a software-prefetched next tile).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T>
__device__ __forceinline__ T block_excl_scan_lds(T v, T *scratch, T *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    T inc = wave_incl_scan(v);
    if (lane == 63) scratch[wave] = inc;
    lds_barrier();
    T base = 0, tot = 0;
    for (int i = 0; i < nw; ++i) {  // every thread sums the (few) wave totals itself: no second barrier
        T t = scratch[i];
        if (i < wave) base += t;
        tot += t;
    }
    *total = tot;
    return inc - v + base;
}

}  // namespace smx
