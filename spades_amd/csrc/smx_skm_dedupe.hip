// spades_amd/csrc/smx_skm_dedupe.hip — the on-chip dedupe stage behind the super-k-mer scan (included at the end of smx_superkmer.hip).
//
// The partition-sorted super-k-mer slots go in, the distinct canonical k-mers of every CHUNK of whole minimizer partitions come out
// (kmer_index/kmer_mph/kmer_splitter.hpp:123-170 — buffer, sort, unique — is what the stage stands in for). A chunk is expanded inside
// LDS: one exact hash set whose entries are fingerprint | reference (slot, offset) — the k-mers themselves stay in the staged slots —
// and the winners are appended to a record array in HBM. All copies of a k-mer (either strand) share their canonical minimizer, hence
// their partition, hence — unless the partition is cut by the chunk capacity — their chunk: winners of whole partitions are exactly
// distinct ("clean", front of the output); winners of a cut partition may recur ("dirty", stacked from the back; the host uniques
// that part before the two are joined). MODE 0: plain records. MODE 1 (EXT): every instance also knows the bases next to it inside
// its read (the slot, or the slot's neighbour sets at its two ends) — the extensions the (K+1)-mers around it give its k-mer
// (InOutMask bits in the k-mer's canonical frame: out bits 0-3 by next base, in bits 4-7 by previous base;
// kmer_extension_index_builder.hpp:45-60, inout_mask.hpp:92-131); table entries are 8-bit fingerprint | 8 extension bits (OR of all
// copies) | reference, and the winners leave in the EXT layout (smx_device.hpp). MODE 2 (PM): EXT + partition-major output for the
// construction route that never sorts the k-mers (PmOut, smx_superkmer.hip; smx_pm.hip).
//
// Round 4. The round-3 kernel (one workgroup per item of 256 partitions, planning its chunks as it went, one instance per thread and
// pass) spent 20 us per chunk of ~1400 instances and was bound by VALU issue on the busiest SIMDs (rocprofv3 SQ counters:
// profiles/r04). What changed:
//   * the chunks are PLANNED by a kernel of their own (k_skm_plan) into one list that the dedupe kernel strides over: it knows its
//     next chunk while it works on this one and fetches its slots a chunk ahead;
//   * identical super-k-mers are FOLDED by the plan before anything is expanded (the scan stores every slot on the strand of its
//     canonical minimizer, so error-free copies from reads of either strand are equal word for word): a quarter of the instances of
//     a 30x read set never reach the hash set, and the chunks are cut AFTER the fold, so every wave of every chunk is full;
//   * a thread owns a SEGMENT of up to SEG consecutive instances of one super-k-mer and ROLLS the k-mer and its reverse complement from
//     one instance to the next (two shifts) instead of extracting and reverse-complementing every instance from the staged slot;
//   * the winners are rolled out of the segment a second time after the table is complete and staged through LDS in table-slot order,
//     so that they leave in full lines; local links are written in the same loop; occupancy words by ballot; every barrier orders
//     LDS only (the prefetch of the next chunk stays in flight across them).
#pragma once

namespace smx {

#ifndef SMX_WPE
#define SMX_WPE 5  // waves per SIMD the dedupe kernel is compiled for (96 VGPRs, a few spilled: 119.0 -> 115.8 ms against 4 at config 3)
#endif
#ifndef SMX_SB
#define SMX_SB 2
#endif
struct SkmChunk {
    unsigned long long a;  // first slot (40 bits) | slots << 40 (10 bits) | dirty << 50 | segments << 51 (12 bits)
    unsigned long long b;  // first partition (32 bits) | partitions << 32 (0: none to be entered into the partition table)
};
constexpr unsigned long long SKM_M40 = (1ull << 40) - 1;
constexpr uint32_t SKM_PLAN_BLOCK = 512;   // list entries a planning workgroup reserves at a time (the unused rest stays zero = a hole)
constexpr uint32_t SKM_PLAN_BUF = 256;
// Chunk plan. One workgroup per item of 256 consecutive partitions ("keys"); thread t holds key t: its slots (slot_off) and its
// segments (kseg, counted by the scan). The item is cut greedily into chunks of whole keys (<= maxseg segments, <= maxslots slots):
// every thread finds, by binary search over the prefix sums, where a chunk that starts with its key would end; thread 0 follows that
// chain. A key that does not fit a chunk on its own is cut into pieces by the whole workgroup, reading its slots ("dirty" chunks:
// their winners may recur, the host uniques them). skip_slots: a key (partition) of more slots than this is left out — one workgroup would chew on a homopolymer
// partition of 10^7 instances for 100 ms while the others idle; its slot range is cut into pieces that a second launch deals out as items
// of their own (slot_off = one pseudo-key per piece with rows of item_stride = 257 offsets, kseg == nullptr, force_dirty = 1: every
// chunk of such a piece is a dirty one and the partition table is not touched).
constexpr uint32_t SKM_FOLD_TAB = 4096;        // slot set of the fold (entries of 4 B), per planning workgroup
constexpr uint32_t SKM_FOLD_ITEM_MAX = 3328;   // most slots of an item that are folded (beyond: the item is planned unfolded)
constexpr uint32_t SKM_FOLD_KEY_MAX = 1024;    // a partition of more slots is left alone (low complexity: every window a slot of its own)
template <int NW>
__global__ void __launch_bounds__(BLK) k_skm_plan(uint64_t *__restrict__ slots, const unsigned long long *__restrict__ slot_off,
                                                  const uint32_t *__restrict__ kseg, uint32_t fold, uint32_t nitems, uint32_t item_stride, uint32_t force_dirty,
                                                  unsigned long long skip_slots, uint32_t maxseg, uint32_t maxslots, SkmChunk *list,
                                                  unsigned long long *list_alloc /* [0] next free entry, [1] clean chunks, [2] overflow, [3] folded instances */,
                                                  unsigned long long list_cap, unsigned long long *pinfo) {
    constexpr int SW = 2 * NW, SEG = SkmSeg<NW>::value;
    static_assert(SKM_KEYS_PER_ITEM == BLK, "one key per thread");
    __shared__ unsigned long long koff[SKM_KEYS_PER_ITEM + 1];
    __shared__ uint32_t pseg[SKM_KEYS_PER_ITEM + 1], pslot[SKM_KEYS_PER_ITEM + 1];  // exclusive prefix sums over the keys (barrier keys count 0)
    __shared__ uint16_t nxt[SKM_KEYS_PER_ITEM];  // end (exclusive) of the chunk that starts with key t
    __shared__ unsigned long long bar[BLK / 64], cutm[BLK / 64];  // keys no chunk may span (skipped or cut) / keys to be cut, bit per key
    __shared__ SkmChunk buf[SKM_PLAN_BUF];
    __shared__ uint32_t cpre[2 * BLK + 2];
    __shared__ uint32_t scr[BLK / 64 + 2];
    __shared__ uint32_t s_nbuf, s_lleft, s_nfit, s_nclean;
    __shared__ unsigned long long s_lbase, s_folded;
    __shared__ __attribute__((aligned(16))) uint32_t ftab[SKM_FOLD_TAB];  // fold: (tag << 16 | slot of the item), 0xFFFFFFFF = empty
    __shared__ uint32_t ksg[SKM_KEYS_PER_ITEM];  // segments per key after the fold
    const uint32_t t = threadIdx.x;
    if (t == 0) {
        s_nbuf = 0;
        s_lleft = 0;
        s_lbase = 0;
        s_nclean = 0;
        s_folded = 0;
    }
    auto flush = [&]() {  // all threads
        __syncthreads();
        const uint32_t nb = s_nbuf;
        if (nb) {
            if (t == 0 && s_lleft < nb) {
                const unsigned long long b = atomicAdd(&list_alloc[0], (unsigned long long)SKM_PLAN_BLOCK);
                if (b + SKM_PLAN_BLOCK > list_cap) {
                    list_alloc[2] = 1;
                    s_lbase = ~0ull;
                    s_lleft = 0x7FFFFFFFu;
                } else {
                    s_lbase = b;
                    s_lleft = SKM_PLAN_BLOCK;
                }
            }
            __syncthreads();
            const unsigned long long base = s_lbase;
            if (base != ~0ull)
                for (uint32_t i = t; i < nb; i += BLK) list[base + i] = buf[i];
            __syncthreads();
            if (t == 0) {
                if (s_lbase != ~0ull) {
                    s_lbase += nb;
                    s_lleft -= nb;
                }
                s_nbuf = 0;
            }
        }
        __syncthreads();
    };
    auto nseg_of = [&](unsigned long long s) -> uint32_t {
        const uint32_t c = (uint32_t)(slots[s * SW + SW - 1] >> 56);
        return (c + SEG - 1) / SEG;
    };
    for (uint32_t item = blockIdx.x; item < nitems; item += gridDim.x) {
        __syncthreads();
        for (uint32_t i = t; i <= SKM_KEYS_PER_ITEM; i += BLK) koff[i] = slot_off[(uint64_t)item * item_stride + i];
        __syncthreads();
        const unsigned long long n = koff[t + 1] - koff[t];
        uint32_t sg = kseg && n ? kseg[(uint64_t)item * SKM_KEYS_PER_ITEM + t] : 0u;
        // ---- Fold. The scan stores every slot on the strand of its canonical minimizer, so the copies of an error-free stretch of the
        // genome that reads of either strand hold in full are equal word for word (and equal slots have equal minimizers: they meet in
        // this item). One copy keeps its instances and takes the others' neighbour bases (sets: an OR); the others are left with no
        // window (count byte 0: the dedupe kernel stages them and gives them no segment). A quarter of the instances of a 30x read set
        // go this way BEFORE the chunks are cut, so that the chunks are full of what is left.
        const unsigned long long S0 = koff[0], SN = koff[SKM_KEYS_PER_ITEM] - koff[0];
        if (fold && kseg && SN <= SKM_FOLD_ITEM_MAX) {  // (uniform)
            for (uint32_t i = t; i < SKM_FOLD_TAB / 4; i += BLK) ((uint4 *)ftab)[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
            ksg[t] = n > SKM_FOLD_KEY_MAX ? sg : 0u;
            __syncthreads();
            unsigned nfold = 0;
            constexpr int FB = 4;  // slots per thread in flight: their words are loaded together (the loop is bound by the latency of those loads)
            for (uint32_t i0 = t; i0 < (uint32_t)SN; i0 += FB * BLK) {
                uint64_t wq[FB][SW];
#pragma unroll
                for (int f = 0; f < FB; ++f) {
                    const uint32_t i = i0 + (uint32_t)f * BLK;
                    if (i < (uint32_t)SN) {
                        const uint64_t *ps = slots + (S0 + i) * SW;
#pragma unroll
                        for (int q = 0; q < SW; ++q) wq[f][q] = ps[q];
                    }
                }
#pragma unroll
                for (int f = 0; f < FB; ++f) {
                    const uint32_t i = i0 + (uint32_t)f * BLK;
                    if (i >= (uint32_t)SN) break;
                    uint32_t lo = 0, hi = SKM_KEYS_PER_ITEM;  // key of slot S0 + i: koff[lo] <= S0 + i < koff[lo + 1]
                    while (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (koff[mid] <= S0 + i) lo = mid;
                        else hi = mid;
                    }
                    if (koff[lo + 1] - koff[lo] > SKM_FOLD_KEY_MAX) continue;
                    uint64_t w[SW];
#pragma unroll
                    for (int q = 0; q < SW; ++q) w[q] = wq[f][q];
                    const uint32_t c = (uint32_t)(w[SW - 1] >> 56), sets = (uint32_t)(w[SW - 1] >> 48) & 0xFFu;
                    w[SW - 1] &= ~(0xFFull << 48);  // what must be equal: the bases and the window count
                    uint32_t h = 0x9E3779B1u;
#pragma unroll
                    for (int q = 0; q < SW; ++q) h = (h ^ (uint32_t)w[q] ^ __builtin_rotateleft32((uint32_t)(w[q] >> 32), 13)) * 0x85EBCA6Bu;
                    h ^= h >> 15;
                    h *= 0xC2B2AE35u;
                    h ^= h >> 16;
                    const uint32_t ent = (h & 0xFFFF0000u) | i;
                    h &= SKM_FOLD_TAB - 1;
                    bool dup = false;
                    for (;;) {
                        const uint32_t o = atomicCAS(&ftab[h], 0xFFFFFFFFu, ent);
                        if (o == 0xFFFFFFFFu) break;
                        if ((o ^ ent) >> 16 == 0) {  // same tag: compare with the slot that sits there (L2 has it)
                            const uint64_t *po = slots + (S0 + (o & 0xFFFFu)) * SW;
                            bool eq = true;
#pragma unroll
                            for (int q = 0; q < SW; ++q) eq &= (q == SW - 1 ? (po[q] & ~(0xFFull << 48)) : po[q]) == w[q];
                            if (eq) {
                                if (sets) atomicOr((unsigned long long *)(po + SW - 1), (unsigned long long)sets << 48);
                                dup = true;
                                break;
                            }
                        }
                        h = (h + 1) & (SKM_FOLD_TAB - 1);
                    }
                    if (dup) {
                        slots[(S0 + i) * SW + SW - 1] = w[SW - 1] & ~(0xFFull << 56);  // no window of its own any more
                        nfold += c;
                    } else {
                        atomicAdd(&ksg[lo], (c + SEG - 1) / SEG);
                    }
                }
            }
            if (nfold) atomicAdd(&s_folded, (unsigned long long)nfold);
            __syncthreads();
            sg = n ? ksg[t] : 0u;
        }
        const bool skipped = skip_slots != ~0ull && n > skip_slots;
        const bool cut = !skipped && n && (!kseg || n > maxslots || sg > maxseg);
        if (skipped && pinfo) pinfo[(uint64_t)item * SKM_KEYS_PER_ITEM + t] = PM_DIRTY;
        if (cut && pinfo && !force_dirty) pinfo[(uint64_t)item * SKM_KEYS_PER_ITEM + t] = PM_DIRTY;
        const bool barrier = skipped || cut;
        {
            const unsigned long long bm = __ballot(barrier), cm = __ballot(cut);
            if ((t & 63u) == 0) {
                bar[t >> 6] = bm;
                cutm[t >> 6] = cm;
            }
        }
        const uint32_t wsg = barrier ? 0u : sg, wsl = barrier ? 0u : (uint32_t)n;
        uint32_t tot;
        const uint32_t es = block_excl_scan<uint32_t>(wsg, scr, &tot);
        pseg[t] = es;
        if (t == BLK - 1) pseg[BLK] = tot;
        const uint32_t el = block_excl_scan<uint32_t>(wsl, scr, &tot);
        pslot[t] = el;
        if (t == BLK - 1) pslot[BLK] = tot;
        __syncthreads();
        {  // end of the chunk that starts here: the largest e in (t, lim] with seg(t..e) <= maxseg and slots(t..e) <= maxslots, lim = next barrier
            uint32_t lim = SKM_KEYS_PER_ITEM;
#pragma unroll
            for (uint32_t w = 0; w < BLK / 64; ++w) {
                unsigned long long m = bar[w];
                if (w < (t >> 6)) m = 0;
                else if (w == (t >> 6)) m &= ~0ull << (t & 63u);
                if (m && lim == SKM_KEYS_PER_ITEM) lim = w * 64 + (uint32_t)__ffsll(m) - 1;
            }
            uint32_t lo = t, hi = lim;  // invariant: [t, lo) fits
            if (!barrier) {
                lo = t + 1;  // (a key that is no barrier fits on its own)
                while (lo < hi) {
                    const uint32_t mid = (lo + hi + 1) >> 1;
                    if (pseg[mid] - es <= maxseg && pslot[mid] - el <= maxslots) lo = mid;
                    else hi = mid - 1;
                }
            }
            nxt[t] = (uint16_t)lo;
        }
        __syncthreads();
        if (t == 0) {
            uint32_t nb = 0, ncl = 0, k = 0;
            while (k < SKM_KEYS_PER_ITEM) {
                if ((bar[k >> 6] >> (k & 63u)) & 1ull) {
                    ++k;
                    continue;
                }
                const uint32_t e = nxt[k], nsl = pslot[e] - pslot[k];
                if (nsl) {
                    SkmChunk c;
                    c.a = koff[k] | ((unsigned long long)nsl << 40) | ((unsigned long long)(force_dirty ? 1u : 0u) << 50) | ((unsigned long long)(pseg[e] - pseg[k]) << 51);
                    c.b = force_dirty ? 0ull : (((unsigned long long)item * SKM_KEYS_PER_ITEM + k) | ((unsigned long long)(e - k) << 32));
                    buf[nb++] = c;
                    ncl += force_dirty ? 0u : 1u;
                }
                k = e;
            }
            s_nbuf = nb;
            s_nclean += ncl;
        }
        flush();
        // cut keys: pieces of <= maxslots slots and <= maxseg segments, by the whole workgroup (two consecutive slots per thread)
        for (uint32_t w = 0; w < BLK / 64; ++w) {
            for (unsigned long long m = cutm[w]; m; m &= m - 1) {
                const uint32_t k = w * 64 + (uint32_t)__ffsll(m) - 1;
                unsigned long long pos = koff[k];
                const unsigned long long end = koff[k + 1];
                while (pos < end) {
                    const uint32_t lim = maxslots < 2u * BLK ? maxslots : 2u * BLK;
                    const uint32_t nst = end - pos < lim ? (uint32_t)(end - pos) : lim;
                    const uint32_t i0 = 2 * t, i1 = i0 + 1;
                    const uint32_t c0 = i0 < nst ? nseg_of(pos + i0) : 0, c1 = i1 < nst ? nseg_of(pos + i1) : 0;
                    uint32_t tot2;
                    const uint32_t ex = block_excl_scan<uint32_t>(c0 + c1, scr, &tot2);
                    cpre[i0] = ex;
                    cpre[i1] = ex + c0;
                    if (t == BLK - 1) cpre[2 * BLK] = tot2;
                    __syncthreads();
                    for (uint32_t q = t + 1; q <= nst; q += BLK)  // largest q in [1, nst] with cpre[q] <= maxseg
                        if (cpre[q] <= maxseg && (q == nst || cpre[q + 1] > maxseg)) s_nfit = q;
                    __syncthreads();
                    const uint32_t nfit = s_nfit;
                    if (t == 0) {
                        SkmChunk c;
                        c.a = pos | ((unsigned long long)nfit << 40) | (1ull << 50) | ((unsigned long long)cpre[nfit] << 51);
                        c.b = 0;
                        buf[s_nbuf++] = c;
                    }
                    pos += nfit;
                    __syncthreads();
                    if (s_nbuf == SKM_PLAN_BUF) flush();
                }
            }
        }
        flush();
    }
    __syncthreads();
    if (t == 0 && s_nclean) atomicAdd(&list_alloc[1], (unsigned long long)s_nclean);
    if (t == 0 && s_folded) atomicAdd(&list_alloc[3], s_folded);  // instances that were folded away (statistics)
}

template <int NW>
__device__ __forceinline__ Rec<NW> rec_roll_fw(const Rec<NW> &x, unsigned K, uint64_t b) {  // x[1..K-1] + b
    Rec<NW> r;
#pragma unroll
    for (int i = 0; i < NW; ++i) r.w[i] = (x.w[i] >> 2) | (i + 1 < NW ? x.w[i + 1] << 62 : 0ull);
    r.w[NW - 1] |= b << (((K - 1) & 31u) << 1);
    return r;
}
template <int NW>
__device__ __forceinline__ Rec<NW> rec_roll_rc(const Rec<NW> &y, unsigned K, uint64_t b) {  // RC(x[1..K-1] + b) from y = RC(x): (3 - b) + y[0..K-2]
    Rec<NW> r;
#pragma unroll
    for (int i = NW - 1; i >= 0; --i) r.w[i] = (y.w[i] << 2) | (i > 0 ? y.w[i - 1] >> 62 : (3ull - b));
    const unsigned tail = (K & 31u) << 1;
    if (tail) r.w[NW - 1] &= (1ull << tail) - 1;
    return r;
}

// LDS (dynamic): sl[scap * SW] u64 | tab[T] u32 | grp[NT] u32 | segl[NT] u16 | seglast[NT] u16 | nbv[scap -> x4] u8 | cl[scap] u8
// NT threads = segments per chunk at most; T = 16 * NT table slots (thread g owns occupancy group g); scap <= min(NT, 512) slots.
// MODE: 0 plain, 1 EXT, 2 EXT + partition-major output (see the head of this file), 3 partition-major output with PLAIN k-mer records ("nx": k without
// 8 spare bits in the last record word — the byte a winner gathered goes to the mask array alone, for the winners of cut partitions too).
template <int NW, int MODE, int NT>
__global__ void __launch_bounds__(NT, SMX_WPE) k_skm_dedupe2(const uint64_t *__restrict__ slots, const unsigned long long *__restrict__ slot_off, unsigned K,
                                                    const SkmChunk *__restrict__ chunks, uint32_t nlist, uint32_t scap, void *out_, unsigned long long out_cap,
                                                    unsigned long long clean_cap, unsigned long long dirty_cap, unsigned long long *out_count,
                                                    unsigned long long *dirty_count, unsigned long long *err, unsigned long long *prof, PmOut pm) {
    constexpr int SW = 2 * NW, SEG = SkmSeg<NW>::value, SB = SMX_SB;
    constexpr bool EXT = MODE >= 1, PM = MODE >= 2, NX = MODE == 3;
    constexpr uint32_t T = 16u * NT, EMPTY = 0xFFFFFFFFu;
    static_assert(SEG % SB == 0, "segments are worked through in sub-batches");
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    uint64_t *sl = lds64;
    uint32_t *tab = (uint32_t *)(sl + (size_t)scap * SW);
    uint32_t *grp = tab + T;
    uint16_t *segl = (uint16_t *)(grp + NT);
    uint16_t *seglast = segl + NT;
    uint8_t *nbv = (uint8_t *)(seglast + NT);  // per slot 8 bits: the bases seen before the run (bits 0-3, one bit per base) and behind it (4-7)
    uint8_t *cl = nbv + ((scap + 3u) & ~3u);
    __shared__ uint32_t scr[NT / 64 + 2];
    __shared__ uint32_t s_skip, s_cid, s_nhead;
    __shared__ unsigned long long s_gbase;
    Rec<NW> *out = (Rec<NW> *)out_;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    // staging area of the output (over the slots and the table, both dead by then): records | mask bytes
    const uint32_t stage_bytes = scap * SW * 8 + T * 4;
    __shared__ unsigned long long s_pt[10];  // SMX_DEBUG: 100 MHz ticks per phase seen by thread 0 (in LDS: registers are dear here)
    unsigned npal = 0;
    unsigned long long tbits = 0;  // extension bits of the winners whose node entries this thread wrote (fused node table)
#define SKM_T(i)                                       \
    if (prof && t == 0) {                              \
        const unsigned long long t1 = wall_clock64();  \
        s_pt[i] += t1 - s_pt[6];                       \
        s_pt[6] = t1;                                  \
    }
    if (prof && t == 0) {
        for (int i = 0; i < 10; ++i) s_pt[i] = 0;
        s_pt[6] = wall_clock64();
    }
    const SkmChunk none{0, 0};
    uint32_t ci = blockIdx.x;
    SkmChunk d1 = ci < nlist ? chunks[ci] : none;
    SkmChunk d2 = (uint64_t)ci + gridDim.x < nlist ? chunks[ci + gridDim.x] : none;
    uint64_t pf[SW];
    auto fetch_slots = [&](const SkmChunk &c) {
        const uint32_t nt = (uint32_t)(c.a >> 40) & 0x3FFu;
        if (t < nt) {
            const uint64_t *p = slots + ((c.a & SKM_M40) + t) * SW;
#pragma unroll
            for (int i = 0; i < SW; ++i) pf[i] = ld_pol<7>(p + i);
        }
    };
    fetch_slots(d1);
    for (; ci < nlist; ci += gridDim.x) {
        const SkmChunk cur = d1;
        const uint32_t ntake = (uint32_t)(cur.a >> 40) & 0x3FFu;
        const bool dirty = (cur.a >> 50) & 1u;
        uint32_t nseg = (uint32_t)(cur.a >> 51) & 0xFFFu;  // (the plan's figure; checked against the slots below)
        uint32_t my_c = 0;
        if (t < ntake) {
            const uint64_t last = pf[SW - 1];
            my_c = (uint32_t)(last >> 56);
            cl[t] = (uint8_t)my_c;
            nbv[t] = (uint8_t)(last >> 48);  // the bases seen next to the run (bits 0-3 before it, 4-7 behind it; complete after the fold)
#pragma unroll
            for (int i = 0; i < SW; ++i) sl[(size_t)t * SW + i] = i == SW - 1 ? (last & ~(0xFFFFull << 48)) : pf[i];
        }
        {
            const uint4 e4 = make_uint4(EMPTY, EMPTY, EMPTY, EMPTY);
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) *(uint4 *)&tab[(i * NT + t) * 4] = e4;
        }
        // the chunk after this one: its slots start to arrive now, its descriptor was fetched a chunk ago
        d1 = d2;
        fetch_slots(d1);
        d2 = (uint64_t)ci + 2ull * gridDim.x < nlist ? chunks[ci + 2 * gridDim.x] : none;
        if (!ntake) continue;  // a hole of the list
        {  // segment list: (slot << 5) | piece
            const uint32_t ns = (my_c + SEG - 1) / SEG;
            uint32_t tot;
            const uint32_t ex = block_excl_scan_lds<uint32_t>(ns, scr, &tot);
            if (tot != nseg || tot > NT) {  // never expected: the plan and the slots disagree — no k-mer may be dropped silently
                if (t == 0) atomicOr(err, 1ull);
            }
            nseg = tot < (uint32_t)NT ? tot : (uint32_t)NT;
            for (uint32_t q = 0; q < ns; ++q)
                if (ex + q < (uint32_t)NT) segl[ex + q] = (uint16_t)((t << 5) | q);
        }
        lds_barrier();
        SKM_T(0)
        // ---- inserts: one segment per thread ----
        // per instance 16 bits: table slot (14) | stored orientation is the canonical one << 14 | won << 15; instance j at bits 16 j of
        // (hp0, hp1). The loops over the instances of a segment are NOT unrolled: unrolled, the compiler rolls all SEG k-mers ahead of
        // their use (64 VGPRs) and the body with the palindrome test is large.
        uint64_t hp0 = 0, hp1 = 0;
        uint32_t seg_n = 0, lastv = 0;
        if (t < nseg) {
            const uint32_t sd = segl[t], s = sd >> 5, q = sd & 31u;
            const uint32_t c = cl[s], nb = nbv[s], j0 = q * SEG;  // (nb: the neighbour sets, complete after the fold)
            const uint32_t n = min((uint32_t)SEG, c - j0);
            seg_n = n;
            const uint64_t *ss = sl + (size_t)s * SW;
            Rec<NW> x = skm_extract<NW>(ss, j0, K);
            Rec<NW> y = rec_rc<NW>(x, K);
            uint32_t nxt;  // the SEG bases behind the first k-mer
            {
                const uint32_t p = j0 + K, sh = (p & 31u) << 1;
                uint64_t v = ss[p >> 5] >> sh;
                if (sh) v |= ss[(p >> 5) + 1] << (64 - sh);
                nxt = (uint32_t)v & 0xFFFFu;
            }
            // bases seen before the first instance (one bit per base; inside the slot: the base there)
            uint32_t Lset = j0 > 0 ? 1u << ((uint32_t)(ss[(j0 - 1) >> 5] >> (((j0 - 1) & 31u) << 1)) & 3u) : (nb & 15u);
#pragma unroll 1
            for (uint32_t sb = 0; sb < n; sb += SB) {
                {
                    uint64_t hw = 0;
                    Rec<NW> cxs[SB], oth[SB];  // canonical strand, other strand (a copy in a slot may be either)
                    uint32_t hs[SB], ent[SB], old[SB], ebs[SB];
                    bool fw[SB];
#pragma unroll
                    for (int u = 0; u < SB; ++u) {
                        const uint32_t j = sb + (uint32_t)u;
                        const bool fwd = rc_ge<NW>(y, x);
                        fw[u] = fwd;
                        Rec<NW> cx;
#pragma unroll
                        for (int i = 0; i < NW; ++i) {
                            cx.w[i] = fwd ? x.w[i] : y.w[i];
                            oth[u].w[i] = fwd ? y.w[i] : x.w[i];
                        }
                        cxs[u] = cx;
                        const uint32_t hh = rec_hash32<NW>(cx);
                        hs[u] = hh & (T - 1);
                        const uint32_t fp = EXT ? (((hh >> 24) == 0xFFu) ? 0xFEu : (hh >> 24)) : (hh >> 16);
                        const uint32_t ref = (s << 7) | (j0 + j);
                        ent[u] = EXT ? ((fp << 24) | ref) : ((fp << 16) | ref);
                        const uint32_t b = nxt & 3u;
                        nxt >>= 2;
                        if constexpr (EXT) {
                            // InOutMask bits in the canonical frame: out bits 0-3 by next base, in bits 4-7 by previous base; on the other
                            // strand the roles swap and the bases are complemented (bit b -> bit 3 - b: a 4-bit reversal)
                            const uint32_t Rset = j0 + j + 1 < c ? 1u << b : (nb >> 4);
                            // (the other strand's byte is the 8-bit reversal of this strand's: one v_bfrev + shift + select, no branch)
                            const uint32_t e8 = Rset | (Lset << 4);
                            ebs[u] = fwd ? e8 : (__brev(e8) >> 24);
                            Lset = 1u << ((uint32_t)x.w[0] & 3u);
                        }
                        x = rec_roll_fw<NW>(x, K, (uint64_t)b);
                        y = rec_roll_rc<NW>(y, K, (uint64_t)b);
                    }
#pragma unroll
                    for (int u = 0; u < SB; ++u)
                        old[u] = sb + (uint32_t)u < n ? atomicCAS(&tab[hs[u]], EMPTY, ent[u]) : 0u;
#pragma unroll
                    for (int u = 0; u < SB; ++u) {
                        if (sb + (uint32_t)u < n) {
                            uint32_t h = hs[u], o = old[u];
                            bool won = false;
                            for (;;) {
                                if (o == EMPTY) {
                                    won = true;
                                    break;
                                }
                                if ((EXT ? (o >> 24) : (o >> 16)) == (EXT ? (ent[u] >> 24) : (ent[u] >> 16))) {
                                    const Rec<NW> xo = skm_extract<NW>(sl + (size_t)((o & 0xFFFFu) >> 7) * SW, o & 127u, K);
                                    if (rec_eq<NW>(xo, cxs[u]) || rec_eq<NW>(xo, oth[u])) break;  // the canonical k-mer is there already
                                }
                                h = (h + 1) & (T - 1);
                                o = atomicCAS(&tab[h], EMPTY, ent[u]);
                            }
                            if constexpr (EXT)
                                if (ebs[u]) atomicOr(&tab[h], ebs[u] << 16);
                            const uint32_t v = h | (fw[u] ? 0x4000u : 0u) | (won ? 0x8000u : 0u);
                            hw |= (uint64_t)v << (16 * u);
                            lastv = v;
                        }
                    }
                    static_assert(4 % SB == 0, "a sub-batch's codes stay inside one 64-bit word");
                    if (sb < 4) hp0 |= hw << (16 * sb);
                    else hp1 |= hw << (16 * (sb - 4));
                }
            }
            if constexpr (PM) seglast[t] = (uint16_t)(lastv & 0x7FFFu);
        }
        lds_barrier();
        SKM_T(1)
        // ---- occupancy: thread g owns group g (16 table slots); a wave reads 64 slots at a time and ballots ----
        uint32_t occ = 0;
#pragma unroll 4
        for (uint32_t it = 0; it < 16; ++it) {
            const unsigned long long b = __ballot(tab[wave * 1024u + it * 64u + lane] != EMPTY);
            if ((lane >> 2) == it) occ = (uint32_t)(b >> ((lane & 3u) << 4)) & 0xFFFFu;
        }
        uint32_t wcount;
        const uint32_t pre = block_excl_scan_lds<uint32_t>(__popc(occ), scr, &wcount);
        grp[t] = pre | (occ << 16);
        // The chunk's place in the output: ONE counter serves every chunk of the launch (~88 atomics per microsecond on one address: at 4.6 M
        // chunks the answer takes ~3 us to come back; round-5 phase ticks: "occupancy + allocation" 17 % of a chunk's time, every wave of the
        // workgroup waiting for thread 0 at the next barrier). Nothing before the copy-out needs the answer — the winners are staged in LDS
        // by rank first — so thread 0 ISSUES the atomic at the top of the first staging round (issue_place) and publishes what it returned
        // (publish_place) behind that round's loop, just before the barrier in front of the copy-out: the staging runs while the atomic is
        // on its way. (Between the two there is no other vector-memory instruction — not even a register reload from scratch: any of those
        // would make the wave wait for the atomic first, vmcnt counts in order.)
        unsigned long long place_raw = 0;
        auto issue_place = [&]() {  // thread 0, once per chunk with winners
            // (The counter's address goes through an empty asm statement: a pointer the compiler knows to be the same in every lane makes its
            // atomic optimizer wrap the atomic in a wave reduction + v_readfirstlane of the result, i.e. an s_waitcnt right behind it — the
            // very wait this is about. A "divergent" pointer is left alone, and the wait lands at the first use, in publish_place.)
            unsigned long long *ctr = dirty ? dirty_count : out_count;
            asm volatile("" : "+v"(ctr));
            place_raw = atomicAdd(ctr, PM && !dirty ? ((unsigned long long)wcount | (1ull << PM_BASE_BITS)) : (unsigned long long)wcount);
        };
        auto publish_place = [&]() {  // thread 0, once per chunk
            asm volatile("" : "+v"(place_raw));  // (the value is READ here and not before: none of what follows may be hoisted in front of the staging loop)
            s_skip = 0;
            if (!wcount) s_gbase = 0;
            else if (!dirty) {
                if constexpr (PM) {
                    s_gbase = place_raw & PM_BASE_MASK;
                    s_cid = (uint32_t)(place_raw >> PM_BASE_BITS);
                    s_skip = s_gbase + wcount > clean_cap;
                    if (s_cid >= pm.max_chunks) {
                        s_skip = 1;
                        *pm.overflow = 1;
                    }
                } else {
                    s_gbase = place_raw;
                    s_skip = s_gbase + wcount > clean_cap;
                }
            } else {
                s_skip = place_raw + wcount > dirty_cap;
                s_gbase = s_skip ? 0 : out_cap - place_raw - wcount;
            }
        };
        // The winners' bytes leave the table, and the first k-mer of the segment is taken from the slot once more, before the staging area
        // overwrites both: the winners are rolled out of it again below (holding the segment's canonical k-mers in registers across the
        // barriers instead cost 60 VGPRs, i.e. half the waves).
        uint64_t ebq = 0;  // 8 bits per instance
        Rec<NW> x0, y0;
#pragma unroll
        for (int i = 0; i < NW; ++i) x0.w[i] = y0.w[i] = 0;
        uint32_t nxt0 = 0;
        if (t < nseg) {
            const uint32_t sd = segl[t], s = sd >> 5, j0 = (sd & 31u) * SEG;
            const uint64_t *ss = sl + (size_t)s * SW;
            x0 = skm_extract<NW>(ss, j0, K);
            y0 = rec_rc<NW>(x0, K);
            const uint32_t p = j0 + K, sh = (p & 31u) << 1;
            uint64_t v = ss[p >> 5] >> sh;
            if (sh) v |= ss[(p >> 5) + 1] << (64 - sh);
            nxt0 = (uint32_t)v & 0xFFFFu;
            if constexpr (EXT) {
                uint64_t a0 = hp0, a1 = hp1;
#pragma unroll 1
                for (uint32_t j = 0; j < seg_n; ++j) {
                    const uint32_t hv = (uint32_t)a0 & 0xFFFFu;
                    a0 = (a0 >> 16) | (a1 << 48);
                    a1 >>= 16;
                    if (hv & 0x8000u) ebq |= (uint64_t)((tab[hv & 0x3FFFu] >> 16) & 0xFFu) << (8 * j);
                }
            }
        }
        lds_barrier();  // (grp[] is complete: rank_of below)
        SKM_T(2)
        auto rank_of = [&](uint32_t h) -> uint32_t {
            const uint32_t g = grp[h >> 4];
            return (g & 0xFFFFu) + __popc((g >> 16) & ((1u << (h & 15u)) - 1u));
        };
        // Staging area (over the slots and the table, both dead now): [link words, 4 B per winner (PM)] | records | mask bytes (PM).
        // Local links: instances j, j+1 of one super-k-mer are k-mers side by side in a read, i.e. a de Bruijn edge between the nodes of
        // their table entries (and the reverse one between the other strands). A node with ONE outgoing extension has one successor,
        // whichever read shows it; where several reads disagree the node has several extensions and nobody reads the link. A thread
        // knows the pairs inside its segment, and the pair across the boundary to the segment before it from that segment's last entry.
        bool links = false, fuse = false;
        uint32_t lnk_bytes = 0, mk_bytes = 0;
        if constexpr (PM) {
            links = !dirty && wcount && (size_t)wcount * 8 <= (size_t)stage_bytes;  // (a chunk that does not fit the output any more — skip, known at the copy-out — stages its links for nothing)
            lnk_bytes = links ? ((wcount * 4 + 15u) & ~15u) : 0u;
            // the node table of the chunk is written from here (PmOut::tab): the bytes of ALL its winners stay in LDS behind the links (mk), whatever the
            // number of output rounds; the rounds stage their records behind them
            fuse = links && pm.tab != nullptr;
            mk_bytes = fuse ? ((wcount + 15u) & ~15u) + 8u * (T / 32) : 0u;  // ... and two bitmaps of the chunk's nodes (hp, rm: see the node table below)
        }
        constexpr uint32_t WPC = T / 32;  // words of a node bitmap = words of rbits per chunk (= maxn / 16, smx_pm.hpp)
        uint16_t *lnk = (uint16_t *)lds64;
        uint8_t *mk = (uint8_t *)lds64 + lnk_bytes;
        uint32_t *hp = (uint32_t *)(mk + ((wcount + 15u) & ~15u));  // bit nd: some local link leads to node nd
        uint32_t *rm = hp + WPC;                                     // bit nd: the successor of node nd is not in this chunk
        Rec<NW> *stg = (Rec<NW> *)((uint8_t *)lds64 + lnk_bytes + mk_bytes);
        const uint32_t R = fuse ? (((stage_bytes - lnk_bytes - mk_bytes) / (uint32_t)sizeof(Rec<NW>)) & ~15u)
                                : (((stage_bytes - lnk_bytes) / (uint32_t)(sizeof(Rec<NW>) + 1)) & ~15u);  // records per output round
        if constexpr (PM) {
            if (links) {
                for (uint32_t i = t; i < lnk_bytes / 16; i += NT) ((uint4 *)lnk)[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
                if (fuse) {
                    for (uint32_t i = t; i < 2 * WPC; i += NT) hp[i] = 0;
                    if (t == 0) s_nhead = 0;
                }
                lds_barrier();
            }
        }
        // ---- winners: to their place in table-slot order, staged through LDS in rounds of R records; links in the first round ----
        for (uint32_t r0 = 0; r0 < wcount; r0 += R) {
            if (r0) lds_barrier();
            uint8_t *stm = fuse ? mk + r0 : (uint8_t *)(stg + R);  // the bytes of this round's records
            Rec<NW> x = x0, y = y0;
            uint32_t nxt = nxt0;
            uint64_t a0 = hp0, a1 = hp1, e0 = ebq;
            uint32_t prev = 0xFFFFFFFFu;  // node of the instance before (2 * rank + orientation)
            if constexpr (PM) {
                if (links && r0 == 0 && t < nseg && (segl[t] & 31u) != 0) {
                    const uint32_t v = seglast[t - 1];
                    prev = 2 * rank_of(v & 0x3FFFu) + ((v & 0x4000u) ? 0u : 1u);
                }
            }
            if (r0 == 0 && t == 0) issue_place();  // (wcount > 0 in here)
#pragma unroll 1
            for (uint32_t j = 0; j < seg_n; ++j) {
                const uint32_t hv = (uint32_t)a0 & 0xFFFFu;
                a0 = (a0 >> 16) | (a1 << 48);
                a1 >>= 16;
                const uint32_t eb = (uint32_t)e0 & 0xFFu;
                e0 >>= 8;
                const bool fwd = hv & 0x4000u;
                uint32_t r = 0xFFFFFFFFu;  // (losers and other rounds: out of range)
                if (PM ? (links && r0 == 0) || (hv & 0x8000u) : (hv & 0x8000u) != 0) {
                    const uint32_t rk = rank_of(hv & 0x3FFFu);
                    if (hv & 0x8000u) r = rk - r0;
                    if constexpr (PM) {
                        if (links && r0 == 0) {
                            const uint32_t node = 2 * rk + (fwd ? 0u : 1u);
                            if (prev != 0xFFFFFFFFu) {
                                lnk[prev] = (uint16_t)node;
                                lnk[node ^ 1u] = (uint16_t)(prev ^ 1u);
                            }
                            prev = node;
                        }
                    }
                }
                if (r < R) {
                    Rec<NW> cx;
#pragma unroll
                    for (int i = 0; i < NW; ++i) cx.w[i] = fwd ? x.w[i] : y.w[i];
                    if constexpr (NX) {
                        if (dirty) stm[r] = (uint8_t)eb;  // (nx: the byte of a cut partition's winner travels in the mask array as well)
                    }
                    if constexpr (PM) {
                        if (!dirty) {
                            stm[r] = (uint8_t)eb;
                            // Palindromic (k+1)-mers among the extensions (k_ext_split's second figure): cx + c is its own
                            // reverse complement iff c complements cx[0] and cx[1..K-1] = RC(cx)[0..K-2]; b + cx likewise with cx[0..K-2] =
                            // RC(cx)[1..K-1]. The other strand is at hand; the first 15 bases of either equation (30 bits, 4^-15 by chance)
                            // are compared first, so that a wave all but never enters the full test.
                            const uint32_t rlo = (uint32_t)(fwd ? y.w[0] : x.w[0]), rlo2 = (uint32_t)((fwd ? y.w[0] : x.w[0]) >> 2);
                            const uint32_t clo = (uint32_t)cx.w[0], clo2 = (uint32_t)(cx.w[0] >> 2);
                            if (((clo2 ^ rlo) & 0x3FFFFFFFu) == 0 || ((clo ^ rlo2) & 0x3FFFFFFFu) == 0) {
                                const unsigned c0 = (unsigned)cx.w[0] & 3u, cl_ = (unsigned)(cx.w[NW - 1] >> (((K - 1) & 31u) << 1)) & 3u;
                                Rec<NW> rx;
#pragma unroll
                                for (int i = 0; i < NW; ++i) rx.w[i] = fwd ? y.w[i] : x.w[i];
                                Rec<NW> xsf, rsf, xp = cx, rp = rx;  // suffixes (drop base 0) and prefixes (drop base K-1)
#pragma unroll
                                for (int i = 0; i < NW; ++i) {
                                    xsf.w[i] = (cx.w[i] >> 2) | (i + 1 < NW ? cx.w[i + 1] << 62 : 0ull);
                                    rsf.w[i] = (rx.w[i] >> 2) | (i + 1 < NW ? rx.w[i + 1] << 62 : 0ull);
                                }
                                const uint64_t topm = ~(3ull << (((K - 1) & 31u) << 1));
                                xp.w[NW - 1] &= topm;
                                rp.w[NW - 1] &= topm;
                                if (((eb >> (3 - c0)) & 1) && rec_eq<NW>(xsf, rp)) ++npal;
                                if (((eb >> (7 - cl_)) & 1) && rec_eq<NW>(xp, rsf)) ++npal;
                            }
                        }
                    }
                    if constexpr (EXT && !NX) cx.w[NW - 1] = (cx.w[NW - 1] << EXT_BITS) | eb;
                    stg[r] = cx;
                }
                {
                    const uint64_t b = nxt & 3u;
                    nxt >>= 2;
                    x = rec_roll_fw<NW>(x, K, b);
                    y = rec_roll_rc<NW>(y, K, b);
                }
            }
            if (r0 == 0 && t == 0) publish_place();
            lds_barrier();
            const unsigned long long gb = s_gbase;
            const bool skip = s_skip != 0;
            if constexpr (PM) {
                if (r0 == 0 && !dirty && !skip) {  // (wcount > 0 in here) the chunk's group words, its descriptor, and its partitions' words
                    const uint32_t cid = s_cid;
                    pm.meta[(size_t)cid * NT + t] = pre | (occ << 16);
                    if (t == 0) pm.cinfo[cid] = gb | ((unsigned long long)wcount << PM_BASE_BITS);
                    const uint32_t nkeys = (uint32_t)(cur.b >> 32), k0 = (uint32_t)cur.b;
                    for (uint32_t k = t; k < nkeys; k += NT)  // the partitions that lie in this chunk
                        if (slot_off[(uint64_t)k0 + k + 1] > slot_off[(uint64_t)k0 + k]) pm.pinfo[(uint64_t)k0 + k] = gb | ((unsigned long long)cid << PM_BASE_BITS);
                }
            }
            if (!skip) {
                const uint32_t nr = min(R, wcount - r0);
                Rec<NW> *dst = out + gb + r0;
                for (uint32_t i = t; i < nr; i += NT) {
                    const Rec<NW> v = stg[i];
#pragma unroll
                    for (int j = 0; j < NW; ++j) st_pol<6>(&dst[i].w[j], v.w[j]);
                }
                if constexpr (PM) {
                    if (!dirty || NX) {
                        uint8_t *md = pm.mask + gb + r0;
                        for (uint32_t i = t; i < nr; i += NT) st_pol<6>(md + i, stm[i]);
                    }
                    if (links && r0 == 0 && pm.llink != nullptr) {  // (no link array where this stage writes the node table itself)
                        uint32_t *gl = pm.llink + gb;
                        for (uint32_t i = t; i < wcount; i += NT) gl[i] = ((const uint32_t *)lnk)[i];
                    }
                    // ---- the chunk's node table (k_pm_tab of smx_pm.hip, from LDS), first half, beside the copy-out of the LAST round (every winner's byte is in mk by
                    // now, the links since the first round): per node its outgoing extensions and, where there is exactly one, the successor by the local link or
                    // TAB_NODE_MASK + its bit in rm (k_pm_remote looks it up). The links become the chain links in place (0xFFFF: the chain ends here).
                    if (fuse && r0 + R >= wcount) {
                        const uint32_t nn = 2 * wcount;
                        uint32_t *lnk32 = (uint32_t *)lnk;
                        for (uint32_t r = t; r < wcount; r += NT) {
                            const unsigned m = mk[r];
                            const uint32_t ll = lnk32[r];
                            tbits += __popc(m);
                            const bool junction = mask_junction(m);
                            node_t e[2];
                            uint32_t w2 = 0;
#pragma unroll
                            for (unsigned o = 0; o < 2; ++o) {
                                const unsigned mo = (o ? brev8(m) : m) & 15u;
                                const uint32_t l = o ? (ll >> 16) : (ll & 0xFFFFu);
                                uint32_t w = 0xFFFFu;  // no local successor
                                e[o] = (node_t)mo << TAB_OUT_SHIFT;
                                if (uniq4(mo)) {
                                    if (l != 0xFFFFu && l < nn) {
                                        e[o] |= 2 * gb + l;
                                        if (!junction) {
                                            w = l;
                                            atomicOr(&hp[w >> 5], 1u << (w & 31u));
                                        }
                                    } else {  // not next to it in any super-k-mer of the chunk
                                        e[o] |= TAB_NODE_MASK;
                                        atomicOr(&rm[(2 * r + o) >> 5], 1u << ((2 * r + o) & 31u));
                                    }
                                }
                                w2 |= w << (16 * o);
                            }
                            lnk32[r] = w2;
                            __builtin_nontemporal_store(smx_ull2{e[0], e[1]}, reinterpret_cast<smx_ull2 *>(pm.tab + 2 * (gb + r)));
                        }
                    }
                }
            }
        }
        lds_barrier();
        SKM_T(3)
        if constexpr (PM) {
            // ---- the chunk's node table, second half: the chains inside the chunk, once each from their heads (the nodes no local link leads to): jump words.
            // LDS over the staged records (all copied out): the head list. (Tried and dropped, round 6: pointer jumping over all nodes instead of one thread per
            // chain — to convergence, ~9 rounds of a dense pass + barrier: 13 us per chunk in here where this takes 5; 2 / 3 / 4 rounds in front of the walks: each
            // round costs ~0.7 us and the walks get no shorter on the clock — profiles/r06/fused_node_table_pointer_jumping_ab.log, ..._chain_doubling_ab.log. Nor did
            // it matter that the first half moved beside the copy-out, two barriers less: a chunk's time in here is its instructions and LDS round trips.)
            if (fuse && !s_skip) {
                const unsigned long long gb = s_gbase;
                const uint32_t cid = s_cid, nn = 2 * wcount;
                uint16_t *list = (uint16_t *)stg;  // (two nodes may share a successor: up to nn heads)
                for (uint32_t i = t; i < WPC; i += NT) pm.rbits[(size_t)cid * WPC + i] = rm[i];
                for (uint32_t nd = t; nd < nn; nd += NT) {  // chain heads: a local successor, no local predecessor; every other node's jump word is 0
                    if (lnk[nd] != 0xFFFFu && !((hp[nd >> 5] >> (nd & 31u)) & 1u)) list[atomicAdd(&s_nhead, 1u)] = (uint16_t)nd;
                    else __builtin_nontemporal_store(0u, pm.jmp + 2 * gb + nd);
                }
                lds_barrier();
                SKM_T(7)
                const uint32_t nhead = s_nhead;
                for (uint32_t i = t; i < nhead; i += NT) {  // every chain once, from its head
                    const uint32_t h = list[i];
                    uint32_t cur = h, st = 0;
                    for (uint32_t a; (a = lnk[cur]) != 0xFFFFu && st < nn; ++st) cur = a;
                    // (bit 31: the chain's last node is a junction — k_pm_walk_len then ends its path there without reading that node's entry; smx_pm.hip)
                    __builtin_nontemporal_store(((cur - h) & 0xFFFFu) | (st << 16) | (mask_junction(mk[cur >> 1]) ? 0x80000000u : 0u), pm.jmp + 2 * gb + h);
                }
                lds_barrier();  // (the next chunk's slots and table go over all of this)
            }
            SKM_T(8)
        }
        if (prof && t == 0) {
            s_pt[4] += 1;
            s_pt[5] += ntake;
        }
    }
    if (prof && t == 0)
    {
        for (int i = 0; i < 6; ++i) atomicAdd(&prof[i], s_pt[i]);
        atomicAdd(&prof[7], s_pt[7]);
        atomicAdd(&prof[8], s_pt[8]);
    }
    if constexpr (PM) {
        if (npal) atomicAdd(pm.pals, (unsigned long long)npal);
        if (pm.tab != nullptr) {
            for (int o = 32; o > 0; o >>= 1) tbits += __shfl_down(tbits, o, 64);
            if (lane == 0 && tbits) atomicAdd(pm.tab_stats, tbits);
        }
    }
#undef SKM_T
}

// nx (MODE 3): the winners of cut partitions left the stage as plain k-mers with their bytes beside them; the k-mers have been sorted and made
// unique since (bucket-major in DB hash buckets, `sorted`), and every copy's byte has to reach its k-mer: bucket of the copy, binary search in the
// bucket's slice, atomic OR into a word array (4 bytes per word; cleared by the caller).
template <int NW>
__global__ void __launch_bounds__(BLK) k_nx_dirty_masks(const void *copies_, const uint8_t *bytes, uint64_t n, const void *sorted_, const unsigned long long *boff, uint32_t DB,
                                                        uint32_t *words, uint32_t *err) {
    const Rec<NW> *copies = (const Rec<NW> *)copies_, *sorted = (const Rec<NW> *)sorted_;
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLK) {
        const Rec<NW> x = copies[i];
        const uint32_t b = bucket_of(xxh3_rec<NW>(x), DB);
        uint64_t lo = boff[b], hi = boff[b + 1];
        while (lo < hi) {  // first record >= x
            const uint64_t mid = (lo + hi) >> 1;
            if (rec_less<NW>(sorted[mid], x)) lo = mid + 1;
            else hi = mid;
        }
        if (lo >= boff[b + 1] || !rec_eq<NW>(sorted[lo], x)) {
            atomicAdd(err, 1u);
            continue;
        }
        atomicOr(&words[lo >> 2], (uint32_t)bytes[i] << (8u * (unsigned)(lo & 3u)));
    }
}

}  // namespace smx
