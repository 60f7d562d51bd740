// spades_amd/csrc/smx_spill_split.hpp — host-only (no HIP in here: tests/test_spill_split_cpu.py compiles it with g++).
//
// Out-of-core merge, the case the bucket ranges cannot solve: ONE bucket whose slices of the spilled runs together exceed what the
// HBM budget can merge at once (spades-kmercount has B = 16 fixed, kmercount.cpp:220 — "use more buckets" is not an option there).
// The reference merges a bucket's runs with a loser tree in one streaming pass (kmer_index_builder.hpp:346-430); here the device
// merges (concatenate + one pass of the count pipeline), so the bucket is cut by KEY RANGE: every run's slice of the bucket is
// sorted-unique in record order (word 0 most significant, adt/array_vector.hpp:332-350), a splitter key cuts all of them by binary
// search, the copies of one key (at most one per run) always fall into the same part, parts are disjoint key intervals in ascending
// order — so the merged parts, one after the other, ARE the bucket's sorted-unique set.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace smx_split {

struct Slice {
    const char *p;  // n records of nw words each, strictly increasing
    uint64_t n;
};

inline int rec_cmp(const uint64_t *a, const uint64_t *b, unsigned nw) {
    for (unsigned i = 0; i < nw; ++i)
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
}

inline const uint64_t *rec_at(const Slice &s, uint64_t i, unsigned nw) { return (const uint64_t *)(s.p + i * (uint64_t)nw * 8); }

// first position in [lo, hi) of s whose record is >= key (upper = false) or > key (upper = true)
inline uint64_t bound(const Slice &s, uint64_t lo, uint64_t hi, const uint64_t *key, unsigned nw, bool upper) {
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2;
        const int c = rec_cmp(rec_at(s, mid, nw), key, nw);
        if (c < 0 || (upper && c == 0)) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// Cut positions of the parts: cuts[p][r] = first record of run r that belongs to part p (cuts.front() = zeros, cuts.back() = the
// slice lengths; part p = [cuts[p][r], cuts[p + 1][r]) of every run r). Every part holds at most max(max_part, #runs) records (a
// part cannot be smaller than the copies of one key). Bisection at the median of the largest sub-slice down to a quarter of max_part,
// then neighbouring leaves are joined while they fit: parts come out 3/4 full or better, whatever the key distribution.
inline std::vector<std::vector<uint64_t>> plan(const std::vector<Slice> &runs, unsigned nw, uint64_t max_part) {
    const size_t R = runs.size();
    std::vector<std::vector<uint64_t>> leaves;  // boundaries in ascending key order
    std::vector<uint64_t> lo(R, 0), hi(R);
    for (size_t r = 0; r < R; ++r) hi[r] = runs[r].n;
    leaves.push_back(lo);
    if (max_part < 1) max_part = 1;
    const uint64_t leaf_max = max_part >= 4 ? max_part / 4 : 1;
    // explicit stack of intervals still to be cut, processed in key order
    struct Iv {
        std::vector<uint64_t> lo, hi;
    };
    std::vector<Iv> stack;
    stack.push_back({lo, hi});
    while (!stack.empty()) {
        Iv iv = std::move(stack.back());
        stack.pop_back();
        uint64_t tot = 0, big = 0;
        size_t rbig = 0;
        for (size_t r = 0; r < R; ++r) {
            const uint64_t n = iv.hi[r] - iv.lo[r];
            tot += n;
            if (n > big) {
                big = n;
                rbig = r;
            }
        }
        if (tot <= leaf_max || big <= 1) {  // small enough, or at most one record per run left (tot <= #runs)
            leaves.push_back(iv.hi);
            continue;
        }
        // pivot: the median record of the largest sub-slice (big >= 2: its lower bound there is lo + big / 2, so either side is non-empty)
        const uint64_t *key = rec_at(runs[rbig], iv.lo[rbig] + big / 2, nw);
        std::vector<uint64_t> cut(R);
        uint64_t left = 0;
        for (size_t r = 0; r < R; ++r) {
            cut[r] = bound(runs[r], iv.lo[r], iv.hi[r], key, nw, false);
            left += cut[r] - iv.lo[r];
        }
        if (left == 0 || left == tot) {  // (big >= 2 puts at least one record on either side; kept as a guard against unsorted input)
            leaves.push_back(iv.hi);
            continue;
        }
        stack.push_back({cut, iv.hi});  // right half after the left one
        stack.push_back({iv.lo, cut});
    }
    // join consecutive leaves while they fit
    std::vector<std::vector<uint64_t>> cuts;
    cuts.push_back(leaves.front());
    uint64_t cur = 0;
    for (size_t i = 1; i < leaves.size(); ++i) {
        uint64_t n = 0;
        for (size_t r = 0; r < R; ++r) n += leaves[i][r] - leaves[i - 1][r];
        if (cur && cur + n > max_part) {
            cuts.push_back(leaves[i - 1]);
            cur = 0;
        }
        cur += n;
    }
    if (cuts.back() != leaves.back()) cuts.push_back(leaves.back());
    if (cuts.size() == 1) cuts.push_back(leaves.back());  // an empty bucket: one empty part
    return cuts;
}

}  // namespace smx_split
