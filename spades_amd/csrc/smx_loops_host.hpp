// spades_amd/csrc/smx_loops_host.hpp — perfect loops on the host, on PACKED k-mers and on all cores (host-only code: no HIP in here,
// tests/test_loops_host_cpu.py compiles it with g++ next to a string-level restatement of the reference and compares them).
//
// Reference: CollectLoops / FindMinimalKMerInLoop / ConstructLoopFromVertex / SplitLoop,
// /root/reference/src/common/assembly_graph/construction/debruijn_graph_constructor.hpp:252-293,359-397. The reference visits the k-mers
// in index order; the first non-junction k-mer that no unitig and no earlier loop holds starts a loop: the loop is rotated to its minimal
// k-mer (over both strands, RtSeq operator<: nucleotide-lexicographic), written from there until the first edge comes back, split at its
// first palindromic (k+1)-mer if it has one, and every part is emitted as max(part, RC(part)).
//
// Rounds 1-3 did this with std::string k-mers in an unordered_map, one loop after the other: ~6 us per loop k-mer, minutes for a
// metagenome with 10 000 plasmids (50 M loop k-mers). Here: the k-mers stay 2-bit packed; an open-addressing index and the successor
// of every oriented k-mer are made by all threads; ONE thread follows the successor array in file order to find the cycles (array
// chasing only: the order of the loops is the order in which the reference meets them); the loops themselves — rotation, sequence,
// palindrome split, orientation — are then independent and go to all threads again.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace smxl {

struct PackedLoop {
    std::vector<uint64_t> words;  // 2 bits per nucleotide, nucleotide t at bits 2 (t mod 32) of word t / 32 (the unitig layout of the graph)
    uint64_t len = 0;             // nucleotides
    uint64_t start_node = 0, end_node = 0;  // (rank << 1 | strand) of the first / last k-mer of the emitted sequence
    uint8_t self_rc = 0;          // sequence == RC(sequence)
};

constexpr uint64_t NONE = ~0ull;

template <class Fn>
inline void parallel_for_blocks(uint64_t n, unsigned threads, uint64_t min_block, const Fn &fn) {
    unsigned nt = threads ? threads : std::max(1u, std::min(std::thread::hardware_concurrency(), 64u));
    nt = (unsigned)std::min<uint64_t>(nt, std::max<uint64_t>(1, n / std::max<uint64_t>(min_block, 1)));
    if (nt <= 1) {
        fn((uint64_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back([&, t] { fn(n * t / nt, n * (t + 1) / nt); });
    for (auto &x : th) x.join();
}

inline uint64_t rev2(uint64_t w) {  // the 32 2-bit groups of a word in reverse order
    w = ((w >> 2) & 0x3333333333333333ull) | ((w & 0x3333333333333333ull) << 2);
    w = ((w >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((w & 0x0F0F0F0F0F0F0F0Full) << 4);
    return __builtin_bswap64(w);
}

struct KmerOps {
    unsigned k, nw;
    // RC(x): reverse the groups of all words, complement, drop the padding that has moved to the bottom
    void rc(const uint64_t *x, uint64_t *r) const {
        uint64_t t[8];
        for (unsigned i = 0; i < nw; ++i) t[i] = ~rev2(x[nw - 1 - i]);
        const unsigned pad = 64 * nw - 2 * k;  // < 64 (nw = ceil(k / 32))
        for (unsigned i = 0; i < nw; ++i) {
            if (pad == 0) r[i] = t[i];
            else r[i] = (t[i] >> pad) | (i + 1 < nw ? t[i + 1] << (64 - pad) : 0ull);
        }
        const unsigned tail = (2 * k) & 63u;
        if (tail) r[nw - 1] &= (1ull << tail) - 1;
    }
    // nucleotide-lexicographic order (RtSeq operator<): the first nucleotide that differs decides
    int cmp(const uint64_t *a, const uint64_t *b) const {
        for (unsigned i = 0; i < nw; ++i) {
            const uint64_t x = a[i] ^ b[i];
            if (x) {
                const unsigned p = (unsigned)__builtin_ctzll(x) & ~1u;
                return ((a[i] >> p) & 3) < ((b[i] >> p) & 3) ? -1 : 1;
            }
        }
        return 0;
    }
    unsigned nucl(const uint64_t *x, unsigned j) const { return (unsigned)(x[j >> 5] >> ((j & 31u) << 1)) & 3u; }
    // x[1..k-1] + c
    void next(const uint64_t *x, unsigned c, uint64_t *r) const {
        for (unsigned i = 0; i < nw; ++i) r[i] = (x[i] >> 2) | (i + 1 < nw ? x[i + 1] << 62 : 0ull);
        r[(k - 1) >> 5] |= (uint64_t)c << (((k - 1) & 31u) << 1);
    }
    uint64_t hash(const uint64_t *x) const {
        uint64_t h = 0x9E3779B97F4A7C15ull;
        for (unsigned i = 0; i < nw; ++i) {
            h ^= x[i];
            h *= 0xBF58476D1CE4E5B9ull;
            h ^= h >> 29;
        }
        h *= 0x94D049BB133111EBull;
        return h ^ (h >> 32);
    }
};

inline uint8_t invert_mask(uint8_t a) {  // the InOutMask of the other strand: the byte bit-reversed (inout_mask.hpp:92-131)
    a = (uint8_t)((a >> 4) | (a << 4));
    a = (uint8_t)(((a >> 2) & 0x33) | ((a & 0x33) << 2));
    return (uint8_t)(((a >> 1) & 0x55) | ((a & 0x55) << 1));
}
inline bool uniq4(unsigned m) { return m && !(m & (m - 1)); }

// kmers: n canonical k-mers of nw words each (bits above 2k zero) in k-mer-file order; ranks: their positions in the file (ascending);
// masks: their InOutMask bytes (out bits 0-3 by next nucleotide, in bits 4-7 by previous one, in the canonical frame). Every k-mer must
// be a non-junction k-mer that lies on a perfect loop whose k-mers are all in the set (what is left after the unbranching paths).
// Returns 0, or a negative number when the set is not closed under the walk (never expected: the caller reports an inconsistent index).
inline int collect_loops(const uint64_t *kmers, const uint64_t *ranks, const uint8_t *masks, uint64_t n, unsigned k, std::vector<PackedLoop> &out,
                         unsigned threads = 0, uint64_t grain = 0 /* tests: items per thread at least (0: the defaults) */) {
    out.clear();
    if (!n) return 0;
    const unsigned nw = (k + 31) / 32;
    if (nw > 8 || k < 1) return -1;
    const KmerOps op{k, nw};
    // ---- index: open addressing, value = position + 1 ----
    uint64_t tsize = 16;
    while (tsize < 2 * n) tsize <<= 1;
    std::vector<uint64_t> table(tsize, 0);
    parallel_for_blocks(n, threads, grain ? grain : 1 << 14, [&](uint64_t a, uint64_t b) {
        for (uint64_t i = a; i < b; ++i) {
            uint64_t h = op.hash(kmers + i * nw) & (tsize - 1);
            for (;;) {
                uint64_t expect = 0;
                if (__atomic_compare_exchange_n(&table[h], &expect, i + 1, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break;
                h = (h + 1) & (tsize - 1);
            }
        }
    });
    auto find = [&](const uint64_t *x) -> uint64_t {
        uint64_t h = op.hash(x) & (tsize - 1);
        for (;;) {
            const uint64_t v = table[h];
            if (!v) return NONE;
            if (!memcmp(kmers + (v - 1) * nw, x, nw * 8)) return v - 1;
            h = (h + 1) & (tsize - 1);
        }
    };
    // ---- successor of every oriented k-mer: node = 2 * position + strand (0: as stored) ----
    std::vector<uint64_t> succ(2 * n);
    parallel_for_blocks(2 * n, threads, grain ? grain : 1 << 14, [&](uint64_t a, uint64_t b) {
        uint64_t x[8], y[8], r[8];
        for (uint64_t v = a; v < b; ++v) {
            const uint64_t i = v >> 1;
            const uint8_t m = (v & 1) ? invert_mask(masks[i]) : masks[i];
            if (!uniq4(m & 15u) || !uniq4((m >> 4) & 15u)) {  // (a junction: the walks below stop with an error if they ever get here)
                succ[v] = NONE;
                continue;
            }
            if (v & 1) op.rc(kmers + i * nw, x);
            else memcpy(x, kmers + i * nw, nw * 8);
            op.next(x, (unsigned)__builtin_ctz(m & 15u), y);
            op.rc(y, r);
            const bool minimal = op.cmp(r, y) >= 0;  // IsMinimal: y <= RC(y)
            const uint64_t j = find(minimal ? y : r);
            succ[v] = j == NONE ? NONE : 2 * j + (minimal ? 0 : 1);
        }
    });
    // ---- cycles in the order in which the reference meets them: by their first k-mer in file order, walked from its stored strand.
    // Every k-mer walks its cycle until it meets a k-mer that comes earlier in the file (then it is not the first of its loop: expected
    // after O(log length) steps in a file in hash order) or comes back to itself (then it is, and the walk was the whole cycle): no thread
    // waits for another, the leaders are found by all cores, and the result does not depend on the schedule (the marks in `seen` only
    // spare k-mers of finished cycles their walk). A cycle and its reverse complement hold the same stored k-mers, so one of the two is
    // walked: the one on which the first k-mer lies as stored — the reference's choice.
    std::vector<uint8_t> seen(n, 0);
    struct Block {
        std::vector<uint64_t> nodes, off;
    };
    unsigned cnt = threads ? threads : std::max(1u, std::min(std::thread::hardware_concurrency(), 64u));
    cnt = (unsigned)std::min<uint64_t>(cnt, std::max<uint64_t>(1, n / (grain ? grain : 4096)));
    const uint64_t nblocks = cnt == 1 ? 1 : (uint64_t)cnt * 8;
    std::vector<Block> blocks(nblocks);
    std::atomic<uint64_t> nextb{0};
    std::atomic<int> chase_err{0};
    auto chase = [&]() {
        std::vector<uint64_t> local;
        for (;;) {
            const uint64_t bi = nextb.fetch_add(1);
            if (bi >= nblocks) break;
            Block &blk = blocks[bi];
            for (uint64_t si = n * bi / nblocks, se = n * (bi + 1) / nblocks; si < se; ++si) {
                if (__atomic_load_n(&seen[si], __ATOMIC_RELAXED)) continue;
                local.clear();
                uint64_t v = 2 * si;
                bool leader = true;
                do {
                    if ((v >> 1) < si) {
                        leader = false;
                        break;
                    }
                    local.push_back(v);
                    v = succ[v];
                    if (v == NONE || local.size() > 2 * n) {  // a successor that is a junction or not in the set, or no way back: no perfect loop
                        chase_err = 1;
                        return;
                    }
                } while (v != 2 * si);
                if (!leader) continue;
                for (uint64_t x : local) __atomic_store_n(&seen[x >> 1], (uint8_t)1, __ATOMIC_RELAXED);
                blk.off.push_back(blk.nodes.size());
                blk.nodes.insert(blk.nodes.end(), local.begin(), local.end());
            }
            blk.off.push_back(blk.nodes.size());
        }
    };
    if (cnt <= 1) chase();
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < cnt; ++t) th.emplace_back(chase);
        for (auto &t : th) t.join();
    }
    if (chase_err) return -2;
    std::vector<uint64_t> cyc;      // nodes of all cycles, one after the other, leaders in file order
    std::vector<uint64_t> cyc_off;  // start of every cycle in cyc
    cyc.reserve(n);
    for (auto &blk : blocks) {
        for (size_t c = 0; c + 1 < blk.off.size(); ++c) cyc_off.push_back(cyc.size() + blk.off[c]);
        cyc.insert(cyc.end(), blk.nodes.begin(), blk.nodes.end());
        std::vector<uint64_t>().swap(blk.nodes);
    }
    cyc_off.push_back(cyc.size());
    const uint64_t nc = cyc_off.size() - 1;
    // ---- the loops ----
    std::vector<PackedLoop> res(2 * nc);  // two parts per cycle at most; empty ones (len 0) are dropped below
    std::atomic<uint64_t> nextc{0};
    std::atomic<int> bad{0};
    auto strand0_if_palindrome = [&](uint64_t v) -> uint64_t {  // a k-mer that is its own reverse complement (even k) has one node: strand 0
        if (!(v & 1) || (k & 1)) return v;
        uint64_t r[8];
        op.rc(kmers + (v >> 1) * nw, r);
        return op.cmp(r, kmers + (v >> 1) * nw) == 0 ? v ^ 1ull : v;
    };
    auto work = [&]() {
        std::vector<uint64_t> w;
        std::vector<uint8_t> s, p, q;
        for (;;) {
            const uint64_t c = nextc.fetch_add(1);
            if (c >= nc) break;
            const uint64_t *cn = cyc.data() + cyc_off[c];
            const uint64_t L = cyc_off[c + 1] - cyc_off[c];
            // FindMinimalKMerInLoop: min over the k-mers of the cycle and their RCs = the smallest STORED (canonical) k-mer on it
            uint64_t best = cn[0] >> 1;
            for (uint64_t t = 1; t < L; ++t) {
                const uint64_t i = cn[t] >> 1;
                if (op.cmp(kmers + i * nw, kmers + best * nw) < 0) best = i;
            }
            // ConstructLoopFromVertex: from that k-mer (as stored: it may lie on the reverse-complement cycle) once around
            w.resize(L + 1);
            w[0] = 2 * best;
            bool ok = true;
            for (uint64_t t = 0; t < L && ok; ++t) ok = (w[t + 1] = succ[w[t]]) != NONE;
            if (!ok || w[L] != w[0]) {
                bad = 1;
                continue;
            }
            s.resize(k + L);
            for (unsigned j = 0; j < k; ++j) s[j] = (uint8_t)op.nucl(kmers + best * nw, j);
            for (uint64_t t = 1; t <= L; ++t) {  // last nucleotide of the oriented k-mer w[t]
                const uint64_t i = w[t] >> 1;
                s[k - 1 + t] = (w[t] & 1) ? (uint8_t)(3u - op.nucl(kmers + i * nw, 0)) : (uint8_t)op.nucl(kmers + i * nw, k - 1);
            }
            // first palindromic (k+1)-mer: the edge w[t] -> w[t+1] is its own reverse complement iff w[t+1] is w[t] on the other strand
            uint64_t pos = NONE;
            for (uint64_t t = 0; t < L; ++t)
                if (w[t + 1] == (w[t] ^ 1ull)) {
                    pos = t;
                    break;
                }
            struct Part {
                uint64_t a, b;  // first and last node of the part
            } parts[2];
            int np = 0;
            for (int pi = 0; pi < 2; ++pi) {
                if (pos == NONE) {
                    if (pi) break;
                    p = s;
                    parts[0] = {w[0], w[L]};
                } else if (pi == 0) {  // SplitLoop: the palindromic (k+1)-mer on its own ...
                    p.assign(s.begin() + pos, s.begin() + pos + k + 1);
                    parts[0] = {w[pos], w[pos + 1]};
                } else {  // ... and the rest, from the k-mer behind it around to the k-mer it starts with
                    p.assign(s.begin() + pos + 1, s.begin() + L);
                    p.insert(p.end(), s.begin(), s.begin() + pos + k);
                    parts[1] = {w[pos + 1], w[pos]};
                }
                ++np;
                const size_t n_ = p.size();
                q.resize(n_);
                for (size_t t = 0; t < n_; ++t) q[t] = (uint8_t)(3u - p[n_ - 1 - t]);
                const int c3 = memcmp(p.data(), q.data(), n_);  // codes 0..3: byte order = nucleotide order
                const std::vector<uint8_t> &e = c3 < 0 ? q : p;  // max(part, RC(part))
                PackedLoop &pl = res[2 * c + pi];
                pl.len = n_;
                pl.self_rc = c3 == 0;
                pl.words.assign((n_ + 31) / 32, 0);
                for (size_t t = 0; t < n_; ++t) pl.words[t >> 5] |= (uint64_t)e[t] << ((t & 31) << 1);
                const uint64_t fa = strand0_if_palindrome(c3 < 0 ? (parts[pi].b ^ 1ull) : parts[pi].a);
                const uint64_t fb = strand0_if_palindrome(c3 < 0 ? (parts[pi].a ^ 1ull) : parts[pi].b);
                pl.start_node = (ranks[fa >> 1] << 1) | (fa & 1);
                pl.end_node = (ranks[fb >> 1] << 1) | (fb & 1);
            }
            (void)np;
        }
    };
    {
        unsigned nt = threads ? threads : std::max(1u, std::min(std::thread::hardware_concurrency(), 64u));
        nt = (unsigned)std::min<uint64_t>(nt, std::max<uint64_t>(1, n / (grain ? grain : 4096)));
        if (nt <= 1) work();
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; ++t) th.emplace_back(work);
            for (auto &t : th) t.join();
        }
    }
    if (bad) return -4;
    out.reserve(nc);
    for (auto &pl : res)
        if (pl.len) out.push_back(std::move(pl));
    return 0;
}

}  // namespace smxl
