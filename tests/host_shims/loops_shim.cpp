// tests/host_shims/loops_shim.cpp — C entries for tests/test_loops_host_cpu.py: the product's packed perfect-loop collector
// (spades_amd/csrc/smx_loops_host.hpp, as it is) and the string-level restatement it is checked against (loops_string_ref.hpp).
#include "../../spades_amd/csrc/smx_loops_host.hpp"
#include "loops_string_ref.hpp"

#include <cstring>

namespace {
struct Flat {
    std::vector<std::string> seq;
    std::vector<uint64_t> start, end;
    std::vector<uint8_t> self;
};
// result layout: returns the number of loops (or a negative error); lens/starts/ends/selfs[cap]; ASCII sequences one after the other in text[text_cap]
int deliver(const Flat &f, uint64_t *lens, uint64_t *starts, uint64_t *ends, uint8_t *selfs, uint64_t cap, char *text, uint64_t text_cap) {
    if (f.seq.size() > cap) return -100;
    uint64_t at = 0;
    for (size_t i = 0; i < f.seq.size(); ++i) {
        if (at + f.seq[i].size() > text_cap) return -101;
        memcpy(text + at, f.seq[i].data(), f.seq[i].size());
        at += f.seq[i].size();
        lens[i] = f.seq[i].size();
        starts[i] = f.start[i];
        ends[i] = f.end[i];
        selfs[i] = f.self[i];
    }
    return (int)f.seq.size();
}
}  // namespace

extern "C" int loops_packed(const uint64_t *kmers, const uint64_t *ranks, const uint8_t *masks, uint64_t n, unsigned k, unsigned threads, uint64_t grain,
                            uint64_t *lens, uint64_t *starts, uint64_t *ends, uint8_t *selfs, uint64_t cap, char *text, uint64_t text_cap) {
    std::vector<smxl::PackedLoop> loops;
    const int rc = smxl::collect_loops(kmers, ranks, masks, n, k, loops, threads, grain);
    if (rc) return rc;
    Flat f;
    for (auto &l : loops) {
        std::string s(l.len, 'A');
        for (uint64_t t = 0; t < l.len; ++t) s[t] = "ACGT"[(l.words[t >> 5] >> ((t & 31) << 1)) & 3];
        f.seq.push_back(s);
        f.start.push_back(l.start_node);
        f.end.push_back(l.end_node);
        f.self.push_back(l.self_rc);
    }
    return deliver(f, lens, starts, ends, selfs, cap, text, text_cap);
}

// what smx_construct.hpp: append_loops did with the string collector in rounds 1-3
extern "C" int loops_string(const uint64_t *kmers, const uint64_t *ranks, const uint8_t *masks, uint64_t n, unsigned k, uint64_t *lens, uint64_t *starts,
                            uint64_t *ends, uint8_t *selfs, uint64_t cap, char *text, uint64_t text_cap) {
    const unsigned nw = (k + 31) / 32;
    std::vector<smxh::LoopNode> nodes(n);
    for (uint64_t t = 0; t < n; ++t) {
        nodes[t].rank = ranks[t];
        nodes[t].kmer.resize(k);
        for (unsigned j = 0; j < k; ++j) nodes[t].kmer[j] = "ACGT"[(kmers[(size_t)t * nw + (j >> 5)] >> ((j & 31) << 1)) & 3];
        nodes[t].mask = masks[t];
    }
    smxh::LoopCollector lc(nodes, k);
    std::vector<std::string> loops;
    lc.collect(loops);
    Flat f;
    for (auto &sq : loops) {
        f.seq.push_back(sq);
        f.start.push_back(lc.node_of(sq.substr(0, k)));
        f.end.push_back(lc.node_of(sq.substr(sq.size() - k)));
        f.self.push_back(sq == smxh::revcomp(sq) ? 1 : 0);
    }
    return deliver(f, lens, starts, ends, selfs, cap, text, text_cap);
}
