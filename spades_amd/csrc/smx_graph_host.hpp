// spades_amd/csrc/smx_graph_host.hpp — host side of the construction path: the (rare, inherently serial) perfect-loop
// collection, link records / vertex numbering, and the GFA / FASTA writers. Included by smx_api.hip.
//
// Reference (paths relative to /root/reference/src/common):
//   CollectLoops / FindMinimalKMerInLoop / ConstructLoopFromVertex / SplitLoop
//                                  assembly_graph/construction/debruijn_graph_constructor.hpp:252-293,359-397
//   LinkRecord, ConstructGraph     same file :422-568 ; ids: assembly_graph/core/graph_core.hpp:234 (bias 3),
//                                  out-edge lists sorted by id :199-202, IncomingEdges = conj(OutgoingEdges(conj v)) :625-628
//   GFAWriter                      io/graph/gfa_writer.cpp:19-47,73-87,113-116
//   unitig FASTA                   projects/spades_tools/gbuilder.cpp:191-200, io/reads/header_naming.hpp:15-21
#pragma once
#include <atomic>
#include <unistd.h>
#include <thread>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

namespace smxh {

struct GraphHost {
    unsigned k = 0;
    // edges in the reference's enumeration order (unbranching paths, then loops)
    std::vector<uint64_t> eoff;    // [n_edges+1] offsets into seq
    std::string seq;               // ACGT
    std::vector<uint64_t> estart;  // node = 2*rank + rc of the first k-mer (rank: size_t in the reference, 64 bits here)
    std::vector<uint64_t> eend;    // node of the last k-mer
    std::vector<uint8_t> eself;    // s == RC(s)
    std::vector<uint32_t> ecov;    // raw coverage per edge (filled by smx_graph_fill_coverage; empty = no -c)
    std::vector<uint32_t> eflank_s, eflank_e;  // flanking raw coverage of the edge and of its conjugate (first / last 50 (k+1)-mers)
    uint64_t n_paths = 0, n_loops = 0, n_vertices = 0, n_links = 0;
    // link structure
    struct Rec {
        uint64_t hash_and_mask, edge;
    };
    std::vector<Rec> recs;
    std::vector<size_t> vstart;  // first record of every vertex, in vertex-id order
    size_t n_edges() const { return eoff.empty() ? 0 : eoff.size() - 1; }
};

inline char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; }
inline std::string revcomp(const std::string &s) {
    std::string r(s.size(), 'A');
    for (size_t i = 0; i < s.size(); ++i) r[i] = comp(s[s.size() - 1 - i]);
    return r;
}

// ---- perfect loops: smx_loops_host.hpp (packed k-mers, all cores); the string-level version of rounds 1-3 is its checker now
// (tests/host_shims/loops_string_ref.hpp) ----

// ---- spades-core edge order (DeBruijnGraphExtentionConstructor::ConstructGraph, debruijn_graph_constructor.hpp:590-604):
// unitigs sorted by Sequence::RawCompare (sequence/sequence.hpp:605-624: length, then packed 64-bit words from word 0)
// fn(begin, end) over [0, n) on several host threads (contiguous blocks)
template <class Fn>
inline void parallel_blocks(size_t n, size_t min_block, const Fn &fn) {
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt ? std::min(nt, 32u) : 1u;
    if (const char *e = getenv("SMX_WRITE_GRAIN")) min_block = std::max<size_t>(1, (size_t)atoll(e));  // tests: force the threaded path
    nt = (unsigned)std::min<size_t>(nt, std::max<size_t>(1, n / std::max<size_t>(min_block, 1)));
    if (nt <= 1) {
        fn((size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back([&, t] { fn(n * t / nt, n * (t + 1) / nt); });
    for (auto &x : th) x.join();
}
// std::sort in blocks on several threads + rounds of pairwise std::inplace_merge
template <class T, class Cmp>
inline void parallel_sort(std::vector<T> &v, const Cmp &cmp) {
    const size_t n = v.size();
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt ? std::min(nt, 16u) : 1u;
    size_t min_block = (size_t)1 << 16;
    if (const char *e = getenv("SMX_WRITE_GRAIN")) min_block = std::max<size_t>(1, (size_t)atoll(e));
    while (nt > 1 && n / nt < min_block) nt >>= 1;
    unsigned p2 = 1;
    while (p2 * 2 <= nt) p2 *= 2;
    nt = p2;
    if (nt <= 1) {
        std::sort(v.begin(), v.end(), cmp);
        return;
    }
    std::vector<size_t> cut(nt + 1);
    for (unsigned t = 0; t <= nt; ++t) cut[t] = n * t / nt;
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t) th.emplace_back([&, t] { std::sort(v.begin() + cut[t], v.begin() + cut[t + 1], cmp); });
        for (auto &x : th) x.join();
    }
    for (unsigned w = 1; w < nt; w *= 2) {
        std::vector<std::thread> th;
        for (unsigned t = 0; t + w < nt; t += 2 * w)
            th.emplace_back([&, t, w] { std::inplace_merge(v.begin() + cut[t], v.begin() + cut[t + w], v.begin() + cut[std::min(nt, t + 2 * w)], cmp); });
        for (auto &x : th) x.join();
    }
}

inline void sort_edges_raw(GraphHost &g) {
    const size_t ne = g.n_edges();
    if (ne < 2) return;
    std::vector<uint64_t> woff(ne + 1, 0);
    for (size_t i = 0; i < ne; ++i) woff[i + 1] = woff[i] + (g.eoff[i + 1] - g.eoff[i] + 31) / 32;
    std::vector<uint64_t> words(woff[ne], 0);
    parallel_blocks(ne, (size_t)1 << 14, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            const char *sq = g.seq.data() + g.eoff[i];
            const uint64_t len = g.eoff[i + 1] - g.eoff[i];
            for (uint64_t t = 0; t < len; ++t) {
                const char ch = sq[t];
                const uint64_t code = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3;
                words[woff[i] + (t >> 5)] |= code << ((t & 31) << 1);
            }
        }
    });
    std::vector<size_t> perm(ne);
    for (size_t i = 0; i < ne; ++i) perm[i] = i;
    parallel_sort(perm, [&](size_t a, size_t b) {
        const uint64_t la = g.eoff[a + 1] - g.eoff[a], lb = g.eoff[b + 1] - g.eoff[b];
        if (la != lb) return la < lb;
        const uint64_t nw = woff[a + 1] - woff[a];
        for (uint64_t w = 0; w < nw; ++w)
            if (words[woff[a] + w] != words[woff[b] + w]) return words[woff[a] + w] < words[woff[b] + w];
        return false;
    });
    GraphHost o;
    o.k = g.k;
    o.eoff.assign(ne + 1, 0);
    for (size_t i = 0; i < ne; ++i) o.eoff[i + 1] = o.eoff[i] + (g.eoff[perm[i] + 1] - g.eoff[perm[i]]);
    o.seq.resize(g.seq.size());
    o.estart.resize(ne);
    o.eend.resize(ne);
    o.eself.resize(ne);
    parallel_blocks(ne, (size_t)1 << 14, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            const size_t s = perm[i];
            memcpy(&o.seq[o.eoff[i]], g.seq.data() + g.eoff[s], (size_t)(g.eoff[s + 1] - g.eoff[s]));
            o.estart[i] = g.estart[s];
            o.eend[i] = g.eend[s];
            o.eself[i] = g.eself[s];
        }
    });
    o.n_paths = g.n_paths;
    o.n_loops = g.n_loops;
    g = std::move(o);
}

// LSD radix sort of 64-bit keys, 16-bit digits, passes whose digit is constant are skipped
inline void radix_sort_u64(std::vector<uint64_t> &a) {
    const size_t n = a.size();
    if (n < 2) return;
    std::vector<uint64_t> b(n);
    std::vector<size_t> cnt(65536);
    for (int pass = 0; pass < 4; ++pass) {
        const int sh = pass * 16;
        std::fill(cnt.begin(), cnt.end(), 0);
        for (size_t i = 0; i < n; ++i) cnt[(a[i] >> sh) & 0xFFFF]++;
        if (cnt[(a[0] >> sh) & 0xFFFF] == n) continue;
        size_t run = 0;
        for (size_t d = 0; d < 65536; ++d) {
            size_t c = cnt[d];
            cnt[d] = run;
            run += c;
        }
        for (size_t i = 0; i < n; ++i) b[cnt[(a[i] >> sh) & 0xFFFF]++] = a[i];
        a.swap(b);
    }
}

// ---- link records + vertices ----------------------------------------------------------------------------
// sorter: sorts distinct 64-bit keys ascending (host radix sort by default; the library passes its device pipeline)
using KeySorter = std::function<void(std::vector<uint64_t> &)>;

inline void build_links(GraphHost &g, const KeySorter &sorter = radix_sort_u64) {
    const uint64_t min_id = 3;
    const size_t ne = g.n_edges();
    g.recs.assign(ne * 2, GraphHost::Rec{0, 0});
    for (size_t i = 0; i < ne; ++i) {
        const uint64_t edge = min_id + 2 * i;
        g.recs[2 * i] = {((uint64_t)(g.estart[i] >> 1) << 2) | ((uint64_t)(g.estart[i] & 1) << 1) | 1ull, edge};
        if (g.eself[i]) g.recs[2 * i + 1] = {~0ull, 0};  // LinkRecord(): invalid
        else g.recs[2 * i + 1] = {((uint64_t)(g.eend[i] >> 1) << 2) | ((uint64_t)(g.eend[i] & 1) << 1), edge};
    }
    auto eam = [](const GraphHost::Rec &r) { return (r.edge << 2) | (r.hash_and_mask & 3); };
    // CompareByVertexKMerEdgeIdAndMask: (rank, EdgeAndMask). rank < 2^31 and edge id < 2^31, so the pair packs into one
    // 64-bit key (rank << 33 | EdgeAndMask); invalid records (self-conjugate ends) sort last as ~0.
    uint64_t max_rank = 0;
    for (size_t i = 0; i < ne; ++i) max_rank = std::max(max_rank, std::max(g.estart[i], g.eend[i]) >> 1);
    const bool packable = ne < (1ull << 29) && max_rank < (1ull << 31);
    if (packable) {
        std::vector<uint64_t> keys(g.recs.size());
        for (size_t i = 0; i < g.recs.size(); ++i) {
            const GraphHost::Rec &r = g.recs[i];
            const bool invalid = r.hash_and_mask + 1 == 0 && r.edge == 0;
            keys[i] = invalid ? ~0ull : ((r.hash_and_mask >> 2) << 33) | eam(r);
        }
        {  // valid keys are distinct; the invalid ones (all ~0) are set aside so that the sorter sees distinct keys only
            size_t nvalid = 0;
            for (size_t i = 0; i < keys.size(); ++i)
                if (keys[i] != ~0ull) keys[nvalid++] = keys[i];
            const size_t ninv = keys.size() - nvalid;
            keys.resize(nvalid);
            sorter(keys);
            keys.resize(nvalid + ninv, ~0ull);
        }
        for (size_t i = 0; i < keys.size(); ++i) {
            if (keys[i] == ~0ull) g.recs[i] = {~0ull, 0};
            else {
                const uint64_t e = keys[i] & ((1ull << 33) - 1);
                g.recs[i] = {((keys[i] >> 33) << 2) | (e & 3), e >> 2};
            }
        }
    } else {
        parallel_sort(g.recs, [&](const GraphHost::Rec &a, const GraphHost::Rec &b) {
            uint64_t ha = a.hash_and_mask >> 2, hb = b.hash_and_mask >> 2;
            if (ha != hb) return ha < hb;
            return eam(a) < eam(b);
        });
    }
    g.vstart.clear();
    for (size_t i = 0; i < g.recs.size(); ++i) {
        const bool invalid = g.recs[i].hash_and_mask + 1 == 0 && g.recs[i].edge == 0;
        if ((i == 0 || (g.recs[i].hash_and_mask >> 2) != (g.recs[i - 1].hash_and_mask >> 2)) && !invalid) g.vstart.push_back(i);
    }
    // vertices by their smallest EdgeAndMask (unique per vertex): key = EdgeAndMask << 31 | position in vstart
    if (packable && g.vstart.size() < (1ull << 31)) {
        std::vector<uint64_t> keys(g.vstart.size());
        for (size_t i = 0; i < keys.size(); ++i) keys[i] = (eam(g.recs[g.vstart[i]]) << 31) | i;
        sorter(keys);
        std::vector<size_t> vs(keys.size());
        for (size_t i = 0; i < keys.size(); ++i) vs[i] = g.vstart[(size_t)(keys[i] & ((1ull << 31) - 1))];
        g.vstart.swap(vs);
    } else {
        std::sort(g.vstart.begin(), g.vstart.end(), [&](size_t a, size_t b) { return eam(g.recs[a]) < eam(g.recs[b]); });
    }
    g.n_vertices = g.vstart.size();
}

// out-edge lists of v and conj(v) for vertex number vn (sorted by edge id)
inline void vertex_edges(const GraphHost &g, size_t vn, uint64_t *outv, size_t &no, uint64_t *outc, size_t &nc) {
    const uint64_t min_id = 3;
    no = nc = 0;
    const size_t i0 = g.vstart[vn];
    const uint64_t h = g.recs[i0].hash_and_mask >> 2;
    for (size_t j = i0; j < g.recs.size() && (g.recs[j].hash_and_mask >> 2) == h; ++j) {
        const uint64_t e = g.recs[j].edge;
        const size_t ei = (size_t)((e - min_id) >> 1);
        const uint64_t ce = g.eself[ei] ? e : e + 1;
        const bool is_rc = (g.recs[j].hash_and_mask >> 1) & 1, is_start = g.recs[j].hash_and_mask & 1;
        if (is_start) {
            if (!is_rc) outv[no++] = e; else outc[nc++] = e;
        } else {
            if (!is_rc) outc[nc++] = ce; else outv[no++] = ce;
        }
    }
    std::sort(outv, outv + no);
    std::sort(outc, outc + nc);
}

class BufWriter {
    FILE *f_;
    std::vector<char> buf_;
    size_t n_ = 0;
    bool ok_ = true;

  public:
    explicit BufWriter(FILE *f) : f_(f), buf_((size_t)8 << 20) {}
    void add(const char *s, size_t n) {
        if (n_ + n > buf_.size()) flush();
        if (n > buf_.size()) {
            ok_ &= fwrite(s, 1, n, f_) == n;
            return;
        }
        memcpy(buf_.data() + n_, s, n);
        n_ += n;
    }
    void add(const char *s) { add(s, strlen(s)); }
    void num(uint64_t v) {
        char t[24];
        int p = 24;
        do {
            t[--p] = (char)('0' + v % 10);
            v /= 10;
        } while (v);
        add(t + p, (size_t)(24 - p));
    }
    void flush() {
        if (n_) ok_ &= fwrite(buf_.data(), 1, n_, f_) == n_;
        n_ = 0;
    }
    bool ok() const { return ok_; }
};

// GFAWriter::WriteSegmentsAndLinks without coverage (DP:f:0, KC:i:0)
// Formats items [0, n) with fmt(begin, end, out) on several host threads (blocks of `grain` items, one output string per
// block) and writes the blocks in order: the text writers are the slowest step of a large construction otherwise (one thread
// formats ~0.9 GB/s). Small inputs stay on the calling thread.
template <class Fmt>
inline bool parallel_write(FILE *f, size_t n, size_t grain, const Fmt &fmt) {
    if (const char *e = getenv("SMX_WRITE_GRAIN")) grain = std::max<size_t>(1, (size_t)atoll(e));  // tests: force many small blocks
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt ? std::min(nt, 64u) : 1u;
    if (n < 4 * grain) nt = 1;
    bool ok = true;
    int fd = -1;
    off_t pos = -1;
    if (nt > 1) {  // (a stream that cannot seek — a pipe — is written block after block by this thread)
        if (fflush(f) != 0) return false;
        fd = fileno(f);
        pos = ftello(f);
        if (fd < 0 || pos < 0) nt = 1;
    }
    if (nt == 1) {
        std::string out;
        for (size_t b = 0; b < n; b += grain) {
            out.clear();
            fmt(b, std::min(n, b + grain), out);
            if (!out.empty()) ok &= fwrite(out.data(), 1, out.size(), f) == out.size();
        }
        return ok;
    }
    // Rounds of nt blocks: every thread formats its block, the block offsets follow from the sizes, and every thread writes its own block
    // at its place (pwrite: page allocation and copy of a 2 GB text run on all threads, not on one; measured at 12 M unitigs: the single
    // writing thread was 60 % of spades-gbuilder-mi355x's wall). The stream position is carried by hand and restored at the end.
    std::vector<std::string> out(nt);
    std::vector<off_t> at(nt);
    std::atomic<bool> wok{true};
    for (size_t base = 0; base < n; base += (size_t)nt * grain) {
        {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; ++t) {
                const size_t b = std::min(n, base + (size_t)t * grain), e = std::min(n, b + grain);
                out[t].clear();
                if (b >= e) continue;
                th.emplace_back([&, t, b, e] { fmt(b, e, out[t]); });
            }
            for (auto &x : th) x.join();
        }
        for (unsigned t = 0; t < nt; ++t) {
            at[t] = pos;
            pos += (off_t)out[t].size();
        }
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t) {
            if (out[t].empty()) continue;
            th.emplace_back([&, t] {
                size_t done = 0;
                while (done < out[t].size()) {
                    const ssize_t w = pwrite(fd, out[t].data() + done, out[t].size() - done, at[t] + (off_t)done);
                    if (w <= 0) {
                        wok = false;
                        return;
                    }
                    done += (size_t)w;
                }
            });
        }
        for (auto &x : th) x.join();
    }
    if (fseeko(f, pos, SEEK_SET) != 0) return false;
    return ok && wok.load();
}
inline void append_num(std::string &o, uint64_t v) {
    char t[24];
    int p = 24;
    do {
        t[--p] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    o.append(t + p, (size_t)(24 - p));
}

inline bool write_gfa(GraphHost &g, FILE *f, const char *flavour_version) {
    const uint64_t min_id = 3;
    bool ok = true;
    {
        std::string h = std::string("H\tsp:Z:") + flavour_version + "\n";
        ok &= fwrite(h.data(), 1, h.size(), f) == h.size();
    }
    const size_t ne = g.n_edges();
    const bool cov = g.ecov.size() == ne;
    ok &= parallel_write(f, ne, (size_t)1 << 16, [&](size_t b, size_t e, std::string &o) {
        o.reserve((size_t)(g.eoff[e] - g.eoff[b]) + (e - b) * 48);
        for (size_t i = b; i < e; ++i) {
            o += "S\t";
            append_num(o, min_id + 2 * i);
            o += '\t';
            o.append(g.seq.data() + g.eoff[i], (size_t)(g.eoff[i + 1] - g.eoff[i]));
            if (!cov) {
                o += "\tDP:f:0\tKC:i:0\n";
            } else {  // "DP:f:" << float(cov) (ostream default = %g, 6 significant digits) ; cov = raw / #(k+1)-mers
                char t[64];
                const double c = (double)g.ecov[i] / (double)(g.eoff[i + 1] - g.eoff[i] - g.k);
                int tn = snprintf(t, sizeof t, "\tDP:f:%g\tKC:i:%u\n", (double)(float)c, g.ecov[i]);
                o.append(t, (size_t)tn);
            }
        }
    });
    std::atomic<uint64_t> n_links{0};
    ok &= parallel_write(f, g.vstart.size(), (size_t)1 << 16, [&](size_t vb, size_t ve, std::string &o) {
        uint64_t nl = 0;
        for (size_t vn = vb; vn < ve; ++vn) {
            uint64_t outv[8], outc[8];
            size_t no, nc;
            vertex_edges(g, vn, outv, no, outc, nc);
            for (size_t a = 0; a < nc; ++a) {
                const uint64_t oc = outc[a];
                const size_t ei = (size_t)((oc - min_id) >> 1);
                const uint64_t inc = g.eself[ei] ? oc : (((oc - min_id) & 1) ? oc - 1 : oc + 1);
                for (size_t c = 0; c < no; ++c) {
                    const uint64_t oe = outv[c];
                    const uint64_t cin = min_id + (((inc - min_id) >> 1) << 1), cout = min_id + (((oe - min_id) >> 1) << 1);
                    o += "L\t";
                    append_num(o, cin);
                    o += inc == cin ? "\t+\t" : "\t-\t";
                    append_num(o, cout);
                    o += oe == cout ? "\t+\t" : "\t-\t";
                    append_num(o, g.k);
                    o += "M\n";
                    ++nl;
                }
            }
        }
        n_links += nl;
    });
    g.n_links = n_links.load();
    return ok;
}

// ---- SPAdes internal graph format (gbuilder --spades; BasicGraphIO::Save = GraphIO + CoverageIO) ---------------------
// io/binary/graph.hpp:27-74: ULEB128 unsigned integers (io/binary/binary.hpp impl::Encoding), raw 1-byte bool, Sequence =
// raw size_t length + packed 2-bit words (sequence/sequence.hpp:797-830).
//   vreserved, ereserved, link_size, vertex_cnt ; for every vertex id in increasing order: SaveVertex(v) ; for every outgoing
//   edge e1 (sorted by id) with e1 <= conj(e1): e1, conj(e1), SaveVertex(EdgeEnd(e1)), EdgeNucls(e1) ; 0.
//   SaveVertex(v) = v, conj(v) and, the first time either of them is met, complex=false + overlap k.
//   vreserved = 2V + V/100, ereserved = 2E + E/100 (ConstructGraph, debruijn_graph_constructor.hpp:513,554).
// io/binary/coverage.hpp:23-29 (.cvr): (canonical edge id, raw coverage)* 0.
inline void put_uleb(BufWriter &w, uint64_t v) {
    char b[10];
    int n = 0;
    do {
        uint8_t byte = v & 0x7f;
        v >>= 7;
        if (v) byte |= 0x80;
        b[n++] = (char)byte;
    } while (v);
    w.add(b, (size_t)n);
}

// end vertex id of every edge orientation: canonical edge i ends at v (end record, !rc) or conj(v) (rc);
// conj(e) ends at conj(start vertex of e)   (LinkEdge, debruijn_graph_constructor.hpp:491-500)
inline void edge_end_vertices(const GraphHost &g, std::vector<uint64_t> &end_c, std::vector<uint64_t> &end_r) {
    const uint64_t min_id = 3;
    const size_t ne = g.n_edges(), nv = g.vstart.size();
    end_c.assign(ne, 0);
    end_r.assign(ne, 0);
    for (size_t vn = 0; vn < nv; ++vn) {
        const size_t i0 = g.vstart[vn];
        const uint64_t h = g.recs[i0].hash_and_mask >> 2;
        for (size_t j = i0; j < g.recs.size() && (g.recs[j].hash_and_mask >> 2) == h; ++j) {
            const size_t ei = (size_t)((g.recs[j].edge - min_id) >> 1);
            const bool is_rc = (g.recs[j].hash_and_mask >> 1) & 1, is_start = g.recs[j].hash_and_mask & 1;
            const uint64_t v = min_id + 2 * vn, cv = v + 1;
            if (is_start) {
                end_r[ei] = is_rc ? v : cv;
                if (g.eself[ei]) end_c[ei] = end_r[ei];
            } else {
                end_c[ei] = is_rc ? cv : v;
            }
        }
    }
}

inline void put_uleb(std::string &o, uint64_t v) {
    do {
        uint8_t byte = v & 0x7f;
        v >>= 7;
        if (v) byte |= 0x80;
        o += (char)byte;
    } while (v);
}
inline bool write_grseq(const GraphHost &g, FILE *f) {
    const uint64_t min_id = 3;
    const size_t ne = g.n_edges(), nv = g.vstart.size();
    bool ok = true;
    {
        std::string h;
        put_uleb(h, 2 * nv + nv / 100);
        put_uleb(h, 2 * ne + ne / 100);
        put_uleb(h, 0);
        put_uleb(h, 2 * nv);
        ok &= fwrite(h.data(), 1, h.size(), f) == h.size();
    }
    std::vector<uint64_t> end_c, end_r;
    edge_end_vertices(g, end_c, end_r);
    // SaveVertex writes "complex = false, overlap k" the first time a vertex (either orientation) is mentioned in the traversal
    // (vertex vn, orientation o, then the end vertices of its canonical out-edges in id order). That first mention is the vertex's
    // own turn unless an edge of an EARLIER vertex ends in it; one serial pass finds it, the formatting then runs in blocks.
    // mention key = ((2 vn + o) << 4) | (1 + index of the edge in the out-list), 0 for the vertex's own mention
    std::vector<uint64_t> first(nv);
    for (size_t vn = 0; vn < nv; ++vn) first[vn] = (uint64_t)(2 * vn) << 4;
    for (size_t vn = 0; vn < nv; ++vn) {
        uint64_t outv[8], outc[8];
        size_t no, nc;
        vertex_edges(g, vn, outv, no, outc, nc);
        for (int o = 0; o < 2; ++o) {
            const uint64_t *lst = o ? outc : outv;
            const size_t n = o ? nc : no;
            for (size_t a = 0; a < n; ++a) {
                const uint64_t e1 = lst[a];
                if ((e1 - min_id) & 1) continue;
                const size_t un = (size_t)((end_c[(size_t)((e1 - min_id) >> 1)] - min_id) >> 1);
                const uint64_t key = ((uint64_t)(2 * vn + o) << 4) | (a + 1);
                if (key < first[un]) first[un] = key;
            }
        }
    }
    ok &= parallel_write(f, nv, (size_t)1 << 15, [&](size_t vb, size_t ve, std::string &out) {
        auto save_vertex = [&](uint64_t vid, uint64_t key) {
            const size_t vn = (size_t)((vid - min_id) >> 1);
            const uint64_t v = min_id + 2 * vn;
            put_uleb(out, vid);
            put_uleb(out, vid == v ? v + 1 : v);
            if (key != first[vn]) return;
            out += (char)0;  // complex = false
            put_uleb(out, g.k);
        };
        std::vector<uint64_t> words;
        for (size_t vn = vb; vn < ve; ++vn) {
            uint64_t outv[8], outc[8];
            size_t no, nc;
            vertex_edges(g, vn, outv, no, outc, nc);
            for (int o = 0; o < 2; ++o) {
                save_vertex(min_id + 2 * vn + o, (uint64_t)(2 * vn + o) << 4);
                const uint64_t *lst = o ? outc : outv;
                const size_t n = o ? nc : no;
                for (size_t a = 0; a < n; ++a) {
                    const uint64_t e1 = lst[a];
                    const size_t ei = (size_t)((e1 - min_id) >> 1);
                    const bool canon = ((e1 - min_id) & 1) == 0;
                    if (!canon) continue;  // conj(e1) < e1
                    const uint64_t e2 = g.eself[ei] ? e1 : e1 + 1;
                    put_uleb(out, e1);
                    put_uleb(out, e2);
                    save_vertex(end_c[ei], ((uint64_t)(2 * vn + o) << 4) | (a + 1));
                    const uint64_t len = g.eoff[ei + 1] - g.eoff[ei];
                    out.append((const char *)&len, 8);
                    words.assign((size_t)((len + 31) / 32), 0);
                    const char *sq = g.seq.data() + g.eoff[ei];
                    for (uint64_t t = 0; t < len; ++t) {
                        const char ch = sq[t];
                        const uint64_t code = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3;
                        words[(size_t)(t >> 5)] |= code << ((t & 31) << 1);
                    }
                    // short-sequence representation (< 60 nt, sequence.hpp:196-251): the inline 2-word buffer keeps its
                    // metadata byte [size:6 | rtl:1 | is_short:1] in the top byte of word 1, and BinWrite dumps it with the data
                    if (len > 32 && len < 60) words[1] |= (uint64_t)(((len & 0x3F) << 2) | 1) << 56;  // is_short_rep: size < 60
                    out.append((const char *)words.data(), words.size() * 8);
                }
                put_uleb(out, 0);
            }
        }
    });
    return ok;
}

inline bool write_cvr(const GraphHost &g, FILE *f) {
    BufWriter w(f);
    const size_t ne = g.n_edges();
    for (size_t i = 0; i < ne; ++i) {
        put_uleb(w, 3 + 2 * i);
        put_uleb(w, g.ecov.size() == ne ? g.ecov[i] : 0);
    }
    put_uleb(w, 0);
    w.flush();
    return w.ok();
}

// gbuilder --fastg: io::FastgWriter::WriteSegmentsAndLinks (io/graph/fastg_writer.cpp:21-48). Every edge (both orientations,
// id order): ">" name(e) [":" names of the outgoing edges of EdgeEnd(e), as a sorted std::set<std::string>, ","-joined] ";"
// name(e) = EDGE_<canonical id>_length_<nt>_cov_<std::to_string(double)> + "'" for the non-canonical orientation
// (io/utils/edge_namer.hpp:31-37,72-92, io/reads/header_naming.hpp:15-25); sequence wrapped at 60.
inline bool write_fastg(const GraphHost &g, FILE *f) {
    const uint64_t min_id = 3;
    const size_t ne = g.n_edges();
    std::vector<uint64_t> end_c, end_r;
    edge_end_vertices(g, end_c, end_r);
    auto name = [&](uint64_t e) {
        const size_t ei = (size_t)((e - min_id) >> 1);
        const uint64_t len = g.eoff[ei + 1] - g.eoff[ei];
        const double cov = g.ecov.size() == ne ? (double)g.ecov[ei] / (double)(len - g.k) : 0.0;
        char t[96];
        snprintf(t, sizeof t, "EDGE_%llu_length_%llu_cov_%f", (unsigned long long)(min_id + 2 * ei), (unsigned long long)len, cov);
        std::string s(t);
        if ((e - min_id) & 1) s += "'";
        return s;
    };
    return parallel_write(f, ne, (size_t)1 << 15, [&](size_t b, size_t e_, std::string &out) {
        for (size_t i = b; i < e_; ++i) {
            for (int o = 0; o < (g.eself[i] ? 1 : 2); ++o) {
                const uint64_t e = min_id + 2 * i + o;
                const uint64_t endv = o ? end_r[i] : end_c[i];
                std::vector<std::string> next;
                if (endv >= min_id) {
                    uint64_t outv[8], outc[8];
                    size_t no, nc;
                    vertex_edges(g, (size_t)((endv - min_id) >> 1), outv, no, outc, nc);
                    const uint64_t *lst = ((endv - min_id) & 1) ? outc : outv;
                    const size_t n = ((endv - min_id) & 1) ? nc : no;
                    for (size_t a = 0; a < n; ++a) next.push_back(name(lst[a]));
                    std::sort(next.begin(), next.end());
                    next.erase(std::unique(next.begin(), next.end()), next.end());
                }
                out += '>';
                out += name(e);
                for (size_t a = 0; a < next.size(); ++a) {
                    out += a ? ',' : ':';
                    out += next[a];
                }
                out += ";\n";
                const uint64_t len = g.eoff[i + 1] - g.eoff[i];
                std::string sq(g.seq.data() + g.eoff[i], (size_t)len);
                if (o) sq = revcomp(sq);
                for (uint64_t p = 0; p < len; p += 60) {
                    out.append(sq.data() + p, (size_t)std::min<uint64_t>(60, len - p));
                    out += '\n';
                }
            }
        }
    });
}

// gbuilder --unitigs: ">EDGE_<i>_length_<len>" + sequence wrapped at 60 (gbuilder.cpp:191-200)
inline bool write_unitigs_fasta(const GraphHost &g, FILE *f) {
    return parallel_write(f, g.n_edges(), (size_t)1 << 16, [&](size_t b, size_t e, std::string &out) {
        for (size_t i = b; i < e; ++i) {
            const uint64_t len = g.eoff[i + 1] - g.eoff[i];
            out += ">EDGE_";
            append_num(out, i + 1);
            out += "_length_";
            append_num(out, len);
            out += '\n';
            for (uint64_t p = 0; p < len; p += 60) {
                out.append(g.seq.data() + g.eoff[i] + p, (size_t)std::min<uint64_t>(60, len - p));
                out += '\n';
            }
        }
    });
}

}  // namespace smxh
