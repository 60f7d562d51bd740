// spades_amd/tools/read_input.hpp — host-side FASTA/FASTQ(.gz) front-end of the CLI clones.
// The reference keeps parsing on the CPU too (kseq + zlib-ng: common/io/reads/parser.cpp); only sequences are used here.
// Reads are handed to the library in ASCII; smx_submit_reads_ascii applies the longest-ACGT-run rule.
#pragma once
#include <sys/stat.h>
#include "../../include/smx.h"
#include "bgzf_reader.hpp"
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cerrno>
#include <mutex>
#include <thread>
#include <unistd.h>
#include <cstring>
#include <string>
#include <vector>

namespace smxtool {

struct ReadBatch {
    std::string bases;
    std::vector<uint64_t> off{0};
    void add(const std::string &s) {
        bases += s;
        off.push_back(bases.size());
    }
    uint64_t size() const { return off.size() - 1; }
    void clear() {
        bases.clear();
        off.assign(1, 0);
    }
};

class LineReader {  // gzopen reads plain files transparently as well
    gzFile f_;
    std::vector<char> buf_;
    size_t pos_ = 0, len_ = 0;

  public:
    explicit LineReader(const std::string &path) : f_(gzopen(path.c_str(), "rb")), buf_(1 << 20) {
        if (f_) gzbuffer(f_, 1 << 20);
    }
    ~LineReader() {
        if (f_) gzclose(f_);
    }
    bool ok() const { return f_ != nullptr; }
    bool getline(std::string &out) {
        out.clear();
        for (;;) {
            if (pos_ == len_) {
                int n = gzread(f_, buf_.data(), (unsigned)buf_.size());
                if (n <= 0) return !out.empty();
                len_ = (size_t)n;
                pos_ = 0;
            }
            size_t i = pos_;
            while (i < len_ && buf_[i] != '\n') ++i;
            out.append(buf_.data() + pos_, i - pos_);
            if (i < len_) {
                pos_ = i + 1;
                if (!out.empty() && out.back() == '\r') out.pop_back();
                return true;
            }
            pos_ = len_;
        }
    }
};

// Streams every sequence of a FASTA or FASTQ file to cb(seq). Returns false when the file cannot be opened,
// throws std::string on malformed input.
template <class F>
bool for_each_sequence(const std::string &path, F cb) {
    LineReader in(path);
    if (!in.ok()) return false;
    std::string line, seq;
    if (!in.getline(line)) return true;
    if (!line.empty() && line[0] == '>') {  // FASTA (multi-line)
        bool have = true;
        while (in.getline(line)) {
            if (!line.empty() && line[0] == '>') {
                cb(seq);
                seq.clear();
            } else {
                seq += line;
            }
        }
        if (have) cb(seq);
    } else if (!line.empty() && line[0] == '@') {  // FASTQ (4-line records; multi-line FASTQ is handled by length)
        for (;;) {
            seq.clear();
            std::string qual;
            bool got_plus = false;
            while (in.getline(line)) {
                if (!line.empty() && line[0] == '+') {
                    got_plus = true;
                    break;
                }
                seq += line;
            }
            if (!got_plus) throw std::string("malformed FASTQ (no '+' line): ") + path;
            while (qual.size() < seq.size() && in.getline(line)) qual += line;
            cb(seq);
            if (!in.getline(line)) break;
            if (line.empty()) {
                if (!in.getline(line)) break;
            }
            if (line.empty() || line[0] != '@') throw std::string("malformed FASTQ (expected '@'): ") + path;
        }
    } else {
        throw std::string("unknown input format (neither FASTA nor FASTQ): ") + path;
    }
    return true;
}

// Dataset description (the YAML that spades.py writes and `spades-gbuilder <file>.yaml` / `spades-kmercount -d` read;
// io::DataSet::load, common/library/library.cpp): a list of libraries, each a mapping with `type`, `orientation` and the read lists
// `left reads` / `right reads` / `single reads` / `interlaced reads` / `merged reads` (block or flow sequences). Only that subset of
// YAML is understood. Relative paths are taken from the directory of the YAML file, as the reference does.
struct DatasetLibrary {
    std::string type;
    std::vector<std::string> files;
    bool graph_constructable() const {  // SequencingLibrary::is_graph_constructable, common/library/library.hpp:197-202
        return type == "paired-end" || type == "single" || type == "hq-mate-pairs" || type == "clouds10x";
    }
};
inline bool load_dataset_yaml(const std::string &path, std::vector<DatasetLibrary> &libs) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::string text;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    fclose(f);
    std::string dir = path;
    const size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
    auto unquote = [](std::string v) {
        while (!v.empty() && (v.back() == ' ' || v.back() == '\r' || v.back() == ',')) v.pop_back();
        size_t a = 0;
        while (a < v.size() && v[a] == ' ') ++a;
        v = v.substr(a);
        if (v.size() >= 2 && (v.front() == '"' || v.front() == '\'') && v.back() == v.front()) v = v.substr(1, v.size() - 2);
        return v;
    };
    auto add_path = [&](DatasetLibrary &lib, const std::string &raw) {
        std::string v = unquote(raw);
        if (v.empty()) return;
        if (v[0] != '/') v = dir + "/" + v;
        lib.files.push_back(v);
    };
    auto is_reads_key = [](const std::string &k) {
        return k == "left reads" || k == "right reads" || k == "single reads" || k == "interlaced reads" || k == "merged reads";
    };
    bool in_reads = false;
    size_t pos = 0;
    while (pos < text.size()) {
        size_t e = text.find('\n', pos);
        if (e == std::string::npos) e = text.size();
        std::string line = text.substr(pos, e - pos);
        pos = e + 1;
        const size_t hash = line.find(" #");
        if (hash != std::string::npos) line = line.substr(0, hash);
        size_t ind = 0;
        while (ind < line.size() && line[ind] == ' ') ++ind;
        if (ind == line.size() || line[ind] == '#') continue;
        std::string body = line.substr(ind);
        bool new_item = false;
        if (body.compare(0, 2, "- ") == 0 || body == "-") {
            if (ind == 0) {  // a new library
                libs.emplace_back();
                in_reads = false;
                new_item = true;
                body = body.size() > 2 ? body.substr(2) : std::string();
                if (body.empty()) continue;
            } else if (in_reads && !libs.empty()) {  // an entry of the current read list
                add_path(libs.back(), body.substr(body.size() > 1 ? 2 : 1));
                continue;
            }
        }
        (void)new_item;
        const size_t colon = body.find(':');
        if (colon == std::string::npos || libs.empty()) continue;
        const std::string key = unquote(body.substr(0, colon));
        std::string val = body.substr(colon + 1);
        in_reads = false;
        if (key == "type") libs.back().type = unquote(val);
        else if (is_reads_key(key)) {
            const size_t lb = val.find('[');
            if (lb != std::string::npos) {  // flow sequence on one line
                const size_t rb = val.find(']', lb);
                std::string inner = val.substr(lb + 1, (rb == std::string::npos ? val.size() : rb) - lb - 1);
                size_t p = 0;
                while (p <= inner.size()) {
                    size_t c = inner.find(',', p);
                    if (c == std::string::npos) c = inner.size();
                    add_path(libs.back(), inner.substr(p, c - p));
                    p = c + 1;
                }
            } else {
                in_reads = true;
            }
        }
    }
    return true;
}

// fread(dst, 1, n, f) for big pieces of a plain file: the bytes [ftello(f), +n) are fetched by several threads with pread (a read of a
// page-cache / tmpfs file is a copy, 6-8 GB/s from one thread; different threads copy different pages at once) and the stream is moved
// on. Short reads (end of file) come back short, an error as 0 — like fread. SMX_IO_GRAIN = bytes per thread at least (tests: small).
inline size_t read_plain(FILE *f, char *dst, size_t n) {
    size_t grain = (size_t)16 << 20;
    if (const char *e = getenv("SMX_IO_GRAIN")) grain = (size_t)std::max(1, atoi(e));
    const unsigned nt = (unsigned)std::min<size_t>(std::min(io_threads(), 8u), n / grain);
    const off_t at = nt > 1 ? ftello(f) : (off_t)-1;
    if (nt <= 1 || at < 0) return fread(dst, 1, n, f);
    const int fd = fileno(f);
    std::vector<size_t> got(nt, 0);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            const size_t a = n * t / nt, b = n * (t + 1) / nt;
            size_t g = 0;
            while (a + g < b) {
                const ssize_t r = pread(fd, dst + a + g, b - a - g, at + (off_t)(a + g));
                if (r < 0 && errno == EINTR) continue;
                if (r <= 0) break;  // end of file (or an error: the caller sees a short read)
                g += (size_t)r;
            }
            got[t] = g;
        });
    for (auto &x : th) x.join();
    size_t total = 0;  // the bytes up to the first slice that came back short
    for (unsigned t = 0; t < nt; ++t) {
        total += got[t];
        if (got[t] < n * (t + 1) / nt - n * t / nt) break;
    }
    if (fseeko(f, at + (off_t)total, SEEK_SET) != 0) return 0;
    return total;
}

// One input file -> library. Uncompressed 4-line FASTQ goes to HBM as raw bytes and is cut into reads on the device
// (smx_submit_fastq_text: page-locked chunks, complete records only, the tail is carried over); everything else
// (gzip, FASTA, multi-line FASTQ) takes the host parser above. Returns 0, an smx error code, or -1 when the file
// cannot be opened; throws std::string on malformed input.
// mu (optional): serialises the library calls when several files are read by several host threads (a context is used by one
// thread at a time; reading and inflating — the slow part — run in parallel).
// The device arena maps its physical memory the first time an address is used (~17 ms per GiB): a tool that builds one graph or makes one
// count per process would pay that inside its first stage. Asked for here, it happens on the library's helper thread while the input is read.
// The figures: bytes of arena per byte of (uncompressed) input text that a construction / a both-strands count at 30x ends up using, split
// between the temporary and the long-lived region as measured on the MI355X (SMX_PREWARM_X="bottom,top" overrides them; 0,0 turns it off).
inline void prewarm_for_inputs(smx_ctx *ctx, const std::vector<std::string> &files, double bottom_x, double top_x) {
    if (const char *e = getenv("SMX_PREWARM_X")) {
        if (sscanf(e, "%lf,%lf", &bottom_x, &top_x) != 2) return;
    }
    double bytes = 0;
    for (const std::string &f : files) {
        struct stat st;
        if (stat(f.c_str(), &st) != 0) continue;
        const bool gz = f.size() > 3 && f.compare(f.size() - 3, 3, ".gz") == 0;
        bytes += (double)st.st_size * (gz ? 4.0 : 1.0);
    }
    if (bytes <= 0) return;
    (void)smx_prewarm(ctx, (size_t)(bytes * bottom_x), (size_t)(bytes * top_x));
}

inline int submit_file(smx_ctx *ctx, const std::string &path, std::mutex *mu = nullptr) {
    auto locked = [&](auto &&fn) {
        if (!mu) return fn();
        std::lock_guard<std::mutex> g(*mu);
        return fn();
    };
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return -1;
    unsigned char head[2] = {0, 0};
    const size_t nh = fread(head, 1, 2, f);
    const bool gz = nh == 2 && head[0] == 0x1f && head[1] == 0x8b;
    fseek(f, 0, SEEK_END);
    const long fsize = ftell(f);
    rewind(f);
    // The raw text goes to the device in page-locked chunks: plain files through fread, gzip files through zlib (inflate is the
    // bottleneck there, ~0.4 GB/s of text per stream; cutting the text into reads on the device takes the parser off that thread).
    gzFile gzf = nullptr;
    if (gz) {
        fclose(f);
        f = nullptr;
        gzf = gzopen(path.c_str(), "rb");
        if (!gzf) return -1;
        gzbuffer(gzf, 1 << 20);
    }
    bool device_path = !getenv("SMX_HOST_PARSE");  // env: force the host parser (measurements)
    if (device_path && !gz) device_path = nh == 2 && head[0] == '@';
    if (device_path) {
        // page-locking memory costs ~0.3 s per GiB: take what the file needs, in 256 MiB chunks at most
        const size_t want = gz ? (size_t)256 << 20 : (size_t)(fsize > 0 ? fsize : 0) + 4096;
        const size_t chunk_bytes = std::min<size_t>((size_t)256 << 20, std::max<size_t>(want, (size_t)1 << 20));
        char *buf = (char *)smx_pinned_alloc(chunk_bytes);
        bool pinned = buf != nullptr;
        if (!buf) buf = (char *)malloc(chunk_bytes);
        size_t have = 0;
        bool fallback = false, any = false, eof = false;
        int rc = 0;
        // BGZF (blocked gzip: what BCL Convert / DRAGEN and bgzip write): the blocks are inflated in parallel (bgzf_reader.hpp); an ordinary
        // gzip stream has one thread's worth of work by construction
        BgzfReader bz;
        bool bgzf = gz && !getenv("SMX_NO_BGZF") && BgzfReader::is_bgzf(path) && bz.open(path, io_threads());
        for (;;) {
            while (!eof && have < chunk_bytes) {  // fill the chunk
                size_t got;
                if (bgzf) {  // whole blocks, inflated by several threads straight into the chunk
                    if (chunk_bytes - have < 65536) break;
                    got = bz.read(buf + have, chunk_bytes - have);
                    if (got == BgzfReader::kError) {
                        // Not BGZF from here on — an ordinary gzip member behind BGZF blocks (`cat a.bgzf.gz b.gz`), a header flag the block
                        // parser does not take: zlib reads on from the boundary of the last good block (a failed call consumes nothing), one
                        // stream like any other .gz; a damaged file fails there with the format error as before.
                        gzclose(gzf);
                        gzf = nullptr;
                        const int fd2 = ::open(path.c_str(), O_RDONLY);
                        if (fd2 >= 0 && lseek(fd2, (off_t)bz.pos(), SEEK_SET) == (off_t)bz.pos()) gzf = gzdopen(fd2, "rb");
                        if (!gzf) {
                            if (fd2 >= 0) ::close(fd2);
                            rc = SMX_INVALID_INPUT_FORMAT;
                            break;
                        }
                        gzbuffer(gzf, 1 << 20);
                        bgzf = false;
                        continue;
                    }
                } else if (gz) {
                    const int g = gzread(gzf, buf + have, (unsigned)std::min<size_t>(chunk_bytes - have, (size_t)1 << 30));
                    if (g < 0) {
                        rc = SMX_INVALID_INPUT_FORMAT;
                        break;
                    }
                    got = (size_t)g;
                } else {
                    got = read_plain(f, buf + have, chunk_bytes - have);
                }
                have += got;
                if (got == 0) eof = true;
            }
            if (rc || have == 0) break;
            if (!any && buf[0] != '@') {  // not FASTQ (a gzipped FASTA, ...): the host parser decides
                fallback = true;
                break;
            }
            uint64_t n = 0, used = 0;
            rc = locked([&] { return smx_submit_fastq_text(ctx, buf, have, eof ? 1 : 0, &n, &used); });
            if (rc == SMX_INVALID_INPUT_FORMAT && !any) {  // not strict 4-line FASTQ: nothing was submitted, let the host parser decide
                fallback = true;
                rc = 0;
                break;
            }
            if (rc) break;
            any = any || n > 0;
            if (used == 0 && !eof && (have == chunk_bytes || (bgzf && chunk_bytes - have < 65536))) {  // a single record larger than the chunk
                fallback = !any;
                if (!fallback) rc = SMX_INVALID_INPUT_FORMAT;
                break;
            }
            memmove(buf, buf + used, have - used);
            have -= used;
            if (eof) break;
        }
        if (pinned) smx_pinned_free(buf); else free(buf);
        if (!fallback) {
            if (f) fclose(f);
            if (gzf) gzclose(gzf);
            return rc;
        }
    }
    if (gzf) gzclose(gzf);
    if (f) fclose(f);
    ReadBatch batch;
    int rc = 0;
    bool ok = for_each_sequence(path, [&](const std::string &s) {
        batch.add(s);
        if (batch.bases.size() > ((size_t)1 << 30) && !rc) {
            rc = locked([&] { return smx_submit_reads_ascii(ctx, batch.bases.data(), batch.off.data(), batch.size()); });
            batch.clear();
        }
    });
    if (!ok) return -1;
    if (!rc) rc = locked([&] { return smx_submit_reads_ascii(ctx, batch.bases.data(), batch.off.data(), batch.size()); });
    return rc;
}

}  // namespace smxtool
