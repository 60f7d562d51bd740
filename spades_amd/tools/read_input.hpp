// spades_amd/tools/read_input.hpp — host-side FASTA/FASTQ(.gz) front-end of the CLI clones.
// The reference keeps parsing on the CPU too (kseq + zlib-ng: common/io/reads/parser.cpp); only sequences are used here.
// Reads are handed to the library in ASCII; smx_submit_reads_ascii applies the longest-ACGT-run rule.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
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

}  // namespace smxtool
