// spades_amd/tools/fastq_split.hpp — byte ranges of one uncompressed 4-line FASTQ file for several readers (the ranks of
// spades-gbuilder-mi355x --gpus N each read their own range; the reference reads one stream, io/reads/file_reader.hpp, so the split
// has no counterpart there — k-mer counting and graph construction do not depend on the order of the reads).
// No dependencies: the CPU test tier compiles this header alone.
#pragma once
#include <sys/types.h>

#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

namespace smxtool {

// Is the head of the file strict 4-line FASTQ ('@' line, sequence, '+' line, quality of the same length)? Every reader looks at the
// same bytes, so every reader decides the same way. Looks at up to `max_records` records within the first `max_bytes` bytes.
inline bool fastq_head_is_four_line(const std::string &path, size_t max_records = 256, size_t max_bytes = (size_t)4 << 20) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<char> buf(max_bytes);
    const size_t n = fread(buf.data(), 1, buf.size(), f);
    const bool whole = n < buf.size();
    fclose(f);
    size_t pos = 0, records = 0;
    auto line = [&](size_t &b, size_t &e) {  // next line [b, e) without its terminator; false when the buffer ends inside it
        if (pos >= n) return false;
        b = pos;
        while (pos < n && buf[pos] != '\n') ++pos;
        if (pos == n && !whole) return false;
        e = pos;
        if (e > b && buf[e - 1] == '\r') --e;
        if (pos < n) ++pos;
        return true;
    };
    while (records < max_records) {
        size_t b0, e0, b1, e1, b2, e2, b3, e3;
        if (!line(b0, e0)) break;
        if (e0 == b0 && pos >= n) break;  // trailing empty line
        if (!line(b1, e1) || !line(b2, e2) || !line(b3, e3)) {
            if (whole) return false;  // a record cut short by the end of the file
            break;                    // ... by the end of the window: what was seen so far decides
        }
        if (buf[b0] != '@' || e2 == b2 || buf[b2] != '+' || e1 - b1 != e3 - b3) return false;
        ++records;
    }
    return records > 0;
}

// Start of the first record at or after byte `from`: a line that begins with '@' whose next-but-one line begins with '+'. (A quality
// line may begin with '@' too, but then the next-but-one line is a sequence line, which never begins with '+'.) Returns `fsize`
// when there is none.
inline long long fastq_record_at_or_after(FILE *f, long long from, long long fsize) {
    if (from <= 0) return 0;
    if (from >= fsize) return fsize;
    size_t window = (size_t)1 << 20;
    for (;;) {
        const long long base = from - 1;  // the byte before: a '\n' there makes `from` itself a line start
        const size_t want = (size_t)std::min<long long>((long long)window, fsize - base);
        std::vector<char> buf(want);
        if (fseeko(f, (off_t)base, SEEK_SET) != 0) return fsize;
        const size_t n = fread(buf.data(), 1, want, f);
        const bool to_end = base + (long long)n >= fsize;
        // line starts inside the window
        std::vector<size_t> ls;
        for (size_t i = 0; i + 1 < n; ++i)
            if (buf[i] == '\n') ls.push_back(i + 1);
        for (size_t j = 0; j < ls.size(); ++j) {
            if (buf[ls[j]] != '@') continue;
            if (j + 2 < ls.size()) {
                if (buf[ls[j + 2]] == '+') return base + (long long)ls[j];
            } else if (!to_end) {
                break;  // the deciding line lies beyond the window: look again with a bigger one
            }
        }
        if (to_end) return fsize;
        if (window >= ((size_t)1 << 34)) return fsize;
        window <<= 2;
    }
}

// [begin, end) of part `part` of `parts`: cut points at records, part 0 starts at 0, the last part ends at the end of the file; the
// parts are disjoint and cover every record.
inline bool fastq_part_range(const std::string &path, unsigned part, unsigned parts, long long *begin, long long *end) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseeko(f, 0, SEEK_END);
    const long long fsize = (long long)ftello(f);
    const long long a = fsize / (long long)parts * (long long)part, b = part + 1 == parts ? fsize : fsize / (long long)parts * (long long)(part + 1);
    *begin = fastq_record_at_or_after(f, a, fsize);
    *end = part + 1 == parts ? fsize : fastq_record_at_or_after(f, b, fsize);
    fclose(f);
    return true;
}

}  // namespace smxtool
