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

// Is this text (the head of a file; `whole`: all of it) strict 4-line FASTQ ('@' line, sequence, '+' line, quality of the same length)?
// Looks at up to `max_records` records.
inline bool fastq_text_is_four_line(const char *buf, size_t n, bool whole, size_t max_records = 256) {
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

// The same for the head of a file. Every reader looks at the same bytes, so every reader decides the same way.
inline bool fastq_head_is_four_line(const std::string &path, size_t max_records = 256, size_t max_bytes = (size_t)4 << 20) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<char> buf(max_bytes);
    const size_t n = fread(buf.data(), 1, buf.size(), f);
    fclose(f);
    return fastq_text_is_four_line(buf.data(), n, n < buf.size(), max_records);
}

// A window of text whose first byte is the byte BEFORE the position asked for: offset (>= 1) of the first record start in it — a line
// that begins with '@' whose next-but-one line begins with '+' (a quality line may begin with '@' too, but then the next-but-one line is
// a sequence line, which never begins with '+') —, kNoRecord when the window reaches the end of the text (`to_end`) and holds none,
// kNeedMore when the deciding line lies beyond the window.
constexpr long long kNoRecord = -2, kNeedMore = -1;
inline long long fastq_record_in_window(const char *buf, size_t n, bool to_end) {
    std::vector<size_t> ls;  // line starts inside the window
    for (size_t i = 0; i + 1 < n; ++i)
        if (buf[i] == '\n') ls.push_back(i + 1);
    for (size_t j = 0; j < ls.size(); ++j) {
        if (buf[ls[j]] != '@') continue;
        if (j + 2 < ls.size()) {
            if (buf[ls[j + 2]] == '+') return (long long)ls[j];
        } else if (!to_end) {
            return kNeedMore;
        }
    }
    return to_end ? kNoRecord : kNeedMore;
}

// Start of the first record at or after byte `from` of a text of `size` bytes that `read(offset, n, dst)` hands out (a file, the text
// of a BGZF file): `size` when there is none.
template <class ReadAt>
inline long long fastq_record_at_or_after_in(ReadAt read, long long from, long long size) {
    if (from <= 0) return 0;
    if (from >= size) return size;
    size_t window = (size_t)1 << 20;
    for (;;) {
        const long long base = from - 1;  // the byte before: a '\n' there makes `from` itself a line start
        const size_t want = (size_t)std::min<long long>((long long)window, size - base);
        std::vector<char> buf(want);
        if (!read(base, want, buf.data())) return size;
        const bool to_end = base + (long long)want >= size;
        const long long r = fastq_record_in_window(buf.data(), want, to_end);
        if (r >= 0) return base + r;
        if (r == kNoRecord || window >= ((size_t)1 << 34)) return size;
        window <<= 2;
    }
}

inline long long fastq_record_at_or_after(FILE *f, long long from, long long fsize) {
    return fastq_record_at_or_after_in(
        [&](long long off, size_t n, char *dst) { return fseeko(f, (off_t)off, SEEK_SET) == 0 && fread(dst, 1, n, f) == n; }, from, fsize);
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
