// spades_amd/tools/read_share.hpp — ONE rank's share of an input file, for the multi-GPU hosts (gbuilder_mgpu.hpp, kmercount_mgpu.hpp):
// a byte range of an uncompressed 4-line FASTQ file cut at records (fastq_split.hpp), every parts-th sequence of anything else.
// (The reference reads one stream, io/reads/file_reader.hpp; k-mer counting and graph construction do not depend on the order of the reads.)
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/smx.h"
#include "fastq_split.hpp"
#include "read_input.hpp"
#include "rank_watchdog.hpp"

namespace smxtool {

// part / parts: the share (rank * sub + i of world * sub; SMX_MGPU_PARTS = sub > 1 makes every rank read its share in `sub` pieces — a
// test hook that runs the range reader with one rank).
inline int submit_fastq_range(smx_ctx *ctx, const std::string &path, long long begin, long long end) {
    if (begin >= end) return 0;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return -1;
    if (fseeko(f, (off_t)begin, SEEK_SET) != 0) {
        fclose(f);
        return -1;
    }
    size_t chunk_bytes = (size_t)std::min<long long>((long long)256 << 20, std::max<long long>(end - begin + 4096, (long long)1 << 20));
    if (const char *e = getenv("SMX_MGPU_CHUNK")) chunk_bytes = (size_t)std::max(1024, atoi(e));  // test hook: the carry-over between chunks on small files
    char *buf = (char *)smx_pinned_alloc(chunk_bytes);
    const bool pinned = buf != nullptr;
    if (!buf) buf = (char *)malloc(chunk_bytes);
    long long pos = begin;
    size_t have = 0;
    int rc = 0;
    for (;;) {
        while (pos < end && have < chunk_bytes) {
            const size_t got = read_plain(f, buf + have, (size_t)std::min<long long>((long long)(chunk_bytes - have), end - pos));
            if (got == 0) {
                rc = SMX_IO_ERROR;
                break;
            }
            have += got;
            pos += (long long)got;
        }
        if (rc || have == 0) break;
        const bool last = pos >= end;
        uint64_t n = 0, used = 0;
        rc = smx_submit_fastq_text(ctx, buf, have, last ? 1 : 0, &n, &used);
        RankWatch::tick();
        if (rc) break;
        if (used == 0 && !last && have == chunk_bytes) {  // a single record larger than the chunk
            rc = SMX_INVALID_INPUT_FORMAT;
            break;
        }
        memmove(buf, buf + used, have - used);
        have -= used;
        if (last) break;
    }
    if (pinned) smx_pinned_free(buf); else free(buf);
    fclose(f);
    return rc;
}

// The same for a BGZF-compressed FASTQ file (bgzf_reader.hpp): the share is a range of the TEXT, cut at records; only the blocks that
// cover it are inflated (block-parallel), so N ranks inflate the file once between them — every rank on the zlib stream of an
// ordinary gzip file inflates all of it.
inline int submit_bgzf_range(smx_ctx *ctx, const BgzfText &bz, long long begin, long long end) {
    if (begin >= end) return 0;
    size_t chunk_bytes = (size_t)std::min<long long>((long long)256 << 20, std::max<long long>(end - begin + 4096, (long long)1 << 20));
    if (const char *e = getenv("SMX_MGPU_CHUNK")) chunk_bytes = (size_t)std::max(1024, atoi(e));
    char *buf = (char *)smx_pinned_alloc(chunk_bytes);
    const bool pinned = buf != nullptr;
    if (!buf) buf = (char *)malloc(chunk_bytes);
    const unsigned nt = io_threads();
    long long pos = begin;
    size_t have = 0;
    int rc = 0;
    for (;;) {
        if (pos < end && have < chunk_bytes) {
            const size_t n = (size_t)std::min<long long>((long long)(chunk_bytes - have), end - pos);
            if (!bz.read((uint64_t)pos, n, buf + have, nt)) {
                rc = SMX_INVALID_INPUT_FORMAT;
                break;
            }
            have += n;
            pos += (long long)n;
        }
        if (have == 0) break;
        const bool last = pos >= end;
        uint64_t n = 0, used = 0;
        rc = smx_submit_fastq_text(ctx, buf, have, last ? 1 : 0, &n, &used);
        RankWatch::tick();
        if (rc) break;
        if (used == 0 && !last && have == chunk_bytes) {  // a single record larger than the chunk
            rc = SMX_INVALID_INPUT_FORMAT;
            break;
        }
        memmove(buf, buf + used, have - used);
        have -= used;
        if (last) break;
    }
    if (pinned) smx_pinned_free(buf); else free(buf);
    return rc;
}

// 0, an smx error code, or -1 when the file cannot be read; throws std::string on malformed input (host parser)
inline int submit_share(smx_ctx *ctx, const std::string &path, unsigned part, unsigned parts) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return -1;
    unsigned char head[2] = {0, 0};
    const size_t nh = fread(head, 1, 2, f);
    fclose(f);
    const bool gz = nh == 2 && head[0] == 0x1f && head[1] == 0x8b;
    if (!gz && nh == 2 && head[0] == '@' && !getenv("SMX_HOST_PARSE") && fastq_head_is_four_line(path)) {
        long long b = 0, e = 0;
        if (!fastq_part_range(path, part, parts, &b, &e)) return -1;
        return submit_fastq_range(ctx, path, b, e);
    }
    if (gz && !getenv("SMX_HOST_PARSE") && !getenv("SMX_NO_BGZF") && BgzfReader::is_bgzf(path)) {
        BgzfText bz;
        if (bz.open(path) && bz.size() > 0) {
            // (every rank looks at the same head of the text, so every rank decides the same way)
            const size_t hn = (size_t)std::min<uint64_t>(bz.size(), (uint64_t)4 << 20);
            std::vector<char> head_text(hn);
            if (bz.read(0, hn, head_text.data(), io_threads()) && head_text[0] == '@' && fastq_text_is_four_line(head_text.data(), hn, hn == bz.size())) {
                const long long T = (long long)bz.size();
                auto at = [&](long long off, size_t n, char *dst) { return bz.read((uint64_t)off, n, dst, 1); };
                const long long a = T / (long long)parts * (long long)part, b = part + 1 == parts ? T : T / (long long)parts * (long long)(part + 1);
                const long long begin = fastq_record_at_or_after_in(at, a, T), end = part + 1 == parts ? T : fastq_record_at_or_after_in(at, b, T);
                return submit_bgzf_range(ctx, bz, begin, end);
            }
        }
    }
    // (ordinary) gzip, FASTA, multi-line FASTQ: every rank parses the file and keeps every parts-th sequence
    ReadBatch batch;
    int rc = 0;
    uint64_t idx = 0;
    const bool ok = for_each_sequence(path, [&](const std::string &s) {
        if (idx++ % parts != part) return;
        batch.add(s);
        if (batch.bases.size() > ((size_t)1 << 30) && !rc) {
            rc = smx_submit_reads_ascii(ctx, batch.bases.data(), batch.off.data(), batch.size());
            RankWatch::tick();
            batch.clear();
        }
    });
    if (!ok) return -1;
    if (!rc) rc = smx_submit_reads_ascii(ctx, batch.bases.data(), batch.off.data(), batch.size());
    return rc;
}

}  // namespace smxtool
