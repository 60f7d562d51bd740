// spades_amd/tools/bgzf_reader.hpp — block-parallel inflate of BGZF files (the blocked gzip variant of the SAM/BAM specification, section 4.1,
// which is what Illumina's BCL Convert / DRAGEN write for *.fastq.gz by default): every block is a gzip member of its own of at most
// 64 KiB whose header says how long it is (extra subfield 'B','C': BSIZE) and whose trailer says how much it inflates to (ISIZE), so the
// blocks of a slab can be inflated by several threads straight to their final places. The reference reads *.gz through zlib on one thread
// (io/reads/parser.cpp -> kseq/gzread); an ordinary gzip file is not BGZF and keeps that path here too.
// No dependencies besides zlib: the CPU test tier compiles this header alone.
#pragma once
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace smxtool {

class BgzfReader {
    struct Block {
        size_t in_off, in_len, out_off;  // deflate data inside the slab; place of the text in the destination
        uint32_t isize, crc;
    };
    int fd_ = -1;
    uint64_t fsize_ = 0, pos_ = 0;
    unsigned nthreads_ = 1;
    std::vector<unsigned char> slab_;

    // header of the block at p (n bytes available): total block length and header length; 0 = not a BGZF block header / not enough bytes
    static size_t parse_header(const unsigned char *p, size_t n, size_t *hdr_len) {
        if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
        const size_t xlen = (size_t)p[10] | ((size_t)p[11] << 8);
        if (n < 12 + xlen) return 0;
        size_t q = 12, bsize = 0;
        while (q + 4 <= 12 + xlen) {  // extra subfields: SI1 SI2 SLEN(2) data
            const size_t slen = (size_t)p[q + 2] | ((size_t)p[q + 3] << 8);
            if (p[q] == 'B' && p[q + 1] == 'C' && slen == 2 && q + 6 <= 12 + xlen) bsize = ((size_t)p[q + 4] | ((size_t)p[q + 5] << 8)) + 1;
            q += 4 + slen;
        }
        if (p[3] & ~4) return 0;  // (file name, comment, header CRC: BGZF writers do not set them)
        *hdr_len = 12 + xlen;
        return bsize >= *hdr_len + 8 ? bsize : 0;
    }

  public:
    static size_t header(const unsigned char *p, size_t n, size_t *hdr_len) { return parse_header(p, n, hdr_len); }
    ~BgzfReader() {
        if (fd_ >= 0) close(fd_);
    }
    static bool is_bgzf(const std::string &path) {
        unsigned char h[64];
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        const ssize_t n = pread(fd, h, sizeof h, 0);
        close(fd);
        size_t hl = 0;
        return n >= 18 && parse_header(h, (size_t)n, &hl) != 0;
    }
    bool open(const std::string &path, unsigned nthreads) {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) return false;
        struct stat st;
        if (fstat(fd_, &st) != 0) return false;
        fsize_ = (uint64_t)st.st_size;
        pos_ = 0;
        nthreads_ = nthreads ? nthreads : 1;
        return true;
    }
    bool eof() const { return pos_ >= fsize_; }
    uint64_t pos() const { return pos_; }  // file offset of the next block (a failed read() leaves it where it was)
    static constexpr size_t kError = ~(size_t)0;

    // Text of the next whole blocks, as many as fit into cap bytes (cap >= 64 KiB): returns the number of bytes, 0 at the end of the
    // file, kError on a malformed or damaged file (CRC32 and ISIZE of every block are checked, as gzread does).
    size_t read(char *dst, size_t cap) {
        if (eof()) return 0;
        if (cap < 65536) return kError;
        // compressed slab: FASTQ deflates to a quarter or so, so a third of cap bytes of input fills most of cap (what does not fit is
        // read again by the next call; incompressible input just gives shorter returns)
        const size_t want = (size_t)std::min<uint64_t>(fsize_ - pos_, (uint64_t)cap / 3 + 65536);
        slab_.resize(want);
        size_t got = 0;
        while (got < want) {
            const ssize_t r = pread(fd_, slab_.data() + got, want - got, (off_t)(pos_ + got));
            if (r <= 0) return kError;
            got += (size_t)r;
        }
        const bool more = pos_ + got < fsize_;  // the file goes on behind the slab
        std::vector<Block> blocks;
        size_t p = 0, out = 0;
        while (p < got) {
            size_t hl = 0;
            const size_t bl = parse_header(slab_.data() + p, got - p, &hl);
            if (bl == 0) {
                if (more && !blocks.empty()) break;  // a header cut by the end of the slab: the next call starts there
                return kError;
            }
            if (p + bl > got) {
                if (!more || blocks.empty()) return kError;  // the file ends inside a block / a block larger than the slab
                break;
            }
            const unsigned char *t = slab_.data() + p + bl - 8;
            const uint32_t crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
            const uint32_t isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
            if (isize > 65536) return kError;
            if (out + isize > cap) break;
            blocks.push_back({p + hl, bl - hl - 8, out, isize, crc});
            out += isize;
            p += bl;
        }
        if (blocks.empty()) return kError;
        std::atomic<size_t> next{0};
        std::atomic<bool> bad{false};
        auto work = [&]() {
            z_stream z;
            memset(&z, 0, sizeof z);
            if (inflateInit2(&z, -15) != Z_OK) {
                bad = true;
                return;
            }
            for (;;) {
                const size_t b0 = next.fetch_add(16);  // a few blocks at a time
                if (b0 >= blocks.size() || bad) break;
                for (size_t b = b0; b < std::min(b0 + 16, blocks.size()); ++b) {
                    const Block &k = blocks[b];
                    inflateReset(&z);
                    z.next_in = slab_.data() + k.in_off;
                    z.avail_in = (uInt)k.in_len;
                    z.next_out = (Bytef *)dst + k.out_off;
                    z.avail_out = k.isize;
                    const int rc = inflate(&z, Z_FINISH);
                    if (rc != Z_STREAM_END || z.avail_out != 0 || z.avail_in != 0 ||
                        (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)dst + k.out_off, k.isize) != k.crc) {
                        bad = true;
                        break;
                    }
                }
            }
            inflateEnd(&z);
        };
        const unsigned nt = (unsigned)std::min<size_t>(nthreads_, (blocks.size() + 15) / 16);
        std::vector<std::thread> th;
        for (unsigned i = 1; i < nt; ++i) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
        if (bad) return kError;
        pos_ += p;
        if (out == 0 && !eof()) return read(dst, cap);  // (only empty blocks so far: 0 means the end of the file to the caller)
        return out;
    }
};

// Random access to the TEXT of a BGZF file: the block index (where every block starts, how much text lies in front of it) comes from
// one walk over the headers and trailers; read() inflates only the blocks that cover the range asked for. This is how several readers
// (the ranks of the multi-GPU hosts) take disjoint parts of one *.fastq.gz.
class BgzfText {
    int fd_ = -1;
    std::vector<uint64_t> coff_, toff_;  // compressed offset of block b; text in front of block b (toff_[n] = all of it)
    std::vector<uint32_t> clen_, hlen_;

  public:
    ~BgzfText() {
        if (fd_ >= 0) close(fd_);
    }
    uint64_t size() const { return toff_.empty() ? 0 : toff_.back(); }
    size_t blocks() const { return clen_.size(); }
    bool open(const std::string &path) {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) return false;
        struct stat st;
        if (fstat(fd_, &st) != 0) return false;
        const uint64_t fsize = (uint64_t)st.st_size;
        std::vector<unsigned char> slab((size_t)16 << 20);
        uint64_t pos = 0, text = 0;
        toff_.assign(1, 0);
        while (pos < fsize) {
            const size_t want = (size_t)std::min<uint64_t>(fsize - pos, slab.size());
            size_t got = 0;
            while (got < want) {
                const ssize_t r = pread(fd_, slab.data() + got, want - got, (off_t)(pos + got));
                if (r <= 0) return false;
                got += (size_t)r;
            }
            size_t p = 0;
            while (p < got) {
                size_t hl = 0;
                const size_t bl = BgzfReader::header(slab.data() + p, got - p, &hl);
                if (bl == 0 || p + bl > got) break;  // cut by the end of the slab (or not a block: decided below)
                const unsigned char *t = slab.data() + p + bl - 4;
                const uint32_t isize = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
                if (isize > 65536) return false;
                coff_.push_back(pos + p);
                clen_.push_back((uint32_t)bl);
                hlen_.push_back((uint32_t)hl);
                text += isize;
                toff_.push_back(text);
                p += bl;
            }
            if (p == 0) return false;  // not a block header, or a block that ends behind the file
            pos += p;
        }
        return true;
    }
    // text bytes [off, off + n) -> dst; false on a damaged file or a range outside the text
    bool read(uint64_t off, size_t n, char *dst, unsigned nthreads = 1) const {
        if (off + n > size()) return false;
        if (n == 0) return true;
        const size_t b0 = (size_t)(std::upper_bound(toff_.begin(), toff_.end(), off) - toff_.begin()) - 1;
        size_t b1 = b0;
        while (toff_[b1 + 1] < off + n) ++b1;
        std::atomic<size_t> next{b0};
        std::atomic<bool> bad{false};
        auto work = [&]() {
            z_stream z;
            memset(&z, 0, sizeof z);
            if (inflateInit2(&z, -15) != Z_OK) {
                bad = true;
                return;
            }
            std::vector<unsigned char> in(65536 + 64), text(65536);
            for (;;) {
                const size_t b = next.fetch_add(1);
                if (b > b1 || bad) break;
                const uint32_t isize = (uint32_t)(toff_[b + 1] - toff_[b]);
                if (isize == 0) continue;
                size_t got = 0;
                while (got < clen_[b]) {
                    const ssize_t r = pread(fd_, in.data() + got, clen_[b] - got, (off_t)(coff_[b] + got));
                    if (r <= 0) break;
                    got += (size_t)r;
                }
                const unsigned char *t = in.data() + clen_[b] - 8;
                const uint32_t crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
                // a block that lies inside the range goes straight to its place, the two at the ends through a buffer
                const uint64_t lo = std::max<uint64_t>(toff_[b], off), hi = std::min<uint64_t>(toff_[b + 1], off + n);
                const bool inside = lo == toff_[b] && hi == toff_[b + 1];
                unsigned char *out = inside ? (unsigned char *)dst + (toff_[b] - off) : text.data();
                inflateReset(&z);
                z.next_in = in.data() + hlen_[b];
                z.avail_in = (uInt)(clen_[b] - hlen_[b] - 8);
                z.next_out = out;
                z.avail_out = isize;
                if (got != clen_[b] || inflate(&z, Z_FINISH) != Z_STREAM_END || z.avail_out != 0 ||
                    (uint32_t)crc32(crc32(0L, Z_NULL, 0), out, isize) != crc) {
                    bad = true;
                    break;
                }
                if (!inside) memcpy(dst + (lo - off), text.data() + (lo - toff_[b]), (size_t)(hi - lo));
            }
            inflateEnd(&z);
        };
        const unsigned nt = (unsigned)std::min<size_t>(nthreads ? nthreads : 1, (b1 - b0) / 8 + 1);
        std::vector<std::thread> th;
        for (unsigned i = 1; i < nt; ++i) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
        return !bad;
    }
};

// threads for the block-parallel inflate of one file (SMX_IO_THREADS; default: the cores, at most 16)
inline unsigned io_threads() {
    if (const char *e = getenv("SMX_IO_THREADS")) return (unsigned)std::max(1, atoi(e));
    const unsigned hc = std::thread::hardware_concurrency();
    return std::max(1u, std::min(16u, hc ? hc : 1u));
}

}  // namespace smxtool
