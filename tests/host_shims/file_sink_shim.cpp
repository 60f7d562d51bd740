// tests/host_shims/file_sink_shim.cpp — C entry for tests/test_file_sink_cpu.py: the product's output sink as it is (spades_amd/csrc/smx_file_sink.hpp).
#include "../../spades_amd/csrc/smx_file_sink.hpp"

#include <fcntl.h>

// writes src[0, n) to path in blocks of `block` bytes, the blocks in the order `order` (block indices), as the writers do (offset = index * block
// + head; the head bytes first); returns 1 if the sink mapped the file, 0 if it used pwrite, negative on failure
extern "C" int file_sink_write(const char *path, const char *head, uint64_t head_n, const char *src, uint64_t n, uint64_t block, const uint64_t *order,
                               uint64_t nblocks, int rdwr) {
    const int fd = open(path, (rdwr ? O_RDWR : O_WRONLY) | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return -1;
    smxio::FileSink sink;
    sink.begin(fd, head_n + n);
    const int mapped = sink.mapped() ? 1 : 0;
    bool ok = sink.put(head, head_n, 0);
    for (uint64_t i = 0; i < nblocks && ok; ++i) {
        const uint64_t o = order[i] * block, m = std::min<uint64_t>(block, n - o);
        ok = sink.put(src + o, m, (off_t)(head_n + o));
    }
    if (!sink.end()) ok = false;
    if (close(fd) != 0) ok = false;
    return ok ? mapped : -2;
}
