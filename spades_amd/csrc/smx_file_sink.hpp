// spades_amd/csrc/smx_file_sink.hpp — where the big outputs (final_kmers: 16 B per k-mer, tens of GB; the GFA text) meet the file
// system. Host-only code (no HIP in here: tests/test_file_sink_cpu.py compiles it with g++ and compares files).
//
// The tools' outputs are measured on tmpfs, and there one writer thread was the wall: pwrite() on shmem allocates and copies under the
// inode lock — 6.4 GB/s from one thread, 3.4 GB/s from eight (measured on the GPU box in round 4), i.e. 6.3 s of the 7.7 s that
// spades-kmercount-mi355x takes for 20 M reads are this copy. A shared mapping of the file has no such lock: page faults on shmem
// allocate under the page's own lock, so several threads can fill DIFFERENT pages of one file at once. FileSink therefore maps a tmpfs
// output of known final size (ftruncate + mmap MAP_SHARED) and copies every block with a few threads; any other file system, a small
// file, a descriptor without read access, or a tmpfs without the room (a store into a mapped page that cannot be allocated is a
// SIGBUS, not an error code — so the room is checked first) keeps the pwrite path. SMX_WRITE_MMAP=0 / =1 forces the choice, -1 decides by
// the file system; SMX_WRITE_THREADS sets the copy threads (default 8). DEFAULT since round 5: pwrite (see begin()).
#pragma once
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/vfs.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace smxio {

class FileSink {
    int fd_ = -1;
    char *map_ = nullptr;
    size_t map_len_ = 0;
    bool ok_ = true;
    // copy threads (mapped mode): a block is cut into page-aligned slices, one per thread
    std::vector<std::thread> pool_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    uint64_t gen_ = 0;
    unsigned pending_ = 0;
    bool stop_ = false;
    const char *job_src_ = nullptr;
    char *job_dst_ = nullptr;
    size_t job_n_ = 0;

    void worker(unsigned t, unsigned nt) {
        uint64_t seen = 0;
        for (;;) {
            const char *src;
            char *dst;
            size_t n;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_work_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                src = job_src_;
                dst = job_dst_;
                n = job_n_;
            }
            // slice t of nt, cut at 4 KB boundaries of the DESTINATION (two threads never touch one page)
            const uintptr_t d0 = (uintptr_t)dst;
            auto cut = [&](unsigned i) -> size_t {
                if (i == 0) return 0;
                if (i >= nt) return n;
                const uintptr_t p = (d0 + (uintptr_t)(n * (uint64_t)i / nt)) & ~(uintptr_t)4095;
                return p <= d0 ? 0 : std::min<size_t>((size_t)(p - d0), n);
            };
            const size_t a = cut(t), b = cut(t + 1);
            if (b > a) memcpy(dst + a, src + a, b - a);
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (--pending_ == 0) cv_done_.notify_one();
            }
        }
    }

  public:
    FileSink() = default;
    FileSink(const FileSink &) = delete;
    FileSink &operator=(const FileSink &) = delete;
    ~FileSink() { (void)end(); }

    bool mapped() const { return map_ != nullptr; }

    // fd: the output file, open for writing (for the mapped mode: O_RDWR); file_size: its final size, all of which will be put()
    void begin(int fd, size_t file_size) {
        fd_ = fd;
        ok_ = true;
        // Round 5: the mapping is opt-in. On the GPU box (2 x EPYC 9575F, 256 threads) the driver measured 11.2 s for 38.5 GB through the
        // mapping with 8 threads against 6.3 s through one pwrite thread (BENCH_r04 vs the builder's run before the change): the 4.1–4.5 GB/s
        // of the mapping had been measured in an 8-core sandbox. SMX_WRITE_MMAP=-1 restores "decide by the file system".
        int want = 0;
        if (const char *e = getenv("SMX_WRITE_MMAP")) want = atoi(e) < 0 ? -1 : atoi(e) ? 1 : 0;
        if (want == 0 || file_size == 0) return;
        struct statfs sf;
        if (fstatfs(fd, &sf) != 0) return;
        const bool tmpfs = (unsigned long)sf.f_type == 0x01021994ul;  // TMPFS_MAGIC
        if (want < 0 && (!tmpfs || file_size < ((size_t)64 << 20))) return;
        if (tmpfs) {  // the pages must exist when they are stored to: leave the room for them (and a margin) or stay with pwrite's ENOSPC
            struct stat st;
            const uint64_t have = fstat(fd, &st) == 0 ? (uint64_t)st.st_blocks * 512ull : 0ull;
            const uint64_t room = (uint64_t)sf.f_bavail * (uint64_t)sf.f_bsize + have;
            if (room < (uint64_t)file_size + ((uint64_t)64 << 20)) return;
        }
        if (ftruncate(fd, (off_t)file_size) != 0) return;
        void *m = mmap(nullptr, file_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) return;  // (e.g. the descriptor is write-only)
        map_ = (char *)m;
        map_len_ = file_size;
        unsigned nt = 8;
        if (const char *e = getenv("SMX_WRITE_THREADS")) nt = (unsigned)std::max(1, atoi(e));
        nt = std::min(nt, std::max(1u, std::thread::hardware_concurrency()));
        if (nt > 1)
            for (unsigned t = 0; t < nt; ++t) pool_.emplace_back([this, t, nt] { worker(t, nt); });
    }

    // n bytes at file offset off; blocks until they are in the file's pages
    bool put(const char *src, size_t n, off_t off) {
        if (!ok_) return false;
        if (!n) return true;
        if (map_) {
            if ((uint64_t)off + n > map_len_) return ok_ = false;
            if (pool_.empty() || n < ((size_t)1 << 20)) {
                memcpy(map_ + off, src, n);
                return true;
            }
            std::unique_lock<std::mutex> lk(mu_);
            job_src_ = src;
            job_dst_ = map_ + off;
            job_n_ = n;
            pending_ = (unsigned)pool_.size();
            ++gen_;
            cv_work_.notify_all();
            cv_done_.wait(lk, [&] { return pending_ == 0; });
            return true;
        }
        size_t w = 0;
        while (w < n) {
            const ssize_t r = pwrite(fd_, src + w, n - w, off + (off_t)w);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) return ok_ = false;
            w += (size_t)r;
        }
        return true;
    }

    // everything put: releases the mapping and the threads; false if any put failed
    bool end() {
        if (!pool_.empty()) {
            {
                std::lock_guard<std::mutex> lk(mu_);
                stop_ = true;
            }
            cv_work_.notify_all();
            for (auto &t : pool_) t.join();
            pool_.clear();
            stop_ = false;
        }
        if (map_) {
            if (munmap(map_, map_len_) != 0) ok_ = false;
            map_ = nullptr;
            map_len_ = 0;
        }
        return ok_;
    }
};

}  // namespace smxio
