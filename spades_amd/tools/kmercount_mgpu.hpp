// spades_amd/tools/kmercount_mgpu.hpp — spades-kmercount on N GPUs of one node, C++ host over librccl.
// One process per GPU (forked by the tool itself before any HIP call), bucket-range owners, ONE exchange:
//   own input files -> smx_extract_partition_owned (local pre-dedupe, records grouped by owner) -> counts by ncclAllGather ->
//   grouped ncclSend / ncclRecv between all pairs (every pair has its own xGMI link; no ring) -> smx_count_records on the owner ->
//   every rank writes its bucket range into <workdir>/final_kmers at its byte offset (buckets are contiguous per rank, so the file
//   is the concatenation of the rank outputs: KMerDiskStorage::merge, kmer_index_builder.hpp:190-203).
// The ncclUniqueId travels through a file in the work directory. Input: whole files round-robin when there is one per rank, else every
// file cut among the ranks.
#pragma once
#include <signal.h>
#include <fcntl.h>
#include <sys/prctl.h>
#include "rank_watchdog.hpp"
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "../../include/smx.h"
#include "read_share.hpp"

namespace smxtool {

#define MG_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "[rank %d] %s failed: %s\n", rank, #call, hipGetErrorString(e_));     \
            return SMX_DEVICE_ERROR;                                                              \
        }                                                                                         \
    } while (0)
#define MG_NCCL(call)                                                                             \
    do {                                                                                          \
        ncclResult_t r_ = (call);                                                                 \
        if (r_ != ncclSuccess) {                                                                  \
            fprintf(stderr, "[rank %d] %s failed: %s\n", rank, #call, ncclGetErrorString(r_));    \
            return SMX_DEVICE_ERROR;                                                              \
        }                                                                                         \
    } while (0)

inline int sharded_rank_main(int rank, int world, unsigned K, const std::string &workdir, const std::vector<std::string> &input) {
    smx_ctx *ctx = nullptr;
    smxtool::RankWatch::arm(rank);
    smxtool::RankWatch::mark("smx_create");
    if (int rc = smx_create(&ctx, rank, 0)) {
        fprintf(stderr, "[rank %d] no usable MI355X device %d (smx_create -> %d)\n", rank, rank, rc);
        return rc;
    }
    // communicator: rank 0 publishes the id through the work directory
    const std::string idfile = workdir + "/.smx_nccl_id";
    ncclUniqueId id;
    smxtool::RankWatch::mark("communicator: ncclGetUniqueId / ncclCommInitRank");
    if (rank == 0) {
        MG_NCCL(ncclGetUniqueId(&id));
        const std::string tmp = idfile + ".tmp";
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(&id, sizeof id, 1, f) != 1) return SMX_IO_ERROR;
        fclose(f);
        if (rename(tmp.c_str(), idfile.c_str()) != 0) return SMX_IO_ERROR;
    } else {
        bool got = false;
        for (int t = 0; t < 12000 && !got; ++t) {  // up to 2 minutes
            FILE *f = fopen(idfile.c_str(), "rb");
            if (f) {
                got = fread(&id, sizeof id, 1, f) == 1;
                fclose(f);
            }
            if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
        if (!got) {
            fprintf(stderr, "[rank %d] no communicator id from rank 0\n", rank);
            return SMX_IO_ERROR;
        }
    }
    MG_HIP(hipSetDevice(rank));
    ncclComm_t comm;
    MG_NCCL(ncclCommInitRank(&comm, world, id, rank));
    hipStream_t stream;
    MG_HIP(hipStreamCreate(&stream));
    smxtool::RankWatch::mark("communicator up: reading the input");
    // this rank's share of the input: whole files dealt out round-robin when there is at least one per rank (nothing is parsed twice);
    // otherwise (R1 / R2 on eight GPUs) every file is cut among all ranks (read_share.hpp: byte ranges of plain FASTQ, every world-th
    // read of anything else)
    const bool whole_files = input.size() >= (size_t)world;
    for (size_t i = 0; i < input.size(); ++i) {
        if (whole_files && (int)(i % (size_t)world) != rank) continue;
        int rc;
        try {
            rc = whole_files ? submit_file(ctx, input[i]) : submit_share(ctx, input[i], (unsigned)rank, (unsigned)world);
        } catch (const std::string &e) {
            fprintf(stderr, "%s\n", e.c_str());
            return SMX_INVALID_INPUT_FORMAT;
        }
        if (rc == -1) {
            fprintf(stderr, "File %s doesn't exist or can't be read!\n", input[i].c_str());
            return SMX_INPUT_FILE_NOT_FOUND;
        }
        if (rc) {
            fprintf(stderr, "%s\n", smx_last_error(ctx));
            return rc;
        }
    }
    smxtool::RankWatch::mark("input submitted: local extraction");
    const unsigned NB = 16, nw = (K + 31) / 32;  // 16 buckets: kmercount.cpp:220
    // records grouped by owner in a buffer of the library's pool, sized after the local pre-dedupe (not one record per k-mer instance)
    uint64_t *d_send = nullptr, *d_recv = nullptr, *d_cnt = nullptr, *d_all = nullptr;
    std::vector<uint64_t> counts(world, 0);
    {
        const void *p = nullptr;
        if (int rc = smx_extract_partition_owned(ctx, K, SMX_MODE_ALL, NB, (unsigned)world, &p, counts.data())) {
            fprintf(stderr, "%s\n", smx_last_error(ctx));
            return rc;
        }
        d_send = (uint64_t *)const_cast<void *>(p);
    }
    // counts of every pair
    smxtool::RankWatch::mark("extracted: all-gather of the counts");
    MG_HIP(hipMalloc((void **)&d_cnt, (size_t)world * 8));
    MG_HIP(hipMalloc((void **)&d_all, (size_t)world * world * 8));
    MG_HIP(hipMemcpy(d_cnt, counts.data(), (size_t)world * 8, hipMemcpyHostToDevice));
    MG_NCCL(ncclAllGather(d_cnt, d_all, (size_t)world, ncclUint64, comm, stream));
    MG_HIP(hipStreamSynchronize(stream));
    std::vector<uint64_t> all((size_t)world * world);
    MG_HIP(hipMemcpy(all.data(), d_all, all.size() * 8, hipMemcpyDeviceToHost));
    std::vector<uint64_t> soff(world + 1, 0), roff(world + 1, 0);
    for (int p = 0; p < world; ++p) {
        soff[p + 1] = soff[p] + counts[p];
        roff[p + 1] = roff[p] + all[(size_t)p * world + rank];  // what rank p sends to me
    }
    const uint64_t n_recv = roff[world];
    {  // receive side in the pool as well: smx_count_records sorts it in place
        void *p = nullptr;
        if (int rc = smx_exchange_buffer(ctx, n_recv * nw, &p)) {
            fprintf(stderr, "%s\n", smx_last_error(ctx));
            return rc;
        }
        d_recv = (uint64_t *)p;
    }
    // the exchange: all pairs at once, every pair on its own xGMI link, in rounds of <= 1 GiB per pair (one transfer of tens of GB was
    // seen to stop short on this stack: dist.py, _a2a); the segment that stays here is a device copy (SMX_MGPU_SELF_RCCL=1 sends it
    // through ncclSend / ncclRecv as well: the one-rank tests exercise the RCCL calls that way)
    smxtool::RankWatch::mark("exchange (grouped ncclSend / ncclRecv)");
    {
        const bool self_rccl = getenv("SMX_MGPU_SELF_RCCL") != nullptr;
        const uint64_t LIM = getenv("SMX_MGPU_ROUND_WORDS") ? (uint64_t)std::max(1LL, atoll(getenv("SMX_MGPU_ROUND_WORDS"))) : (uint64_t)1 << 27;  // words (env: test hook)
        uint64_t mx = 0;
        for (size_t i = 0; i < all.size(); ++i) mx = std::max<uint64_t>(mx, all[i] * nw);
        const uint64_t rounds = std::max<uint64_t>(1, (mx + LIM - 1) / LIM);
        for (uint64_t r = 0; r < rounds && (world > 1 || self_rccl); ++r) {
            MG_NCCL(ncclGroupStart());
            for (int p = 0; p < world; ++p) {
                if (p == rank && !self_rccl) continue;
                const uint64_t s1 = soff[p + 1] * nw, a = std::min(soff[p] * nw + r * LIM, s1), b = std::min(a + LIM, s1);
                const uint64_t r1 = roff[p + 1] * nw, e = std::min(roff[p] * nw + r * LIM, r1), f = std::min(e + LIM, r1);
                if (b > a) MG_NCCL(ncclSend(d_send + a, b - a, ncclUint64, p, comm, stream));
                if (f > e) MG_NCCL(ncclRecv(d_recv + e, f - e, ncclUint64, p, comm, stream));
            }
            MG_NCCL(ncclGroupEnd());
        smxtool::RankWatch::tick();
        }
        if (!self_rccl && counts[rank])
            MG_HIP(hipMemcpyAsync(d_recv + roff[rank] * nw, d_send + soff[rank] * nw, (size_t)counts[rank] * nw * 8, hipMemcpyDeviceToDevice, stream));
    }
    MG_HIP(hipStreamSynchronize(stream));
    smxtool::RankWatch::mark("exchanged: owner-side count");
    smx_extract_release(ctx);
    if (int rc = smx_count_records(ctx, K, NB, d_recv, n_recv)) {
        fprintf(stderr, "%s\n", smx_last_error(ctx));
        return rc;
    }
    // file offsets: records per rank
    smxtool::RankWatch::mark("counted: all-gather of the file offsets, writing");
    uint64_t n_mine = 0;
    smx_count_info(ctx, &n_mine, nullptr, nullptr);
    const std::string out = workdir + "/final_kmers";
    if (rank == 0) {  // created before the collective below, which orders it before everybody's writes
        FILE *f = fopen(out.c_str(), "wb");
        if (!f) {
            fprintf(stderr, "Cannot open %s for writing\n", out.c_str());
            return SMX_IO_ERROR;
        }
        fclose(f);
    }
    MG_HIP(hipMemcpy(d_cnt, &n_mine, 8, hipMemcpyHostToDevice));
    MG_NCCL(ncclAllGather(d_cnt, d_all, 1, ncclUint64, comm, stream));
    MG_HIP(hipStreamSynchronize(stream));
    std::vector<uint64_t> per(world);
    MG_HIP(hipMemcpy(per.data(), d_all, (size_t)world * 8, hipMemcpyDeviceToHost));
    uint64_t before = 0, total = 0;
    for (int p = 0; p < world; ++p) {
        if (p < rank) before += per[p];
        total += per[p];
    }
    {
        const int fd = open(out.c_str(), O_WRONLY);
        if (fd < 0) return SMX_IO_ERROR;
        std::vector<uint64_t> sizes(NB);
        smx_bucket_sizes(ctx, sizes.data());
        std::vector<char> buf;
        uint64_t at = before * nw * 8;
        for (unsigned b = 0; b < NB; ++b) {
            if (!sizes[b]) continue;
            buf.resize(sizes[b] * nw * 8);
            if (int rc = smx_copy_bucket(ctx, b, buf.data())) return rc;
            size_t done = 0;
            while (done < buf.size()) {
                const ssize_t w = pwrite(fd, buf.data() + done, buf.size() - done, (off_t)(at + done));
                if (w <= 0) {
                    close(fd);
                    return SMX_IO_ERROR;
                }
                done += (size_t)w;
            }
            at += buf.size();
        }
        close(fd);
    }
    // everybody has written before rank 0 reports
    smxtool::RankWatch::mark("written: last all-gather");
    MG_NCCL(ncclAllGather(d_cnt, d_all, 1, ncclUint64, comm, stream));
    MG_HIP(hipStreamSynchronize(stream));
    if (rank == 0) {
        printf("K-mer counting done. There are %llu kmers in total.\n", (unsigned long long)total);
        printf("K-mer counting done, kmers saved to \"%s\"\n", out.c_str());
        unlink(idfile.c_str());
    }
    // (the work is done: should giving the resources back not finish, the rank leaves with success after a grace period)
    smxtool::RankWatch::teardown_begins();
    smxtool::RankWatch::mark("teardown: hipFree");
    (void)hipFree(d_cnt);
    (void)hipFree(d_all);
    smxtool::RankWatch::mark("teardown: ncclCommDestroy");
    ncclCommDestroy(comm);
    smxtool::RankWatch::mark("teardown: hipStreamDestroy");
    (void)hipStreamDestroy(stream);
    smxtool::RankWatch::mark("teardown: smx_destroy");
    smx_destroy(ctx);
    smxtool::RankWatch::done();
    return 0;
}

// fork one process per GPU (nothing of HIP has been touched yet in this process) and wait for them
inline int run_sharded(int world, unsigned K, const std::string &workdir, const std::vector<std::string> &input) {
    // a rank that leaves between two collectives (unreadable input, a failed count, a write error) would leave the others waiting in
    // the next one for ever: what can be checked before the fork is checked here, and the first rank that fails takes the others down
    for (const std::string &f : input) {
        FILE *t = fopen(f.c_str(), "rb");
        if (!t) {
            fprintf(stderr, "File %s doesn't exist or can't be read!\n", f.c_str());
            return SMX_INPUT_FILE_NOT_FOUND;
        }
        fclose(t);
    }
    mkdir(workdir.c_str(), 0777);
    // RCCL between processes of one node shares device memory through dmabuf handles: the host driver of these boxes supports nothing else
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    unlink((workdir + "/.smx_nccl_id").c_str());
    std::vector<pid_t> kids;
    for (int r = 0; r < world; ++r) {
        const pid_t pid = fork();
        if (pid < 0) {
            for (pid_t k : kids) kill(k, SIGKILL);
            for (pid_t k : kids) waitpid(k, nullptr, 0);
            return SMX_DEVICE_ERROR;
        }
        if (pid == 0) {
            prctl(PR_SET_PDEATHSIG, SIGKILL);  // a rank never outlives the tool
            _exit(sharded_rank_main(r, world, K, workdir, input));
        }
        kids.push_back(pid);
    }
    int rc = 0;
    size_t left = kids.size();
    while (left) {
        int st = 0;
        const pid_t pid = waitpid(-1, &st, 0);
        if (pid < 0) break;
        bool ours = false;
        for (pid_t &k : kids)
            if (k == pid) {
                k = -1;
                ours = true;
            }
        if (!ours) continue;
        --left;
        const int code = WIFEXITED(st) ? WEXITSTATUS(st) : SMX_DEVICE_ERROR;
        if (code && !rc) {
            rc = code;
            fprintf(stderr, "a rank failed with code %d: stopping the other ranks\n", code);
            for (pid_t k : kids)
                if (k > 0) kill(k, SIGKILL);
        }
    }
    return rc;
}

}  // namespace smxtool
