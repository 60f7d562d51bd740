// integration/kmer_gpu_counter.hpp — the reference-side binding of libspades_mi355x.so (include/smx.h).
//
// This is the ONE class a SPAdes maintainer adds to drop the GPU path in: a kmers::KMerCounter<RtSeq>
// (/root/reference/src/common/kmer_index/kmer_mph/kmer_index_builder.hpp:259-282) whose Count() runs on the MI355X and hands back
// the reference's own result type, a KMerDiskStorage<RtSeq> whose bucket files kmers_XXXXXX.<b> hold the same bytes
// KMerDiskCounter::Count (:306-332) would have written. Every consumer downstream — KMerDiskStorage::merge (final_kmers),
// KMerIndexBuilder::BuildIndex, DeBruijnExtensionIndexBuilder::BuildExtensionIndexFromKPOMers, the Construction phases — keeps
// working unchanged; integration/Makefile compiles this header against the reference tree, and the two drivers next to it are
// run on the GPU box by tests/test_integration_gpu.py.
//
// Reads are handed to the library by a feeder callback, because where they come from differs per call site (FASTQ files for
// spades-kmercount, the binary .seq streams inside spades-core): the feeder calls smx_submit_reads_* on the context it is given.
#pragma once
extern "C" {
#include "smx.h"
}
#include "kmer_index/ph_map/kmer_maps.hpp"  // kmer_index_traits<RtSeq>
#include "kmer_index/kmer_mph/kmer_index_builder.hpp"
#include "kmer_index/kmer_mph/kmer_buckets.hpp"
#include "sequence/rtseq.hpp"
#include "utils/filesystem/temporary.hpp"
#include "utils/logger/logger.hpp"

#include <fstream>
#include <functional>
#include <stdexcept>
#include <vector>

namespace kmers {

class KMerGpuCounter : public KMerCounter<RtSeq> {
  public:
    using ReadFeeder = std::function<int(smx_ctx *)>;  // submits the reads (forward strand only); returns an smx error code

    // canonical_only: the construction's splitter (StoringTypeFilter<InvertableStoring>, kmer_splitters.hpp:28-44);
    // false: every k-mer of read and RC(read) (spades-kmercount, kmercount.cpp:65-83)
    KMerGpuCounter(fs::TmpDir work_dir, unsigned k, ReadFeeder feeder, bool canonical_only, int device = 0)
            : KMerCounter<RtSeq>(k), work_dir_(work_dir), feeder_(std::move(feeder)),
              mode_(canonical_only ? SMX_MODE_CANONICAL : SMX_MODE_ALL) {
        if (int rc = smx_create(&ctx_, device, 0))
            throw std::runtime_error("KMerGpuCounter: no usable MI355X (smx_create returned " + std::to_string(rc) + ")");
    }
    KMerGpuCounter(const std::filesystem::path &work_dir, unsigned k, ReadFeeder feeder, bool canonical_only, int device = 0)
            : KMerGpuCounter(fs::tmp::make_temp_dir(work_dir, "kmer_counter"), k, std::move(feeder), canonical_only, device) {}
    KMerGpuCounter(const KMerGpuCounter &) = delete;
    ~KMerGpuCounter() override { smx_destroy(ctx_); }

    size_t kmer_size() const override { return RtSeq::GetDataSize(this->k()) * sizeof(RtSeq::DataType); }
    smx_ctx *context() { return ctx_; }

    KMerDiskStorage<RtSeq> Count(unsigned num_buckets, unsigned num_threads) override {
        INFO("Counting k-mer instances into " << num_buckets << " buckets on the MI355X (" << num_threads << " host threads are not needed)");
        check(smx_reads_clear(ctx_));
        check(feeder_(ctx_));
        check(smx_count(ctx_, this->k(), mode_, num_buckets));
        // the reference's result object: bucket b = mulhi(XXH3(record), num_buckets), sorted unique records inside
        KMerDiskStorage<RtSeq> res(work_dir_, this->k(), kmer::KMerSegmentPolicy<RtSeq>(num_buckets));
        std::vector<uint64_t> sizes(num_buckets);
        check(smx_bucket_sizes(ctx_, sizes.data()));
        size_t kmers = 0;
        std::vector<char> buf;
        for (unsigned b = 0; b < num_buckets; ++b) {
            auto file = res.create(b);  // kmers_XXXXXX.<b>
            buf.resize(sizes[b] * kmer_size());
            if (sizes[b]) check(smx_copy_bucket(ctx_, b, buf.data()));
            std::ofstream os(file->file(), std::ios::out | std::ios::binary);
            os.write(buf.data(), (std::streamsize)buf.size());
            if (!os) throw std::runtime_error("I/O error writing " + file->file().string());
            kmers += sizes[b];
        }
        INFO("K-mer counting done. There are " << kmers << " kmers in total. ");
        return res;
    }

    KMerDiskStorage<RtSeq> CountAll(unsigned num_buckets, unsigned num_threads, bool merge = true) override {
        auto storage = Count(num_buckets, num_threads);
        if (merge) storage.merge();
        return storage;
    }

  private:
    void check(int rc) {
        if (rc) throw std::runtime_error(std::string("libspades_mi355x: ") + smx_last_error(ctx_) + " (code " + std::to_string(rc) + ")");
    }
    smx_ctx *ctx_ = nullptr;
    fs::TmpDir work_dir_;
    ReadFeeder feeder_;
    int mode_;
};

}  // namespace kmers
