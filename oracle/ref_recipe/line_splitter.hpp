// oracle/ref_recipe/line_splitter.hpp — TEST INFRASTRUCTURE, not product code.
// The one-sequence-per-line front-end shared by the reference harnesses (ref_kmercount, ref_earlytip): it replaces only the FASTQ
// reader of the reference tools; everything behind it (KMerSortingSplitter buffers, sort, unique, merge) is the reference's own code.
#pragma once
#include "kmer_index/ph_map/kmer_maps.hpp"
#include "kmer_index/kmer_mph/kmer_index_builder.hpp"
#include "kmer_index/kmer_mph/kmer_splitter.hpp"
#include "kmer_index/ph_map/storing_traits.hpp"
#include "io/reads/single_read.hpp"
#include "io/reads/longest_valid_wrapper.hpp"
#include "sequence/rtseq.hpp"
#include "utils/logger/log_writers.hpp"

#include <fstream>
#include <iostream>
#include <string>

static void create_console_logger() {
    using namespace logging;
    logger *lg = create_logger("");
    lg->add_writer(std::make_shared<console_writer>());
    attach_logger(lg);
}

class LineSplitter : public kmers::KMerSortingSplitter<RtSeq> {
    std::string file_;
    bool canonical_only_;
    size_t bufsize_;

    bool Fill(const Sequence &seq, unsigned tid) {
        if (seq.size() < this->K_)
            return false;
        bool stop = false;
        RtSeq kmer = seq.start<RtSeq>(this->K_) >> 'A';
        for (size_t j = this->K_ - 1; j < seq.size(); ++j) {
            kmer <<= seq[j];
            if (canonical_only_ && !kmers::StoringTypeFilter<kmers::InvertableStoring>::filter(kmer))
                continue;
            stop |= this->push_back_internal(kmer, tid);
        }
        return stop;
    }

  public:
    using kmers::KMerSortingSplitter<RtSeq>::RawKMers;
    LineSplitter(const std::filesystem::path &workdir, unsigned K, std::string file, bool canonical_only, size_t bufsize)
            : kmers::KMerSortingSplitter<RtSeq>(workdir, K), file_(std::move(file)),
              canonical_only_(canonical_only), bufsize_(bufsize) {}

    // Like ParallelSortingSplitter::Split (kmercount.cpp:96-121): fill per-thread buffers from a
    // batch of reads with `nthreads` workers, dump (sort+unique+append run) whenever a cell overflows.
    RawKMers Split(size_t num_files, unsigned nthreads) override {
        auto out = PrepareBuffers(num_files, nthreads, bufsize_);
        std::ifstream is(file_);
        std::string line;
        std::vector<std::string> batch;
        const size_t batch_reads = 4096 * (size_t)nthreads;
        bool eof = false;
        size_t n = 0;
        while (!eof) {
            batch.clear();
            while (batch.size() < batch_reads) {
                if (!std::getline(is, line)) { eof = true; break; }
                if (!line.empty() && line.back() == '\r') line.pop_back();
                batch.push_back(line);
            }
            bool stop = false;
#           pragma omp parallel for num_threads(nthreads) reduction(|| : stop) schedule(dynamic, 256)
            for (size_t i = 0; i < batch.size(); ++i) {
                if (batch[i].empty()) continue;
                // SingleRead validates lazily; LongestValid cuts to the longest ACGT run (first on ties)
                io::SingleRead r(std::to_string(n + i), batch[i]);
                io::LongestValid(r);
                if (r.size() == 0) continue;  // kmercount.cpp:65-83 never consults IsValid()
                unsigned tid = (unsigned) omp_get_thread_num();
                stop = Fill(r.sequence(), tid) || stop;
                stop = Fill(r.sequence(/* rc */ true), tid) || stop;
            }
            n += batch.size();
            if (stop) DumpBuffers(out);
        }
        DumpBuffers(out);
        this->ClearBuffers();
        return out;
    }
};

