// integration/construction_gpu_counter.cpp — the reference's construction classes consuming the GPU counter.
// The first phase of spades-core's Construction stage ("k+1-mer counting", stages/construction.cpp:215-257) builds
//     kmers::KMerDiskCounter<RtSeq> counter(workdir, Splitter(workdir, k + 1, read_streams, buffer));  kpomers = counter.Count(10 * nthreads, nthreads);
// and the next phases feed `kpomers` to DeBruijnExtensionIndexBuilder::BuildExtensionIndexFromKPOMers (:259-287) and
// UnbranchingPathExtractor (:343-369). Here the counter is kmers::KMerGpuCounter; everything behind it is the reference's own
// code, compiled from /root/reference: its MPHF index is built over bucket files the MI355X wrote.
//   construction_gpu_counter <k> <nthreads> <reads.txt (one sequence per line)> <workdir> <out.txt>
// out.txt: one unitig per line in the extractor's order — equal to what oracle/_ref/ref_earlytip (the same driver with the
// reference's own KMerDiskCounter) writes for the same input.
#include "kmer_gpu_counter.hpp"
#include "kmer_index/extension_index/kmer_extension_index_builder.hpp"
#include "assembly_graph/construction/debruijn_graph_constructor.hpp"
#include "utils/logger/log_writers.hpp"

#include <fstream>
#include <iostream>

static void create_console_logger() {
    using namespace logging;
    logger *lg = create_logger("");
    lg->add_writer(std::make_shared<console_writer>());
    attach_logger(lg);
}

int main(int argc, char **argv) {
    if (argc < 6) {
        std::cerr << "usage: construction_gpu_counter <k> <nthreads> <reads.txt> <workdir> <out.txt>\n";
        return 2;
    }
    const unsigned k = (unsigned)atoi(argv[1]), nthreads = (unsigned)atoi(argv[2]);
    const std::string reads = argv[3], outfile = argv[5];
    const std::filesystem::path workdir = argv[4];
    omp_set_num_threads((int)nthreads);
    create_console_logger();
    std::filesystem::create_directories(workdir);
    try {
        auto tmp = fs::tmp::make_temp_dir(workdir, "construction_gpu");
        kmers::DeBruijnExtensionIndex<> index(k);
        {
            auto feeder = [&](smx_ctx *ctx) -> int {  // one sequence per line; N handling (longest valid run) happens in the library
                std::ifstream is(reads);
                std::string line, bases;
                std::vector<uint64_t> off{0};
                while (std::getline(is, line)) {
                    bases += line;
                    off.push_back(bases.size());
                }
                return smx_submit_reads_ascii(ctx, bases.data(), off.data(), off.size() - 1);
            };
            kmers::KMerGpuCounter counter(workdir, k + 1, feeder, /*canonical_only=*/true);
            auto kpomers = counter.Count(10 * nthreads, nthreads);
            kmers::DeBruijnExtensionIndexBuilder().BuildExtensionIndexFromKPOMers(tmp, index, kpomers, nthreads, 0);
        }
        auto seqs = debruijn_graph::UnbranchingPathExtractor(index, k).ExtractUnbranchingPathsAndLoops(10 * nthreads);
        std::ofstream os(outfile);
        for (const auto &s : seqs) os << s.str() << "\n";
    } catch (std::exception const &e) {
        std::cerr << e.what() << std::endl;
        return EINTR;
    }
    return 0;
}
