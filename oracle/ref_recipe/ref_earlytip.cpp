// oracle/ref_recipe/ref_earlytip.cpp — TEST INFRASTRUCTURE, not product code.
//
// The reference's extension index, early tip clipper and unbranching-path extractor, compiled from the sources where they lie
// under /root/reference and driven on a one-sequence-per-line read file:
//   canonical (k+1)-mers              KMerDiskCounter + LineSplitter (mode B)                      kmer_index_builder.hpp:284-431
//   extension index                   DeBruijnExtensionIndexBuilder::BuildExtensionIndexFromKPOMers kmer_extension_index_builder.hpp:80-107
//   early tip clipper (bound > 0)     EarlyTipClipperProcessor(index, bound).ClipTips()            early_simplification.hpp:38-162
//   unitigs + perfect loops           UnbranchingPathExtractor::ExtractUnbranchingPathsAndLoops     debruijn_graph_constructor.hpp:399-406
// This is what spades-core's Construction stage runs between "Extension index construction" and "Condensing graph"
// (stages/construction.cpp:289-305, 345-369) with length_bound = RL - K by default.
//
//   early A/T remover (last argument "at"; RNA pipelines only, before the tip clipper):
//                                     EarlyLowComplexityClipperProcessor(index, 0.8, 10, 200).RemoveATEdges() + RemoveATTips()
//                                                                                                  early_simplification.hpp:164-347, construction.cpp:317-326
//   spades-core edge order ("sorted" among the trailing arguments): the unitigs sorted by Sequence::RawCompare, as
//                                     DeBruijnGraphExtentionConstructor::ConstructGraph does before ids are assigned   :590-604
//   k-mer file + InOutMask bytes ("kmers=<path>"): the index's k-mers in file order (kmer_begin .. kmer_end, the order the MPHF indexes)
//                                     as RtSeq words -> <path>, their masks (out bits 0-3, in bits 4-7) -> <path>.masks; "nounitigs" stops there
//   ref_earlytip <k> <nthreads> <tip_length_bound|0> <reads.txt> <workdir> <out.txt> [at] [sorted] [noloops] [kmers=<path>] [nounitigs]
//   out.txt: one edge sequence per line in the extractor's order (nthreads = 1 makes the order and the clipping deterministic)
#include "line_splitter.hpp"
#include "kmer_index/extension_index/kmer_extension_index_builder.hpp"
#include "assembly_graph/construction/early_simplification.hpp"
#include "assembly_graph/construction/debruijn_graph_constructor.hpp"

int main(int argc, char **argv) {
    if (argc < 7) {
        std::cerr << "usage: ref_earlytip <k> <nthreads> <tip_length_bound|0> <reads.txt> <workdir> <out.txt>\n";
        return 2;
    }
    unsigned k = (unsigned) atoi(argv[1]);
    unsigned nthreads = (unsigned) atoi(argv[2]);
    size_t bound = strtoull(argv[3], nullptr, 10);
    std::string reads = argv[4];
    std::filesystem::path workdir = argv[5];
    std::string outfile = argv[6];
    omp_set_num_threads((int) nthreads);
    create_console_logger();
    std::filesystem::create_directories(workdir);
    auto tmp = fs::tmp::make_temp_dir(workdir, "ref_earlytip");

    kmers::DeBruijnExtensionIndex<> index(k);
    {
        LineSplitter splitter(workdir, k + 1, reads, /*canonical_only=*/true, 0);
        kmers::KMerDiskCounter<RtSeq> counter(workdir, std::move(splitter));
        auto kpomers = counter.Count(10 * nthreads, nthreads);
        kmers::DeBruijnExtensionIndexBuilder().BuildExtensionIndexFromKPOMers(tmp, index, kpomers, nthreads, 0);
    }
    bool at = false, sorted = false, keep_loops = true, unitigs = true;
    std::string dump;
    for (int i = 7; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "at") at = true;
        if (a == "sorted") sorted = true;
        if (a == "noloops") keep_loops = false;
        if (a == "nounitigs") unitigs = false;
        if (a.rfind("kmers=", 0) == 0) dump = a.substr(6);
    }
    if (!dump.empty()) {  // before any clipper touches the masks: the extension index as BuildExtensionIndexFromKPOMers left it
        std::ofstream ok(dump, std::ios::binary), om(dump + ".masks", std::ios::binary);
        const size_t nw = RtSeq::GetDataSize(k);
        auto its = index.kmer_begin(1);  // (one part: the whole file, in order — as CollectLoops / ExtractUnbranchingPaths iterate it, :353-370)
        for (auto &it = its.front(); it.good(); ++it) {
            RtSeq kmer(k, *it);
            auto kwh = index.ConstructKWH(kmer);
            const auto mask = index.get_value(kwh);
            unsigned char b = 0;
            for (char c = 0; c < 4; ++c) {
                if (mask.CheckOutgoing(c)) b |= (unsigned char) (1u << c);
                if (mask.CheckIncoming(c)) b |= (unsigned char) (16u << c);
            }
            ok.write((const char *) kmer.data(), (std::streamsize) (nw * sizeof(RtSeq::DataType)));
            om.put((char) b);
        }
    }
    if (!unitigs) return 0;
    if (at) {
        debruijn_graph::EarlyLowComplexityClipperProcessor at_processor(index, 0.8, 10, 200);
        at_processor.RemoveATEdges();
        at_processor.RemoveATTips();
    }
    if (bound)
        debruijn_graph::EarlyTipClipperProcessor(index, bound).ClipTips();
    const unsigned nchunks = sorted ? 16 * nthreads : 10 * nthreads;  // :592 vs gbuilder
    auto seqs = keep_loops ? debruijn_graph::UnbranchingPathExtractor(index, k).ExtractUnbranchingPathsAndLoops(nchunks)
                           : debruijn_graph::UnbranchingPathExtractor(index, k).ExtractUnbranchingPaths(nchunks);
    if (sorted) std::sort(seqs.begin(), seqs.end(), Sequence::RawCompare);
    std::ofstream os(outfile);
    for (const auto &s : seqs) os << s.str() << "\n";
    return 0;
}
