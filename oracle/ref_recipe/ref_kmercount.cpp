// oracle/ref_recipe/ref_kmercount.cpp — TEST INFRASTRUCTURE, not product code.
//
// A thin driver around the REAL reference classes, compiled from the sources where they lie
// under /root/reference (see Makefile): RtSeq (sequence/rtseq.hpp), XXH3 (ext/include/xxh),
// KMerSegmentPolicy (kmer_mph/kmer_buckets.hpp), KMerSortingSplitter (kmer_mph/kmer_splitter.hpp),
// KMerDiskCounter/KMerDiskStorage (kmer_mph/kmer_index_builder.hpp), pdqsort_pod, loser_tree,
// io::SingleRead + io::LongestValid (io/reads/longest_valid_wrapper.hpp) and, for mode B,
// StoringTypeFilter<InvertableStoring> (ph_map/storing_traits.hpp).
//
// It replaces only the FASTQ front-end (kseq/zlib-ng, MPMC ReadProcessor) of
// spades_tools/kmercount.cpp:48-122 by a one-sequence-per-line text reader, so the whole
// split -> sort -> unique -> k-way merge -> concat path that fixes the bytes of `final_kmers`
// is the reference's own code.
//
//   ref_kmercount <A|B> <K> <num_buckets> <bufsize|0> <reads.txt> <workdir> <out_file> [nthreads]
//     A: every K-mer of read and RC(read)            (spades-kmercount, kmercount.cpp:65-83)
//     B: only K-mers with IsMinimal(), read + RC     (DeBruijnReadKMerSplitter, kmer_splitters.hpp:28-44)
//   out_file = buckets 0..B-1 concatenated (KMerDiskStorage::merge) ; bucket sizes -> out_file.sizes
#include "line_splitter.hpp"

int main(int argc, char **argv) {
    if (argc < 8) {
        std::cerr << "usage: ref_kmercount <A|B> <K> <num_buckets> <bufsize|0> <reads.txt> <workdir> <out_file> [nthreads]\n";
        return 2;
    }
    bool canonical = argv[1][0] == 'B';
    unsigned K = (unsigned) atoi(argv[2]);
    unsigned nb = (unsigned) atoi(argv[3]);
    size_t bufsize = strtoull(argv[4], nullptr, 10);
    std::string reads = argv[5];
    std::filesystem::path workdir = argv[6];
    std::string outfile = argv[7];
    unsigned nthreads = argc > 8 ? (unsigned) atoi(argv[8]) : 1;

    create_console_logger();
    std::filesystem::create_directories(workdir);
    LineSplitter splitter(workdir, K, reads, canonical, bufsize);
    kmers::KMerDiskCounter<RtSeq> counter(workdir, std::move(splitter));
    auto res = counter.Count(nb, nthreads);
    {
        std::ofstream sz(outfile + ".sizes");
        for (size_t i = 0; i < res.num_buckets(); ++i) sz << res.bucket_size(i) << "\n";
    }
    res.merge();
    auto final_kmers = res.final_kmers();
    std::filesystem::rename(final_kmers->file(), outfile);
    return 0;
}
