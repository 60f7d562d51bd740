// integration/construction_gpu.cpp — spades-core's Construction stage on the MI355X, as a LINK-TIME replacement.
//
// This translation unit defines debruijn_graph::Construction — the class declared in the reference's own header
// (/root/reference/src/common/stages/construction.hpp) and registered by its pipeline (projects/spades/pipeline.cpp:193,
// SPAdes.add<debruijn_graph::Construction>()). Linked in front of the reference's archives it takes the place of
// common/stages/construction.cpp (libstages.a is only searched for symbols that are still undefined), so the spades-core binary
// integration/Makefile links (target spades-core-gpu, from the objects of the reference's own build tree) runs the whole
// assembly pipeline with the graph construction on the GPU; nothing of the reference is modified or copied.
//
// One phase replaces KMerCounting / ExtensionIndexBuilder / EarlyATClipper / EarlyTipClipper / GraphCondenser /
// PHMCoverageFiller (stages/construction.cpp:215-453): reads go to libspades_mi355x.so as the .seq files the read conversion
// wrote, the graph comes back through the reference's own graph_pack reader, coverage, flanking coverage and the multiplicity
// histogram through the C ABI. The "k-mer multiplicity estimation" phase (read_cov_threshold, CoverageFilter :172-213) has no
// counterpart: the stage refuses to run with it.
extern "C" {
#include "smx.h"
}
#include "stages/construction.hpp"

#include "alignment/edge_index.hpp"
#include "assembly_graph/graph_support/detail_coverage.hpp"
#include "configs/config_struct.hpp"
#include "io/binary/graph_pack.hpp"
#include "io/dataset_support/read_converter.hpp"
#include "pipeline/genomic_info.hpp"
#include "pipeline/graph_pack.hpp"
#include "pipeline/graph_pack_helpers.h"
#include "utils/filesystem/temporary.hpp"

namespace debruijn_graph {

struct ConstructionStorage {  // nothing is kept between phases: the graph lives in the library's context until it is loaded
    fs::TmpDir workdir;
};

void Construction::init(graph_pack::GraphPack &gp, const char *) {
    init_storage();
    storage().workdir = fs::tmp::make_temp_dir(gp.workdir(), "construction");
    auto &dataset = cfg::get_writable().ds;
    // dataset statistics the later stages rely on (restated from Construction::init, stages/construction.cpp:129-156)
    VERIFY(dataset.RL == 0 && dataset.aRL == 0.);
    size_t merged_max_len = 0, read_count = 0;
    uint64_t total_nucls = 0;
    for (size_t i = 0; i < dataset.reads.lib_count(); ++i) {
        if (!dataset.reads[i].is_graph_constructable()) continue;
        const auto &lib_data = dataset.reads[i].data();
        if (lib_data.unmerged_read_length == 0)
            FATAL_FORMAT_ERROR("Failed to determine read length for library #" << lib_data.lib_index << ". "
                               "Check that not only merged reads are present.");
        dataset.no_merge_RL = std::max(dataset.no_merge_RL, lib_data.unmerged_read_length);
        merged_max_len = std::max(merged_max_len, lib_data.merged_read_length);
        total_nucls += lib_data.total_nucls;
        read_count += lib_data.read_count;
    }
    dataset.RL = std::max(dataset.no_merge_RL, merged_max_len);
    INFO("Max read length " << dataset.RL);
    if (merged_max_len > 0) INFO("Max read length without merged " << dataset.no_merge_RL);
    dataset.aRL = double(total_nucls) / double(read_count);
    INFO("Average read length " << dataset.aRL);
}

void Construction::fini(graph_pack::GraphPack &) { reset_storage(); }

Construction::~Construction() {}

namespace {

class GpuConstruction : public Construction::Phase {
  public:
    GpuConstruction() : Construction::Phase("Graph construction on the MI355X", "construction_gpu") {}
    virtual ~GpuConstruction() = default;

    void run(graph_pack::GraphPack &gp, const char *) override {
        const auto &params = cfg::get().con;
        if (params.read_cov_threshold) FATAL_ERROR("read_cov_threshold (k-mer multiplicity estimation) is not available in the MI355X construction stage");
        const unsigned k = unsigned(gp.k()), nthreads = unsigned(cfg::get().max_threads);
        smx_ctx *ctx = nullptr;
        if (int rc = smx_create(&ctx, 0, 0)) FATAL_ERROR("no usable MI355X (smx_create returned " << rc << ")");
        auto check = [&](int rc) {
            if (rc) FATAL_ERROR("libspades_mi355x: " << smx_last_error(ctx) << " (code " << rc << ")");
        };
        auto &dataset = cfg::get_writable().ds;
        // behaviour switches of the reference's phases (stages/construction.cpp:289-326,343-369,446-448)
        check(smx_set_option(ctx, "sort_edges", 1));  // DeBruijnGraphExtentionConstructor::ConstructGraph sorts by Sequence::RawCompare
        check(smx_set_option(ctx, "keep_perfect_loops", params.keep_perfect_loops ? 1 : 0));
        if (config::PipelineHelper::IsRNAPipeline(cfg::get().mode)) check(smx_set_option(ctx, "early_at_remover", 1));
        if (params.early_tc.enable && !cfg::get().gap_closer_enable) {
            const size_t bound = params.early_tc.length_bound ? *params.early_tc.length_bound : dataset.RL - k;
            check(smx_set_option(ctx, "early_tip_bound", int64_t(bound)));
        }
        check(smx_set_option(ctx, "flank_range", int64_t(gp.get<omnigraph::FlankingCoverage<Graph>>().averaging_range())));
        // reads: the .seq files io::ReadConverter::ConvertToBinary wrote (binary_converter.cpp:83-151); trusted contigs and the
        // contigs of the previous k shape the graph but are not counted in the coverage (stages/construction.cpp:89-117)
        auto submit = [&](const std::string &prefix) {
            if (!prefix.empty() && std::filesystem::exists(prefix + ".seq")) check(smx_submit_reads_binary(ctx, (prefix + ".seq").c_str()));
        };
        for (size_t i = 0; i < dataset.reads.lib_count(); ++i) {
            auto &lib = dataset.reads[i];
            const bool contigs = lib.type() == io::LibraryType::TrustedContigs;
            if (!lib.is_graph_constructable() && !contigs) continue;
            io::ReadConverter::ConvertToBinary(lib);
            check(smx_set_option(ctx, "submit_contigs", contigs ? 1 : 0));
            const auto &info = lib.data().binary_reads_info;
            submit(info.paired_read_prefix);
            submit(info.merged_read_prefix);
            submit(info.single_read_prefix);
        }
        if (cfg::get().use_additional_contigs) {
            INFO("Contigs from previous K will be used: " << cfg::get().additional_contigs);
            check(smx_set_option(ctx, "submit_contigs", 1));
            submit((std::filesystem::path(cfg::get().additional_contigs) / "contigs").string());
        }
        check(smx_set_option(ctx, "submit_contigs", 0));
        INFO("Counting (k+1)-mers, building the extension index and condensing the graph on the MI355X");
        check(smx_build_graph(ctx, k, 10 * nthreads));  // bucket count of kmer_extension_index_builder.hpp:75
        {   // which of the library's routes the build took (with the default configuration — early_tip_clipper on — it is the one the bench measures)
            uint64_t rs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            check(smx_graph_route_stats(ctx, rs));
            INFO("Construction route: " << (rs[0] == 0 ? "partition-major (k-mers never sorted)" : rs[0] == 1 ? "k-mers + masks from one count, sorted" : "(k+1)-mer file first")
                 << ", " << rs[4] << " junction k-mers, " << rs[5] << " start de-edges");
        }
        check(smx_graph_fill_coverage(ctx));            // PHMCoverageFiller, stages/construction.cpp:371-435
        uint64_t info[8];
        check(smx_graph_info(ctx, info));
        INFO("Graph: " << info[2] << " edges (" << info[3] << " perfect loops) from " << info[0] << " (k+1)-mers and " << info[1] << " k-mers");
        // hand-over: the graph and its coverage through the graph_pack files the reference reads itself
        const std::filesystem::path base = storage().workdir->dir() / "graph";
        check(smx_graph_write_spades(ctx, base.c_str()));
        auto &index = gp.get_mutable<EdgeIndex<Graph>>();
        if (index.IsAttached()) index.Detach();
        auto &graph = gp.get_mutable<Graph>();
        io::binary::BasicGraphIO<Graph> gio;
        if (!gio.Load(base.string(), graph)) FATAL_ERROR("cannot load " << base);
        // flanking coverage of both orientations of every edge (FillCoverageAndFlankingFromPHM, coverage_filling.hpp:17-96)
        const uint64_t ne = info[2];
        std::vector<uint32_t> fl_e(ne), fl_c(ne);
        check(smx_graph_copy_flanking(ctx, fl_e.data(), fl_c.data()));
        auto &flanking = gp.get_mutable<omnigraph::FlankingCoverage<Graph>>();
        for (uint64_t i = 0; i < ne; ++i) {
            const EdgeId e(3 + 2 * i);  // ids as written: edge i -> 3 + 2 i, its conjugate the next id
            flanking.SetRawCoverage(e, fl_e[i]);
            const EdgeId ce = graph.conjugate(e);
            if (ce != e) flanking.SetRawCoverage(ce, fl_c[i]);
        }
        // multiplicity histogram (stages/construction.cpp:414-431: two k-mers per canonical record, index = multiplicity - 1)
        uint64_t nh = 0;
        check(smx_graph_coverage_histogram(ctx, nullptr, 0, &nh));
        std::vector<uint64_t> h(nh);
        check(smx_graph_coverage_histogram(ctx, h.data(), nh, &nh));
        std::vector<size_t> hist(nh > 0 ? nh - 1 : 0, 0);
        for (uint64_t c = 1; c < nh; ++c) hist[c - 1] = 2 * h[c];
        gp.get_mutable<GenomicInfo>().set_cov_histogram(hist);
        smx_destroy(ctx);
    }

    void load(graph_pack::GraphPack &, const std::filesystem::path &, const char *) override { VERIFY_MSG(false, "implement me"); }
    void save(const graph_pack::GraphPack &, const std::filesystem::path &, const char *) const override {}
};

}  // namespace

Construction::Construction() : spades::CompositeStageDeferred<ConstructionStorage>("de Bruijn graph construction", "construction") {
    add<GpuConstruction>();
}

}  // namespace debruijn_graph
