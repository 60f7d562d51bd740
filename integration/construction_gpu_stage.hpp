// integration/construction_gpu_stage.hpp — spades-core's Construction stage on the MI355X.
//
// Drop-in for debruijn_graph::Construction (/root/reference/src/common/stages/construction.cpp:439-453), in the manner of
// hpcSPAdes' ConstructionMPI which replaces all phases at once (projects/hpcspades/mpi/stages/construction_mpi.cpp:303-412,758-762):
// one AssemblyStage whose run() hands the binary read files of the dataset to libspades_mi355x.so, builds the graph there
// (canonical (k+1)-mers -> k-mer file -> extension masks -> [early A/T remover, early tip clipper] -> unitigs in
// Sequence::RawCompare order -> link records -> coverage incl. flanking) and loads the result into the GraphPack through the
// reference's own reader of its graph_pack files (.grseq + .cvr, io::binary::BasicGraphIO).
// A pipeline registers it instead of Construction:      SPAdes.add<debruijn_graph::ConstructionGPU>()   (pipeline.cpp: add<Construction>()).
// integration/Makefile type-checks this header against the reference tree (target stage_check); linking it needs the whole of
// spades-core, which only the reference's own build produces.
#pragma once
extern "C" {
#include "smx.h"
}
#include "io/binary/graph_pack.hpp"
#include "io/dataset_support/read_converter.hpp"
#include "configs/config_struct.hpp"
#include "pipeline/graph_pack.hpp"
#include "pipeline/graph_pack_helpers.h"
#include "pipeline/stage.hpp"
#include "utils/filesystem/temporary.hpp"

namespace debruijn_graph {

class ConstructionGPU : public spades::AssemblyStage {
  public:
    ConstructionGPU() : spades::AssemblyStage("de Bruijn graph construction (MI355X)", "construction") {}

    void run(graph_pack::GraphPack &gp, const char *) override {
        const auto &params = cfg::get().con;
        const unsigned k = unsigned(gp.k()), nthreads = unsigned(cfg::get().max_threads);
        smx_ctx *ctx = nullptr;
        if (int rc = smx_create(&ctx, 0, 0)) FATAL_ERROR("ConstructionGPU: no usable MI355X (smx_create returned " << rc << ")");
        auto check = [&](int rc) {
            if (rc) FATAL_ERROR("libspades_mi355x: " << smx_last_error(ctx) << " (code " << rc << ")");
        };
        // behaviour switches of the reference's phases (stages/construction.cpp:289-326,343-369,446-448)
        check(smx_set_option(ctx, "sort_edges", 1));  // DeBruijnGraphExtentionConstructor::ConstructGraph sorts by Sequence::RawCompare
        check(smx_set_option(ctx, "keep_perfect_loops", params.keep_perfect_loops ? 1 : 0));
        if (config::PipelineHelper::IsRNAPipeline(cfg::get().mode)) check(smx_set_option(ctx, "early_at_remover", 1));
        auto &dataset = cfg::get_writable().ds;
        if (params.early_tc.enable && !cfg::get().gap_closer_enable) {
            const size_t bound = params.early_tc.length_bound ? *params.early_tc.length_bound : dataset.RL - k;
            check(smx_set_option(ctx, "early_tip_bound", int64_t(bound)));
        }
        // reads: the .seq files io::ReadConverter::ConvertToBinary wrote for this run (binary_converter.cpp:83-151); contigs of the
        // previous k and trusted contigs shape the graph but are not counted in the coverage (construction.cpp:89-117)
        for (size_t i = 0; i < dataset.reads.lib_count(); ++i) {
            auto &lib = dataset.reads[i];
            const bool contigs = lib.type() == io::LibraryType::TrustedContigs;
            if (!lib.is_graph_constructable() && !contigs) continue;
            io::ReadConverter::ConvertToBinary(lib);
            check(smx_set_option(ctx, "submit_contigs", contigs ? 1 : 0));
            const auto &info = lib.data().binary_reads_info;
            if (lib.has_paired()) check(smx_submit_reads_binary(ctx, (info.paired_read_prefix + ".seq").c_str()));
            if (lib.has_merged()) check(smx_submit_reads_binary(ctx, (info.merged_read_prefix + ".seq").c_str()));
            if (lib.has_single()) check(smx_submit_reads_binary(ctx, (info.single_read_prefix + ".seq").c_str()));
        }
        check(smx_set_option(ctx, "submit_contigs", 0));
        check(smx_build_graph(ctx, k, 10 * nthreads));  // bucket count of kmer_extension_index_builder.hpp:75
        check(smx_graph_fill_coverage(ctx));            // PHMCoverageFiller, construction.cpp:371-435
        // hand-over through the graph_pack surface the reference loads itself
        auto tmp = fs::tmp::make_temp_dir(cfg::get().tmp_dir, "construction_gpu");
        const std::filesystem::path base = tmp->dir() / "graph";
        check(smx_graph_write_spades(ctx, base.c_str()));
        smx_destroy(ctx);
        auto &graph = gp.get_mutable<Graph>();
        io::binary::BasicGraphIO<Graph> gio;
        if (!gio.Load(base.string(), graph)) FATAL_ERROR("ConstructionGPU: cannot load " << base);
    }
};

}  // namespace debruijn_graph
