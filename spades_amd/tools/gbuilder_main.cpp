// spades_amd/tools/gbuilder_main.cpp — drop-in CLI for `spades-gbuilder`
// (reference: projects/spades_tools/gbuilder.cpp:66-245; docs/standalone.md) over libspades_mi355x.so.
//   spades-gbuilder-mi355x <fasta/fastq[.gz]> <out> [-k 21] [-c] [-t N] [-tmp-dir d] [-b n] [--unitigs|--fastg|--gfa|--spades] [--gpus N]
// --gpus N: one process per GPU, every rank reads its share of the input, k-mers and masks meet at their bucket owners by one RCCL
// exchange, the compact structure is gathered where the graph is built (gbuilder_mgpu.hpp). Same bytes as the single-GPU run.
// -t selects the bucket count 10*t and therefore the unitig/segment numbering of the reference run being
// reproduced (SURVEY.md finding 3); default = the reference's default (cores/2+1 is host dependent, so 1 here).
// Not in this build: YAML datasets.
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/smx.h"
#include "read_input.hpp"
#include "gbuilder_mgpu.hpp"
#include <chrono>
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define STAGE(what)                                                                 \
    if (getenv("SMX_DEBUG")) {                                                      \
        double t_ = now_s();                                                        \
        fprintf(stderr, "[tool] %-12s %7.3f s\n", what, t_ - t_stage);             \
        t_stage = t_;                                                               \
    }

int main(int argc, char **argv) {
    unsigned k = 21, nthreads = 1;
    int gpus = 0;
    std::string file, outfile;
    enum { UNITIGS, GFA, SPADES, FASTG } mode = UNITIGS;
    bool coverage = false;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto need = [&]() -> const char * {
            if (i + 1 >= argc) {
                fprintf(stderr, "Invalid command line arguments\n");
                exit(SMX_INVALID_PARAMETER);
            }
            return argv[++i];
        };
        if (a == "-k") k = (unsigned)atoi(need());
        else if (a == "-t") nthreads = (unsigned)atoi(need());
        else if (a == "-tmp-dir" || a == "-b") (void)need();
        else if (a == "--gpus") gpus = atoi(need());
        else if (a == "--unitigs" || a == "-unitigs") mode = UNITIGS;
        else if (a == "--gfa" || a == "-gfa") mode = GFA;
        else if (a == "-c") coverage = true;
        else if (a == "--spades" || a == "-spades") mode = SPADES;
        else if (a == "--fastg" || a == "-fastg") mode = FASTG;
        else if (!a.empty() && a[0] == '-') {
            fprintf(stderr, "Invalid command line arguments\n");
            return SMX_INVALID_PARAMETER;
        } else pos.push_back(a);
    }
    if (pos.size() != 2) {
        fprintf(stderr, "usage: %s <dataset description (in YAML) or input FASTA file> <output filename> [-k value] [-t value] "
                        "[-tmp-dir dir] [-b value] [--unitigs|--fastg|--gfa|--spades]\n", argv[0]);
        return SMX_INVALID_PARAMETER;
    }
    file = pos[0];
    outfile = pos[1];
    if (k < 1) { fprintf(stderr, "k-mer size %u is too low\n", k); return SMX_INVALID_PARAMETER; }
    if (k >= 128) { fprintf(stderr, "k-mer size %u is too high\n", k); return SMX_INVALID_PARAMETER; }
    if (k % 2 == 0) { fprintf(stderr, "k-mer size must be odd\n"); return SMX_INVALID_PARAMETER; }
    if (gpus < 0 || gpus > 64) { fprintf(stderr, "Invalid command line arguments\n"); return SMX_INVALID_PARAMETER; }
    if (gpus > 0) {  // one process per GPU; nothing of HIP may be touched in this process before the fork
        smxtool::GbOptions o;
        o.k = k, o.nthreads = nthreads, o.coverage = coverage, o.mode = mode == GFA ? 1 : mode == SPADES ? 2 : mode == FASTG ? 3 : 0, o.outfile = outfile;
        if (file.size() > 5 && file.compare(file.size() - 5, 5, ".yaml") == 0) {
            std::vector<smxtool::DatasetLibrary> libs;
            if (!smxtool::load_dataset_yaml(file, libs)) {
                fprintf(stderr, "Dataset description file: %s does not exist or is not a valid YAML file\n", file.c_str());
                return SMX_INPUT_FILE_NOT_FOUND;
            }
            for (const auto &lib : libs)
                if (lib.graph_constructable()) o.files.insert(o.files.end(), lib.files.begin(), lib.files.end());
        } else {
            o.files.push_back(file);
        }
        printf("K-mer length set to %u\n", k);
        fflush(stdout);
        const int rc = smxtool::gb_run_sharded(gpus, o);
        if (!rc) printf("SPAdes standalone graph builder finished\n");
        return rc;
    }
    smx_ctx *ctx = nullptr;
    double t_stage = now_s();
    if (int rc = smx_create(&ctx, 0, 0)) {
        fprintf(stderr, "No usable MI355X device (smx_create -> %d)\n", rc);
        return rc;
    }
    printf("K-mer length set to %u\n", k);
    int rc = 0;
    try {
        STAGE("device init")
        std::vector<std::string> files;
        if (file.size() > 5 && file.compare(file.size() - 5, 5, ".yaml") == 0) {  // LoadDataset, gbuilder.cpp:98-110
            std::vector<smxtool::DatasetLibrary> libs;
            if (!smxtool::load_dataset_yaml(file, libs)) {
                fprintf(stderr, "Dataset description file: %s does not exist or is not a valid YAML file\n", file.c_str());
                smx_destroy(ctx);
                return SMX_INPUT_FILE_NOT_FOUND;
            }
            for (const auto &lib : libs)
                if (lib.graph_constructable()) files.insert(files.end(), lib.files.begin(), lib.files.end());
        } else {
            files.push_back(file);
        }
        smxtool::prewarm_for_inputs(ctx, files, 6.0, 6.0);
        for (const auto &fn : files) {
            rc = smxtool::submit_file(ctx, fn);
            if (rc) break;
        }
        STAGE("read input")
        if (rc == -1) {
            fprintf(stderr, "Dataset description file: %s does not exist or is not a valid YAML file\n", file.c_str());
            smx_destroy(ctx);
            return SMX_INPUT_FILE_NOT_FOUND;
        }
        if (!rc) rc = smx_build_graph(ctx, k, 10 * nthreads);
        STAGE("build graph")
        if (!rc && coverage && mode != UNITIGS) {
            printf("Filling coverage index\n");
            rc = smx_graph_fill_coverage(ctx);
        }
        if (!rc) {
            uint64_t info[8];
            smx_graph_info(ctx, info);
            printf("Extracting unbranching paths finished. %llu sequences extracted\n", (unsigned long long)(info[2] - info[3]));
            printf("Collecting perfect loops finished. %llu loops collected\n", (unsigned long long)info[3]);
            printf("Saving %s to %s\n", mode == GFA ? "graph" : "unitigs", outfile.c_str());
            rc = mode == GFA ? smx_graph_write_gfa(ctx, outfile.c_str(), "SPAdes-4.3.0-dev")
                 : mode == SPADES ? smx_graph_write_spades(ctx, outfile.c_str())
                 : mode == FASTG ? smx_graph_write_fastg(ctx, outfile.c_str()) : smx_graph_write_unitigs(ctx, outfile.c_str());
            STAGE("write output")
        }
        if (rc) fprintf(stderr, "%s\n", smx_last_error(ctx));
    } catch (const std::string &s) {
        fprintf(stderr, "%s\n", s.c_str());
        rc = EINTR;
    }
    smx_destroy(ctx);
    if (!rc) printf("SPAdes standalone graph builder finished\n");
    return rc;
}
