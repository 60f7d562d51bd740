// spades_amd/tools/kmercount_main.cpp — drop-in CLI for `spades-kmercount`
// (reference: projects/spades_tools/kmercount.cpp:134-229; docs/standalone.md:5-45) over libspades_mi355x.so.
//   spades-kmercount-mi355x [-k 21] [-t N] [-w dir] [-b bytes] [--gpus N] files...   ->  <dir>/final_kmers
// --gpus N: one process per GPU, k-mers redistributed by bucket owner with one RCCL exchange (kmercount_mgpu.hpp).
// -t and -b are accepted for command-line compatibility; the result does not depend on them (SURVEY.md finding 3).
// Exit codes follow common/utils/logger/error_codes.hpp (64-68).
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/smx.h"
#include "read_input.hpp"
#include "kmercount_mgpu.hpp"
#include <chrono>
#include <mutex>
#include <thread>
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define STAGE(what)                                                                 \
    if (getenv("SMX_DEBUG")) {                                                      \
        double t_ = now_s();                                                        \
        fprintf(stderr, "[tool] %-12s %7.3f s\n", what, t_ - t_stage);             \
        t_stage = t_;                                                               \
    }

static void usage(const char *a0) {
    printf("SYNOPSIS\n        %s [-k <value>] [-t <value>] [-w <dir>] [-b <value>] [-h] [<input files>...]\n\n"
           "DESCRIPTION\n        SPAdes k-mer counting engine (MI355X)\n\n"
           "        Output: <output_dir>/final_kmers - unordered set of kmers in binary format. Kmers from both forward and\n"
           "        reverse-complementary reads are taken into account.\n", a0);
}

int main(int argc, char **argv) {
    unsigned K = 21, nthreads = 1;
    int gpus = 0;
    std::string workdir = ".";
    std::vector<std::string> input;
    std::string dataset;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto need = [&](const char *what) -> const char * {
            if (i + 1 >= argc) {
                fprintf(stderr, "Invalid command line arguments: %s needs a value\n", what);
                exit(SMX_INVALID_PARAMETER);
            }
            return argv[++i];
        };
        if (a == "-k" || a == "--kmer") K = (unsigned)atoi(need("-k"));
        else if (a == "-t" || a == "--threads") nthreads = (unsigned)atoi(need("-t"));
        else if (a == "-w" || a == "--workdir") workdir = need("-w");
        else if (a == "-b" || a == "--bufsize") (void)need("-b");
        else if (a == "-d" || a == "--dataset") dataset = need("-d");
        else if (a == "--gpus") gpus = atoi(need("--gpus"));
        else if (a == "-h" || a == "--help") {
            usage(argv[0]);
            return 0;
        } else if (!a.empty() && a[0] == '-') {
            usage(argv[0]);
            fprintf(stderr, "Invalid command line arguments\n");
            return SMX_INVALID_PARAMETER;
        } else input.push_back(a);
    }
    (void)nthreads;
    if (!dataset.empty()) {  // "Dataset description (in YAML), input files ignored", kmercount.cpp:141,210-214
        std::vector<smxtool::DatasetLibrary> libs;
        if (!smxtool::load_dataset_yaml(dataset, libs)) {
            fprintf(stderr, "Dataset description file: %s does not exist or is not a valid YAML file\n", dataset.c_str());
            return SMX_INPUT_FILE_NOT_FOUND;
        }
        input.clear();
        for (const auto &lib : libs) input.insert(input.end(), lib.files.begin(), lib.files.end());
    }
    if (input.empty()) {
        fprintf(stderr, "No input files were specified\n");
        return SMX_INVALID_PARAMETER;
    }
    if (gpus > 0) {  // the sharded path (also with one GPU: same code, self-exchange)
        printf("K-mer length set to %u\n", K);
        return smxtool::run_sharded(gpus, K, workdir, input);
    }
    smx_ctx *ctx = nullptr;
    double t_stage = now_s();
    if (int rc = smx_create(&ctx, 0, 0)) {
        fprintf(stderr, "No usable MI355X device (smx_create -> %d)\n", rc);
        return rc;
    }
    printf("K-mer length set to %u\n", K);
    try {
        // one host thread per input file (up to 8 at a time): reading / inflating is the slow part of the ingest and R1/R2 files are
        // independent streams; the library calls are serialised
        STAGE("device init")
        smxtool::prewarm_for_inputs(ctx, input, 6.0, 6.0);
        std::mutex mu;
        std::vector<int> rcs(input.size(), 0);
        std::vector<std::string> errs(input.size());
        for (size_t base = 0; base < input.size(); base += 8) {
            std::vector<std::thread> th;
            for (size_t i = base; i < std::min(input.size(), base + 8); ++i) {
                printf("Processing \"%s\"\n", input[i].c_str());
                th.emplace_back([&, i] {
                    try {
                        rcs[i] = smxtool::submit_file(ctx, input[i], &mu);
                    } catch (const std::string &e) {
                        rcs[i] = -2;
                        errs[i] = e;
                    }
                });
            }
            for (auto &t : th) t.join();
        }
        for (size_t i = 0; i < input.size(); ++i) {
            if (rcs[i] == -1) {
                fprintf(stderr, "File %s doesn't exist or can't be read!\n", input[i].c_str());
                smx_destroy(ctx);
                return SMX_INPUT_FILE_NOT_FOUND;
            }
            if (rcs[i] == -2) throw errs[i];
            if (rcs[i]) throw std::string(smx_last_error(ctx));
        }
        STAGE("read input")
        // Count + merge with the destination known from the start (16 buckets: kmercount.cpp:220): a count that goes out of core streams its merged
        // bucket ranges into final_kmers as the reference's merge does (kmer_index_builder.hpp:346-430) — host memory never holds the merged result
        std::string out = workdir + "/final_kmers";
        if (int rc = smx_count_to_file(ctx, K, SMX_MODE_ALL, 16, out.c_str())) {
            fprintf(stderr, "%s\n", smx_last_error(ctx));
            smx_destroy(ctx);
            return rc;
        }
        STAGE("count+write")
        uint64_t n = 0;
        smx_count_info(ctx, &n, nullptr, nullptr);
        printf("K-mer counting done. There are %llu kmers in total.\n", (unsigned long long)n);
        printf("K-mer counting done, kmers saved to \"%s\"\n", out.c_str());
    } catch (const std::string &s) {
        fprintf(stderr, "%s\n", s.c_str());
        smx_destroy(ctx);
        return EINTR;  // kmercount.cpp:226-229
    }
    smx_destroy(ctx);
    return 0;
}
