#!/bin/bash
# Round 5, the recorded runs on the final kernels: rocprofv3 kernel stats + PMC passes (tools/profile_bench.sh), the default bench line,
# the whole GPU tier.
out=gpurun_out/r5f; mkdir -p $out; exec > $out/log.txt 2>&1
set -x
bash tools/profile_bench.sh r05 --sharded-construct 0 --distributed-walks 0 --sync-upload
ls -la gpurun_out/prof_r05
timeout 1500 python bench.py > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err; python tools/bench_summary.py $out/bench.json
timeout 1300 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $out/gpu_tests.log 2>&1; tail -5 $out/gpu_tests.log
