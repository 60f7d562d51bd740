#!/bin/bash
# round-2 closing measurements (GPU box): default bench line, the sharded path on one rank, a 2 M-read construction
mkdir -p gpurun_out/final; exec > gpurun_out/final/log.txt 2>&1
timeout 1500 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; tail -3 gpurun_out/final/bench.err
timeout 600 python bench.py --gpus 1 --force-sharded --reads 20e6 --genome 100e6 --steps 3 --warmup 1 > gpurun_out/final/bench_sharded_1rank.json 2> gpurun_out/final/bench_sharded.err; tail -3 gpurun_out/final/bench_sharded.err
timeout 600 python bench.py --count-only --reads 20e6 --genome 100e6 --steps 3 --warmup 1 --no-cpu-baseline --extra-kmercount 0 > gpurun_out/final/bench_count_20M.json 2>> gpurun_out/final/bench_sharded.err
echo "=== 2 M reads construction"; SMX_DEBUG=1 timeout 300 python tools/scale_probe.py 2e6 10e6 graph 2>&1 | grep -E "^\[smx\] g:|build k=55|coverage"
