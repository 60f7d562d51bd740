#!/bin/bash
# Round 6, call 29: per-kernel times of the step with the early tip clipper (bound 95) at config 3, closing build
out=$(pwd)/gpurun_out/r6z5; mkdir -p $out; root=$(pwd)
cd /tmp && export TMPDIR=/tmp
common="--no-cpu-baseline --extra-kmercount 0 --end-to-end 0 --sharded-construct 0 --distributed-walks 0 --no-file-on-demand --early-tip-extra 0 --scaling-reference 0 --sync-upload"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python $root/bench.py --steps 2 --warmup 0 $common --opt early_tip_bound=95 > $out/kt.bench.json 2> $out/kt.err
f=$(find $out/kt -name "*kernel_stats.csv" | head -1); cp "$f" $out/tip95_config3_kernel_stats.csv; rm -rf $out/kt
head -25 $out/tip95_config3_kernel_stats.csv | cut -c1-200
