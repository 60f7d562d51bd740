#!/bin/bash
# one quick bench line on the in-tree library (kernel experiments between the recorded calls)
out=gpurun_out/r5q; mkdir -p $out; exec > $out/log.txt 2>&1
timeout 400 python bench.py --no-cpu-baseline --extra-kmercount 0 --end-to-end 0 --sharded-construct 0 --distributed-walks 0 --steps 3 --warmup 1 $BENCH_EXTRA > $out/q.json 2> $out/q.err
python tools/bench_summary.py $out/q.json 2>&1 | sed -n 1,5p; tail -2 $out/q.err
