#!/bin/bash
# the default bench line on the final build, with this build's PMC table in place under profiles/r05 (so that the recorded line quotes its traffic)
out=gpurun_out/r5j; mkdir -p $out; exec > $out/log.txt 2>&1
timeout 1500 python bench.py > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err; python tools/bench_summary.py $out/bench.json
