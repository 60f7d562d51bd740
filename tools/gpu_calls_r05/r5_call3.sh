#!/bin/bash
# Round 5, third GPU call: dedupe variants (allocation atomic off the critical path; one-multiply hash; 4-window segments)
out=gpurun_out/r5c; mkdir -p $out; exec > $out/log.txt 2>&1
common="--no-cpu-baseline --extra-kmercount 0 --end-to-end 0 --sharded-construct 0 --distributed-walks 0 --steps 3 --warmup 1"
for v in base alloc hash1 seg4; do
  [ -f tools/ab/lib_$v.so ] || continue
  SMX_BENCH_LIB=tools/ab/lib_$v.so timeout 400 python bench.py $common > $out/ab_$v.json 2> $out/ab_$v.err
  echo "== $v"; python tools/bench_summary.py $out/ab_$v.json 2>&1 | sed -n 1,4p
done
SMX_BENCH_LIB=tools/ab/lib_alloc.so SMX_DEBUG=1 timeout 400 python bench.py $common --steps 1 --warmup 0 > $out/debug.json 2> $out/debug.err
grep -E "dedupe chunks" $out/debug.err | tail -2
