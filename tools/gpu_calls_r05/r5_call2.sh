#!/bin/bash
# Round 5, second GPU call: kernel variants A/B on one box (tools/ab/lib_*.so, built here; same host code), then the phase profile of the
# dedupe / scan kernels (SMX_DEBUG ticks) on the current build.
out=gpurun_out/r5b; mkdir -p $out; exec > $out/log.txt 2>&1
common="--no-cpu-baseline --extra-kmercount 0 --end-to-end 0 --sharded-construct 0 --distributed-walks 0 --steps 3 --warmup 1"
for v in r4 base sb4 wpe4 wpe6 scan8 scan4; do
  [ -f tools/ab/lib_$v.so ] || continue
  SMX_BENCH_LIB=tools/ab/lib_$v.so timeout 400 python bench.py $common > $out/ab_$v.json 2> $out/ab_$v.err
  echo "== $v"; python tools/bench_summary.py $out/ab_$v.json 2>&1 | sed -n 1,4p
done
SMX_BENCH_LIB=tools/ab/lib_base.so SMX_DEBUG=1 timeout 400 python bench.py $common --steps 1 --warmup 0 > $out/debug.json 2> $out/debug.err
grep -E "dedupe chunks|skm_scan phase|prededupe:|pm_tab|g:" $out/debug.err | sort | uniq -c | sort -rn | head -40
