#!/bin/bash
# Round 6, call 14: fused node table with pointer-jumping chains + walk_len's junction shortcut: parity tests, bench (fused / unfused), phase ticks.
out=gpurun_out/r6n; mkdir -p $out; exec > $out/log.txt 2>&1
timeout 900 python -m pytest tests/test_pm_route_gpu.py tests/test_graph_gpu.py tests/test_golden_gpu.py -m gpu -x -q -n 4 2>&1 | tail -5
common="--no-cpu-baseline --extra-kmercount 0 --end-to-end 0 --sharded-construct 0 --distributed-walks 0 --no-file-on-demand --early-tip-extra 0 --scaling-reference 0 --steps 3 --warmup 1"
run() {  # name, lib, extra flags
  SMX_BENCH_LIB=tools/ab/lib_$2.so timeout 400 python bench.py $common $3 > $out/ab_$1.json 2> $out/ab_$1.err
  echo "== $1"; python tools/bench_summary.py $out/ab_$1.json 2>&1 | sed -n 1,5p; tail -2 $out/ab_$1.err | grep -v amdgpu.ids | cut -c1-300
}
run fused base ""
run unfused base "--opt pm_fuse_tab=0"
run fused_again base ""
SMX_BENCH_LIB=tools/ab/lib_base.so SMX_DEBUG=1 timeout 400 python bench.py $common --steps 1 --warmup 0 > $out/debug.json 2> $out/debug.err
grep -E "dedupe chunks" $out/debug.err | tail -2
run fused_tip95 base "--opt early_tip_bound=95"
