#!/bin/bash
# generic round-2 GPU call: $1 = tag, rest = what to run (tests | probe args...)
tag=$1; shift
mkdir -p gpurun_out/$tag; exec > gpurun_out/$tag/log.txt 2>&1
for step in "$@"; do
  case $step in
    tests) timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/tests.txt 2>&1; grep -v "^  File" gpurun_out/$tag/tests.txt | tail -25 ;;
    g10) SMX_DEBUG=1 timeout 600 python tools/scale_probe.py 10e6 50e6 graph 2>&1 | grep -v "big leaf\|^\[smx\] \(mark+alloc\|level1\|levels2+\|leaf sort\|compact\|leaves\|skm_scan\|dedupe\|prededupe\)" ;;
    g100) SMX_DEBUG=1 timeout 900 python tools/scale_probe.py 100e6 500e6 graph 2>&1 | grep -v "big leaf\|^\[smx\] \(skm_scan\|dedupe\)" ;;
    c100) SMX_DEBUG=1 timeout 900 python tools/scale_probe.py 100e6 500e6 count 2>&1 | grep -v "big leaf\|^\[smx\] \(skm_scan\|dedupe\)" ;;
    scale) timeout 900 python -m pytest tests/test_scale_gpu.py -m gpu -x -q 2>&1 | tail -15 ;;
    bench) timeout 1500 python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err; tail -5 gpurun_out/$tag/bench.err; cat gpurun_out/$tag/bench.json ;;
    benchq) timeout 1500 python bench.py --no-cpu-baseline --extra-kmercount 0 > gpurun_out/$tag/benchq.json 2> gpurun_out/$tag/benchq.err; tail -5 gpurun_out/$tag/benchq.err; cat gpurun_out/$tag/benchq.json ;;
    benchq2g) SMX_ARENA_CHUNK_MB=2048 timeout 1500 python bench.py --no-cpu-baseline --extra-kmercount 0 > gpurun_out/$tag/benchq2g.json 2> gpurun_out/$tag/benchq2g.err; tail -3 gpurun_out/$tag/benchq2g.err ;;
    benchq64) SMX_ARENA_CHUNK_MB=64 timeout 1500 python bench.py --no-cpu-baseline --extra-kmercount 0 > gpurun_out/$tag/benchq64.json 2> gpurun_out/$tag/benchq64.err; tail -3 gpurun_out/$tag/benchq64.err ;;
    shard100) SMX_DEBUG=1 timeout 900 python bench.py --gpus 1 --force-sharded --steps 2 --warmup 1 2>&1 | grep -v "big leaf\|^\[smx\] \(skm_scan\|dedupe\)" | tail -40 ;;
    tdist) timeout 900 python -m pytest tests/test_dist_gpu.py tests/test_cli_gpu.py -m gpu -x -q > gpurun_out/$tag/tdist.txt 2>&1; grep -v "^  File" gpurun_out/$tag/tdist.txt | tail -40 ;;
    text) timeout 900 python -m pytest tests/test_ext_route_gpu.py -m gpu -x -q > gpurun_out/$tag/text.txt 2>&1; grep -v "^  File" gpurun_out/$tag/text.txt | tail -40 ;;
    tgraph) timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_prededupe_gpu.py tests/test_count_gpu.py -m gpu -x -q 2>&1 | tail -15 ;;
    benchopt=*) o=${step#benchopt=}; timeout 1500 python bench.py --no-cpu-baseline --extra-kmercount 0 --steps 2 --opt $o > gpurun_out/$tag/bench_$o.json 2> gpurun_out/$tag/bench_$o.err; tail -2 gpurun_out/$tag/bench_$o.err; python tools/bench_summary.py gpurun_out/$tag/bench_$o.json ;;
    *) echo "unknown step $step" ;;
  esac
done
