#!/bin/bash
# Round 6, third GPU call: successor table on a side stream (pm_overlap) A/B on the bench line; the library's distributed walks again at config 5's
# per-rank share after the first measurements (own segment as a device copy, one LDS atomic per wave and owner, no all-reduce per exchange).
#   gpurun --timeout 1800 -- 'bash tools/gpu_calls_r06/r6_call3.sh'
out=gpurun_out/r6c; mkdir -p $out; exec > $out/log.txt 2>&1
set -x
timeout 600 python -m pytest tests/test_pm_route_gpu.py tests/test_graph_gpu.py tests/test_dist_gpu.py -m gpu -q -x -p no:cacheprovider > $out/gpu_tests_pm_graph_dist.log 2>&1; tail -5 $out/gpu_tests_pm_graph_dist.log
for v in 1 0 1 0; do
  timeout 600 python bench.py --no-cpu-baseline --extra-kmercount 0 --end-to-end 0 --scaling-reference 0 --steps 5 --opt pm_overlap=$v > $out/bench_overlap$v.json 2> $out/bench_overlap$v.err
  python - $out/bench_overlap$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("pm_overlap", sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], d["step_breakdown_ms"], "fingerprint", d["construct"]["checks"]["graph_fingerprint"])
PY
done
SMX_DEBUG=1 timeout 900 python tools/dwalk_probe.py 62.5e6 312.5e6 55 16 --no-reference > $out/dwalk_62M.log 2>&1; grep -E "walks:|distributed walks|torch peak|fingerprint walks" $out/dwalk_62M.log | tail -40
timeout 600 bash tools/cli_bimodality_probe.sh 20000000 > $out/cli_probe.log 2>&1; grep -E "gpus 1|walks:|wall|identical" $out/cli_probe.log
