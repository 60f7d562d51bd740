#!/bin/bash
# Round 6, seventh GPU call: the early tip clipper on route 0 at 100 M reads (incremental node-table renewal; full renewal as A/B), where the route's memory goes
# when it gives up, the PMC / kernel-stat passes of the round's bench step, SQ counters of the on-chip dedupe kernel.
#   gpurun --timeout 2400 -- 'bash tools/gpu_calls_r06/r6_call7.sh'
out=gpurun_out/r6g; mkdir -p $out; exec > $out/log.txt 2>&1
set -x
timeout 600 python -m pytest tests/test_pm_route_gpu.py tests/test_graph_gpu.py -m gpu -q -p no:cacheprovider > $out/gpu_tests.log 2>&1; tail -4 $out/gpu_tests.log
B="--no-cpu-baseline --extra-kmercount 0 --end-to-end 0 --scaling-reference 0 --steps 5"
for tag in "tip95_pm:--opt early_tip_bound=95" "tip95_pm_full_retab:--opt early_tip_bound=95 --opt pm_full_retab=1" "tip95_pm_20M:--opt early_tip_bound=95 --reads 20e6 --genome 100e6" "plain_pm_20M:--reads 20e6 --genome 100e6"; do
  name=${tag%%:*}; args=${tag#*:}
  SMX_DEBUG_BAIL=1 timeout 900 python bench.py $B $args > $out/bench_$name.json 2> $out/bench_$name.err; grep -E "gives up" $out/bench_$name.err | head -2
  python - $out/bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    st = d["roofline"]["stages_ms"]
    print(sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], d["step_breakdown_ms"], "unitigs", d["construct"]["n_unitigs"], "fingerprint", d["construct"]["checks"]["graph_fingerprint"])
    print("   ", {k: round(v, 1) for k, v in st.items() if v > 3 and not k.startswith("kmers:")})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
timeout 1200 bash tools/profile_bench.sh r06 --sync-upload; ls gpurun_out/prof_r06
timeout 900 bash tools/sq_counters.sh k_skm_dedupe2 > $out/dedupe2_sq_counters_20M.txt 2>&1; cat $out/dedupe2_sq_counters_20M.txt
