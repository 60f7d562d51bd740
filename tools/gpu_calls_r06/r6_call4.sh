#!/bin/bash
# Round 6, fourth GPU call: early clippers on route 0 (tests, spades-core with the GPU stage, the bench line with early_tip_bound = 95 on both routes and
# the sorted route re-measured on round-6 sources), the C++ host's distributed walks at config 5's per-rank share.
#   gpurun --timeout 3000 -- 'bash tools/gpu_calls_r06/r6_call4.sh'
out=gpurun_out/r6d; mkdir -p $out; exec > $out/log.txt 2>&1
set -x
timeout 900 python -m pytest tests/test_pm_route_gpu.py tests/test_graph_gpu.py tests/test_integration_gpu.py tests/test_dist_gpu.py -m gpu -q -p no:cacheprovider --durations=5 > $out/gpu_tests.log 2>&1; tail -12 $out/gpu_tests.log
B="--no-cpu-baseline --extra-kmercount 0 --end-to-end 0 --scaling-reference 0 --steps 5"
for tag in "tip95_pm:--opt early_tip_bound=95" "tip95_sorted:--sorted-route --opt early_tip_bound=95" "plain_sorted:--sorted-route" "plain_pm:"; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 900 python bench.py $B $args > $out/bench_$name.json 2> $out/bench_$name.err; tail -2 $out/bench_$name.err
  python - $out/bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    st = d["roofline"]["stages_ms"]
    print(sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], d["step_breakdown_ms"], "unitigs", d["construct"]["n_unitigs"], "fingerprint", d["construct"]["checks"]["graph_fingerprint"])
    print("   ", {k: round(v, 1) for k, v in st.items() if v > 4 and not k.startswith("kmers:")})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
timeout 1500 bash tools/cpp_walks_at_size.sh 62500000 > $out/cpp_walks_62M.log 2>&1; cat $out/cpp_walks_62M.log
