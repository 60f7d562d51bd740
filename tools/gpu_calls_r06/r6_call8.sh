#!/bin/bash
# Round 6, eighth GPU call: early tip clipper on route 0 after the junction list lost its isolated k-mers (100 M and 20 M reads), per-kernel times of that step.
#   gpurun --timeout 1800 -- 'bash tools/gpu_calls_r06/r6_call8.sh'
out=gpurun_out/r6k; mkdir -p $out; exec > $out/log.txt 2>&1
set -x
timeout 600 python -m pytest tests/test_pm_route_gpu.py tests/test_graph_gpu.py tests/test_integration_gpu.py -m gpu -q -p no:cacheprovider > $out/gpu_tests.log 2>&1; tail -4 $out/gpu_tests.log
B="--no-cpu-baseline --extra-kmercount 0 --end-to-end 0 --scaling-reference 0 --steps 5"
for tag in "tip95_pm:--opt early_tip_bound=95" "tip95_pm_20M:--opt early_tip_bound=95 --reads 20e6 --genome 100e6" "plain_pm:"; do
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
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/prof -- python $OLDPWD/bench.py --no-cpu-baseline --extra-kmercount 0 --end-to-end 0 --scaling-reference 0 --steps 2 --warmup 0 --opt early_tip_bound=95 --reads 20e6 --genome 100e6 > /dev/null 2> $OLDPWD/$out/prof.err )
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); head -24 "$f" | cut -c1-230; cp "$f" $out/tip95_20M_kernel_stats.csv; rm -rf $out/prof
