#!/bin/bash
# Round 6, sixth GPU call: the tip clipper with one lane per branch (both routes, 100 M reads), the walks' grouping kernels without global atomics
# (per-kernel times again; the C++ host at config 5's per-rank share), tests of what changed.
#   gpurun --timeout 2400 -- 'bash tools/gpu_calls_r06/r6_call6.sh'
out=gpurun_out/r6f; mkdir -p $out; exec > $out/log.txt 2>&1
set -x
timeout 900 python -m pytest tests/test_pm_route_gpu.py tests/test_graph_gpu.py tests/test_spill_gpu.py tests/test_dist_gpu.py tests/test_zz_cli_rccl_gpu.py -m gpu -q -p no:cacheprovider > $out/gpu_tests.log 2>&1; tail -6 $out/gpu_tests.log
B="--no-cpu-baseline --extra-kmercount 0 --end-to-end 0 --scaling-reference 0 --steps 5"
for tag in "tip95_pm:--opt early_tip_bound=95" "tip95_sorted:--sorted-route --opt early_tip_bound=95"; do
  name=${tag%%:*}; args=${tag#*:}
  SMX_DEBUG_BAIL=1 timeout 900 python bench.py $B $args > $out/bench_$name.json 2> $out/bench_$name.err; grep -E "gives up" $out/bench_$name.err | head -2
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
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/prof_dwalk -- python $OLDPWD/tools/dwalk_probe.py 20e6 100e6 55 16 --no-reference > $OLDPWD/$out/dwalk_20M_rocprof.log 2>&1 )
f=$(find $out/prof_dwalk -name "*kernel_stats.csv" | head -1); head -16 "$f" | cut -c1-200; cp "$f" $out/dwalk_20M_kernel_stats.csv; rm -rf $out/prof_dwalk
grep -E "distributed walks:|identical" $out/dwalk_20M_rocprof.log
timeout 1500 bash tools/cpp_walks_at_size.sh 62500000 > $out/cpp_walks_62M.log 2>&1; cat $out/cpp_walks_62M.log
