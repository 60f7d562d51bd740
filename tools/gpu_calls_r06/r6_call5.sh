#!/bin/bash
# Round 6, fifth GPU call: early clippers on route 0 after the stage-timer fix (tests; why the 100 M-read step with early_tip_bound left the route),
# per-kernel times of the library's distributed walks (rocprofv3 on the one-rank probe), out-of-core count streamed to its file at 60 M reads / 32 GB.
#   gpurun --timeout 2400 -- 'bash tools/gpu_calls_r06/r6_call5.sh'
out=gpurun_out/r6e; mkdir -p $out; exec > $out/log.txt 2>&1
set -x
timeout 900 python -m pytest tests/test_pm_route_gpu.py tests/test_graph_gpu.py tests/test_spill_gpu.py -m gpu -q -p no:cacheprovider > $out/gpu_tests.log 2>&1; tail -6 $out/gpu_tests.log
SMX_DEBUG=1 timeout 600 python bench.py --no-cpu-baseline --extra-kmercount 0 --end-to-end 0 --scaling-reference 0 --steps 1 --warmup 0 --opt early_tip_bound=95 > $out/bench_tip95_pm_debug.json 2> $out/bench_tip95_pm_debug.err
grep -E "gives up|pm_tab:|g:|arena" $out/bench_tip95_pm_debug.err | tail -30
python - $out/bench_tip95_pm_debug.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("tip95 (debug run) ms/step", d["ms_per_step"], {k: round(v, 1) for k, v in d["roofline"]["stages_ms"].items() if v > 4 and not k.startswith("kmers:")})
except Exception as e:
    print("FAILED", e)
PY
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$out/prof_dwalk -- python $OLDPWD/tools/dwalk_probe.py 20e6 100e6 55 16 --no-reference > $OLDPWD/$out/dwalk_20M_rocprof.log 2>&1 )
f=$(find $out/prof_dwalk -name "*kernel_stats.csv" | head -1); head -30 "$f"; cp "$f" $out/dwalk_20M_kernel_stats.csv; rm -rf $out/prof_dwalk
timeout 1200 python tools/verify_spill.py 60e6 32 > $out/spill_60M_budget32G_streamed.log 2>&1; cat $out/spill_60M_budget32G_streamed.log
