#!/bin/bash
# Round 6, closing call after the second half of the round (fused node table, packed scan, short-path walk_write, junction flag in the jump words):
# PMC / kernel-stat passes of the step, the bench line that quotes them, the one-rank point of the N > 1 workload.
#   gpurun --timeout 2400 -- 'bash tools/gpu_calls_r06/r6_final3.sh'
out=gpurun_out/r6w; mkdir -p $out; exec > $out/log.txt 2>&1
set -x
timeout 1200 bash tools/profile_bench.sh r06 --sync-upload; ls gpurun_out/prof_r06; cat gpurun_out/prof_r06/summary.log
cp gpurun_out/prof_r06/pmc_hbm_traffic.csv profiles/r06/config3_pm_pmc_hbm_traffic.csv; cp gpurun_out/prof_r06/kernel_stats.csv profiles/r06/config3_pm_kernel_stats.csv
timeout 1200 python bench.py > $out/bench_config3.json 2> $out/bench_config3.err; tail -3 $out/bench_config3.err; python tools/bench_summary.py $out/bench_config3.json
python - $out/bench_config3.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("roofline", {k: v for k, v in d["roofline"].items() if k != "stages_ms"})
print("construct", d["construct"]["roofline"])
print("dominant", d["dominant_kernel"])
print("early_tip_clipper", d.get("early_tip_clipper"))
PY
timeout 600 python bench.py --gpus 1 --scaling > $out/bench_config4_share_1rank.json 2> $out/bench_config4_share_1rank.err; tail -2 $out/bench_config4_share_1rank.err; head -c 600 $out/bench_config4_share_1rank.json
