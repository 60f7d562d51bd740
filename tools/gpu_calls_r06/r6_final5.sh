#!/bin/bash
# Round 6, the record after the clipper-path changes: the default bench line (its kernels are those of the PMC table of r6_final4; the early_tip_clipper extra is new),
# the one-rank point of the N > 1 workload (smoke() and the GPU tier: a call of their own, r6_call27.sh).
#   gpurun --timeout 3000 -- 'bash tools/gpu_calls_r06/r6_final5.sh'
out=gpurun_out/r6w; mkdir -p $out; exec > $out/log5.txt 2>&1
timeout 1200 python bench.py > $out/bench_config3.json 2> $out/bench_config3.err; tail -3 $out/bench_config3.err; python tools/bench_summary.py $out/bench_config3.json
python - $out/bench_config3.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("roofline", {k: v for k, v in d["roofline"].items() if k != "stages_ms"})
print("construct", d["construct"]["roofline"])
print("dominant", d["dominant_kernel"])
print("early_tip_clipper", d.get("early_tip_clipper"))
PY
timeout 600 python bench.py --gpus 1 --scaling > $out/bench_config4_share_1rank.json 2> $out/bench_config4_share_1rank.err; head -c 300 $out/bench_config4_share_1rank.json; echo

