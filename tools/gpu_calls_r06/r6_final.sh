#!/bin/bash
# Round 6, closing call: the whole GPU tier on the final sources, smoke, the default bench line (the record), the PMC / kernel-stat passes of that step,
# the one-rank point of the N > 1 workload.
#   gpurun --timeout 3000 -- 'bash tools/gpu_calls_r06/r6_final.sh'
out=gpurun_out/r6z; mkdir -p $out; exec > $out/log.txt 2>&1
set -x
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 > $out/gpu_tests.log 2>&1; tail -20 $out/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt
timeout 1200 python bench.py > $out/bench_config3.json 2> $out/bench_config3.err; tail -3 $out/bench_config3.err; python tools/bench_summary.py $out/bench_config3.json
python - $out/bench_config3.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k in ("kmer_file_on_demand", "scaling_reference", "kmercount_mode", "early_tip_clipper", "dominant_kernel"):
    print(k, d.get(k))
print("construct roofline", d["construct"]["roofline"])
print("roofline", {k: v for k, v in d["roofline"].items() if k != "stages_ms"})
print("e2e", {k: (v.get("seconds"), v.get("stages_s")) for k, v in d.get("end_to_end", {}).items() if isinstance(v, dict)})
PY
timeout 1200 bash tools/profile_bench.sh r06 --sync-upload; ls gpurun_out/prof_r06
timeout 600 python bench.py --gpus 1 --scaling --sharded-construct 10e6 --distributed-walks 2e6 > $out/bench_config4_share_1rank.json 2> $out/bench_config4_share_1rank.err; tail -3 $out/bench_config4_share_1rank.err; python - $out/bench_config4_share_1rank.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d.get("construct_sharded"))
PY
