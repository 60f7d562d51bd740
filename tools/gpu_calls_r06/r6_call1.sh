#!/bin/bash
# Round 6, first GPU call: the tier on the round's first sources (arena prewarm helper, plain-record k-mer file that fails without losing the graph,
# bucket accessor on the device), the bench line with the round's accounting (route-0 bytes, k-mer file on demand, kmercount through the merge,
# scaling reference = config 4's per-GPU share), the CLI bimodality probe and 200 one-rank RCCL launches (item 9).
#   gpurun --timeout 3000 -- 'bash tools/gpu_calls_r06/r6_call1.sh'
out=gpurun_out/r6a; mkdir -p $out; exec > $out/log.txt 2>&1
set -x
nproc; free -g | head -2; rocm-smi --showmeminfo vram | head -8
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > $out/gpu_tests.log 2>&1; tail -30 $out/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt
timeout 900 python bench.py > $out/bench_config3.json 2> $out/bench_config3.err; tail -3 $out/bench_config3.err; python tools/bench_summary.py $out/bench_config3.json
python - $out/bench_config3.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k in ("kmer_file_on_demand", "scaling_reference", "kmercount_mode"):
    print(k, d.get(k))
print("construct roofline", d["construct"]["roofline"])
print("e2e", {k: (v.get("seconds"), v.get("stages_s")) for k, v in d.get("end_to_end", {}).items() if isinstance(v, dict)})
PY
timeout 600 bash tools/cli_bimodality_probe.sh 20000000 > $out/cli_bimodality.log 2>&1; cat $out/cli_bimodality.log
timeout 1500 python tools/rccl_launch_loop.py 200 20 > $out/rccl_200_launches.log 2>&1; tail -5 $out/rccl_200_launches.log
timeout 600 python bench.py --gpus 1 --scaling --sharded-construct 0 > $out/bench_config4_share_1rank.json 2> $out/bench_config4_share_1rank.err; tail -3 $out/bench_config4_share_1rank.err; head -c 1500 $out/bench_config4_share_1rank.json
