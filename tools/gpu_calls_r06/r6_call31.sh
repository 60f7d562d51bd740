#!/bin/bash
# Round 6, call 31: the tip clipper branch walks use bit 31 of the jump words at chunk crossings (one line instead of three);
out=gpurun_out/r6z7; mkdir -p $out; exec > $out/log.txt 2>&1
timeout 1200 python -m pytest tests/test_pm_route_gpu.py tests/test_graph_gpu.py tests/test_ext_route_gpu.py tests/test_integration_gpu.py -m gpu -x -q -n 4 2>&1 | tail -3
common="--no-cpu-baseline --extra-kmercount 0 --end-to-end 0 --sharded-construct 0 --distributed-walks 0 --no-file-on-demand --early-tip-extra 0 --scaling-reference 0 --steps 3 --warmup 1"
run() {  # name, extra flags
  timeout 400 python bench.py $common $2 > $out/ab_$1.json 2> $out/ab_$1.err
  echo "== $1"; python tools/bench_summary.py $out/ab_$1.json 2>&1 | sed -n 2,4p | cut -c1-220
  python - $out/ab_$1.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print("fingerprint", d["construct"]["checks"].get("graph_fingerprint"), "unitigs", d["construct"]["n_unitigs"], d["config"]["route"][:60])
except Exception as e: print("no line:", e)
PY
}
run tip95 "--opt early_tip_bound=95"
run tip95_again "--opt early_tip_bound=95"

run default ""
