#!/bin/bash
# Round 6, call 23: k_pm_remote works a per-wave queue off 64 lookups at a time (all modes); with it, one end of an edge answering for both (pm_remote_mirror = 1)
# against every end (0); route parity tests.
out=gpurun_out/r6v; mkdir -p $out; exec > $out/log.txt 2>&1
timeout 900 python -m pytest tests/test_pm_route_gpu.py tests/test_graph_gpu.py -m gpu -x -q -n 4 2>&1 | tail -3
common="--no-cpu-baseline --extra-kmercount 0 --end-to-end 0 --sharded-construct 0 --distributed-walks 0 --no-file-on-demand --early-tip-extra 0 --scaling-reference 0 --steps 3 --warmup 1"
run() {  # name, extra flags
  SMX_BENCH_LIB=tools/ab/lib_base.so timeout 400 python bench.py $common $2 > $out/ab_$1.json 2> $out/ab_$1.err
  echo "== $1"; python tools/bench_summary.py $out/ab_$1.json 2>&1 | sed -n 2,4p | cut -c1-220
  python - $out/ab_$1.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print("fingerprint", d["construct"]["checks"].get("graph_fingerprint"), "unitigs", d["construct"]["n_unitigs"], d["config"]["route"][:60])
except Exception as e: print("no line:", e)
PY
}
run queue_each_end ""
run queue_mirror "--opt pm_remote_mirror=1"
run queue_each_end_again ""
run queue_mirror_again "--opt pm_remote_mirror=1"
