#!/bin/bash
# Round 6, call 16: pointer doubling (2 / 3 / 4 rounds) in front of the chain walks of the fused node table, phase ticks of each.
out=gpurun_out/r6p; mkdir -p $out; exec > $out/log.txt 2>&1
common="--no-cpu-baseline --extra-kmercount 0 --end-to-end 0 --sharded-construct 0 --distributed-walks 0 --no-file-on-demand --early-tip-extra 0 --scaling-reference 0 --steps 3 --warmup 1"
for v in ticks dbl2 dbl3 dbl4 ticks; do
  SMX_BENCH_LIB=tools/ab/lib_$v.so timeout 400 python bench.py $common > $out/ab_$v.json 2> $out/ab_$v.err
  echo "== $v"; python tools/bench_summary.py $out/ab_$v.json 2>&1 | sed -n 2,5p | cut -c1-200
  python - $out/ab_$v.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print("fingerprint", d["construct"]["checks"].get("graph_fingerprint"), "unitigs", d["construct"]["n_unitigs"])
except Exception as e: print("no line:", e)
PY
  SMX_BENCH_LIB=tools/ab/lib_$v.so SMX_DEBUG=1 timeout 400 python bench.py $common --steps 1 --warmup 0 > $out/debug_$v.json 2> $out/debug_$v.err
  grep -E "dedupe chunks" $out/debug_$v.err | tail -1
done
