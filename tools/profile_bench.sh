#!/bin/bash
# GPU box: the three rocprofv3 passes behind profiles/rNN (run from the repo root via gpurun).
#   1. --kernel-trace --stats   per-kernel durations of `bench.py --steps 2 --warmup 0` (setup pass + 2 steps = 3 passes of the hot path)
#   2. --pmc FETCH_SIZE         HBM read KB per dispatch   (own pass, kernel-trace only: see MI355X_MICROARCH.md, HBM section)
#   3. --pmc WRITE_SIZE         HBM write KB per dispatch  (own pass)
# usage: tools/profile_bench.sh <tag> [extra bench.py args]
set -u
tag=${1:-r02}; shift || true
root=$(pwd); out=$root/gpurun_out/prof_$tag; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
common="--no-cpu-baseline --extra-kmercount 0 --end-to-end 0 --scaling-reference 0 --early-tip-extra 0 --no-file-on-demand $*"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python "$root/bench.py" --steps 2 --warmup 0 $common > "$out/kt.bench.json" 2> "$out/kt.err"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/$c" -- python "$root/bench.py" --steps 1 --warmup 0 $common > "$out/$c.bench.json" 2> "$out/$c.err"
done
cd "$root" && python tools/pmc_summary.py "$out" > "$out/summary.log" 2>&1; tail -5 "$out/summary.log"
# keep only the condensed tables (the raw traces are hundreds of MB)
rm -rf "$out/kt" "$out/FETCH_SIZE" "$out/WRITE_SIZE"
