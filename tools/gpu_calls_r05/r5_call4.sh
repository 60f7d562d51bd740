#!/bin/bash
# Round 5, fourth GPU call (second attempt: the first lost its box in the 20 M-read / 8 GB out-of-core leg — nothing came back — so the
# out-of-core legs at size are NOT repeated): the both-strands leg of the bench inside the bench process vs a fresh context.
out=gpurun_out/r5d; mkdir -p $out; exec > $out/log.txt 2>&1
set -x
cat /sys/fs/cgroup/memory.max /sys/fs/cgroup/memory/memory.limit_in_bytes 2>/dev/null
SMX_DEBUG=1 timeout 600 python bench.py --no-cpu-baseline --end-to-end 0 --sharded-construct 0 --distributed-walks 0 --steps 1 --warmup 0 > $out/bench_modeA.json 2> $out/bench_modeA.err
grep -E "two strands|count_reads" $out/bench_modeA.err | tail -20
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5d/bench_modeA.json"))
print("kmercount_mode", d.get("kmercount_mode"))
PY
SMX_DEBUG=1 timeout 600 python tools/modeA_probe.py > $out/modeA_fresh.json 2> $out/modeA_fresh.err; grep -E "two strands" $out/modeA_fresh.err | tail -5; tail -c 1500 $out/modeA_fresh.json
