#!/bin/bash
# where do the ~115 device copies and ~60 fills per step (17 + 6 ms under rocprofv3) come from? one kernel trace, condensed to the copy / fill
# dispatches with the kernels launched just before and after them
root=$(pwd); out=$root/gpurun_out/r5q; mkdir -p $out; exec > $out/log.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out/kt -- python $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --extra-kmercount 0 --end-to-end 0 --sharded-construct 0 --distributed-walks 0 --sync-upload > $out/b.json 2> $out/b.err
cd $root
python - <<'PY'
import csv, glob, os
f = glob.glob("gpurun_out/r5q/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last pass of the hot path: from the last k_mark_windows on
marks = [i for i, r in enumerate(rows) if "k_mark_windows" in r["Kernel_Name"]]
seg = rows[marks[-1]:]
def nm(r): return r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
tot = {}
lines = []
for i, r in enumerate(seg):
    n = nm(r)
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if "copyBuffer" in n or "fillBuffer" in n:
        tot[n] = tot.get(n, 0) + d
        if d >= 0.05:
            prev = next((nm(x) for x in reversed(seg[:i]) if "Buffer" not in nm(x)), "-")
            nxt = next((nm(x) for x in seg[i + 1:] if "Buffer" not in nm(x)), "-")
            lines.append(f"{d:7.3f} ms {n:32s} after {prev:40s} before {nxt}")
print("\n".join(lines))
print("totals in the last pass:", {k: round(v, 2) for k, v in tot.items()}, "dispatches", len(seg))
PY
rm -rf $out/kt
