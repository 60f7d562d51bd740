#!/bin/bash
# per-dispatch durations of selected kernels over several steps (rocprofv3 --kernel-trace)
set -u
root=$(pwd); O=$root/gpurun_out/r4kt; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $root/bench.py --no-cpu-baseline --end-to-end 0 --extra-kmercount 0 --steps 4 --warmup 1 "$@" > $O/kt.json 2> $O/kt.err
f=$(ls $O/kt/*/*kernel_trace.csv | head -1)
python - "$f" <<'PY'
import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"]
    for key in ("k_pm_walk_write","k_pm_walk_len","k_pm_tab","k_pm_remote","k_skm_dedupe2","k_skm_permute","k_skm_plan","k_pm_keep","k_pm_jrank","k_pm_junc_write","k_pm_cand_expand"):
        if key in n: d[key].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
for k,v in d.items(): print(k, len(v), " ".join("%.1f"%x for x in v))
PY
rm -rf $O/kt
