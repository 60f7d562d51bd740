#!/bin/bash
# round-4 experiment 1: correctness of the v2 dedupe kernel + A/B timings at config 3 (run on the GPU box via gpurun)
set -u
O=gpurun_out/r4e1; mkdir -p $O
C=spades_amd/csrc
B="python bench.py --no-cpu-baseline --end-to-end 0 --extra-kmercount 0 --steps 3 --warmup 1"
cp $C/variants/lib_sb2.so $C/libspades_mi355x.so
timeout 900 python -m pytest tests/test_prededupe_gpu.py tests/test_pm_route_gpu.py tests/test_ext_route_gpu.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -5 $O/tests.log
SMX_DEBUG=1 timeout 600 $B --steps 1 --warmup 0 > $O/dbg_sb2.json 2> $O/dbg_sb2.err
timeout 600 $B > $O/b_sb2.json 2> $O/b_sb2.err
timeout 600 $B --opt skm_cap=4096 > $O/b_sb2_cap4096.json 2> $O/b_sb2_cap4096.err
cp $C/variants/lib_sb4.so $C/libspades_mi355x.so
timeout 600 $B > $O/b_sb4.json 2> $O/b_sb4.err
timeout 600 $B --opt skm_v2=0 > $O/b_v1.json 2> $O/b_v1.err
for f in $O/b_*.json; do echo $f; python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["stages_ms"], d.get("construct",{}).get("checks"))
except Exception as e: print("ERR", e)
PY
done
grep -h "dedupe chunks\|skm_scan phase\|prededupe" $O/dbg_sb2.err | head -20
