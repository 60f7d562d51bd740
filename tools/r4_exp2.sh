#!/bin/bash
set -u
O=gpurun_out/r4e2; mkdir -p $O
B="python bench.py --no-cpu-baseline --end-to-end 0 --extra-kmercount 0 --steps 3 --warmup 1"
timeout 900 python -m pytest tests/test_prededupe_gpu.py tests/test_pm_route_gpu.py tests/test_ext_route_gpu.py tests/test_count_gpu.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -5 $O/tests.log
SMX_DEBUG=1 timeout 600 $B --steps 1 --warmup 0 > $O/dbg.json 2> $O/dbg.err
timeout 600 $B > $O/b_fold.json 2> $O/b_fold.err
timeout 600 $B --opt skm_fold=0 > $O/b_nofold.json 2> $O/b_nofold.err

for f in $O/b_*.json; do echo $f; python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    st=d["roofline"]["stages_ms"]
    print(d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:v for k,v in st.items() if v>2}, d.get("construct",{}).get("checks",{}).get("graph_fingerprint"))
except Exception as e: print("ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
grep -h "dedupe chunks\|skm_scan phase\|prededupe" $O/dbg.err | head -8
