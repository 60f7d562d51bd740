#!/bin/bash
set -u
O=gpurun_out/r4e4; mkdir -p $O
C=spades_amd/csrc
B="python bench.py --no-cpu-baseline --end-to-end 0 --extra-kmercount 0 --steps 3 --warmup 1"
cp $C/libspades_mi355x.so $C/variants/lib_default.so
timeout 600 $B > $O/b_wpe4.json 2> $O/b_wpe4.err
cp $C/variants/lib_wpe5.so $C/libspades_mi355x.so
timeout 600 $B > $O/b_wpe5.json 2> $O/b_wpe5.err
timeout 600 $B --opt skm_fold=0 > $O/b_wpe5_nofold.json 2> $O/b_wpe5_nofold.err
cp $C/variants/lib_default.so $C/libspades_mi355x.so
for f in $O/b_*.json; do echo $f; python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    st=d["roofline"]["stages_ms"]
    print(d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:v for k,v in st.items() if v>2 and k.startswith("kmers")}, d.get("construct",{}).get("checks",{}).get("graph_fingerprint"))
except Exception as e: print("ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
