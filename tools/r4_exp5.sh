#!/bin/bash
set -u
O=gpurun_out/r4e5; mkdir -p $O
C=spades_amd/csrc
B="python bench.py --no-cpu-baseline --end-to-end 0 --extra-kmercount 0 --steps 3 --warmup 1"
cp $C/libspades_mi355x.so $C/variants/lib_default.so
for v in default scan5 scan6 scan8; do
  cp $C/variants/lib_$v.so $C/libspades_mi355x.so
  timeout 600 $B > $O/b_$v.json 2> $O/b_$v.err
done
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
