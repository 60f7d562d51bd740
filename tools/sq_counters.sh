#!/bin/bash
# SQ counters of one kernel (substring $1) for a 20 M-read step: where do the dedupe kernel's wave cycles go?
set -u
pat=${1:-k_skm_dedupe2}; shift || true
root=$(pwd); O=$root/gpurun_out/sqpmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $root/bench.py --no-cpu-baseline --end-to-end 0 --extra-kmercount 0 --steps 1 --warmup 0 --reads 20e6 --genome 100e6 --scaling-reference 0 --early-tip-extra 0 --no-file-on-demand $*"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- $B > $O/p$i.json 2> $O/p$i.err
  f=$(ls $O/p$i/*/*counter_collection.csv 2>/dev/null | head -1)
  python - "$f" "$pat" <<'PY'
import csv,sys,collections
f,pat=sys.argv[1],sys.argv[2]
acc=collections.defaultdict(float); n=collections.Counter()
try:
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    for k,v in acc.items(): print(k, v, "dispatches", n[k])
except Exception as e: print("ERR",e,f)
PY
  rm -rf $O/p$i
done
