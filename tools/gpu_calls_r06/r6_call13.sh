#!/bin/bash
# Round 6, call 13: the node table written by the dedupe stage itself (pm_fuse_tab, default) against k_pm_tab afterwards (pm_fuse_tab=0); parity tests of the route;
# second batch of non-temporal variants (dedupe copy-out, dedupe slot loads, scan staging stores); FETCH_SIZE calibrated on random reads (tools/ubench_random_access).
#   gpurun --timeout 2400 -- 'bash tools/gpu_calls_r06/r6_call13.sh'
out=gpurun_out/r6m; mkdir -p $out; exec > $out/log.txt 2>&1
timeout 900 python -m pytest tests/test_pm_route_gpu.py tests/test_graph_gpu.py -m gpu -x -q -n 4 2>&1 | tail -5
common="--no-cpu-baseline --extra-kmercount 0 --end-to-end 0 --sharded-construct 0 --distributed-walks 0 --no-file-on-demand --early-tip-extra 0 --scaling-reference 0 --steps 3 --warmup 1"
run() {  # name, lib, extra flags
  SMX_BENCH_LIB=tools/ab/lib_$2.so timeout 400 python bench.py $common $3 > $out/ab_$1.json 2> $out/ab_$1.err
  echo "== $1"; python tools/bench_summary.py $out/ab_$1.json 2>&1 | sed -n 1,5p; tail -2 $out/ab_$1.err | cut -c1-300
}
run fused base ""
run unfused base "--opt pm_fuse_tab=0"
run nt_dd_out nt_dd_out ""
run nt_dd_in nt_dd_in ""
run nt_scan_st nt_scan_st ""
run fused_again base ""
SMX_BENCH_LIB=tools/ab/lib_base.so SMX_DEBUG=1 timeout 400 python bench.py $common --steps 1 --warmup 0 > $out/debug.json 2> $out/debug.err
grep -E "dedupe chunks|pm_tab:" $out/debug.err | tail -3
run fused_tip95 base "--opt early_tip_bound=95"
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
U=$R/tools/ab/ubench_random_access
for pmc in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  d=$R/$out/pmc_$(echo $pmc | tr ' ' '_'); rm -rf $d
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $d -- $U malloc 0 64 > $d.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  echo "== $pmc ($f)"; python3 - "$f" <<'PY'
import csv, sys, collections
if len(sys.argv) < 2 or not sys.argv[1]: sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:40]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k, {c: (v / n[(k, c)]) for c, v in acc[k].items()}, "per launch")
PY
done
echo "== per launch: 256*64*256 lanes * 32 iterations = 134217728 lane accesses (/LPG groups); FETCH_SIZE in KB"
