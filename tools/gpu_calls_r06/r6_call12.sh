#!/bin/bash
# Round 6, call 12: (a) what bounds random reads on this chip — by bytes per access, working set and mapping (hipMalloc vs the arena's VMM chunks), with FETCH_SIZE
# calibrated on those shapes; (b) non-temporal loads / stores on the step's streamed lists and scattered stores (A/B libraries, tools/build_ab.sh).
#   gpurun --timeout 2400 -- 'bash tools/gpu_calls_r06/r6_call12.sh'
out=gpurun_out/r6k; mkdir -p $out; exec > $out/log.txt 2>&1
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
U=$R/tools/ab/ubench_random_access
for w in 0.125 1 8 64 200; do timeout 120 $U malloc 0 $w; done
for c in 512 1024 64; do timeout 200 $U vmm $c 64; done
timeout 200 $U vmm 512 200
timeout 300 $U vmm 2 16
timeout 120 $U malloc 0 16
echo "== counters"
rocprofv3 -L 2>/dev/null | grep -i -E "utcl|tlb|TCC_EA0_RDREQ|TCC_EA0_WRREQ|TCC_MISS|TCC_REQ" | cut -c1-160 | head -40
for pmc in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_MISS_sum TCC_REQ_sum"; do
  d=$R/$out/pmc_$(echo $pmc | tr ' ' '_'); rm -rf $d
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace -d $d -o u -- $U malloc 0 64 > $d.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  echo "== $pmc ($f)"; python3 - "$f" <<'PY'
import csv, sys, collections
if not sys.argv[1]: sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:40]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k, {c: (v / n[(k, c)]) for c, v in acc[k].items()}, "per launch")
PY
done
echo "== per launch: 256*64*256 lanes * 32 iterations = 134217728 lane accesses (/LPG groups)"
cd $R
common="--no-cpu-baseline --extra-kmercount 0 --end-to-end 0 --sharded-construct 0 --distributed-walks 0 --no-file-on-demand --early-tip-extra 0 --scaling-reference 0 --steps 3 --warmup 1"
for v in base nt_permute_st nt_permute_both nt_tab nt_walk_lists nt_remote nt_wwrite nt_all base; do
  [ -f tools/ab/lib_$v.so ] || continue
  SMX_BENCH_LIB=tools/ab/lib_$v.so timeout 400 python bench.py $common > $out/ab_$v.json 2> $out/ab_$v.err
  echo "== $v"; python tools/bench_summary.py $out/ab_$v.json 2>&1 | sed -n 1,5p
done
