#!/bin/bash
# First GPU call of round 5: everything round 4 wrote after its GPU time was spent, checked and measured in one go (≈25–30 GPU-minutes).
#   gpurun --timeout 2400 -- 'bash tools/r5_first_call.sh'
# Results under gpurun_out/r5a/ (copy what is to be judged into profiles/r05/).
out=gpurun_out/r5a; mkdir -p $out; exec > $out/log.txt 2>&1
set -x
# 1. the code that no GPU run stands behind: key-range split of a spilled bucket, loops through the packed host collector (part of the
#    default tier: smoke + graph tests), then the opt-in goldens (10 000 plasmids = 9 937 loops; k = 77 at 20 M reads)
SMX_NEXT=1 timeout 600 python -m pytest tests/test_spill_gpu.py -m gpu -x -q > $out/t_spill_next.txt 2>&1; tail -5 $out/t_spill_next.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt
SMX_NEXT=1 timeout 600 python -m pytest tests/test_graph_gpu.py tests/test_pm_route_gpu.py -m gpu -x -q > $out/t_graph.txt 2>&1; tail -3 $out/t_graph.txt   # (with NEXT: the device loops, option device_loops, against the oracle on every route; production-like partition density on small inputs, option skm_nkey_log2)
SMX_OPTS=device_loops=1 SMX_NEXT=1 SMX_DEBUG=1 timeout 900 python -m pytest tests/test_scale_gpu.py -m gpu -x -q -k "plasmids" > $out/t_scale_device_loops.txt 2>&1; grep -E "g:loops|passed|failed" $out/t_scale_device_loops.txt | tail -12   # 9 937 loops by the kernels; g:loops = their wall time (compare with the run below)
SMX_NEXT=1 SMX_DEBUG=1 timeout 900 python -m pytest tests/test_scale_gpu.py -m gpu -x -q -k "next_scale" > $out/t_scale_next.txt 2>&1; grep -v "^\[smx\] \(skm\|dedupe\|level\|leaf\)" $out/t_scale_next.txt | tail -30
# 2. the whole tier on these sources
timeout 1500 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; tail -4 $out/gpu_tests.log
# 3. writers and readers of the CLI clones: the mapped tmpfs sink / pread threads against the round-4 paths (20 M reads, FASTQ 6.3 GB on tmpfs)
for v in "SMX_WRITE_MMAP=0 SMX_IO_THREADS=1" "SMX_WRITE_MMAP=0" "SMX_IO_THREADS=1" "SMX_WRITE_THREADS=4" "SMX_WRITE_THREADS=16" ""; do
  tag=$(echo "e2e_${v:-default}" | tr ' =' '__')
  env $v timeout 600 python bench.py --reads 20e6 --genome 100e6 --steps 1 --warmup 0 --no-cpu-baseline --extra-kmercount 0 --sharded-construct 0 --distributed-walks 0 > $out/$tag.json 2> $out/$tag.err
  python - "$out/$tag.json" "$v" <<'PY'
import json, sys
try:
    e = json.load(open(sys.argv[1]))["end_to_end"]
    print(sys.argv[2] or "default", {k: (v.get("seconds"), v.get("stages_s")) for k, v in e.items() if isinstance(v, dict)})
except Exception as x:
    print(sys.argv[2], "FAILED", x)
PY
done
# 4. out-of-core at a budget below one bucket's runs (round 3: refused with exit 68 at 8 GB)
timeout 900 python tools/verify_spill.py 20e6 8 > $out/spill_20M_budget8G.log 2>&1; tail -4 $out/spill_20M_budget8G.log
# 5. the bench line
timeout 1500 python bench.py > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err; python tools/bench_summary.py $out/bench.json
