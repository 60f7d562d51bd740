#!/bin/bash
# Round 5, first GPU call: the whole tier on the round's sources (incl. everything that had waited behind SMX_NEXT), NOT stopping at the
# first failure; the device loop collector at 9 937 loops; writers A/B; one quick bench line with the two timed regions.
#   gpurun --timeout 1500 -- 'bash tools/r5_call1.sh'
out=gpurun_out/r5a; mkdir -p $out; exec > $out/log.txt 2>&1
set -x
nproc; free -g | head -2; rocm-smi --showmeminfo vram | head -8
SMX_NEXT=1 timeout 1300 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=25 > $out/gpu_tests.log 2>&1; tail -45 $out/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt
SMX_OPTS=device_loops=1 SMX_NEXT=1 SMX_DEBUG=1 timeout 600 python -m pytest tests/test_scale_gpu.py -m gpu -x -q -k "plasmids" > $out/t_scale_device_loops.txt 2>&1; grep -E "g:loops|passed|failed" $out/t_scale_device_loops.txt | tail -12
SMX_NEXT=1 SMX_DEBUG=1 timeout 600 python -m pytest tests/test_scale_gpu.py -m gpu -x -q -k "plasmids" > $out/t_scale_host_loops.txt 2>&1; grep -E "g:loops|passed|failed" $out/t_scale_host_loops.txt | tail -12
for v in "" "SMX_WRITE_MMAP=1 SMX_WRITE_THREADS=32" "SMX_IO_THREADS=1"; do
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
timeout 900 python bench.py --no-cpu-baseline --extra-kmercount 0 --end-to-end 0 --steps 5 > $out/benchq.json 2> $out/benchq.err; tail -3 $out/benchq.err; python tools/bench_summary.py $out/benchq.json
