#!/bin/bash
# Round 5, fifth GPU call: the bench line once more with the PMC table of these kernels in place (so that the recorded line quotes its
# traffic), the N > 1 code path on one rank (extract, RCCL all-to-all with itself, owner-side count; sharded construction extras), skewed data.
out=gpurun_out/r5g; mkdir -p $out; exec > $out/log.txt 2>&1
set -x
timeout 1500 python bench.py > $out/bench.json 2> $out/bench.err; tail -2 $out/bench.err; python tools/bench_summary.py $out/bench.json
timeout 900 python bench.py --force-sharded --no-cpu-baseline > $out/bench_sharded_1rank.json 2> $out/bench_sharded_1rank.err; tail -3 $out/bench_sharded_1rank.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r5g/bench_sharded_1rank.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "n_gpus", "rccl_world_size", "exchange_ms_max", "owner_count_ms_max")}, d.get("sharded_construct"), d.get("distributed_walks"))
PY
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --extra-kmercount 0 --end-to-end 0 > $out/bench_torchrun_n1.json 2> $out/bench_torchrun_n1.err; tail -2 $out/bench_torchrun_n1.err; python tools/bench_summary.py $out/bench_torchrun_n1.json | head -2
timeout 900 python bench.py --skew --no-cpu-baseline --extra-kmercount 0 --end-to-end 0 --sharded-construct 0 --distributed-walks 0 > $out/bench_skew.json 2> $out/bench_skew.err; tail -2 $out/bench_skew.err; python tools/bench_summary.py $out/bench_skew.json | head -3
