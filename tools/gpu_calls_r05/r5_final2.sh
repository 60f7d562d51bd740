#!/bin/bash
# Round 5, after the plain-record ("nx") variant of the partition-major route went in (k = 29, 31, 61, 63, 93, 95, 125, 127): the whole GPU tier
# (its k = 127 golden of the real spades-gbuilder at 2 M reads now takes the default route), then — kernels of the step changed by a parameter —
# the rocprofv3 passes and the bench line again.
out=gpurun_out/r5i; mkdir -p $out; exec > $out/log.txt 2>&1
set -x
timeout 1300 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $out/gpu_tests.log 2>&1; tail -5 $out/gpu_tests.log
bash tools/profile_bench.sh r05b --sharded-construct 0 --distributed-walks 0 --sync-upload
timeout 1500 python bench.py > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err; python tools/bench_summary.py $out/bench.json
SMX_DEBUG=1 timeout 300 python - > $out/k127.txt 2>&1 <<'PY'
import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import synth, numpy as np
from spades_amd.gbuilder import GraphBuilder
from spades_amd.kmercount import Context
codes = synth.synth_codes(79, 10_000_000, 2_000_000, 0.01, 0.001)
bases, off = synth.ascii_and_offsets(codes)
for nx in (1, 0):
    ctx = Context(); ctx.set_option("nx_route", nx)
    gb = GraphBuilder(127, 8, ctx); gb.reads.push_back_ascii(bases.tobytes(), off)
    gb.build(); t0 = time.time(); 
    for _ in range(3): gb.build()
    dt = (time.time() - t0) / 3
    names = [n for n, _ in ctx.timings()]
    print("nx_route", nx, "pm" if "pm_tab" in names else "kpo", "build s", round(dt, 4), gb.info(), flush=True)
    ctx.close()
PY
grep "nx_route" $out/k127.txt
