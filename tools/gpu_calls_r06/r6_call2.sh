#!/bin/bash
# Round 6, second GPU call: the distributed walks of the LIBRARY (smx_shard_walks) on the MI355X — the dist / CLI test files, then at size on one rank:
# Python host over RCCL (tools/dwalk_probe.py) at 20 M reads and at BASELINE config 5's per-rank share (62.5 M reads, 30x), the C++ host
# (spades-gbuilder-mi355x --gpus 1, SMX_MGPU_WALKS=distributed) at 20 M reads against the single-process GFA; the CLI bimodality probe (item 7).
#   gpurun --timeout 2400 -- 'bash tools/gpu_calls_r06/r6_call2.sh'
out=gpurun_out/r6b; mkdir -p $out; exec > $out/log.txt 2>&1
set -x
timeout 900 python -m pytest tests/test_dist_gpu.py tests/test_zz_cli_rccl_gpu.py tests/test_count_gpu.py -m gpu -q -p no:cacheprovider --durations=8 > $out/gpu_tests_dist_cli.log 2>&1; tail -25 $out/gpu_tests_dist_cli.log
SMX_DEBUG=1 timeout 600 python tools/dwalk_probe.py 20e6 100e6 55 16 > $out/dwalk_20M.log 2>&1; grep -E "walks:|single GPU|distributed walks|torch peak|identical" $out/dwalk_20M.log | tail -40
timeout 300 python tools/dwalk_probe.py 62.5e6 312.5e6 55 16 --reference-only > $out/dwalk_62M_ref.log 2>&1; tail -3 $out/dwalk_62M_ref.log
SMX_DEBUG=1 timeout 900 python tools/dwalk_probe.py 62.5e6 312.5e6 55 16 --no-reference > $out/dwalk_62M.log 2>&1; grep -E "walks:|distributed walks|torch peak|fingerprint walks" $out/dwalk_62M.log | tail -40
timeout 900 bash tools/cli_bimodality_probe.sh 20000000 keep > $out/cli_bimodality.log 2>&1; cat $out/cli_bimodality.log
