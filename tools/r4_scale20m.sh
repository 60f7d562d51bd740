#!/bin/bash
# GPU box: the 20 M-read / 100 Mbp golden of the real spades-gbuilder (tests/golden/scale_20000k_g100000k_s79.json) on all three routes
mkdir -p gpurun_out
timeout 3000 python -m pytest tests/test_scale_gpu.py -x -q -k "20000k" --durations=10 2>&1 | tee gpurun_out/r4_scale20m.log | tail -25
