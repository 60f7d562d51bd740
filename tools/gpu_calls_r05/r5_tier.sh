#!/bin/bash
# the whole GPU tier + smoke on the current build
out=gpurun_out/r5n; mkdir -p $out; exec > $out/log.txt 2>&1
timeout 1300 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $out/gpu_tests.log 2>&1; tail -4 $out/gpu_tests.log | head -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
