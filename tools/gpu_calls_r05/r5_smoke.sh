#!/bin/bash
out=gpurun_out/r5p; mkdir -p $out; exec > $out/log.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
