#!/bin/bash
# Round 6, call 19: the whole GPU tier and smoke() on the build with the fused node table, packed scan, short-path walk_write.
out=gpurun_out/r6s; mkdir -p $out; exec > $out/log.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
