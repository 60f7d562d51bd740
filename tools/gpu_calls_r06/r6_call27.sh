#!/bin/bash
# Round 6, call 27: the GPU tier with test names, whole output kept (the closing run dumped core somewhere and the script had kept only the tail of its log)
out=gpurun_out/r6z3; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
timeout 2400 python -m pytest tests -m gpu -x -v > $out/full.log 2>&1
grep -n "PASSED\|FAILED\|ERROR" $out/full.log | tail -3 > $out/last.txt
grep -n "Fatal\|Abort\|Cannot find\|File \"/tmp/code" $out/full.log | head -40 >> $out/last.txt
tail -5 $out/full.log >> $out/last.txt
