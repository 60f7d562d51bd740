#!/bin/bash
# Round 6, call 28: the GPU tier twice more with the whole output kept (one of six runs of it on the closing builds dumped core; its log had been cut to the tail)
out=gpurun_out/r6z4; mkdir -p $out
for i in 1 2; do
  timeout 2400 python -m pytest tests -m gpu -x -v > $out/full_$i.log 2>&1
  echo "run $i rc=$?" >> $out/last.txt
  grep -n "passed\|failed" $out/full_$i.log | tail -2 >> $out/last.txt
  grep -n "Fatal\|Abort\|Cannot find\|fault\|Current thread" -A12 $out/full_$i.log | grep -v "dist-packages" | head -40 >> $out/last.txt
done
