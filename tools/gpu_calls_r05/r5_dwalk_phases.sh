#!/bin/bash
out=gpurun_out/r5o; mkdir -p $out; exec > $out/log.txt 2>&1
SMX_DEBUG=1 DWALK_SEED=1000 timeout 900 python tools/dwalk_probe.py 100e6 500e6 55 16 --no-reference 2>&1 | grep "\[dist\]\|distributed walks:\|torch peak" | cut -c1-200
