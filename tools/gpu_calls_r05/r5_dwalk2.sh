#!/bin/bash
# distributed walks on ONE rank over RCCL (1) on the bench's own 100 M-read batch (BASELINE config 3's input; 4.29 G k-mers): the graph's fingerprint must be
# the bench line's (f57c4ddcec586f5d6912c6754406fabf); (2) at the per-rank size of BASELINE config 4 (1 B reads on 8 GPUs = 125 M reads per rank)
out=gpurun_out/r5l; mkdir -p $out; exec > $out/log.txt 2>&1
f="^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|socket.cpp\|destroy_process_group"
DWALK_SEED=1000 timeout 1200 python tools/dwalk_probe.py 100e6 500e6 55 16 --no-reference 2>&1 | grep -v "$f" | cut -c1-500 | tail -8
echo "== 125 M reads"
timeout 1200 python tools/dwalk_probe.py 125e6 625e6 55 16 --no-reference 2>&1 | grep -v "$f" | cut -c1-500 | tail -8
