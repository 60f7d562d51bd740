#!/bin/bash
# distributed walks on ONE rank over RCCL at the per-rank size of BASELINE config 4 (1 B reads on 8 GPUs = 125 M reads per rank, 30x over 625 Mbp)
out=gpurun_out/r5m; mkdir -p $out; exec > $out/log.txt 2>&1
f="^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|socket.cpp\|destroy_process_group"
timeout 1200 python tools/dwalk_probe.py 125e6 625e6 55 16 --no-reference 2>&1 | grep -v "$f" | cut -c1-500 | tail -8 | tee $out/walks.txt
timeout 900 python tools/dwalk_probe.py 125e6 625e6 55 16 --reference-only 2>&1 | grep -v "$f" | cut -c1-500 | tail -4 | tee $out/single.txt
a=$(grep "fingerprint walks:" $out/walks.txt | sed 's/.*: //'); b=$(grep "fingerprint single:" $out/single.txt | sed 's/.*: //')
[ -n "$a" ] && [ "$a" = "$b" ] && echo "GRAPHS IDENTICAL" || echo "DIFFERENT OR MISSING"
