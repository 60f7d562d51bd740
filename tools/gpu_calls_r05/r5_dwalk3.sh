#!/bin/bash
# distributed walks on ONE rank over RCCL at the per-rank size of BASELINE config 4 (1 B reads on 8 GPUs = 125 M reads per rank, 30x over 625 Mbp).
# (No single-GPU reference at this size: the default route needs ~286 GB for 125 M reads — "device allocation of 85.9 GB failed", first attempt of this script.)
out=gpurun_out/r5m; mkdir -p $out; exec > $out/log.txt 2>&1
f="^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|socket.cpp\|destroy_process_group"
SMX_DEBUG_DIST=1 timeout 1200 python tools/dwalk_probe.py 125e6 625e6 55 16 --no-reference 2>&1 | grep -v "$f" | cut -c1-500 | tail -8 | tee $out/walks.txt
