#!/bin/bash
# distributed walks on ONE rank over RCCL at the per-rank size of BASELINE config 5 (500 M reads on 8 GPUs = 62.5 M reads per rank, 30x): the walks and
# the single-GPU reference build in processes of their own (the library's arena only grows), fingerprints compared here
out=gpurun_out/r5k; mkdir -p $out; exec > $out/log.txt 2>&1
f="^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|socket.cpp"
timeout 900 python tools/dwalk_probe.py 62.5e6 312.5e6 55 16 --no-reference 2>&1 | grep -v "$f" | cut -c1-400 | tee $out/walks.txt
timeout 900 python tools/dwalk_probe.py 62.5e6 312.5e6 55 16 --reference-only 2>&1 | grep -v "$f" | cut -c1-400 | tee $out/single.txt
a=$(grep "fingerprint walks:" $out/walks.txt | sed 's/.*: //'); b=$(grep "fingerprint single:" $out/single.txt | sed 's/.*: //')
echo "walks  $a"; echo "single $b"; [ -n "$a" ] && [ "$a" = "$b" ] && echo "GRAPHS IDENTICAL" || echo "DIFFERENT OR MISSING"
