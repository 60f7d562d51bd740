#!/bin/bash
set -u
O=gpurun_out/r4e3; mkdir -p $O
B="python bench.py --no-cpu-baseline --end-to-end 0 --extra-kmercount 0 --steps 1 --warmup 0 --reads 20e6 --genome 100e6"
SMX_DEBUG=1 timeout 600 $B > $O/dbg_fold.json 2> $O/dbg_fold.err
SMX_DEBUG=1 timeout 600 $B --opt skm_fold=0 > $O/dbg_nofold.json 2> $O/dbg_nofold.err
grep -h "dedupe \|prededupe:" $O/dbg_fold.err | head -8
grep -h "dedupe \|prededupe:" $O/dbg_nofold.err | head -8
