#!/bin/bash
out=gpurun_out/r5h; mkdir -p $out; exec > $out/log.txt 2>&1
timeout 600 python bench.py --force-sharded --no-cpu-baseline --sharded-construct 0 --distributed-walks 0 --steps 1 --warmup 0 --reads 20e6 --genome 100e6 > $out/sh.json 2> $out/sh.err
echo "rc=$? lines=$(wc -l < $out/sh.json)"; head -c 200 $out/sh.json; echo; tail -3 $out/sh.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --force-sharded --no-cpu-baseline --sharded-construct 0 --distributed-walks 0 --steps 1 --warmup 0 --reads 20e6 --genome 100e6 > $out/sh2.json 2> $out/sh2.err
echo "rc=$? lines=$(wc -l < $out/sh2.json)"; head -c 200 $out/sh2.json; echo; tail -3 $out/sh2.err
