mkdir -p gpurun_out/c52
bash tools/profile_bench.sh r02ext > gpurun_out/c52/prof.log 2>&1
timeout 900 python bench.py --kpomer-route --no-cpu-baseline --extra-kmercount 0 > gpurun_out/c52/bench_kpomer_route.json 2> gpurun_out/c52/kpo.err
timeout 900 python bench.py --gpus 1 --force-sharded --steps 3 --warmup 1 > gpurun_out/c52/bench_sharded_1rank_100M.json 2> gpurun_out/c52/sh.err
tail -3 gpurun_out/c52/prof.log gpurun_out/c52/kpo.err gpurun_out/c52/sh.err
