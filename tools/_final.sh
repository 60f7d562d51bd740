mkdir -p gpurun_out/c56
timeout 900 python -m pytest tests/test_ext_route_gpu.py -m gpu -x -q > gpurun_out/c56/text.txt 2>&1; grep -v "^  File" gpurun_out/c56/text.txt | tail -4
timeout 900 python bench.py --no-cpu-baseline --extra-kmercount 0 > gpurun_out/c56/bench_ext.json 2> gpurun_out/c56/e1.err
timeout 900 python bench.py --kpomer-route --no-cpu-baseline --extra-kmercount 0 > gpurun_out/c56/bench_kpo.json 2> gpurun_out/c56/e2.err
tail -2 gpurun_out/c56/e1.err; tail -2 gpurun_out/c56/e2.err
