mkdir -p gpurun_out/c55
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/c55/tests.txt 2>&1; echo "tests rc=$?"; grep -v "^  File" gpurun_out/c55/tests.txt | tail -6
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c55/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/c55/smoke.txt
