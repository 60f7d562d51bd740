#!/bin/bash
# round-2 call 1: box facts + baseline wall-clock at scale with the round-1 code
mkdir -p gpurun_out/c1; exec > gpurun_out/c1/log.txt 2>&1
nproc; free -g; lscpu | grep -E 'Model name|Socket|Thread|Core'; rocm-smi --showmeminfo vram | head -8
python - <<'PY'
import torch, time
x = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
d = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
for _ in range(2):
    torch.cuda.synchronize(); t = time.time(); d.copy_(x, non_blocking=True); torch.cuda.synchronize(); dt = time.time() - t
    print("H2D pinned 1 GiB: %.1f GB/s" % (1.073 / dt))
    torch.cuda.synchronize(); t = time.time(); x.copy_(d, non_blocking=True); torch.cuda.synchronize(); dt = time.time() - t
    print("D2H pinned 1 GiB: %.1f GB/s" % (1.073 / dt))
PY
export SMX_DEBUG=1
echo "=== graph 10M reads / 50M genome"; timeout 600 python tools/scale_probe.py 10e6 50e6 graph
echo "=== count 100M reads / 500M genome"; timeout 900 python tools/scale_probe.py 100e6 500e6 count
