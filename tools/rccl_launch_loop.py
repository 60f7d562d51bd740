#!/usr/bin/env python3
"""N one-rank launches of the C++ RCCL hosts in a row under SMX_MGPU_WATCHDOG, no retry: the first launch that does not
come back leaves the watchdog's report (rank, last milestone, backtrace of the blocked thread) in the log and the loop goes on counting.
usage: rccl_launch_loop.py [launches=200] [watchdog_seconds=20]   (run on the GPU box; writes to stdout)"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import read_lines  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
WD = sys.argv[2] if len(sys.argv) > 2 else "20"
TOOLS = os.path.join(ROOT, "spades_amd", "tools")
KC, GB = os.path.join(TOOLS, "spades-kmercount-mi355x"), os.path.join(TOOLS, "spades-gbuilder-mi355x")
reads = [r for r in read_lines("reads_small.txt") if r]
td = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
fq = os.path.join(td, "r.fq")
with open(fq, "w") as f:
    for i, r in enumerate(reads):
        f.write(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n")
variants = [
    ([GB, fq, os.path.join(td, "g.gfa"), "-k", "21", "-t", "3", "--gfa", "--gpus", "1"], {"SMX_MGPU_SELF_RCCL": "1"}),
    ([GB, fq, os.path.join(td, "g.gfa"), "-k", "55", "-t", "3", "--gfa", "--gpus", "1", "-c"], {"SMX_MGPU_PARTS": "3"}),
    ([GB, fq, os.path.join(td, "g.gfa"), "-k", "33", "-t", "3", "--gfa", "--gpus", "1"], {"SMX_MGPU_KPOMERS": "1", "SMX_MGPU_SELF_RCCL": "1"}),
    ([KC, "-k", "21", "-w", td, "--gpus", "1", fq], {}),
    ([KC, "-k", "55", "-w", td, "--gpus", "1", fq], {"SMX_MGPU_SELF_RCCL": "1"}),
]
ok = fired = timed_out = other = 0
times = []
first_report = None
for i in range(N):
    argv, env = variants[i % len(variants)]
    t0 = time.time()
    try:
        r = subprocess.run(argv, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, SMX_MGPU_WATCHDOG=WD, **env), timeout=120)
        dt = time.time() - t0
        times.append(dt)
        if r.returncode == 0:
            ok += 1
        elif r.returncode == 75:
            fired += 1
            rep = r.stderr.decode(errors="replace")[-4000:]
            print(f"--- launch {i}: watchdog fired after {dt:.1f} s: {' '.join(argv[-6:])} {env}\n{rep}", flush=True)
            first_report = first_report or rep
        else:
            other += 1
            print(f"--- launch {i}: exit code {r.returncode}: {' '.join(argv[-6:])} {env}\n{r.stderr.decode(errors='replace')[-2000:]}", flush=True)
    except subprocess.TimeoutExpired as e:
        timed_out += 1
        print(f"--- launch {i}: no return in 120 s (the watchdog did not fire either): {' '.join(argv[-6:])} {env}\n{(e.stderr or b'').decode(errors='replace')[-3000:]}", flush=True)
times.sort()
print(f"{N} one-rank launches of the RCCL hosts (watchdog {WD} s per phase, no retry): {ok} ok, {fired} watchdog reports, {timed_out} without return, {other} other failures; "
      f"seconds per launch: median {times[len(times) // 2]:.2f}, max {times[-1]:.2f}" if times else "no launch finished", flush=True)
