#!/usr/bin/env python3
"""GPU box: end-to-end wall time of the CLI clones on an uncompressed FASTQ, device FASTQ parsing vs the host parser."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
d = "/dev/shm/smx_ingest"; os.makedirs(d, exist_ok=True)
fq = os.path.join(d, "r.fq")
rng = np.random.default_rng(1)
G = rng.integers(0, 4, 5_000_000, dtype=np.uint8)
pos = rng.integers(0, len(G) - 150, n)
codes = G[pos[:, None] + np.arange(150)[None, :]]
err = rng.random(codes.shape) < 0.01
codes = np.where(err, (codes + rng.integers(1, 4, codes.shape)) % 4, codes).astype(np.uint8)
rec = np.empty((n, 12 + 150 + 3 + 150 + 1), dtype=np.uint8)  # "@r%09d\n" = 12 bytes
hdr = np.char.zfill(np.arange(n).astype(str), 9)
rec[:, 0] = ord("@"); rec[:, 1] = ord("r")
rec[:, 2:11] = np.frombuffer("".join(hdr).encode(), dtype=np.uint8).reshape(n, 9)
rec[:, 11] = 10
rec[:, 12:162] = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
rec[:, 162] = 10; rec[:, 163] = ord("+"); rec[:, 164] = 10
rec[:, 165:315] = ord("I"); rec[:, 315] = 10
rec.tofile(fq)
print(f"{n} reads, {os.path.getsize(fq) / 1e6:.0f} MB FASTQ")
subprocess.check_call(["gzip", "-1", "-k", fq])
for tool, args in (("spades-gbuilder-mi355x", [fq, os.path.join(d, "o.gfa"), "-k", "55", "-t", "16", "--gfa"]),
                   ("spades-kmercount-mi355x", ["-k", "55", "-w", d, fq]),
                   ("spades-gbuilder-mi355x", [fq + ".gz", os.path.join(d, "o.gfa"), "-k", "55", "-t", "16", "--gfa"])):
    for env_extra, tag in (({}, "device FASTQ parse"), ({"SMX_HOST_PARSE": "1"}, "host parser")):
        env = dict(os.environ, **env_extra)
        exe = os.path.join(ROOT, "spades_amd", "tools", tool)
        best = 1e9
        for it in range(2):
            e2 = dict(env, SMX_DEBUG="1") if it == 1 else env
            t0 = time.time(); r = subprocess.run([exe] + args, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=e2, check=True); best = min(best, time.time() - t0)
        print("   " + " | ".join(l[7:].strip() for l in r.stderr.decode().splitlines() if l.startswith("[tool]")))
        print(f"{tool:26s} {tag:20s} {best:6.2f} s  ({n / best / 1e6:.2f} M reads/s end to end)")
import shutil; shutil.rmtree(d)
