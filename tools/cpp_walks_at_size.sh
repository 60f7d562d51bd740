#!/bin/bash
# BASELINE config 5's per-rank share through the product's C++ host: spades-gbuilder-mi355x --gpus 1 with the k-mer file left sharded
# (SMX_MGPU_WALKS=distributed: smx_shard_walks over the host's own collectives; the rank's own segment is a device copy, as between the ranks of an
# N-GPU run every segment but one goes over xGMI), against the single-process GFA of the same reads. usage: cpp_walks_at_size.sh <reads> [k] [threads]
set -u
N=${1:-62500000}; K=${2:-55}; T=${3:-16}
cd "$(dirname "$0")/.."
D=$(mktemp -d -p /dev/shm)
python - "$N" "$D/r.fq" <<'PY'
import sys, numpy as np
n, path = int(sys.argv[1]), sys.argv[2]
rng = np.random.default_rng(5)
G = n * 150 // 30
genome = rng.integers(0, 4, G, dtype=np.uint8)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
with open(path, "wb") as f:
    for c0 in range(0, n, 1 << 20):
        m = min(1 << 20, n - c0)
        p = rng.integers(0, G - 150, m)
        blk = genome[p[:, None] + np.arange(150)[None, :]]
        err = rng.random(blk.shape) < 0.01
        blk = np.where(err, (blk + rng.integers(1, 4, blk.shape, dtype=np.uint8)) % 4, blk)
        rec = np.empty((m, 12 + 150 + 3 + 150 + 1), dtype=np.uint8)
        rec[:, 0], rec[:, 1] = ord("@"), ord("r")
        rec[:, 2:11] = np.frombuffer("".join(np.char.zfill(np.arange(c0, c0 + m).astype(str), 9)).encode(), dtype=np.uint8).reshape(m, 9)
        rec[:, 11] = 10
        rec[:, 12:162] = lut[blk]
        rec[:, 162], rec[:, 163], rec[:, 164] = 10, ord("+"), 10
        rec[:, 165:315] = ord("I")
        rec[:, 315] = 10
        rec.tofile(f)
PY
ls -la "$D/r.fq"
t0=$(date +%s.%N)
env SMX_DEBUG=1 spades_amd/tools/spades-gbuilder-mi355x "$D/r.fq" "$D/ref.gfa" -k $K -t $T --gfa 2>&1 | grep -E "^\[tool\]"
t1=$(date +%s.%N)
echo "single process: wall $(python3 -c "print(round($t1 - $t0, 2))") s, $(stat -c %s "$D/ref.gfa") bytes of GFA"
for w in distributed gathered; do
  echo "=== spades-gbuilder-mi355x --gpus 1, SMX_MGPU_WALKS=$w (own segment: device copy)"
  t0=$(date +%s.%N)
  env SMX_DEBUG=1 SMX_MGPU_WALKS=$w SMX_MGPU_PARTS=4 SMX_MGPU_WATCHDOG=300 spades_amd/tools/spades-gbuilder-mi355x "$D/r.fq" "$D/o.gfa" -k $K -t $T --gfa --gpus 1 2>&1 | grep -E "walks:|rank 0\] (communicator up|input submitted|owner-side|distributed|graph built|output)|doubling|fit" | tail -40
  t1=$(date +%s.%N)
  echo "wall $(python3 -c "print(round($t1 - $t0, 2))") s; identical to the single-process GFA: $(cmp -s "$D/ref.gfa" "$D/o.gfa" && echo yes || echo NO)"
  rm -f "$D/o.gfa"
done
rm -rf "$D"
