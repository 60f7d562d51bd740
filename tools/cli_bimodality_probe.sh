#!/bin/bash
# "build graph" of spades-gbuilder-mi355x was 1.387 s on the driver's box and 0.131 s in the builder's run of the same 20 M-read leg.
# Fresh processes, one after the other, with the arena's own account of its mapping time (SMX_DEBUG: "[smx] arena: ... mapped in ... s") —
# with the prewarm helper (default) and without it (SMX_PREWARM_X=0,0). usage: cli_bimodality_probe.sh <reads, default 20000000>
set -u
N=${1:-20000000}
cd "$(dirname "$0")/.."
D=$(mktemp -d -p /dev/shm)
python - "$N" "$D/r.fq" <<'PY'
import sys, numpy as np
sys.path.insert(0, "tests")
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
for mode in "default" "0,0" "default" "0,0" "default" "0,0"; do
  echo "=== spades-gbuilder-mi355x, prewarm: $mode"
  if [ "$mode" = "default" ]; then unset SMX_PREWARM_X; else export SMX_PREWARM_X=$mode; fi
  env SMX_DEBUG=1 spades_amd/tools/spades-gbuilder-mi355x "$D/r.fq" "$D/o.gfa" -k 55 -t 16 --gfa 2>&1 | grep -E "^\[tool\]|arena:|g:"
  rm -f "$D/o.gfa"
done
for mode in "default" "0,0" "default" "0,0"; do
  echo "=== spades-kmercount-mi355x, prewarm: $mode"
  if [ "$mode" = "default" ]; then unset SMX_PREWARM_X; else export SMX_PREWARM_X=$mode; fi
  env SMX_DEBUG=1 spades_amd/tools/spades-kmercount-mi355x -k 55 -w "$D" "$D/r.fq" 2>&1 | grep -E "^\[tool\]|arena:"
  rm -f "$D/final_kmers"
done
# the C++ multi-GPU host at one rank on the same input: gathered structure, then the k-mer file left sharded (smx_shard_walks over ncclSend / ncclRecv
# with itself) — both must write the single-process GFA byte for byte
env spades_amd/tools/spades-gbuilder-mi355x "$D/r.fq" "$D/ref.gfa" -k 55 -t 16 --gfa > /dev/null 2>&1
for w in gathered distributed; do
  echo "=== spades-gbuilder-mi355x --gpus 1, SMX_MGPU_WALKS=$w"
  t0=$(date +%s.%N)
  env SMX_DEBUG=1 SMX_MGPU_WALKS=$w SMX_MGPU_SELF_RCCL=1 SMX_MGPU_PARTS=4 SMX_MGPU_WATCHDOG=120 spades_amd/tools/spades-gbuilder-mi355x "$D/r.fq" "$D/o.gfa" -k 55 -t 16 --gfa --gpus 1 2>&1 | grep -E "walks:|rank 0\] (owner|distributed|graph built|output)|doubling" | tail -30
  t1=$(date +%s.%N)
  echo "wall $(python3 -c "print(round($t1 - $t0, 2))") s; identical to the single-process GFA: $(cmp -s "$D/ref.gfa" "$D/o.gfa" && echo yes || echo NO)"
  rm -f "$D/o.gfa"
done
rm -rf "$D"
