#!/usr/bin/env python3
"""GPU box: VERDICT r1 item 7 at its stated size — 100 M PE150 reads, k=55, spades-kmercount mode (all k-mers of read + RC, 16 buckets)
with the context's HBM budget forced to 64 GB, against the run that may use all of HBM (which also has to spill: the result alone is
~140 GB). Both results are served from host memory; they are compared bucket by bucket (count, wrapping sum and xor of all words,
first/last record) and every bucket is checked for strictly increasing records.
usage: verify_spill.py [reads=100e6] [budget_GB=64]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth_reads_device
from spades_amd import KMerDiskCounter, ReadKMerSplitter
from spades_amd.kmercount import Context

n = int(float(sys.argv[1])) // 32 * 32 if len(sys.argv) > 1 else 100_000_000
budget = int(float(sys.argv[2]) * 1e9) if len(sys.argv) > 2 else 64_000_000_000
dev = torch.device("cuda", 0)
words, start, ln, codes = synth_reads_device(1000, 500_000_000, n, dev, n_rate=0.001)
del codes
hw, hs, hl = words.cpu().numpy().view(np.uint64), start.cpu().numpy().view(np.uint64), ln.cpu().numpy().view(np.uint32)
del words, start, ln
torch.cuda.empty_cache()


def signatures(get_bucket, sizes):
    sig, ok = [], True
    for bk in range(16):
        r = get_bucket(bk)
        inc = bool(np.all((r[1:, 0] > r[:-1, 0]) | ((r[1:, 0] == r[:-1, 0]) & (r[1:, 1] > r[:-1, 1])))) if len(r) > 1 else True
        ok &= inc
        sig.append((len(r), int(r.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(r.reshape(-1))), r[0].tolist() if len(r) else None, r[-1].tolist() if len(r) else None))
        del r
    return sig, ok


def run(b, workdir=None):
    """workdir: count + merge with the destination known (smx_count_to_file): an out-of-core result is streamed to <workdir>/final_kmers and host
    memory holds the shrinking runs alone — the leg that fits the 300 GiB memory cgroup of the GPU boxes (round 6); else served from host memory"""
    ctx = Context(hbm_budget=b)
    sp = ReadKMerSplitter(55, "A", ctx)
    sp.push_back_packed(hw[:-8], hs, hl)
    t0 = time.time()
    if workdir:
        st = KMerDiskCounter(workdir, sp).CountAll(16)
        dt = time.time() - t0
        sizes = st.bucket_sizes()
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        mm = np.memmap(os.path.join(workdir, "final_kmers"), dtype=np.uint64, mode="r").reshape(-1, 2)
        sig, ok = signatures(lambda bk: np.array(mm[off[bk]:off[bk + 1]]), sizes)
        where = "streamed to the file" if st.device_ptr() == 0 else "resident, then written"
        del mm
    else:
        st = KMerDiskCounter(None, sp).Count(16)
        dt = time.time() - t0
        sig, ok = signatures(st.bucket, st.bucket_sizes())
        where = "on host" if st.device_ptr() == 0 else "resident"
    rss = max(int(l.split()[1]) for l in open("/proc/self/status") if l.startswith("VmHWM")) / 1e6
    print(f"budget {b / 1e9:g} GB: {dt:.1f} s, {st.total_kmers()} records ({st.total_kmers() * 16 / 1e9:.1f} GB), {where}, "
          f"buckets strictly increasing: {ok}; peak resident host memory of this process so far {rss:.1f} GB", flush=True)
    ctx.close()
    return sig, ok


import tempfile
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
    a, oka = run(0, td)   # all of HBM (the result is held as two strands and written bucket by bucket)
os.makedirs(os.environ.get("SMX_SPILL_WORKDIR", "/dev/shm/smx_spill_wd"), exist_ok=True)
b, okb = run(budget, os.environ.get("SMX_SPILL_WORKDIR", "/dev/shm/smx_spill_wd"))
try:
    os.remove(os.path.join(os.environ.get("SMX_SPILL_WORKDIR", "/dev/shm/smx_spill_wd"), "final_kmers"))
except OSError:
    pass
same = a == b
print("identical signatures:", same)
print("ALL OK" if same and oka and okb else "FAILED")
sys.exit(0 if same and oka and okb else 1)
