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


def run(b):
    ctx = Context(hbm_budget=b)
    sp = ReadKMerSplitter(55, "A", ctx)
    sp.push_back_packed(hw[:-8], hs, hl)
    t0 = time.time()
    st = KMerDiskCounter(None, sp).Count(16)
    dt = time.time() - t0
    sig = []
    ok = True
    for bk in range(16):
        r = st.bucket(bk)
        inc = bool(np.all((r[1:, 0] > r[:-1, 0]) | ((r[1:, 0] == r[:-1, 0]) & (r[1:, 1] > r[:-1, 1])))) if len(r) > 1 else True
        ok &= inc
        sig.append((len(r), int(r.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(r.reshape(-1))), r[0].tolist() if len(r) else None, r[-1].tolist() if len(r) else None))
        del r
    print(f"budget {b / 1e9:g} GB: {dt:.1f} s, {st.total_kmers()} records ({st.total_kmers() * 16 / 1e9:.1f} GB), on host: {st.device_ptr() == 0}, "
          f"buckets strictly increasing: {ok}", flush=True)
    ctx.close()
    return sig, ok


a, oka = run(0)
b, okb = run(budget)
same = a == b
print("identical signatures:", same)
print("ALL OK" if same and oka and okb else "FAILED")
sys.exit(0 if same and oka and okb else 1)
