#!/usr/bin/env python3
"""Quick perf probe on the GPU box: stage timings of smx_count on a synthetic batch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spades_amd import KMerDiskCounter, ReadKMerSplitter
from spades_amd.reads import synth_batch_numpy

n_pairs = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000
for K, mode, nb in ((21, "A", 16), (55, "A", 16), (56, "B", 160)):
    words, start, ln, _ = synth_batch_numpy(1, max(100_000, n_pairs * 10), n_pairs)
    sp = ReadKMerSplitter(K, mode)
    sp.push_back_packed(words, start, ln)
    c = KMerDiskCounter(None, sp)
    for it in range(2):
        t0 = time.time(); st = c.Count(nb); t1 = time.time()
    tm = sp.ctx.timings()
    tot = sum(ms for _, ms in tm)
    print(f"K={K} mode={mode} reads={2*n_pairs} instances={st.kmer_instances()} distinct={st.total_kmers()} wall={t1-t0:.3f}s gpu_ms={tot:.2f} "
          f"Mreads/s(gpu)={2*n_pairs/tot/1e3:.1f}")
    print("   " + " ".join(f"{n}={ms:.2f}" for n, ms in tm))
    sp.ctx.close()
