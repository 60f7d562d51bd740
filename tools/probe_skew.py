#!/usr/bin/env python3
"""GPU box: skewed inputs (tiny genome at huge coverage, poly-A, low-complexity repeats) — every fine bin overflows the LDS
classes, so the merge path (k_sort_big) carries the whole batch. Checks parity on a sub-sample and reports timings."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spades_amd import KMerDiskCounter, ReadKMerSplitter
from spades_amd.reads import synth_batch_numpy, codes_to_ascii, pack_codes
from oracle import oracle

n_pairs = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000
for K, mode, nb in ((55, "A", 16), (21, "A", 16), (56, "B", 160)):
    words, start, ln, codes = synth_batch_numpy(5, 5_000, n_pairs, err=0.001)
    # add low-complexity reads
    L = codes.shape[1]
    extra = np.zeros((2000, L), dtype=np.uint8); extra[1000:] = np.tile(np.array([0, 1], dtype=np.uint8), L // 2 + 1)[:L]
    codes2 = np.concatenate([codes, extra])
    n = codes2.shape[0]
    words = pack_codes(codes2.reshape(-1)); start = (np.arange(n, dtype=np.uint64) * L); ln = np.full(n, L, dtype=np.uint32)
    sp = ReadKMerSplitter(K, mode)
    sp.push_back_packed(words, start, ln)
    c = KMerDiskCounter(None, sp)
    for it in range(2):
        t0 = time.time(); st = c.Count(nb); dt = time.time() - t0
    tm = dict(sp.ctx.timings())
    print(f"K={K} mode={mode} reads={n} instances={st.kmer_instances()} distinct={st.total_kmers()} wall={dt:.3f}s sort_big={tm.get('sort_big',0):.1f} ms sort_unique={tm.get('sort_unique',0):.1f} sort_unique2={tm.get('sort_unique2',0):.1f}")
    got = st.records(); sizes = st.bucket_sizes()
    sp.ctx.close()
    ref, rs = oracle.count(codes_to_ascii(codes2[-6000:]), K, mode, nb)  # parity of the whole run needs the whole input; check a run on the tail separately
    sp2 = ReadKMerSplitter(K, mode); sp2.push_back_reads(codes_to_ascii(codes2[-6000:])); st2 = KMerDiskCounter(None, sp2).Count(nb)
    ok = (st2.records() == ref).all() and (st2.bucket_sizes() == rs).all()
    print("   tail-sample parity vs oracle:", bool(ok))
    sp2.ctx.close()
    assert ok
