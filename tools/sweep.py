import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_reads_device
from spades_amd import KMerDiskCounter, ReadKMerSplitter
from spades_amd.kmercount import Context
dev = torch.device("cuda", 0)
n = 10_000_000 // 32 * 32
words, start, ln, codes = synth_reads_device(1000, 50_000_000, n, dev)
del codes
for K, mode, nb in ((55, "A", 16),):
    for opts in ({}, {"leaf_tab": 1}, {"leaf_grid": 1024}, {"leaf_grid": 1280}, {"leaf_grid": 4096}, {"leaf_tab": 1, "leaf_grid": 1024}, {"leaf_tab": 4}):
        ctx = Context(0)
        for k_, v_ in opts.items():
            ctx.set_option(k_, v_)
        sp = ReadKMerSplitter(K, mode, ctx)
        sp.push_back_device(words.data_ptr(), words.numel() - 8, start.data_ptr(), ln.data_ptr(), n)
        c = KMerDiskCounter(None, sp)
        for _ in range(3):
            st = c.Count(nb)
        d = dict(ctx.timings())
        print(f"K={K} {opts} total={sum(d.values()):.1f} ms sort_unique={d['sort_unique']:.2f}")
        ctx.close()
