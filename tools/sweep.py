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
for K, mode, nb in ((55, "A", 16), (21, "A", 16)):
    for cap, target in ((2048, 1024), (1024, 1024), (1024, 512), (512, 512), (512, 256), (4096, 2048)):
        if K == 21: cap *= 2; target *= 2
        ctx = Context(0)
        ctx.set_option("leaf_cap", cap)
        ctx.set_option("leaf_target", target)
        sp = ReadKMerSplitter(K, mode, ctx)
        sp.push_back_device(words.data_ptr(), words.numel() - 8, start.data_ptr(), ln.data_ptr(), n)
        c = KMerDiskCounter(None, sp)
        for _ in range(2):
            st = c.Count(nb)
        tm = ctx.timings()
        tot = sum(ms for _, ms in tm)
        d = dict(tm)
        print(f"K={K} cap={cap} target={target} total={tot:.1f} ms  sort_wave={d['sort_wave']:.1f} sort_unique={d['sort_unique']:.1f} compact={d['compact']:.1f} levels={[k for k in d if k.endswith('scatter')]}")
        ctx.close()
