import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spades_amd import KMerDiskCounter, ReadKMerSplitter
from spades_amd.reads import synth_batch_numpy
n_pairs = 1_000_000
words, start, ln, _ = synth_batch_numpy(1, 10_000_000, n_pairs)
for K, mode, nb in ((55, "A", 16),):
    for dbg in (0, 1, 2, 3):
        sp = ReadKMerSplitter(K, mode)
        sp.ctx.set_option("dbg", dbg)
        sp.push_back_packed(words, start, ln)
        c = KMerDiskCounter(None, sp)
        for it in range(2):
            st = c.Count(nb)
        tm = dict(sp.ctx.timings())
        print(f"dbg={dbg} sort_unique={tm['sort_unique']:.2f} ms")
        sp.ctx.close()
