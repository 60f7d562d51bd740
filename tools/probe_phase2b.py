import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import bench
from spades_amd import KMerDiskCounter, ReadKMerSplitter
from spades_amd.kmercount import Context
n_reads = 2_000_000; K = 55
dev = torch.device("cuda", 0)
words, start, ln, codes = bench.synth_reads_device(1000, 50_000_000, n_reads, dev)
ctx = Context(0)
sp = ReadKMerSplitter(K, "A", ctx)
sp.push_back_device(words.data_ptr(), words.numel() - 8, start.data_ptr(), ln.data_ptr(), n_reads)
st = KMerDiskCounter(None, sp).Count(16)
D = st.total_kmers(); nw = 2
base = torch.empty(D * nw, dtype=torch.int64, device=dev)
assert ctx.lib.smx_copy_kmers_device(ctx._h, base.data_ptr()) == 0
def small(t): return int(((t.view(D, nw)[:, 0] >= 0) & (t.view(D, nw)[:, 0] < (1 << 49))).sum())
print("D", D, "small-x rows in sorted file", small(base))
for name in ("sorted", "perm", "flip"):
    if name == "sorted": t = base.clone()
    elif name == "perm":
        perm = torch.randperm(D, device=dev); print("perm unique", int(torch.unique(perm).numel()), "max", int(perm.max()))
        t = base.view(D, nw)[perm].contiguous().view(-1)
    else: t = base.view(D, nw).flip(0).contiguous().view(-1)
    print(name, "small-x rows", small(t))
    sys.stderr.write(f"== {name}\n"); sys.stderr.flush()
    assert ctx.lib.smx_count_records(ctx._h, K, 16, t.data_ptr(), D) == 0
    tm = dict(ctx.timings()); print(name, "sort_big", tm["sort_big"])
