#!/usr/bin/env python3
"""What would the sort/place half cost on already-distinct records? (sizing experiment for a pre-dedupe stage)
Counts the bench batch, then re-counts its distinct output through smx_count_records and prints both stage tables."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import bench
from spades_amd import KMerDiskCounter, ReadKMerSplitter
from spades_amd.kmercount import Context

n_reads = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 55
dev = torch.device("cuda", 0)
words, start, ln, codes = bench.synth_reads_device(1000, 50_000_000, n_reads, dev)
del codes
ctx = Context(0)
sp = ReadKMerSplitter(K, "A", ctx)
sp.push_back_device(words.data_ptr(), words.numel() - 8, start.data_ptr(), ln.data_ptr(), n_reads)
c = KMerDiskCounter(None, sp)
for _ in range(2):
    st = c.Count(16)
tm = ctx.timings()
print("full :", round(sum(m for _, m in tm), 2), " ".join(f"{n}={m:.2f}" for n, m in tm))
D = st.total_kmers(); nw = (K + 31) // 32
t = torch.empty(D * nw, dtype=torch.int64, device=dev)
assert ctx.lib.smx_copy_kmers_device(ctx._h, t.data_ptr()) == 0
# shuffle so that the input is not already in order
perm = torch.randperm(D, device=dev)
t = t.view(D, nw)[perm].contiguous().view(-1)
del perm
for _ in range(2):
    assert ctx.lib.smx_count_records(ctx._h, K, 16, t.data_ptr(), D) == 0
tm = ctx.timings()
print(f"distinct-only ({D} records of {st.kmer_instances()} instances):", round(sum(m for _, m in tm), 2), " ".join(f"{n}={m:.2f}" for n, m in tm))
n2 = C.c_uint64(); ctx.lib.smx_count_info(ctx._h, C.byref(n2), None, None)
print("recount distinct", n2.value, "zero rows in input", int((t.view(D, nw) == 0).all(dim=1).sum()))
