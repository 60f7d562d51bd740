#!/usr/bin/env python3
"""GPU box: wall-clock of the count and construction stages at BASELINE-config-3 scale (SMX_DEBUG=1 prints the host sections).
usage: scale_probe.py <reads> <genome> [count|graph|both]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_reads_device
from spades_amd.kmercount import Context, ReadKMerSplitter, KMerDiskCounter
from spades_amd.gbuilder import GraphBuilder

n = int(float(sys.argv[1])) // 32 * 32
genome = int(float(sys.argv[2]))
what = sys.argv[3] if len(sys.argv) > 3 else "both"
dev = torch.device("cuda", 0)
t0 = time.time()
words, start, ln, codes = synth_reads_device(1000, genome, n, dev)
del codes
torch.cuda.synchronize()
print(f"generated {n} reads in {time.time() - t0:.1f} s; free HBM {torch.cuda.mem_get_info()[0] / 1e9:.1f} GB", flush=True)
torch.cuda.empty_cache()
ctx = Context(0)
if what in ("count", "both"):
    sp = ReadKMerSplitter(56, "B", ctx)
    sp.push_back_device(words.data_ptr(), words.numel() - 8, start.data_ptr(), ln.data_ptr(), n)
    for it in range(2):
        t0 = time.time()
        st = KMerDiskCounter(None, sp).Count(160)
        dt = time.time() - t0
        print(f"count B k+1=56: {dt:.3f} s, instances {st.kmer_instances()}, distinct {st.total_kmers()} -> {n / dt / 1e6:.1f} M reads/s", flush=True)
        tm = {}
        for nm, ms in ctx.timings():
            tm[nm] = tm.get(nm, 0) + ms
        print("  stages ms:", {k: round(v, 1) for k, v in tm.items()}, "sum", round(sum(tm.values()), 1), flush=True)
    sp.clear()
if what in ("graph", "both"):
    gb = GraphBuilder(55, 16, ctx)
    gb.push_back_device(words.data_ptr(), words.numel() - 8, start.data_ptr(), ln.data_ptr(), n)
    for it in range(2):
        t0 = time.time()
        info = gb.build()
        dt = time.time() - t0
        print(f"build k=55: {dt:.3f} s {info} -> {n / dt / 1e6:.1f} M reads/s", flush=True)
        tm = {}
        for nm, ms in ctx.timings():
            tm[nm] = tm.get(nm, 0) + ms
        print("  stages ms:", {k: round(v, 1) for k, v in tm.items()}, "sum", round(sum(tm.values()), 1), flush=True)
    t0 = time.time()
    gb.fill_coverage()
    print(f"coverage: {time.time() - t0:.3f} s", flush=True)
ctx.close()
