"""smx_trim at size: build -> trim -> build again on one context must give the same graph (tools, not a test: needs a few GB of HBM).
usage: python tools/trim_probe.py [n_reads=2e6] [genome=10e6] [k=55] [T=16]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from spades_amd.gbuilder import GraphBuilder  # noqa: E402

pos = sys.argv[1:]
n_reads = int(float(pos[0])) // 32 * 32 if pos else 2_000_000
genome = int(float(pos[1])) if len(pos) > 1 else 10_000_000
k = int(pos[2]) if len(pos) > 2 else 55
T = int(pos[3]) if len(pos) > 3 else 16
dev = torch.device("cuda", 0)
words, start, ln, codes = bench.synth_reads_device(1, genome, n_reads, dev, n_rate=0.001)
del codes
torch.cuda.synchronize()
gb = GraphBuilder(k, T)
gb.ctx.set_option("device_links", 2)
gb.push_back_device(words.data_ptr(), n_reads * bench.L // 32, start.data_ptr(), ln.data_ptr(), n_reads)
ok = True
fp0 = None
for it in range(4):
    info = gb.build()
    fp = gb.fingerprint_portable()
    if fp0 is None:
        fp0 = fp
    same = fp == fp0
    ok = ok and same
    print(f"build {it}: {info['n_kmers']} k-mers, {info['n_unitigs']} unitigs, same graph as build 0: {same}", flush=True)
    if it % 2 == 0:
        gb.ctx.graph_clear()
    print(f"  trim ({'after graph_clear' if it % 2 == 0 else 'graph resident'}): {gb.ctx.trim() >> 20} MiB back to the device", flush=True)
gb.ctx.close()
sys.exit(0 if ok else 1)
