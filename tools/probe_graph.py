#!/usr/bin/env python3
"""GPU box: graph construction at scale — stage timings, and parity against the oracle on a sub-sample."""
import sys, os, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spades_amd.gbuilder import GraphBuilder
from spades_amd.reads import synth_batch_numpy, codes_to_ascii

n_pairs = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 55
threads = 16
words, start, ln, codes = synth_batch_numpy(3, max(100_000, n_pairs * 10), n_pairs, n_rate=0.001)
# parity on a sub-sample (oracle is scalar C)
sub = 20000
from oracle import oracle
reads = codes_to_ascii(codes[:sub])
t0 = time.time(); ref = oracle.build_graph(reads, k, 10 * threads); t_or = time.time() - t0
gb = GraphBuilder(k, threads)
gb.push_back_reads(reads)
gb.build()
gb.write_gfa("/tmp/sub.gfa")
same = open("/tmp/sub.gfa").read() == ref["gfa"]
print(f"sub-sample {sub} reads k={k}: unitigs={len(ref['unitigs'])} gfa_identical={same} (oracle {t_or:.1f}s)")
gb.ctx.close()
assert same
gb = GraphBuilder(k, threads)
gb.push_back_packed(words, start, ln)
for it in range(2):
    t0 = time.time(); info = gb.build(); t1 = time.time()
tm = gb.ctx.timings()
print(f"reads={2*n_pairs} k={k} kpomers={info['n_kpomers']} kmers={info['n_kmers']} unitigs={info['n_unitigs']} loops={info['n_loops']} vertices={info['n_vertices']} wall={t1-t0:.3f}s")
print("   " + " ".join(f"{n}={ms:.1f}" for n, ms in tm))
t0 = time.time(); gb.write_gfa("/tmp/full.gfa"); t2 = time.time() - t0
h = hashlib.md5()
with open("/tmp/full.gfa", "rb") as f:
    for blk in iter(lambda: f.read(1 << 24), b""):
        h.update(blk)
print(f"   gfa write {t2:.2f}s size={os.path.getsize('/tmp/full.gfa')/1e6:.1f} MB links={gb.info()['n_links']} md5={h.hexdigest()}")
