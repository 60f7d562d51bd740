#!/usr/bin/env python3
"""GPU box: many small builds in one process, to catch intermittent faults. usage: stress_ctx.py <variant> <n>
variants: A force ext route, build + close; B force ext route, build + every accessor the tests use; C (k+1)-mer route with the
pre-dedupe stage forced, build + close; D defaults, build + close; E like B without timings(); F like A plus timings() only"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from spades_amd.gbuilder import GraphBuilder

variant, n = sys.argv[1], int(sys.argv[2])
rng = np.random.default_rng(3)
g = rng.integers(0, 4, 4000)
reads = []
for _ in range(1500):
    p = int(rng.integers(0, 4000 - 150))
    r = g[p:p + 150].copy()
    e = rng.random(150) < 0.01
    r[e] = (r[e] + 1) % 4
    reads.append("".join(np.array(list("ACGT"))[r]))
opts = {"A": {"prededupe": 1, "ext_route": 1}, "B": {"prededupe": 1, "ext_route": 1}, "C": {"prededupe": 1, "ext_route": 0}, "D": {},
        "E": {"prededupe": 1, "ext_route": 1}, "F": {"prededupe": 1, "ext_route": 1}}[variant]
td = tempfile.mkdtemp()
for i in range(n):
    k = (21, 33, 55)[i % 3]
    gb = GraphBuilder(k, 1 + i % 3)
    for key, v in opts.items():
        gb.ctx.set_option(key, v)
    gb.push_back_reads(reads)
    gb.build()
    if variant in ("B", "F"):
        gb.ctx.timings()
    if variant in ("B", "E"):
        gb.fingerprint()
        if i % 2:
            gb.fill_coverage()
        gb.write_gfa(os.path.join(td, "g.gfa"))
        gb.info()
        gb.unitigs()
        gb.kmers()
    gb.ctx.close()
    if i % 100 == 99:
        print(variant, i + 1, flush=True)
print(variant, "done", flush=True)
