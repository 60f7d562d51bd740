#!/usr/bin/env python3
"""GPU box: build the 2 M-read golden case and leave block fingerprints of the GFA under gpurun_out/ (see gfa_fingerprint.py)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from spades_amd.gbuilder import GraphBuilder
g = json.load(open(os.path.join(ROOT, "tests/golden/scale_2000k_g10000k_s77.json")))
codes = synth.synth_codes(g["seed"], g["genome_len"], g["n_reads"], g["err"], g["n_rate"])
bases, off = synth.ascii_and_offsets(codes)
gb = GraphBuilder(g["k"], g["threads"])
gb.reads.push_back_ascii(bases.tobytes(), off)
print(gb.build())
gb.write_gfa("/tmp/g.gfa")
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools/gfa_fingerprint.py"), "/tmp/g.gfa", sys.argv[1]])
