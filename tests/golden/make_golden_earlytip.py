#!/usr/bin/env python3
"""tests/golden/make_golden_earlytip.py — goldens of the reference's early tip clipper (spades-core Construction stage).

Runs oracle/_ref/ref_earlytip (the reference's DeBruijnExtensionIndex + EarlyTipClipperProcessor + UnbranchingPathExtractor compiled
in place by oracle/ref_recipe/Makefile; needs /root/reference) on the committed read sets and stores the edge sequences, one per
line in the extractor's order, as etc_<reads>_k<k>_t<t>_b<bound>.txt; appends "earlytip" cases to manifest.json."""
import hashlib
import json
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_earlytip")

mf = os.path.join(HERE, "manifest.json")
manifest = json.load(open(mf))
manifest["cases"] = [c for c in manifest["cases"] if c["kind"] != "earlytip"]
for name, K, T, bound in [("small", 21, 1, 129), ("small", 21, 1, 10), ("small", 21, 3, 129), ("small", 33, 1, 117), ("small", 55, 1, 95),
                          ("small", 55, 2, 95), ("mixed", 21, 1, 129), ("mixed", 33, 4, 117), ("tiny", 21, 1, 129), ("tiny", 5, 1, 20),
                          ("loop", 21, 1, 129), ("polyA", 21, 1, 129), ("small", 21, 1, 3)]:
    reads = os.path.join(HERE, f"reads_{name}.txt")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "o.txt")
        subprocess.check_call([REF, str(K), str(T), str(bound), reads, os.path.join(td, "w"), out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        data = open(out).read()
    fn = f"etc_{name}_k{K}_t{T}_b{bound}.txt"
    open(os.path.join(HERE, fn), "w").write(data)
    manifest["cases"].append({"kind": "earlytip", "reads": f"reads_{name}.txt", "K": K, "threads": T, "num_buckets": 10 * T, "bound": bound,
                              "n_edges": data.count("\n"), "md5": hashlib.md5(data.encode()).hexdigest(), "file": fn,
                              "source": "oracle/_ref/ref_earlytip (reference EarlyTipClipperProcessor + UnbranchingPathExtractor)"})
    print(fn, data.count("\n"))
json.dump(manifest, open(mf, "w"), indent=1)
