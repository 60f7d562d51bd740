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

def make_reads_at():
    """reads_small + reads with poly-A / low-complexity tails and heads: A/T tips and length-1 A/T edges for the early A/T remover"""
    import random
    rnd = random.Random(11)
    base = [l.strip() for l in open(os.path.join(HERE, "reads_small.txt")) if l.strip()]
    extra = []
    for _ in range(60):
        r = rnd.choice(base)
        cut = rnd.randint(40, 110)
        tail = rnd.choice(["A", "T", "AT", "AAAT", "TTTTA"]) * 40
        extra.append((r[:cut] + tail)[:150])
        extra.append(("A" * rnd.randint(15, 40) + r[rnd.randint(0, 50):])[:150])
    for _ in range(10):
        extra.append("A" * 30 + rnd.choice("CGT") + "A" * 25 + "".join(rnd.choice("ACGT") for _ in range(60)))
    open(os.path.join(HERE, "reads_at.txt"), "w").write("\n".join(base + extra) + "\n")


make_reads_at()
mf = os.path.join(HERE, "manifest.json")
manifest = json.load(open(mf))
manifest["cases"] = [c for c in manifest["cases"] if c["kind"] != "earlytip"]
for name, K, T, bound, at in [("small", 21, 1, 129, 0), ("small", 21, 1, 10, 0), ("small", 21, 3, 129, 0), ("small", 33, 1, 117, 0),
                              ("small", 55, 1, 95, 0), ("small", 55, 2, 95, 0), ("mixed", 21, 1, 129, 0), ("mixed", 33, 4, 117, 0),
                              ("tiny", 21, 1, 129, 0), ("tiny", 5, 1, 20, 0), ("loop", 21, 1, 129, 0), ("polyA", 21, 1, 129, 0),
                              ("small", 21, 1, 3, 0),
                              # early A/T remover (RNA pipelines), alone and followed by the tip clipper
                              ("at", 21, 1, 0, 1), ("at", 33, 1, 0, 1), ("at", 21, 3, 129, 1), ("at", 55, 1, 95, 1), ("at", 33, 2, 0, 1),
                              ("small", 21, 1, 0, 1), ("polyA", 21, 1, 0, 1), ("at", 21, 1, 129, 0)]:
    reads = os.path.join(HERE, f"reads_{name}.txt")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "o.txt")
        subprocess.check_call([REF, str(K), str(T), str(bound), reads, os.path.join(td, "w"), out] + (["at"] if at else []),
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        data = open(out).read()
    fn = f"etc_{name}_k{K}_t{T}_b{bound}{'_at' if at else ''}.txt"
    open(os.path.join(HERE, fn), "w").write(data)
    manifest["cases"].append({"kind": "earlytip", "reads": f"reads_{name}.txt", "K": K, "threads": T, "num_buckets": 10 * T, "bound": bound, "at": at,
                              "n_edges": data.count("\n"), "md5": hashlib.md5(data.encode()).hexdigest(), "file": fn,
                              "source": "oracle/_ref/ref_earlytip (reference EarlyTipClipperProcessor + UnbranchingPathExtractor)"})
    print(fn, data.count("\n"))
# spades-core edge order: unitigs sorted with the reference's own Sequence::RawCompare (DeBruijnGraphExtentionConstructor, :590-604)
manifest["cases"] = [c for c in manifest["cases"] if c["kind"] != "sorted_edges"]
for name, K, T, loops, at, bound in [("small", 21, 1, 1, 0, 0), ("mixed", 33, 2, 1, 0, 0), ("mixed", 21, 1, 0, 0, 0), ("at", 55, 1, 1, 1, 95),
                                     ("loop", 21, 1, 1, 0, 0), ("loop", 21, 1, 0, 0, 0), ("small", 55, 3, 1, 0, 95)]:
    reads = os.path.join(HERE, f"reads_{name}.txt")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "o.txt")
        subprocess.check_call([REF, str(K), str(T), str(bound), reads, os.path.join(td, "w"), out, "sorted"] + (["at"] if at else []) +
                              ([] if loops else ["noloops"]), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        data = open(out).read()
    fn = f"sorted_{name}_k{K}_t{T}_l{loops}_b{bound}{'_at' if at else ''}.txt"
    open(os.path.join(HERE, fn), "w").write(data)
    manifest["cases"].append({"kind": "sorted_edges", "reads": f"reads_{name}.txt", "K": K, "threads": T, "num_buckets": 10 * T, "keep_loops": loops,
                              "at": at, "bound": bound, "n_edges": data.count("\n"), "md5": hashlib.md5(data.encode()).hexdigest(), "file": fn,
                              "source": "oracle/_ref/ref_earlytip sorted (reference extractor + Sequence::RawCompare)"})
    print(fn, data.count("\n"))
json.dump(manifest, open(mf, "w"), indent=1)
