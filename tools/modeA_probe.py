#!/usr/bin/env python3
"""GPU box: spades-kmercount's workload (all k-mers of read + reverse complement, 16 buckets) at BASELINE config-3 size, inputs
resident in HBM: time, stage split, and size-independent checks of the result (two strands: smx_pipeline.hpp two_strand_finish).
usage: modeA_probe.py [reads=100e6] [k=55] [genome=500e6]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth_reads_device, L
from spades_amd import KMerDiskCounter, ReadKMerSplitter
from spades_amd.kmercount import Context

n = int(float(sys.argv[1])) // 32 * 32 if len(sys.argv) > 1 else 100_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 55
G = int(float(sys.argv[3])) if len(sys.argv) > 3 else 500_000_000
dev = torch.device("cuda", 0)
words, start, ln, codes = synth_reads_device(1000, G, n, dev, n_rate=0.001)
del codes
torch.cuda.synchronize()
torch.cuda.empty_cache()  # (the generator's temporaries go back to the device: the library's arena sizes itself from what is free)
print('free HBM after the reads:', torch.cuda.mem_get_info(dev), file=sys.stderr)
ctx = Context(0)
ctx.set_option("single_batch", 1)  # never batches / host spill here: a probe must not exhaust the box
res = {"reads": n, "k": k}

def count(mode, nb):
    sp = ReadKMerSplitter(k, mode, ctx)
    sp.clear()
    sp.push_back_device(words.data_ptr(), words.numel() - 8, start.data_ptr(), ln.data_ptr(), n)
    c = KMerDiskCounter(None, sp)
    st = c.Count(nb)  # (first call: the arena maps its memory)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        st = c.Count(nb)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    stages = {}
    for name, ms in ctx.timings():
        stages[name] = stages.get(name, 0.0) + ms
    return st, sorted(ts)[1], stages

stB, tB, _ = count("B", 16)
D = stB.total_kmers()
res["canonical_kmers"] = int(D)
stA, tA, stages = count("A", 16)
nA, inst = stA.total_kmers(), stA.kmer_instances()
W = 8 * ((k + 31) // 32)
b_alg = n * L / 4 + 2 * inst * W + nA * W
kernel_ms = sum(stages.values())
res.update({"mode_A_records": int(nA), "expected_2x_canonical": int(2 * D), "kmer_instances": int(inst), "seconds_per_count": round(tA, 4),
            "M_reads_per_s": round(n / tA / 1e6, 2), "kernel_ms": round(kernel_ms, 1), "roofline_frac": round(b_alg / kernel_ms / 1e6 / 8000.0, 4),
            "held_as_two_strands": stA.device_ptr() == 0, "stages_ms": {a: round(b, 1) for a, b in stages.items() if b >= 1.0}})
ok = nA == 2 * D or k % 2 == 0
# one bucket on the host: strictly increasing as (w0, w1, ...) tuples, and every record files under this bucket (XXH3 on a sample)
sizes = stA.bucket_sizes()
b = 3
rec = stA.bucket(b)
key = rec.copy()
lt = np.zeros(len(rec) - 1, dtype=bool); eq = np.ones(len(rec) - 1, dtype=bool)
for w in range(rec.shape[1]):
    lt |= eq & (key[:-1, w] < key[1:, w]); eq &= key[:-1, w] == key[1:, w]
res["bucket3_records"] = int(len(rec)); res["bucket3_strictly_increasing"] = bool(lt.all())
import xxhash
idx = np.random.default_rng(1).integers(0, len(rec), 20000)
good = sum(((xxhash.xxh3_64_intdigest(rec[i].tobytes()) * 16) >> 64) == b for i in idx)
res["bucket3_sample_files_under_bucket3"] = bool(good == len(idx))
ok = ok and res["bucket3_strictly_increasing"] and res["bucket3_sample_files_under_bucket3"] and int(sizes.sum()) == nA
res["ok"] = bool(ok)
print(json.dumps(res))
ctx.close()
sys.exit(0 if ok else 1)
