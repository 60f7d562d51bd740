import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, synth
from spades_amd import KMerDiskCounter, ReadKMerSplitter
from spades_amd.kmercount import Context
codes = synth.synth_codes(77, 10_000_000, 2_000_000)
bases, off = synth.ascii_and_offsets(codes); bases = bases.tobytes()
res = []
for budget in (0, 2 << 30):
    ctx = Context(hbm_budget=budget)
    sp = ReadKMerSplitter(55, "A", ctx); sp.push_back_ascii(bases, off)
    st = KMerDiskCounter(None, sp).Count(16)
    res.append([st.bucket(b) for b in range(16)]); ctx.close()
for b in range(16):
    a, c = res[0][b], res[1][b]
    same = a.shape == c.shape and (a == c).all()
    if same: print("bucket", b, "identical", len(a)); continue
    d = np.nonzero((a != c).any(axis=1))[0]
    key = lambda x: (x[:, 0].astype(object) << 64) | x[:, 1].astype(object)
    srt = bool(np.all((c[1:, 0] > c[:-1, 0]) | ((c[1:, 0] == c[:-1, 0]) & (c[1:, 1] > c[:-1, 1]))))
    sa = set(map(tuple, a[:200000].tolist())) if False else None
    print("bucket", b, "differs at", len(d), "rows; first", d[:5], "strictly increasing:", srt, "same multiset:", bool((np.sort(a.view([('a','<u8'),('b','<u8')]).reshape(-1), order=['a','b']) == np.sort(c.view([('a','<u8'),('b','<u8')]).reshape(-1), order=['a','b'])).all()))
    i = int(d[0]); print("  ref", a[max(0,i-1):i+2].tolist(), "got", c[max(0,i-1):i+2].tolist())
