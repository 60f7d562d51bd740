#!/usr/bin/env python3
"""Block fingerprints of a GFA (S-line lengths, md5 per block of lines) to localise a difference between two big files that cannot
be brought to the same machine. usage: gfa_fingerprint.py <gfa> <out.npz>"""
import hashlib, sys
import numpy as np
BLK = 4096
lens, s_h, l_h = [], [], []
hs, hl, ns, nl = hashlib.md5(), hashlib.md5(), 0, 0
with open(sys.argv[1], "rb") as f:
    for line in f:
        if line[:1] == b"S":
            lens.append(len(line.split(b"\t")[2]))
            hs.update(line); ns += 1
            if ns % BLK == 0: s_h.append(hs.digest()); hs = hashlib.md5()
        elif line[:1] == b"L":
            hl.update(line); nl += 1
            if nl % BLK == 0: l_h.append(hl.digest()); hl = hashlib.md5()
s_h.append(hs.digest()); l_h.append(hl.digest())
np.savez_compressed(sys.argv[2], lens=np.array(lens, dtype=np.uint32), s=np.frombuffer(b"".join(s_h), dtype=np.uint8), l=np.frombuffer(b"".join(l_h), dtype=np.uint8), n=np.array([ns, nl]))
print("S", ns, "L", nl, "sum len", int(np.sum(lens, dtype=np.int64)))
