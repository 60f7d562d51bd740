"""Synthetic read batches (SURVEY.md §8d generator) and 2-bit packing helpers.

Layout produced everywhere: one 2-bit stream (nucleotide g at bits 2*(g mod 32) of word g/32, A0 C1 G2 T3),
read i = nucleotides [start[i], start[i]+len[i]). Reads are stored at full length; when a read
contains N the (start, len) pair selects its longest valid run (first on ties) exactly like
io::LongestValid (common/io/reads/longest_valid_wrapper.hpp:16-53); N positions carry code 0.
"""
from typing import Tuple

import numpy as np


def pack_codes(codes: np.ndarray) -> np.ndarray:
    """uint8 codes (0..3), any length -> uint64 words."""
    n = len(codes)
    pad = (-n) % 32
    c = np.concatenate([codes.astype(np.uint64), np.zeros(pad, dtype=np.uint64)]).reshape(-1, 32)
    shifts = (2 * np.arange(32, dtype=np.uint64))
    return (c << shifts).sum(axis=1, dtype=np.uint64)


def synth_batch_numpy(seed: int, genome_len: int, n_pairs: int, L: int = 150, insert: int = 350, err: float = 0.01,
                      n_rate: float = 0.0) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """-> (words, start, len, codes[n, L] with 4 = N). Paired-end: read1 = genome[p:p+L], read2 = RC(genome[p+ins-L:p+ins])."""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, genome_len, dtype=np.uint8)
    p = rng.integers(0, genome_len - insert + 1, n_pairs)
    idx = np.arange(L)
    r1 = g[p[:, None] + idx[None, :]]
    r2 = 3 - g[(p + insert - 1)[:, None] - idx[None, :]]
    codes = np.empty((2 * n_pairs, L), dtype=np.uint8)
    codes[0::2] = r1
    codes[1::2] = r2
    e = rng.random(codes.shape) < err
    codes[e] = (codes[e] + rng.integers(1, 4, int(e.sum()), dtype=np.uint8)) % 4
    if n_rate > 0:
        codes[rng.random(codes.shape) < n_rate] = 4
    n = codes.shape[0]
    valid = codes < 4
    # longest valid run per read, first on ties
    pos = np.arange(L)[None, :]
    last_bad = np.maximum.accumulate(np.where(valid, -1, pos), axis=1)
    run = np.where(valid, pos - last_bad, 0)
    end = run.argmax(axis=1)  # first maximal value
    ln = run[np.arange(n), end]
    st = end - ln + 1
    st[ln == 0] = 0
    start = (np.arange(n, dtype=np.uint64) * L + st.astype(np.uint64)).astype(np.uint64)
    words = pack_codes(np.where(valid, codes, 0).reshape(-1))
    return words, start, ln.astype(np.uint32), codes


def codes_to_ascii(codes: np.ndarray):
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    return [lut[r].tobytes().decode() for r in codes]
