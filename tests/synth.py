"""Seeded synthetic read sets shared by the golden generator (CPU box, real reference binaries) and the GPU tests that
check the product against the md5s that generator committed. SURVEY.md §8d generator: iid genome, PE150 pairs with a fixed
insert, read 2 = reverse complement of the far end, 1 % substitutions, 0.1 % N (exercises the longest-ACGT-run rule).
numpy's PCG64 stream is stable across numpy versions, so both boxes see the same reads."""
import numpy as np

L = 150
INSERT = 350


def synth_codes(seed, genome_len, n_reads, err=0.01, n_rate=0.001):
    """uint8 [n_reads, L], values 0..3 = ACGT, 4 = N."""
    assert n_reads % 2 == 0
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, genome_len, dtype=np.uint8)
    n_pairs = n_reads // 2
    codes = np.empty((n_reads, L), dtype=np.uint8)
    idx = np.arange(L)
    CH = 1 << 18
    for c0 in range(0, n_pairs, CH):
        c1 = min(n_pairs, c0 + CH)
        p = rng.integers(0, genome_len - INSERT + 1, c1 - c0)
        codes[2 * c0:2 * c1:2] = genome[p[:, None] + idx[None, :]]
        codes[2 * c0 + 1:2 * c1:2] = 3 - genome[(p + INSERT - 1)[:, None] - idx[None, :]]
    for c0 in range(0, n_reads, CH):
        blk = codes[c0:c0 + CH]
        x = rng.random(blk.shape, dtype=np.float32)
        sub = rng.integers(1, 4, blk.shape, dtype=np.uint8)
        blk[:] = np.where(x < err, (blk + sub) % 4, blk)
        blk[(x >= err) & (x < err + n_rate)] = 4
    return codes


def write_fastq(codes, path):
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    n = codes.shape[0]
    qual = b"I" * L
    with open(path, "wb") as f:
        CH = 1 << 16
        for c0 in range(0, n, CH):
            seqs = lut[codes[c0:c0 + CH]]
            f.write(b"".join(b"@r%d\n%s\n+\n%s\n" % (c0 + i, seqs[i].tobytes(), qual) for i in range(seqs.shape[0])))


def ascii_and_offsets(codes):
    """(bytes, uint64 offsets[n+1]) for smx_submit_reads_ascii"""
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    n = codes.shape[0]
    return lut[codes].reshape(-1), (np.arange(n + 1, dtype=np.uint64) * L)
