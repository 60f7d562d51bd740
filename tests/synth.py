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


def skewed_genome(rng, total_len, n_genomes=64, repeat_frac=0.3, lowc_frac=0.01):
    """SURVEY.md §8d config-4 shape: `n_genomes` iid genomes (concatenated; a read never spans two), 30 % of the sequence overwritten by
    copies of random 500..5000-bp segments of the same genome (repeats), 1 % by low-complexity runs (homopolymers, di- and
    trinucleotide repeats of 50..300 bp). Returns (genome uint8[total_len], genome starts int64[n_genomes + 1])."""
    g = rng.integers(0, 4, total_len, dtype=np.uint8)
    cuts = np.sort(rng.choice(np.arange(2000, total_len - 2000), n_genomes - 1, replace=False)) if n_genomes > 1 else np.zeros(0, dtype=np.int64)
    starts = np.concatenate([[0], cuts, [total_len]]).astype(np.int64)
    done = 0
    while done < repeat_frac * total_len:
        gi = int(rng.integers(0, n_genomes))
        a, b = int(starts[gi]), int(starts[gi + 1])
        ln = int(rng.integers(500, 5001))
        if b - a < 2 * ln + 10:
            continue
        src = int(rng.integers(a, b - ln))
        dst = int(rng.integers(a, b - ln))
        seg = g[src:src + ln].copy()
        if rng.random() < 0.5:
            seg = (3 - seg)[::-1]  # inverted repeat
        g[dst:dst + ln] = seg
        done += ln
    done = 0
    while done < lowc_frac * total_len:
        ln = int(rng.integers(50, 301))
        dst = int(rng.integers(0, total_len - ln))
        unit = rng.integers(0, 4, int(rng.integers(1, 4)), dtype=np.uint8)
        g[dst:dst + ln] = np.resize(unit, ln)
        done += ln
    return g, starts


def synth_codes_skewed(seed, total_len, n_reads, err=0.01, n_rate=0.001, n_genomes=64, sigma=1.5):
    """reads from skewed_genome with log-normal abundances over the genomes (sigma of the underlying normal); same read model as synth_codes"""
    assert n_reads % 2 == 0
    rng = np.random.default_rng(seed)
    genome, starts = skewed_genome(rng, total_len, n_genomes)
    size = np.diff(starts).astype(np.float64)
    ab = rng.lognormal(0.0, sigma, n_genomes) * size  # reads per genome ~ abundance x length
    ab /= ab.sum()
    n_pairs = n_reads // 2
    gi = rng.choice(n_genomes, n_pairs, p=ab)
    p = (starts[gi] + (rng.random(n_pairs) * np.maximum(size[gi] - INSERT, 1)).astype(np.int64)).astype(np.int64)
    p = np.minimum(p, starts[gi + 1] - INSERT)
    idx = np.arange(L)
    codes = np.empty((n_reads, L), dtype=np.uint8)
    CH = 1 << 18
    for c0 in range(0, n_pairs, CH):
        c1 = min(n_pairs, c0 + CH)
        q = p[c0:c1]
        codes[2 * c0:2 * c1:2] = genome[q[:, None] + idx[None, :]]
        codes[2 * c0 + 1:2 * c1:2] = 3 - genome[(q + INSERT - 1)[:, None] - idx[None, :]]
    for c0 in range(0, n_reads, CH):
        blk = codes[c0:c0 + CH]
        x = rng.random(blk.shape, dtype=np.float32)
        sub = rng.integers(1, 4, blk.shape, dtype=np.uint8)
        blk[:] = np.where(x < err, (blk + sub) % 4, blk)
        blk[(x >= err) & (x < err + n_rate)] = 4
    return codes


def synth_codes_plasmids(seed, total_len, n_reads, err=0.0, n_rate=0.0, plasmid_len=5000):
    """reads from total_len // plasmid_len CIRCULAR genomes (a read may run across the origin), error-free by default: every plasmid
    that the reads cover completely is a perfect loop of the de Bruijn graph (debruijn_graph_constructor.hpp:252-293) — with errors the
    tips and bubbles of the error k-mers would put junctions on every cycle. Same pair model as synth_codes."""
    assert n_reads % 2 == 0
    rng = np.random.default_rng(seed)
    n_pl = total_len // plasmid_len
    genome = rng.integers(0, 4, n_pl * plasmid_len, dtype=np.uint8).reshape(n_pl, plasmid_len)
    n_pairs = n_reads // 2
    codes = np.empty((n_reads, L), dtype=np.uint8)
    idx = np.arange(L)
    CH = 1 << 18
    for c0 in range(0, n_pairs, CH):
        c1 = min(n_pairs, c0 + CH)
        g = rng.integers(0, n_pl, c1 - c0)
        p = rng.integers(0, plasmid_len, c1 - c0)
        codes[2 * c0:2 * c1:2] = genome[g[:, None], (p[:, None] + idx[None, :]) % plasmid_len]
        codes[2 * c0 + 1:2 * c1:2] = 3 - genome[g[:, None], ((p + INSERT - 1)[:, None] - idx[None, :]) % plasmid_len]
    if err > 0 or n_rate > 0:
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
