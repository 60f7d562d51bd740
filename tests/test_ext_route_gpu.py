"""GPU parity of the construction route that takes k-mers AND extension masks from one count of the reads (option "ext_route";
records carry the InOutMask byte through the pre-dedupe stage and the sort): the real spades-gbuilder goldens, the oracle, and the
(k+1)-mer-file route of the same library must all agree — graph text, k-mer file, mask-derived (k+1)-mer count, coverage."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_manifest, read_lines

pytestmark = pytest.mark.gpu

# small inputs: the pre-dedupe stage (where the bytes are gathered) forced on; pm_route = 0: the k-mers ARE sorted into the file here
# (tests/test_pm_route_gpu.py covers the route that does not)
FORCE = {"prededupe": 1, "ext_route": 1, "pm_route": 0}
LEGACY = {"prededupe": 1, "ext_route": 0}


def _eligible(k):  # 8 spare bits in the last record word, and the pre-dedupe stage (where the bytes are gathered) applies
    nw = (k + 31) // 32
    return k >= 21 and 2 * k + 8 <= 64 * nw


def _build(reads, k, threads, tmp_path, opts, coverage=False):
    from spades_amd.gbuilder import GraphBuilder
    gb = GraphBuilder(k, threads)
    for key, v in opts.items():
        gb.ctx.set_option(key, v)
    gb.push_back_reads(reads)
    gb.build()
    took = any(n == "kmers:ext_merge" for n, _ in gb.ctx.timings())
    fp = gb.fingerprint()  # checksums of the device-resident graph arrays (what bench.py compares between the routes at full size)
    if coverage:
        gb.fill_coverage()
    out = os.path.join(str(tmp_path), "g.gfa")
    gb.write_gfa(out)
    info = dict(gb.info())
    res = dict(info=info, gfa=open(out).read(), unitigs=gb.unitigs(), kmers=gb.kmers(), took_ext_route=took, fp=fp)
    gb.ctx.close()
    return res


def _same_kmers(a, b):  # (k-mer file, InOutMask bytes)
    return all(np.array_equal(x, y) for x, y in zip(a["kmers"], b["kmers"]))


GCASES = [c for c in load_manifest()["cases"] if c["kind"] == "graph" and _eligible(c["K"])]


@pytest.mark.parametrize("case", GCASES, ids=lambda c: f"{c['reads'][6:-4]}-k{c['K']}-t{c['threads']}")
def test_gfa_matches_spades_gbuilder(case, tmp_path):
    reads = [r for r in read_lines(case["reads"]) if r]
    r = _build(reads, case["K"], case["threads"], tmp_path, FORCE)
    assert r["took_ext_route"]
    assert hashlib.md5(r["gfa"].encode()).hexdigest() == case["md5"]
    old = _build(reads, case["K"], case["threads"], tmp_path, LEGACY)
    assert not old["took_ext_route"]
    assert _same_kmers(r, old)
    assert r["fp"] == old["fp"] and any(r["fp"])
    assert r["info"] == old["info"]  # incl. the number of canonical (k+1)-mers, here derived from the mask bits


CCASES = [c for c in load_manifest()["cases"] if c["kind"] == "graph_cov" and _eligible(c["K"])]


@pytest.mark.parametrize("case", CCASES, ids=lambda c: f"{c['reads'][6:-4]}-k{c['K']}-t{c['threads']}")
def test_gfa_with_coverage(case, tmp_path):
    """-c behind this route: there is no (k+1)-mer file, the coverage pass counts it (and checks its size against the mask bits)"""
    reads = [r for r in read_lines(case["reads"]) if r]
    r = _build(reads, case["K"], case["threads"], tmp_path, FORCE, coverage=True)
    assert r["took_ext_route"]
    assert r["gfa"] == open(os.path.join(GOLDEN, case["file"])).read()


def _synth(seed, glen, n, L, err=0.01, nrate=0.002):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, glen)
    reads = []
    for _ in range(n):
        ln = int(L if rng.random() < 0.8 else rng.integers(1, L + 1))
        p = int(rng.integers(0, len(g) - ln + 1))
        r = g[p:p + ln].copy()
        if rng.random() < 0.5:
            r = (3 - r)[::-1]
        e = rng.random(ln) < err
        r[e] = (r[e] + rng.integers(1, 4, int(e.sum()))) % 4
        s = np.array(list("ACGT"))[r]
        s[rng.random(ln) < nrate] = "N"
        reads.append("".join(s))
    return reads


@pytest.mark.parametrize("k", [21, 25, 27, 33, 41, 55, 59, 77, 91, 123])
def test_vs_oracle_seeded(k, tmp_path):
    """ragged reads (1 .. 150 bases, some exactly k and k+1 long), both strands, N, rc-palindromic (k+1)-mers (ACGT / AT repeats),
    homopolymers; several bucket counts"""
    from oracle import oracle
    assert _eligible(k)
    rng = np.random.default_rng(k)
    pal = "".join("ACGT"[i] for i in rng.integers(0, 4, (k + 1) // 2))
    pal = pal + "".join("TGCA"["ACGT".index(c)] for c in reversed(pal))  # its own reverse complement, k+1 long
    assert len(pal) == k + 1
    reads = (_synth(3 * k, 4000, 1500, 150) + ["ACGT" * 40] * 3 + ["AT" * 70] * 2 + ["A" * 140] * 4 + [pal, "G" + pal + "T", pal[:k], pal[1:]]
             + ["C" * k, "C" * (k + 1), "N" * 150, ""])
    for threads in (1, 3):
        ref = oracle.build_graph(reads, k, 10 * threads)
        r = _build(reads, k, threads, tmp_path, FORCE)
        assert r["took_ext_route"]
        assert r["unitigs"] == ref["unitigs"] and r["gfa"] == ref["gfa"]
        old = _build(reads, k, threads, tmp_path, LEGACY)
        assert _same_kmers(r, old) and r["info"] == old["info"] and r["fp"] == old["fp"]


def test_coverage_vs_oracle_seeded(tmp_path):
    from oracle import oracle
    for k, threads in ((21, 2), (55, 1), (77, 2)):
        reads = _synth(5 + k, 6000, 1500, 150) + ["ACGT" * 40] * 3 + ["A" * 100] * 5
        ref = oracle.build_graph(reads, k, 10 * threads, coverage=True)
        r = _build(reads, k, threads, tmp_path, FORCE, coverage=True)
        assert r["gfa"] == ref["gfa"]


def test_cut_partitions_are_merged(tmp_path):
    """a tiny chunk capacity cuts every minimizer partition: copies of a k-mer leave the stage with partial bytes and are ORed
    together after the sort"""
    from oracle import oracle
    k = 55
    reads = _synth(11, 3000, 4000, 150, err=0.002)  # ~200x: partitions far larger than one chunk
    ref = oracle.build_graph(reads, k, 20)
    r = _build(reads, k, 2, tmp_path, dict(FORCE, skm_cap=512))
    assert r["took_ext_route"] and r["gfa"] == ref["gfa"]
    # ... or, when they were not merged before it (the general merge: heads, scan, OR)
    r2 = _build(reads, k, 2, tmp_path, dict(FORCE, skm_cap=512, ext_presort=0))
    assert r2["took_ext_route"] and r2["gfa"] == ref["gfa"] and _same_kmers(r, r2) and r["info"] == r2["info"]


def test_route_is_declined_where_it_does_not_fit(tmp_path):
    """k = 29, 31 / 61, 63 / 127 leave fewer than 8 spare bits in the last record word, k < 21 has no pre-dedupe stage: same results by
    the other route"""
    from oracle import oracle
    for k in (15, 29, 31, 63):
        reads = _synth(k, 3000, 800, 150)
        ref = oracle.build_graph(reads, k, 10)
        r = _build(reads, k, 1, tmp_path, FORCE)
        assert not r["took_ext_route"] and r["gfa"] == ref["gfa"]


def test_several_read_chunks(tmp_path):
    """reads submitted in several calls are separate device chunks: the neighbour bases of a k-mer never come from another chunk"""
    from oracle import oracle
    from spades_amd.gbuilder import GraphBuilder
    k = 33
    reads = _synth(5, 3000, 1200, 150)
    ref = oracle.build_graph(reads, k, 20)
    gb = GraphBuilder(k, 2)
    for key, v in FORCE.items():
        gb.ctx.set_option(key, v)
    for a in range(0, len(reads), 250):
        gb.push_back_reads(reads[a:a + 250])
    gb.build()
    assert any(n == "kmers:ext_merge" for n, _ in gb.ctx.timings())
    out = os.path.join(str(tmp_path), "g.gfa")
    gb.write_gfa(out)
    assert gb.unitigs() == ref["unitigs"] and open(out).read() == ref["gfa"]
    gb.ctx.close()
