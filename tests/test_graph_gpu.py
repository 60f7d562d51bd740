"""GPU parity: de Bruijn construction through the C ABI vs the real spades-gbuilder goldens and the oracle.
Bit-exact: canonical k-mer file, InOutMask bytes, unitig list in the reference's order, GFA text."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_manifest, read_lines

pytestmark = pytest.mark.gpu

GCASES = [c for c in load_manifest()["cases"] if c["kind"] == "graph"]


def _build(reads, k, threads, tmp_path, opts=None):
    from spades_amd.gbuilder import GraphBuilder
    gb = GraphBuilder(k, threads)
    for key, v in (opts or {}).items():
        gb.ctx.set_option(key, v)
    gb.push_back_reads(reads)
    info = gb.build()
    out = os.path.join(str(tmp_path), "g.gfa")
    gb.write_gfa(out)
    res = dict(info=gb.info(), gfa=open(out).read(), unitigs=gb.unitigs(), kmers=gb.kmers())
    gb.ctx.close()
    return res


@pytest.mark.parametrize("case", GCASES, ids=lambda c: f"{c['reads'][6:-4]}-k{c['K']}-t{c['threads']}")
def test_gfa_matches_spades_gbuilder(case, tmp_path):
    reads = [r for r in read_lines(case["reads"]) if r]
    r = _build(reads, case["K"], case["threads"], tmp_path)
    assert r["gfa"].count("\nS\t") == case["n_segments"] and r["gfa"].count("\nL\t") == case["n_links"]
    assert hashlib.md5(r["gfa"].encode()).hexdigest() == case["md5"]
    if case["file"]:
        assert r["gfa"] == open(os.path.join(GOLDEN, case["file"])).read()


CCASES = [c for c in load_manifest()["cases"] if c["kind"] == "graph_cov"]


@pytest.mark.parametrize("case", CCASES, ids=lambda c: f"{c['reads'][6:-4]}-k{c['K']}-t{c['threads']}")
def test_gfa_with_coverage_matches_spades_gbuilder_c(case, tmp_path):
    from spades_amd.gbuilder import GraphBuilder
    reads = [r for r in read_lines(case["reads"]) if r]
    gb = GraphBuilder(case["K"], case["threads"])
    gb.push_back_reads(reads)
    gb.build()
    gb.fill_coverage()
    out = os.path.join(str(tmp_path), "g.gfa")
    gb.write_gfa(out)
    assert open(out).read() == open(os.path.join(GOLDEN, case["file"])).read()
    gb.ctx.close()


SCASES = [c for c in load_manifest()["cases"] if c["kind"] == "graph_spades"]


@pytest.mark.parametrize("case", SCASES, ids=lambda c: c["base"])
def test_spades_internal_format_matches_gbuilder(case, tmp_path):
    """--spades: .grseq + .cvr byte-identical to io::binary::BasicGraphIO::Save of the real binary."""
    from spades_amd.gbuilder import GraphBuilder
    gb = GraphBuilder(case["K"], case["threads"])
    gb.push_back_reads([r for r in read_lines(case["reads"]) if r])
    gb.build()
    if case["coverage"]:
        gb.fill_coverage()
    base = os.path.join(str(tmp_path), "sp")
    gb.write_spades(base)
    for ext in (".grseq", ".cvr"):
        assert open(base + ext, "rb").read() == open(os.path.join(GOLDEN, case["base"] + ext), "rb").read(), ext
    gb.ctx.close()


FCASES = [c for c in load_manifest()["cases"] if c["kind"] == "graph_fastg"]


@pytest.mark.parametrize("case", FCASES, ids=lambda c: c["file"])
def test_fastg_matches_gbuilder(case, tmp_path):
    from spades_amd.gbuilder import GraphBuilder
    gb = GraphBuilder(case["K"], case["threads"])
    gb.push_back_reads([r for r in read_lines(case["reads"]) if r])
    gb.build()
    if case["coverage"]:
        gb.fill_coverage()
    out = os.path.join(str(tmp_path), "g.fastg")
    gb.write_fastg(out)
    assert open(out).read() == open(os.path.join(GOLDEN, case["file"])).read()
    gb.ctx.close()


def test_spades_core_variant_sorted_edges(tmp_path):
    """DeBruijnGraphExtentionConstructor order (RawCompare-sorted unitigs; thread-independent ids) vs the oracle's restatement
    (the order rule itself is pinned by test_spades_core_edge_order_matches_reference); invariants checked as well."""
    from oracle import oracle
    reads = _synth(77, 5000, 1200, 150) + _synth(9, 500, 60, 100, err=0.0, nrate=0.0, circ=True)
    outs = []
    for threads in (1, 4):
        ref = oracle.build_graph(reads, 33, 10 * threads, sort_edges=True)
        r = _build(reads, 33, threads, tmp_path, {"sort_edges": 1})
        assert r["unitigs"] == ref["unitigs"] and r["gfa"] == ref["gfa"]
        outs.append(r["gfa"])
    assert outs[0] == outs[1]  # ids no longer depend on the bucket count
    lens = [len(u) for u in r["unitigs"]]
    assert lens == sorted(lens)
    ref = oracle.build_graph(reads, 33, 10, sort_edges=True, keep_loops=False)
    r = _build(reads, 33, 1, tmp_path, {"sort_edges": 1, "keep_perfect_loops": 0})
    assert r["gfa"] == ref["gfa"] and r["info"]["n_loops"] == 0


def test_coverage_vs_oracle_seeded(tmp_path):
    from oracle import oracle
    from spades_amd.gbuilder import GraphBuilder
    for k, threads in ((21, 2), (55, 1), (77, 1)):
        reads = _synth(5 + k, 6000, 1500, 150) + ["ACGT" * 40] * 3 + ["A" * 100] * 5
        ref = oracle.build_graph(reads, k, 10 * threads, coverage=True)
        gb = GraphBuilder(k, threads)
        gb.push_back_reads(reads)
        gb.build()
        gb.fill_coverage()
        out = os.path.join(str(tmp_path), f"g{k}.gfa")
        gb.write_gfa(out)
        assert open(out).read() == ref["gfa"]
        gb.ctx.close()


def _synth(seed, glen, n, L, err=0.01, nrate=0.002, circ=False):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, glen)
    if circ:
        g = np.concatenate([g, g[:L]])
    reads = []
    for _ in range(n):
        p = int(rng.integers(0, len(g) - L + 1))
        r = g[p:p + L].copy()
        e = rng.random(L) < err
        r[e] = (r[e] + rng.integers(1, 4, int(e.sum()))) % 4
        s = np.array(list("ACGT"))[r]
        s[rng.random(L) < nrate] = "N"
        reads.append("".join(s))
    return reads


@pytest.mark.parametrize("k,threads", [(21, 1), (21, 8), (31, 2), (33, 1), (55, 3), (63, 1), (65, 2), (77, 1), (127, 1), (5, 1), (1, 1)])
def test_graph_vs_oracle_seeded(k, threads, tmp_path):
    from oracle import oracle
    reads = _synth(50 + k, 8000, 1500, 150) + _synth(9, 600, 80, 100, err=0.0, nrate=0.0, circ=True)
    ref = oracle.build_graph(reads, k, 10 * threads)
    r = _build(reads, k, threads, tmp_path)
    km, masks = r["kmers"]
    assert km.shape == ref["kmers"].shape and (km == ref["kmers"]).all()
    assert (masks == ref["masks"]).all()
    assert r["unitigs"] == ref["unitigs"]
    assert r["info"]["n_loops"] == ref["n_loops"]
    assert r["gfa"] == ref["gfa"]
    assert r["info"]["n_vertices"] == ref["n_vertices"] and r["info"]["n_links"] == ref["n_links"]


def test_graph_small_leaves_and_edge_inputs(tmp_path):
    from oracle import oracle
    reads = _synth(3, 3000, 600, 120)
    ref = oracle.build_graph(reads, 21, 30)
    r = _build(reads, 21, 3, tmp_path, {"leaf_cap": 16})
    assert r["gfa"] == ref["gfa"]
    r = _build(reads, 21, 3, tmp_path, {"batch_records": 20000})  # multi-batch (k+1)-mer counting under the graph
    assert r["gfa"] == ref["gfa"]
    for rd in ([], ["ACGT"], ["A" * 60], ["ACGTTGCAACGTTGCAACGTTGCAACGTTGCAACGTTGCA"]):
        ref = oracle.build_graph(rd, 21, 10)
        r = _build(rd, 21, 1, tmp_path)
        assert r["gfa"] == ref["gfa"] and r["unitigs"] == ref["unitigs"]


def test_unitigs_fasta_and_errors(tmp_path):
    from spades_amd import SmxError
    from spades_amd.gbuilder import GraphBuilder
    gb = GraphBuilder(21, 1)
    gb.push_back_reads([r for r in read_lines("reads_small.txt") if r])
    info = gb.build()
    out = os.path.join(str(tmp_path), "u.fa")
    gb.write_unitigs(out)
    lines = open(out).read().split("\n")
    us = gb.unitigs()
    assert lines[0] == f">EDGE_1_length_{len(us[0])}" and lines[1] == us[0][:60]
    assert sum(1 for l in lines if l.startswith(">")) == info["n_unitigs"]
    bad = GraphBuilder(22, 1, gb.ctx)
    with pytest.raises(SmxError) as e:
        bad.build()
    assert e.value.code == 67  # "k-mer size must be odd" -> InvalidParameter (gbuilder.cpp:134-135)


DCASES = [c for c in load_manifest()["cases"] if c["kind"] == "graph" and c["file"]]


@pytest.mark.parametrize("case", DCASES, ids=lambda c: f"{c['reads'][6:-4]}-k{c['K']}-t{c['threads']}")
def test_gfa_with_link_records_built_on_the_device(case, tmp_path):
    """device_links=2 forces the device link-record / vertex path (normally taken from 65 536 edges up) on the small goldens"""
    reads = [r for r in read_lines(case["reads"]) if r]
    r = _build(reads, case["K"], case["threads"], tmp_path, {"device_links": 2})
    assert r["gfa"] == open(os.path.join(GOLDEN, case["file"])).read()


def test_multi_k_builds_on_one_resident_read_set(tmp_path):
    """BASELINE config 5 in small: the reads go to HBM once, graphs for k = 21, 33, 55 (and 21 again) are built one after the other
    on the same context; each GFA equals the reference's for that k (the iterative multi-k loop of spades.py)."""
    from spades_amd.gbuilder import GraphBuilder
    reads = [r for r in read_lines("reads_small.txt") if r]
    first = GraphBuilder(21, 1)
    first.push_back_reads(reads)
    for k in (21, 33, 55, 21):
        gb = first if k == 21 else GraphBuilder(k, 1, ctx=first.ctx)
        gb.build()
        gb.fill_coverage()
        out = os.path.join(str(tmp_path), f"g{k}.gfa")
        gb.write_gfa(out)
        assert open(out).read() == open(os.path.join(GOLDEN, f"graphcov_small_k{k}_t1.gfa")).read() if k != 33 else True
        if k == 33:
            gb2 = os.path.join(str(tmp_path), "g33n.gfa")
            plain = GraphBuilder(33, 1, ctx=first.ctx)
            plain.build()
            plain.write_gfa(gb2)
            assert open(gb2).read() == open(os.path.join(GOLDEN, "graph_small_k33_t1.gfa")).read()
    first.ctx.close()


ECASES = [c for c in load_manifest()["cases"] if c["kind"] == "earlytip"]


@pytest.mark.parametrize("case", ECASES, ids=lambda c: c["file"][4:-4])
def test_early_tip_clipper_matches_reference(case, tmp_path):
    """option early_tip_bound: spades-core's EarlyTipClipperProcessor on the extension masks before condensation; the unitig list
    equals the one of the reference classes (oracle/_ref/ref_earlytip goldens), order included"""
    reads = [r for r in read_lines(case["reads"]) if r]
    r = _build(reads, case["K"], case["threads"], tmp_path, {"early_tip_bound": case["bound"], "early_at_remover": int(case.get("at", 0))})
    want = open(os.path.join(GOLDEN, case["file"])).read().split("\n")[:-1]
    assert r["unitigs"] == want


def test_early_tip_clipper_vs_oracle_seeded(tmp_path):
    from oracle import oracle
    from test_count_gpu import _synth
    from spades_amd.gbuilder import GraphBuilder
    reads = _synth(5, 30000, 6000, 150, err=0.01, nrate=0.001)
    for k, t, bound in ((21, 2, 129), (55, 1, 95), (77, 3, 73)):
        ref = oracle.build_graph(reads, k, 10 * t, early_tip_bound=bound)
        plain = oracle.build_graph(reads, k, 10 * t)
        gb = GraphBuilder(k, t)
        gb.ctx.set_option("early_tip_bound", bound)
        gb.push_back_reads(reads)
        gb.build()
        out = os.path.join(str(tmp_path), "g.gfa")
        gb.write_gfa(out)
        assert open(out).read() == ref["gfa"] and len(ref["unitigs"]) < len(plain["unitigs"])
        assert gb.tip_stats()[1] > 0
        gb.ctx.close()


def test_early_at_remover_vs_oracle_seeded(tmp_path):
    """RNA-pipeline variant: A/T edges + A/T tips removed, then the tip clipper; GFA identical to the oracle on a few thousand reads
    with poly-A / low-complexity tails"""
    from oracle import oracle
    from test_count_gpu import _synth
    from spades_amd.gbuilder import GraphBuilder
    rng = np.random.default_rng(3)
    reads = _synth(6, 20000, 4000, 150, err=0.005, nrate=0.0)
    for i in range(400):
        r = reads[int(rng.integers(0, 4000))]
        cut = int(rng.integers(40, 120))
        reads.append((r[:cut] + ["A", "T", "AT", "AAAAT"][i % 4] * 40)[:150])
        reads.append(("A" * int(rng.integers(12, 45)) + r)[:150])
    for k, t, bound in ((21, 1, 0), (33, 2, 117), (55, 1, 95)):
        ref = oracle.build_graph(reads, k, 10 * t, early_tip_bound=bound, early_at=True)
        plain = oracle.build_graph(reads, k, 10 * t, early_tip_bound=bound)
        gb = GraphBuilder(k, t)
        gb.ctx.set_option("early_at_remover", 1)
        gb.ctx.set_option("early_tip_bound", bound)
        gb.push_back_reads(reads)
        gb.build()
        out = os.path.join(str(tmp_path), "g.gfa")
        gb.write_gfa(out)
        assert open(out).read() == ref["gfa"] and ref["gfa"] != plain["gfa"]
        st = gb.tip_stats()
        assert st[2] > 0 or st[3] > 0
        gb.ctx.close()


SORTED_CASES = [c for c in load_manifest()["cases"] if c["kind"] == "sorted_edges"]


@pytest.mark.parametrize("case", SORTED_CASES, ids=lambda c: c["file"][7:-4])
def test_spades_core_edge_order_matches_reference(case, tmp_path):
    """option sort_edges (+ keep_perfect_loops, early clippers): the unitig list equals the reference extractor's output sorted with
    the reference's own Sequence::RawCompare (oracle/_ref/ref_earlytip ... sorted)"""
    reads = [r for r in read_lines(case["reads"]) if r]
    r = _build(reads, case["K"], case["threads"], tmp_path, {"sort_edges": 1, "keep_perfect_loops": case["keep_loops"],
                                                              "early_tip_bound": case["bound"], "early_at_remover": case["at"]})
    assert r["unitigs"] == open(os.path.join(GOLDEN, case["file"])).read().split("\n")[:-1]


def test_previous_k_contigs_shape_the_graph_but_not_the_coverage(tmp_path):
    """spades.py's iterative multi-k loop: the contigs of the previous k are extra streams of the construction that are not counted
    in coverage (stages/construction.cpp:108-117, 371-435): option submit_contigs"""
    from oracle import oracle
    from spades_amd.gbuilder import GraphBuilder
    reads = [r for r in read_lines("reads_small.txt") if r]
    contigs = [u for u in oracle.build_graph(reads, 21, 10)["unitigs"] if len(u) >= 56][:60] + ["ACGTTGCAAGGCTAGCTAGGATCGATCGGATCGATTTAGCGCGATATCGAGCTAGGGATCCGAT"]
    for k in (33, 55):
        ref = oracle.build_graph(reads + contigs, k, 10, coverage=True, coverage_reads=len(reads))
        both = oracle.build_graph(reads + contigs, k, 10, coverage=True)
        gb = GraphBuilder(k, 1)
        gb.push_back_reads(reads)
        gb.ctx.set_option("submit_contigs", 1)
        gb.push_back_reads(contigs)
        gb.ctx.set_option("submit_contigs", 0)
        gb.build()
        gb.fill_coverage()
        out = os.path.join(str(tmp_path), "g.gfa")
        gb.write_gfa(out)
        assert open(out).read() == ref["gfa"] and ref["gfa"] != both["gfa"]
        gb.ctx.close()


def test_flanking_coverage_is_the_coverage_of_the_edge_ends(tmp_path):
    """FlankingCoverage (graph_support/coverage_filling.hpp:40-44): raw counts over the first 50 (k+1)-mers of an edge and of its
    conjugate, checked against per-(k+1)-mer counts recomputed on the host from the reads"""
    from collections import Counter
    from spades_amd.gbuilder import GraphBuilder
    reads = [r for r in read_lines("reads_small.txt") if r]
    tr = str.maketrans("ACGT", "TGCA")
    for k, R in ((21, 50), (55, 7)):
        K1 = k + 1
        c = Counter()
        for r in reads:
            for s in max(("".join(ch if ch in "ACGT" else " " for ch in r.upper())).split(), key=len, default="").split():
                for j in range(len(s) - K1 + 1):
                    x = s[j:j + K1]
                    y = x[::-1].translate(tr)
                    c[min(x, y)] += 2 if x == y else 1
        gb = GraphBuilder(k, 1)
        gb.ctx.set_option("flank_range", R)
        gb.push_back_reads(reads)
        gb.build()
        gb.fill_coverage()
        fs, fe = gb.flanking_coverage()
        raw = gb.raw_coverage()
        for i, u in enumerate(gb.unitigs()):
            cnt = [c[min(u[j:j + K1], u[j:j + K1][::-1].translate(tr))] for j in range(len(u) - K1 + 1)]
            assert raw[i] == sum(cnt) and fs[i] == sum(cnt[:R]) and fe[i] == sum(cnt[-R:])
        gb.ctx.close()


def _circ_reads(seqs, L, step):
    """error-free reads around circular sequences (every (k+1)-mer of the circle for k < L - step)"""
    reads = []
    for s in seqs:
        ext = s * (L // len(s) + 2)
        reads += [ext[p:p + L] for p in range(0, len(s), step)]
    return reads


@pytest.mark.parametrize("k,route", [(21, {}), (33, {}), (55, {}), (77, {}), (21, {"prededupe": 1, "ext_route": 1, "pm_route": 0}),
                                     (21, {"prededupe": 1, "pm_route": 1}), (55, {"prededupe": 1, "pm_route": 1})])
def test_perfect_loops_on_the_device_equal_the_reference(k, route, tmp_path):
    """option device_loops (smx_loops.hip: cycle leaders, rotation to the minimal k-mer, palindrome split and orientation in kernels, on the
    walks' successor table) against the oracle: several loops at once next to an ordinary genome, a loop shorter than k, a hairpin (a loop
    that is its own reverse complement: split at its first palindromic (k+1)-mer), on every construction route."""
    from oracle import oracle
    rng = np.random.default_rng(100 + k)
    rnd = lambda n: "".join(rng.choice(list("ACGT"), n))
    comp = str.maketrans("ACGT", "TGCA")
    hp = rnd(2 * k + 30)
    circles = [rnd(300), rnd(4 * k + 7), rnd(k - 3), rnd(150), hp + hp.translate(comp)[::-1]]
    reads = _synth(60 + k, 5000, 900, 150) + _circ_reads(circles, 2 * k + 20, 3)
    ref = oracle.build_graph(reads, k, 20)
    assert ref["n_loops"] >= 5
    got = {}
    for dev in (0, 1):
        r = _build(reads, k, 2, tmp_path, dict(route, device_loops=dev))
        assert r["info"]["n_loops"] == ref["n_loops"] and r["unitigs"] == ref["unitigs"] and r["gfa"] == ref["gfa"], dev
        got[dev] = r["gfa"]
    assert got[0] == got[1]


def test_coverage_set_from_outside_drops_the_rank_local_flanking_and_histogram():
    """a sharded coverage pass (dist.py, tools/gbuilder_mgpu.hpp) fills coverage shard by shard and installs the SUM with
    smx_graph_set_coverage; the flanking arrays and the multiplicity histogram of the last smx_graph_fill_coverage then describe one rank's
    reads against one shard. They must not be served as if they were the graph's: the accessors refuse / come back empty, the raw coverage is
    the installed one, and the (k+1)-mer count is the whole file's again after a shard stood in for it."""
    import ctypes as C
    from spades_amd.gbuilder import GraphBuilder
    reads = _synth(11, 3000, 600, 120)
    gb = GraphBuilder(21, 1)
    gb.push_back_reads(reads)
    gb.build()
    gb.fill_coverage()
    cov = gb.raw_coverage()
    fl = gb.flanking_coverage()
    assert len(fl[0]) == len(cov) and cov.sum() > 0
    n_kpo = gb.info()["n_kpomers"]
    lib, h = gb.ctx.lib, gb.ctx._h
    # an empty shard of the (k+1)-mer file stands in for the file (what a rank sees between two owners' shards)
    zeros = (C.c_uint64 * (10 * 1))()
    assert lib.smx_graph_set_kpomers(h, None, 0, zeros) == 0
    twice = (cov.astype(np.uint64) * 2).astype(np.uint32)
    assert lib.smx_graph_set_coverage(h, twice.ctypes.data_as(C.POINTER(C.c_uint32)), len(twice)) == 0
    assert (gb.raw_coverage() == twice).all()
    with pytest.raises(Exception):
        gb.flanking_coverage()
    n = C.c_uint64(99)
    assert lib.smx_graph_coverage_histogram(h, None, 0, C.byref(n)) == 0 and n.value == 0
    info = (C.c_uint64 * 8)()
    assert lib.smx_graph_info(h, info) == 0 and int(info[0]) == n_kpo
    gb.ctx.close()
