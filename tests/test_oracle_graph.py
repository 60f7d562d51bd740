"""CPU: pins the graph-construction restatement (oracle/smx_oracle_graph.c) to the reference:
  * GFA text byte-identical to the real spades-gbuilder binary for several (dataset, k, -t) cases — the GFA
    depends on -t through the 10*t bucket order (SURVEY.md finding 3), perfect loops included;
  * the six k=5 known-answer tests of src/test/debruijn/construction_test.cpp:30-64 (edge sets, order-insensitive).
"""
import hashlib
import os

import pytest

from conftest import GOLDEN, load_manifest, read_lines
from oracle import oracle

GCASES = [c for c in load_manifest()["cases"] if c["kind"] == "graph"]


def _rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


@pytest.mark.parametrize("case", GCASES, ids=lambda c: f"{c['reads'][6:-4]}-k{c['K']}-t{c['threads']}")
def test_gfa_matches_spades_gbuilder(case):
    reads = [r for r in read_lines(case["reads"]) if r]
    g = oracle.build_graph(reads, case["K"], case["num_buckets"])
    assert g["gfa"].count("\nS\t") == case["n_segments"] and g["gfa"].count("\nL\t") == case["n_links"]
    assert hashlib.md5(g["gfa"].encode()).hexdigest() == case["md5"]
    if case["file"]:
        assert g["gfa"] == open(os.path.join(GOLDEN, case["file"])).read()


CCASES = [c for c in load_manifest()["cases"] if c["kind"] == "graph_cov"]


@pytest.mark.parametrize("case", CCASES, ids=lambda c: f"{c['reads'][6:-4]}-k{c['K']}-t{c['threads']}")
def test_gfa_with_coverage_matches_spades_gbuilder_c(case):
    """-c: DP:f / KC:i tags (CoverageHashMapBuilder + FillCoverageAndFlankingFromPHM)."""
    reads = [r for r in read_lines(case["reads"]) if r]
    g = oracle.build_graph(reads, case["K"], case["num_buckets"], coverage=True)
    assert g["gfa"] == open(os.path.join(GOLDEN, case["file"])).read()


# Note: construction_test.cpp:97-106 (SimpleTestEarlyPairedInfo, coverage {CCAC:4, ...}) feeds a forward-only stream
# (no RCWrap), which neither spades-gbuilder nor this boundary can express; coverage is pinned by the real-binary goldens above.


# construction_test.cpp:30-64 (AssertGraph(k, reads, etalon_edges): edge set incl. reverse complements)
KNOWN = [
    ("SimpleThread", ["ACAAACCACCA"], ["ACAAACCACCA"]),
    ("SimpleThread2", ["ACAAACCACCC", "AAACCACCCAC"], ["ACAAACCACCCAC"]),
    ("SplitThread", ["ACAAACCACCA", "ACAAACAACCC"], ["ACAAAC", "CAAACCACCA", "CAAACAACCC"]),
    ("SplitThread2", ["ACAAACCACCA", "ACAAACAACCA"], ["AACCACCA", "ACAAAC", "CAAACCA", "CAAACAACCA"]),
    ("Buldge", ["ACAAAACACCA", "ACAAACCACCA"], ["ACAAAACACCA", "ACAAACCACCA"]),
    ("CondenseSimple", ["CGAAACCAC", "CGAAAACAC", "AACCACACC", "AAACACACC"], ["CGAAAACACAC", "CACACC", "CGAAACCACAC"]),
]


@pytest.mark.parametrize("name,reads,edges", KNOWN, ids=[k[0] for k in KNOWN])
def test_reference_known_answer_graphs(name, reads, edges):
    g = oracle.build_graph(reads, 5, 10)
    got = set()
    for u in g["unitigs"]:
        got.add(u)
        got.add(_rc(u))
    want = set()
    for e in edges:
        want.add(e)
        want.add(_rc(e))
    assert got == want


def test_unitig_invariants():
    reads = [r for r in read_lines("reads_small.txt") if r]
    g = oracle.build_graph(reads, 21, 30)
    assert all(u >= _rc(u) for u in g["unitigs"])  # every S sequence satisfies s >= RC(s)
    # every canonical 22-mer of the reads lies on exactly one unitig (or its RC)
    kp, _ = oracle.count(reads, 22, "B", 30)
    seen = {}
    for u in g["unitigs"]:
        for i in range(len(u) - 21):
            x = u[i:i + 22]
            c = min(x, _rc(x))
            seen[c] = seen.get(c, 0) + 1
    assert len(seen) == len(kp)
    assert all(v == 1 or k == _rc(k) for k, v in seen.items())


ECASES = [c for c in load_manifest()["cases"] if c["kind"] == "earlytip"]


@pytest.mark.parametrize("case", ECASES, ids=lambda c: c["file"][4:-4])
def test_oracle_early_tip_clipper_matches_reference(case):
    """spades-core's EarlyTipClipperProcessor (early_simplification.hpp:38-162): the oracle's sequential restatement gives the edge
    sequences of the reference classes (oracle/_ref/ref_earlytip), in the extractor's order, also for the multi-threaded runs."""
    reads = [r for r in read_lines(case["reads"]) if r]
    g = oracle.build_graph(reads, case["K"], case["num_buckets"], early_tip_bound=case["bound"], early_at=bool(case.get("at")))
    want = open(os.path.join(GOLDEN, case["file"])).read().split("\n")[:-1]
    assert g["unitigs"] == want


SCASES = [c for c in load_manifest()["cases"] if c["kind"] == "sorted_edges"]


@pytest.mark.parametrize("case", SCASES, ids=lambda c: c["file"][7:-4])
def test_oracle_spades_core_edge_order_matches_reference(case):
    """DeBruijnGraphExtentionConstructor::ConstructGraph (debruijn_graph_constructor.hpp:590-604): unitigs (with or without perfect
    loops) sorted by the reference's own Sequence::RawCompare = the oracle's sort_edges order"""
    reads = [r for r in read_lines(case["reads"]) if r]
    g = oracle.build_graph(reads, case["K"], case["num_buckets"], sort_edges=True, keep_loops=bool(case["keep_loops"]),
                           early_tip_bound=case["bound"], early_at=bool(case["at"]))
    assert g["unitigs"] == open(os.path.join(GOLDEN, case["file"])).read().split("\n")[:-1]


def test_oracle_perfect_loops_match_spades_gbuilder_on_plasmids():
    """200 circular 5 kb genomes, error-free reads (tests/synth.py: synth_codes_plasmids): the real spades-gbuilder collects 200 perfect
    loops (debruijn_graph_constructor.hpp:252-293, 359-397: cycle minimum, self-conjugate split) and the C restatement writes the
    same GFA, ± -c. Golden: tests/golden/scale_200k_g1000k_s83_plasmids.json (make_golden_scale.py ... gfaplasmids)."""
    import hashlib
    import json
    import numpy as np
    import synth
    g = json.load(open(os.path.join(GOLDEN, "scale_200k_g1000k_s83_plasmids.json")))
    codes = synth.synth_codes_plasmids(g["seed"], g["genome_len"], g["n_reads"], g["err"], g["n_rate"])
    assert hashlib.md5(codes.tobytes()).hexdigest() == g["codes_md5"]
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    reads = [lut[c].tobytes().decode() for c in codes]
    r = oracle.build_graph(reads, g["k"], 10 * g["effective_threads"], coverage=True)
    assert r["n_loops"] == g["perfect_loops"] == 200 and len(r["unitigs"]) == g["gfa_S_lines"]
    assert hashlib.md5(r["gfa"].encode()).hexdigest() == g["gfa_cov_md5"]
