"""CPU: the host side of the construction path (link records, vertex numbering, GFA / FASTG / .grseq+.cvr / unitig FASTA writers,
spades_amd/csrc/smx_graph_host.hpp) driven through the context-free C-ABI entry smx_host_write_graph, on unitigs produced by the
oracle, against the goldens of the real spades-gbuilder binary. No GPU involved."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_manifest, read_lines
from oracle import oracle
from spades_amd import _lib

CASES = [c for c in load_manifest()["cases"] if c["kind"] in ("graph", "graph_fastg", "graph_spades") and c.get("file", c.get("base"))]


def _rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def _nodes(unitigs, k):
    """2*rank + rc with rank = any injective id of the canonical k-mer."""
    ids = {}
    st, en = [], []
    for u in unitigs:
        for arr, km in ((st, u[:k]), (en, u[-k:])):
            c = min(km, _rc(km))
            r = ids.setdefault(c, len(ids))
            arr.append(2 * r + (0 if km == c else 1))
    return np.array(st, dtype=np.uint64), np.array(en, dtype=np.uint64)


def _write(unitigs, k, fmt, path, cov=None, sort_edges=0):
    lib = _lib.load()
    off = np.zeros(len(unitigs) + 1, dtype=np.uint64)
    if unitigs:
        off[1:] = np.cumsum([len(u) for u in unitigs])
    st, en = _nodes(unitigs, k)
    covp = cov.ctypes.data_as(C.POINTER(C.c_uint32)) if cov is not None else None
    rc = lib.smx_host_write_graph(k, len(unitigs), off.ctypes.data_as(C.POINTER(C.c_uint64)), "".join(unitigs).encode(),
                                  st.ctypes.data_as(C.POINTER(C.c_uint64)), en.ctypes.data_as(C.POINTER(C.c_uint64)), covp,
                                  sort_edges, fmt, path.encode(), b"SPAdes-4.3.0-dev")
    assert rc == 0


def _cov_from_gfa(gfa):
    return np.array([int(l.split("KC:i:")[1]) for l in gfa.split("\n") if l.startswith("S\t")], dtype=np.uint32)


@pytest.mark.parametrize("case", CASES, ids=lambda c: (c.get("file") or c["base"]))
def test_host_writers_match_gbuilder(case, tmp_path):
    reads = [r for r in read_lines(case["reads"]) if r]
    cov = bool(case.get("coverage"))
    g = oracle.build_graph(reads, case["K"], case["num_buckets"], coverage=cov)
    covs = _cov_from_gfa(g["gfa"]) if cov else None
    if case["kind"] == "graph":
        out = str(tmp_path / "g.gfa")
        _write(g["unitigs"], case["K"], 1, out)
        assert open(out).read() == open(os.path.join(GOLDEN, case["file"])).read()
    elif case["kind"] == "graph_fastg":
        out = str(tmp_path / "g.fastg")
        _write(g["unitigs"], case["K"], 2, out, covs)
        assert open(out).read() == open(os.path.join(GOLDEN, case["file"])).read()
    else:
        base = str(tmp_path / "sp")
        _write(g["unitigs"], case["K"], 3, base, covs)
        for ext in (".grseq", ".cvr"):
            assert open(base + ext, "rb").read() == open(os.path.join(GOLDEN, case["base"] + ext), "rb").read(), ext


def test_unitig_fasta_and_sorted_variant(tmp_path):
    reads = [r for r in read_lines("reads_small.txt") if r]
    g = oracle.build_graph(reads, 21, 10)
    out = str(tmp_path / "u.fa")
    _write(g["unitigs"], 21, 0, out)
    lines = open(out).read().split("\n")
    assert lines[0] == f">EDGE_1_length_{len(g['unitigs'][0])}" and lines[1] == g["unitigs"][0][:60]
    gs = oracle.build_graph(reads, 21, 10, sort_edges=True)
    out = str(tmp_path / "s.gfa")
    _write(g["unitigs"], 21, 1, out, sort_edges=1)  # host-side RawCompare sort of the gbuilder-ordered unitigs
    assert open(out).read() == gs["gfa"]


@pytest.mark.parametrize("grain", ["3", "50"])
def test_gfa_writer_blocks_on_several_threads(grain, tmp_path, monkeypatch):
    """SMX_WRITE_GRAIN forces the block-parallel formatting path (normally used from 262 144 segments up) on a golden"""
    monkeypatch.setenv("SMX_WRITE_GRAIN", grain)
    case = [c for c in load_manifest()["cases"] if c["kind"] == "graph" and c["file"] == "graph_small_k21_t3.gfa"][0]
    reads = [r for r in read_lines(case["reads"]) if r]
    g = oracle.build_graph(reads, case["K"], case["num_buckets"])
    out = str(tmp_path / "g.gfa")
    _write(g["unitigs"], case["K"], 1, out)
    assert open(out).read() == open(os.path.join(GOLDEN, case["file"])).read()
    fcase = [c for c in load_manifest()["cases"] if c["kind"] == "graph_fastg"][0]  # the FASTG and FASTA writers use the same block scheme
    reads = [r for r in read_lines(fcase["reads"]) if r]
    gf = oracle.build_graph(reads, fcase["K"], fcase["num_buckets"], coverage=bool(fcase.get("coverage")))
    out = str(tmp_path / "g.fastg")
    _write(gf["unitigs"], fcase["K"], 2, out, _cov_from_gfa(gf["gfa"]) if fcase.get("coverage") else None)
    assert open(out).read() == open(os.path.join(GOLDEN, fcase["file"])).read()
    for sc in [c for c in load_manifest()["cases"] if c["kind"] == "graph_spades"][:3]:  # .grseq: first-mention bookkeeping across blocks
        reads = [r for r in read_lines(sc["reads"]) if r]
        gs = oracle.build_graph(reads, sc["K"], sc["num_buckets"], coverage=bool(sc.get("coverage")))
        base = str(tmp_path / "sp")
        _write(gs["unitigs"], sc["K"], 3, base, _cov_from_gfa(gs["gfa"]) if sc.get("coverage") else None)
        for ext in (".grseq", ".cvr"):
            assert open(base + ext, "rb").read() == open(os.path.join(GOLDEN, sc["base"] + ext), "rb").read(), ext
    freads = [r for r in read_lines(fcase["reads"]) if r]
    gs = oracle.build_graph(freads, fcase["K"], fcase["num_buckets"], sort_edges=True)  # threaded RawCompare sort of the edges
    out = str(tmp_path / "s.gfa")
    _write(gf["unitigs"], fcase["K"], 1, out, sort_edges=1)
    assert open(out).read() == gs["gfa"]
    out = str(tmp_path / "u.fa")
    _write(gf["unitigs"], fcase["K"], 0, out)
    want = "".join(f">EDGE_{i + 1}_length_{len(u)}\n" + "".join(u[p:p + 60] + "\n" for p in range(0, len(u), 60)) for i, u in enumerate(gf["unitigs"]))
    assert open(out).read() == want
