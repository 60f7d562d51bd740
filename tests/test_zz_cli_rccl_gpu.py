"""GPU: the C++ multi-GPU hosts of the CLI clones (spades_amd/tools/kmercount_mgpu.hpp, gbuilder_mgpu.hpp: forked ranks over librccl)
with ONE rank against the real library — same output bytes as the reference tools' goldens. World sizes 2-3 of the same hosts run
in the CPU tier against a shim (tests/test_mgpu_hosts_cpu.py); N > 1 on hardware needs N GPUs (the driver's boxes).
The file sorts last on purpose: these tests start processes of their own, and `pytest -x` should have seen every other test before
one of them can stop the run."""
import gzip
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT, load_manifest, read_lines

pytestmark = pytest.mark.gpu
TOOLS = os.path.join(ROOT, "spades_amd", "tools")
KC = os.path.join(TOOLS, "spades-kmercount-mi355x")
GB = os.path.join(TOOLS, "spades-gbuilder-mi355x")


def _fastq(path, reads, gz=False):
    op = gzip.open if gz else open
    with op(path, "wt") as f:
        for i, r in enumerate(reads):
            if r:
                f.write(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n")


def _launch_ranks(argv, env=None, timeout=90):
    """One launch of a tool with --gpus N. In one of four GPU runs of round 4 a one-rank launch (of 60 in all) did not come back; the place was never
    seen, rounds 4-5 repeated such a launch up to three times. Round 6 ran 200 one-rank launches of both hosts in a row under a 20 s watchdog
    (tools/rccl_launch_loop.py: 200 ok, median 2.7 s, max 3.1 s — profiles/r06/rccl_one_rank_200_launches.log), so the retry is gone: the ranks still
    watch themselves (tools/rank_watchdog.hpp, SMX_MGPU_WATCHDOG: a phase without progress for 60 s prints the rank, its last milestone and a
    backtrace of the blocked thread and leaves with code 75), and a launch that does not come back fails its test with that report."""
    import tempfile
    env = dict(env if env is not None else os.environ, SMX_MGPU_WATCHDOG="60")
    with tempfile.TemporaryFile("w+") as err:
        try:
            r = subprocess.run(argv, stdout=subprocess.DEVNULL, stderr=err, env=env, timeout=timeout)
        except subprocess.TimeoutExpired:
            err.seek(0)
            raise AssertionError(f"{' '.join(argv[:1] + argv[-4:])} did not finish in {timeout} s\n{err.read()[-3000:]}")
        err.seek(0)
        assert r.returncode == 0, f"{argv} -> {r.returncode}\n{err.read()[-3000:]}"


def test_kmercount_cli_rccl_host_one_rank(tmp_path):
    """--gpus 1: the C++ multi-GPU host (forked rank, librccl communicator, grouped ncclSend/ncclRecv to itself, owner-side count,
    pwrite of the bucket range) must write the reference bytes; N > 1 needs N GPUs (the driver's boxes)."""
    cases = [c for c in load_manifest()["cases"] if c["kind"] == "count" and c.get("file") and c["mode"] == "A" and c["num_buckets"] == 16]
    reads = read_lines("reads_tiny.txt")
    f1, f2 = str(tmp_path / "a.fq"), str(tmp_path / "b.fq.gz")
    _fastq(f1, reads[0::2])
    _fastq(f2, reads[1::2], gz=True)
    for i, c in enumerate(cases[:4]):
        wd = tmp_path / f"m{c['K']}"
        wd.mkdir()
        # (alternately: the segment that stays on the rank as a device copy / through ncclSend + ncclRecv to itself)
        _launch_ranks([KC, "-k", str(c["K"]), "-w", str(wd), "--gpus", "1", f1, f2], env=dict(os.environ, **({"SMX_MGPU_SELF_RCCL": "1"} if i % 2 else {})))
        assert open(wd / "final_kmers", "rb").read() == open(os.path.join(GOLDEN, c["file"]), "rb").read()


def test_gbuilder_cli_rccl_host_one_rank(tmp_path):
    """--gpus 1: the C++ multi-GPU host of the construction (forked rank, librccl communicator, k-mers with their mask bytes exchanged
    with itself, owner-side shard, gathered structure, graph, writer) writes the reference's GFA on both routes to the shard, with and
    without -c; N > 1 needs N GPUs (the driver's boxes)."""
    man = load_manifest()["cases"]
    reads = [r for r in read_lines("reads_small.txt") if r]
    fa, fq = str(tmp_path / "r.fa"), str(tmp_path / "r.fq")
    with open(fa, "w") as f:
        for i, r in enumerate(reads):
            f.write(f">r{i}\n{r}\n")
    _fastq(fq, reads)
    out = str(tmp_path / "g.gfa")
    # FASTA: every rank parses the file and keeps its reads; FASTQ: its byte range, in pieces (SMX_MGPU_PARTS) with a small chunk
    # (SMX_MGPU_CHUNK: carry-over of the cut record between chunks); SMX_MGPU_KPOMERS: the route by the sharded (k+1)-mer count;
    # SMX_MGPU_SELF_RCCL: the segment that stays on the rank goes through ncclSend + ncclRecv instead of a device copy.
    # (tools/r4_mgpu_diag.py runs the whole matrix of variants: 26 launches, all equal, profiles/r04/gbuilder_mgpu_one_rank_26_launches.log)
    plan = (("graph", 21, [], fa, {"SMX_MGPU_SELF_RCCL": "1"}), ("graph", 21, [], fq, {"SMX_MGPU_PARTS": "3", "SMX_MGPU_CHUNK": "2048"}),
            ("graph", 33, [], fq, {"SMX_MGPU_KPOMERS": "1", "SMX_MGPU_PARTS": "2", "SMX_MGPU_SELF_RCCL": "1"}),
            ("graph", 55, [], fq, {"SMX_MGPU_PARTS": "3", "SMX_MGPU_CHUNK": "2048", "SMX_MGPU_SELF_RCCL": "1"}),
            ("graph_cov", 21, ["-c"], fq, {"SMX_MGPU_KPOMERS": "1"}), ("graph_cov", 55, ["-c"], fa, {"SMX_MGPU_SELF_RCCL": "1"}),
            # round 6: the k-mer file stays sharded — smx_shard_walks with the host's own collectives (counts by ncclAllGather, grouped ncclSend / ncclRecv,
            # ncclAllReduce), unitigs gathered, smx_build_graph_from_unitigs; -c shard by shard on a graph that has no k-mer file
            ("graph", 21, [], fq, {"SMX_MGPU_WALKS": "distributed", "SMX_MGPU_SELF_RCCL": "1", "SMX_OPTS": "walk_chunk=4096,walk_start_chunk=512"}),
            ("graph", 55, [], fq, {"SMX_MGPU_WALKS": "distributed", "SMX_MGPU_PARTS": "2"}),
            ("graph_cov", 21, ["-c"], fq, {"SMX_MGPU_WALKS": "distributed", "SMX_MGPU_KPOMERS": "1", "SMX_MGPU_SELF_RCCL": "1"}),
            ("graph_cov", 55, ["-c"], fa, {"SMX_MGPU_WALKS": "distributed", "SMX_MGPU_ROUND_WORDS": "4096", "SMX_MGPU_SELF_RCCL": "1"}))
    for kind, K, cov, inp, env in plan:
        c = [c for c in man if c["kind"] == kind and c["file"] and c["reads"] == "reads_small.txt" and c["K"] == K and c["threads"] == 3][0]
        if os.path.exists(out):
            os.remove(out)
        _launch_ranks([GB, inp, out, "-k", str(K), "-t", "3", "--gfa", "--gpus", "1"] + cov, env=dict(os.environ, **env))
        assert open(out).read() == open(os.path.join(GOLDEN, c["file"])).read(), (K, cov, env)


def test_gbuilder_cli_rccl_host_other_outputs(tmp_path):
    """--gpus 1 with --spades (-c) and a missing input: same files, same exit code"""
    c = [c for c in load_manifest()["cases"] if c["kind"] == "graph_spades" and c["base"] == "spades_small_k21_t3_c"][0]
    fq = str(tmp_path / "r.fq")
    _fastq(fq, [r for r in read_lines(c["reads"]) if r])
    out = str(tmp_path / "sp")
    _launch_ranks([GB, fq, out, "-k", "21", "-t", "3", "-c", "--spades", "--gpus", "1"])
    for ext in (".grseq", ".cvr"):
        assert open(out + ext, "rb").read() == open(os.path.join(GOLDEN, c["base"] + ext), "rb").read()
    assert subprocess.call([GB, "/nonexistent.fq", str(tmp_path / "o"), "-k", "21", "--gpus", "1"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) == 65
