"""GPU: the C++ CLI clones (spades_amd/tools) behave like spades-kmercount / spades-gbuilder on the same files:
same output bytes (reference goldens), same exit codes (src/test/integration/test_error_codes.py:107-272)."""
import gzip
import os
import subprocess

import pytest

from conftest import EMU, GOLDEN, ROOT, load_manifest, read_lines

pytestmark = pytest.mark.gpu
TOOLS = os.path.join(ROOT, "spades_amd", "tools")
KC = os.path.join(TOOLS, "spades-kmercount-mi355x")
GB = os.path.join(TOOLS, "spades-gbuilder-mi355x")
if EMU:  # SMX_EMU=1: the same main()s linked against the SIMT stand-in of the library (tests/simt_emu/build_emu.py)
    import build_emu
    KC, GB = build_emu.build_tools()


def _fastq(path, reads, gz=False):
    op = gzip.open if gz else open
    with op(path, "wt") as f:
        for i, r in enumerate(reads):
            if r:
                f.write(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n")


def test_kmercount_cli_matches_reference_bytes(tmp_path):
    cases = [c for c in load_manifest()["cases"] if c["kind"] == "count" and c.get("file") and c["mode"] == "A" and c["num_buckets"] == 16]
    reads = read_lines("reads_tiny.txt")
    fq = str(tmp_path / "r.fq.gz")
    _fastq(fq, reads, gz=True)
    for c in cases:
        wd = tmp_path / f"w{c['K']}"
        wd.mkdir()
        subprocess.check_call([KC, "-k", str(c["K"]), "-t", "3", "-w", str(wd), fq], stdout=subprocess.DEVNULL)
        assert open(wd / "final_kmers", "rb").read() == open(os.path.join(GOLDEN, c["file"]), "rb").read()


def test_gbuilder_cli_matches_reference_gfa(tmp_path):
    for c in [c for c in load_manifest()["cases"] if c["kind"] == "graph" and c["file"] and c["reads"] in ("reads_small.txt", "reads_loop.txt")]:
        reads = [r for r in read_lines(c["reads"]) if r]
        fa = str(tmp_path / "r.fa")
        with open(fa, "w") as f:
            for i, r in enumerate(reads):
                f.write(f">r{i}\n{r}\n")
        out = str(tmp_path / "g.gfa")
        subprocess.check_call([GB, fa, out, "-k", str(c["K"]), "-t", str(c["threads"]), "--gfa"], stdout=subprocess.DEVNULL)
        assert open(out).read() == open(os.path.join(GOLDEN, c["file"])).read()


def test_gbuilder_cli_coverage_flag(tmp_path):
    c = [c for c in load_manifest()["cases"] if c["kind"] == "graph_cov" and c["reads"] == "reads_small.txt" and c["K"] == 21 and c["threads"] == 3][0]
    fq = str(tmp_path / "r.fq")
    _fastq(fq, [r for r in read_lines(c["reads"]) if r])
    out = str(tmp_path / "g.gfa")
    subprocess.check_call([GB, fq, out, "-k", "21", "-t", "3", "-c", "--gfa"], stdout=subprocess.DEVNULL)
    assert open(out).read() == open(os.path.join(GOLDEN, c["file"])).read()


def test_gbuilder_cli_spades_format(tmp_path):
    c = [c for c in load_manifest()["cases"] if c["kind"] == "graph_spades" and c["base"] == "spades_small_k21_t3_c"][0]
    fq = str(tmp_path / "r.fq")
    _fastq(fq, [r for r in read_lines(c["reads"]) if r])
    out = str(tmp_path / "sp")
    subprocess.check_call([GB, fq, out, "-k", "21", "-t", "3", "-c", "--spades"], stdout=subprocess.DEVNULL)
    for ext in (".grseq", ".cvr"):
        assert open(out + ext, "rb").read() == open(os.path.join(GOLDEN, c["base"] + ext), "rb").read()


def test_cli_exit_codes(tmp_path):
    # error_codes.hpp:14-20 — 65 file not found, 67 invalid parameter
    assert subprocess.call([KC, "-k", "21", "-w", str(tmp_path), "/nonexistent.fq"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) == 65
    assert subprocess.call([KC, "-k", "21"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) == 67
    assert subprocess.call([GB, "/nonexistent.fa", str(tmp_path / "o"), "-k", "21"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) == 65
    fa = tmp_path / "r.fa"
    fa.write_text(">a\nACGTACGTACGTACGTACGTACGTACGT\n")
    assert subprocess.call([GB, str(fa), str(tmp_path / "o"), "-k", "22"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) == 67
    assert subprocess.call([GB, str(fa), str(tmp_path / "o"), "-k", "129"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) == 67


def test_cli_tools_with_the_prededupe_stage_forced(tmp_path):
    """SMX_OPTS reaches every context, so the CLI clones can be pushed through the super-k-mer stage on the small goldens."""
    env = dict(os.environ, SMX_OPTS="prededupe=1")
    import hashlib
    tq = str(tmp_path / "t.fq")
    _fastq(tq, read_lines("reads_tiny.txt"))
    for c in [c for c in load_manifest()["cases"] if c["kind"] == "count" and c["mode"] == "A" and c["num_buckets"] == 16
              and c["reads"] == "reads_tiny.txt" and c["K"] >= 21]:
        wd = tmp_path / f"w{c['K']}"
        wd.mkdir()
        subprocess.check_call([KC, "-k", str(c["K"]), "-t", "2", "-w", str(wd), tq], stdout=subprocess.DEVNULL, env=env)
        assert hashlib.md5(open(wd / "final_kmers", "rb").read()).hexdigest() == c["md5"]
    fq = str(tmp_path / "r.fq")
    _fastq(fq, [r for r in read_lines("reads_small.txt") if r])
    c = [c for c in load_manifest()["cases"] if c["kind"] == "graph_cov" and c["reads"] == "reads_small.txt" and c["K"] == 55 and c["threads"] == 3][0]
    out = str(tmp_path / "g.gfa")
    subprocess.check_call([GB, fq, out, "-k", "55", "-t", "3", "-c", "--gfa"], stdout=subprocess.DEVNULL, env=env)
    assert open(out).read() == open(os.path.join(GOLDEN, c["file"])).read()


def test_kmercount_cli_reads_several_files_in_parallel(tmp_path):
    """R1/R2-style inputs (one gzip, one plain, one FASTA) are read by one host thread each; the k-mer file is that of all reads"""
    from oracle import oracle
    import hashlib
    reads = [r for r in read_lines("reads_small.txt") if r]
    a, b, c = reads[0::3], reads[1::3], reads[2::3]
    f1, f2, f3 = str(tmp_path / "a.fq.gz"), str(tmp_path / "b.fq"), str(tmp_path / "c.fa")
    _fastq(f1, a, gz=True)
    _fastq(f2, b)
    with open(f3, "w") as f:
        for i, r in enumerate(c):
            f.write(f">c{i}\n{r}\n")
    subprocess.check_call([KC, "-k", "33", "-w", str(tmp_path), f1, f2, f3], stdout=subprocess.DEVNULL)
    ref, _ = oracle.count(reads, 33, "A", 16)
    assert open(tmp_path / "final_kmers", "rb").read() == ref.tobytes()


def test_cli_tools_take_a_dataset_yaml(tmp_path):
    """`spades-gbuilder <dataset>.yaml` / `spades-kmercount -d <dataset>.yaml` (io::DataSet::load): libraries usable for construction
    are read, a trusted-contigs library is skipped by gbuilder; the graph equals the reference's for the same reads"""
    from oracle import oracle
    reads = [r for r in read_lines("reads_small.txt") if r]
    os.makedirs(tmp_path / "lib")
    _fastq(str(tmp_path / "lib" / "r1.fq.gz"), reads[0::2], gz=True)
    _fastq(str(tmp_path / "r2.fq"), reads[1::2])
    with open(tmp_path / "contigs.fa", "w") as f:
        f.write(">c\n" + "ACGTTGCATTGACCAGT" * 8 + "\n")
    y = tmp_path / "ds.yaml"
    y.write_text(f'- orientation: "fr"\n  type: "paired-end"\n  left reads:\n  - "lib/r1.fq.gz"\n  right reads:\n  - "{tmp_path}/r2.fq"\n'
                 f'- type: "trusted-contigs"\n  single reads:\n  - "contigs.fa"\n')
    out = str(tmp_path / "g.gfa")
    subprocess.check_call([GB, str(y), out, "-k", "21", "-t", "3", "-c", "--gfa"], stdout=subprocess.DEVNULL)
    assert open(out).read() == open(os.path.join(GOLDEN, "graphcov_small_k21_t3.gfa")).read()
    wd = tmp_path / "w"
    wd.mkdir()
    subprocess.check_call([KC, "-k", "21", "-w", str(wd), "-d", str(y)], stdout=subprocess.DEVNULL)
    ref, _ = oracle.count(reads + ["ACGTTGCATTGACCAGT" * 8], 21, "A", 16)  # kmercount takes every library of the dataset
    assert open(wd / "final_kmers", "rb").read() == ref.tobytes()


def test_bgzf_blocks_followed_by_an_ordinary_gzip_member(tmp_path):
    """`cat a.bgzf.gz b.gz` starts like a BGZF file (the block-parallel reader takes it, tools/bgzf_reader.hpp) and goes on as an ordinary
    gzip stream; gzread read such files, the round-4 reader refused them with the format error. The reader hands over to zlib at the boundary of
    the last good block: the k-mer file is the golden of the plain input."""
    from test_bgzf_cpu import bgzf_bytes
    c = [c for c in load_manifest()["cases"] if c["kind"] == "count" and c.get("file") and c["mode"] == "A" and c["num_buckets"] == 16 and c["K"] == 21][0]
    reads = [r for r in read_lines("reads_tiny.txt") if r]
    half = len(reads) // 2
    fq = lambda rs, o: "".join(f"@r{o + i}\n{r}\n+\n{'I' * len(r)}\n" for i, r in enumerate(rs)).encode()
    mixed = str(tmp_path / "mixed.fq.gz")
    with open(mixed, "wb") as f:
        f.write(bgzf_bytes(fq(reads[:half], 0), block=3000)[:-28])  # (without the empty end-of-file block: the file goes on)
        f.write(gzip.compress(fq(reads[half:], half)))
    wd = tmp_path / "w"
    wd.mkdir()
    subprocess.check_call([KC, "-k", "21", "-t", "3", "-w", str(wd), mixed], stdout=subprocess.DEVNULL)
    assert open(wd / "final_kmers", "rb").read() == open(os.path.join(GOLDEN, c["file"]), "rb").read()
