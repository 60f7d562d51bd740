"""CPU: the byte-range splitter of spades-gbuilder-mi355x --gpus N (spades_amd/tools/fastq_split.hpp): the parts are disjoint, cover the
file and start at records — also where quality lines begin with '@' or '+'. The header has no dependencies and is compiled alone."""
import os
import random
import subprocess

from conftest import ROOT

DRIVER = r"""
#include "%s/spades_amd/tools/fastq_split.hpp"
#include <cstdlib>
int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const std::string path = argv[1];
    const unsigned parts = (unsigned)atoi(argv[2]);
    printf("fourline=%%d\n", (int)smxtool::fastq_head_is_four_line(path));
    for (unsigned p = 0; p < parts; ++p) {
        long long b = 0, e = 0;
        if (!smxtool::fastq_part_range(path, p, parts, &b, &e)) return 1;
        printf("%%lld %%lld\n", b, e);
    }
    return 0;
}
"""


def _build(tmp_path):
    src, exe = tmp_path / "drv.cpp", tmp_path / "drv"
    src.write_text(DRIVER % ROOT)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", str(exe), str(src)])
    return str(exe)


def _records(rng, n, lengths):
    recs = []
    for i in range(n):
        ln = rng.choice(lengths)
        seq = "".join(rng.choice("ACGTN") for _ in range(ln))
        qual = "".join(rng.choice("@+I#5") for _ in range(ln))  # quality lines that look like header / separator lines
        recs.append(f"@r{i} x\n{seq}\n+{'r%d' % i if i % 3 == 0 else ''}\n{qual}\n")
    return recs


def test_parts_are_disjoint_cover_the_file_and_start_at_records(tmp_path):
    exe = _build(tmp_path)
    rng = random.Random(5)
    fq = str(tmp_path / "t.fq")
    for trial in range(24):
        recs = _records(rng, rng.choice([1, 2, 3, 7, 50, 400]), [0, 1, 5, 30, 150, 151])
        text = "".join(recs)
        if trial % 4 == 3:
            text = text[:-1]  # no newline at the end of the file
        with open(fq, "w") as f:
            f.write(text)
        starts, o = set(), 0
        for r in recs:
            starts.add(o)
            o += len(r)
        for parts in (1, 2, 3, 5, 16, 64):
            out = subprocess.check_output([exe, fq, str(parts)]).decode().split("\n")
            assert out[0] == "fourline=1"
            rng_ = [tuple(map(int, ln.split())) for ln in out[1:] if ln]
            assert len(rng_) == parts and rng_[0][0] == 0 and rng_[-1][1] == len(text)
            for (b, e), (b2, _) in zip(rng_, rng_[1:]):
                assert e == b2
            for b, e in rng_:
                assert b <= e and (b in starts or b == len(text))


def test_long_records_cross_the_search_window(tmp_path):
    exe = _build(tmp_path)
    rng = random.Random(9)
    recs = _records(rng, 6, [1_500_000, 700_000, 10])  # lines longer than the splitter's first window of 1 MiB
    text = "".join(recs)
    fq = str(tmp_path / "long.fq")
    with open(fq, "w") as f:
        f.write(text)
    starts, o = set(), 0
    for r in recs:
        starts.add(o)
        o += len(r)
    out = subprocess.check_output([exe, fq, "5"]).decode().split("\n")
    rng_ = [tuple(map(int, ln.split())) for ln in out[1:] if ln]
    assert rng_[0][0] == 0 and rng_[-1][1] == len(text)
    for (b, e), (b2, _) in zip(rng_, rng_[1:]):
        assert e == b2
    assert all(b in starts or b == len(text) for b, _ in rng_)


def test_other_formats_are_not_split_by_bytes(tmp_path):
    exe = _build(tmp_path)
    for name, text in (("ml.fq", "@a\nACGT\nACGT\n+\nIIII\nIIII\n"), ("a.fa", ">a\nACGT\n"), ("bad.fq", "@a\nACGT\n+\nII\n"), ("empty.fq", "")):
        p = str(tmp_path / name)
        with open(p, "w") as f:
            f.write(text)
        assert subprocess.check_output([exe, p, "1"]).decode().split("\n")[0] == "fourline=0"
