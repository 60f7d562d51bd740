"""CPU: pins oracle/ (the C restatement) against the reference's golden vectors.

Sources of truth: the worked byte example in kmercount.cpp:160-170, python-xxhash
(libxxhash 0.8.x, same algorithm as ext/include/xxh/xxhash.h), and tests/golden/manifest.json made by
the REAL reference classes (oracle/_ref/ref_kmercount) + the spades-kmercount binary.
"""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_manifest, read_lines
from oracle import oracle

CASES = [c for c in load_manifest()["cases"] if c["kind"] == "count"]


def test_worked_byte_example():
    # kmercount.cpp:160-170 / docs/standalone.md:12-24: AGCTCT -> d8 0d 00 00 00 00 00 00
    rec = oracle.kmer_from_string("AGCTCT")
    assert rec.tobytes() == bytes([0xD8, 0x0D, 0, 0, 0, 0, 0, 0])


def test_xxh3_against_libxxhash():
    xxhash = pytest.importorskip("xxhash")
    rng = np.random.default_rng(5)
    for n in (8, 16, 24, 32):
        for _ in range(200):
            d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
            assert oracle.xxh3_64(d) == xxhash.xxh3_64_intdigest(d)


def test_xxh3_known_answers():
    # frozen from libxxhash 0.8.2 so the check also runs where python-xxhash is absent
    d = bytes(range(7, 7 + 32))
    assert [oracle.xxh3_64(d[:n]) for n in (8, 16, 24, 32)] == [
        0x883FDB8A2206C423, 0x81AC97DB10449AEC, 0xBC592C693060098F, 0x4527E79322C63EC1]


def test_rc_involution_and_minimal():
    rng = np.random.default_rng(7)
    for K in (1, 2, 5, 21, 31, 32, 33, 55, 63, 64, 65, 96, 97, 127, 128):
        for _ in range(50):
            s = "".join(rng.choice(list("ACGT"), K))
            rec = oracle.kmer_from_string(s)
            r = oracle.rc(rec, K)
            comp = s[::-1].translate(str.maketrans("ACGT", "TGCA"))
            assert oracle.kmer_to_string(r, K) == comp
            assert (oracle.rc(r, K) == rec).all()
            assert oracle.is_minimal(rec, K) == (s <= comp)


def test_longest_valid_rule():
    assert oracle.longest_valid("ACGTNACGTAC") == (5, 11)
    assert oracle.longest_valid("ACGTNACGT") == (0, 4)  # first one on ties
    assert oracle.longest_valid("NNNN") == (0, 0)
    assert oracle.longest_valid("") == (0, 0)
    assert oracle.longest_valid("acgtXAC") == (0, 4)


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c['reads'][6:-4]}-{c['mode']}{c['K']}-b{c['num_buckets']}")
def test_count_matches_reference_golden(case):
    reads = read_lines(case["reads"])
    rec, sizes = oracle.count(reads, case["K"], case["mode"], case["num_buckets"])
    assert list(map(int, sizes)) == case["bucket_sizes"]
    assert len(rec) == case["n_records"]
    assert hashlib.md5(rec.tobytes()).hexdigest() == case["md5"]
    if "file" in case:
        assert rec.tobytes() == open(os.path.join(GOLDEN, case["file"]), "rb").read()


def test_output_invariants():
    # bucket nondecreasing, strictly increasing (w0,w1,..) inside a bucket, closed under RC (mode A)
    reads = read_lines("reads_small.txt")
    K = 55
    rec, sizes = oracle.count(reads, K, "A", 16)
    b = np.array([oracle.bucket(r, K, 16) for r in rec])
    assert (np.diff(b) >= 0).all()
    assert (np.bincount(b, minlength=16) == sizes).all()
    for i in range(1, len(rec)):
        if b[i] == b[i - 1]:
            assert tuple(rec[i - 1]) < tuple(rec[i])
    s = {tuple(r) for r in rec.tolist()}
    assert all(tuple(oracle.rc(r, K).tolist()) in s for r in rec[:500])
