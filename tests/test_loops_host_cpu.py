"""Perfect loops on the host (spades_amd/csrc/smx_loops_host.hpp: packed k-mers, hash index + successor array + per-loop work on all
cores) against the string-level collector that rounds 1-3 shipped and pinned to the real spades-gbuilder's loop goldens
(tests/host_shims/loops_string_ref.hpp; CollectLoops, debruijn_graph_constructor.hpp:252-293,359-397). Both are compiled here with g++
as they are. Cases: many loops at once in a shuffled file order, loops shorter than k, length 1 and 2, hairpins (a loop that is its own
reverse complement: split at its first palindromic (k+1)-mer), even k with palindromic k-mers, every record width."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "host_shims", "loops_shim.cpp")
COMP = str.maketrans("ACGT", "TGCA")


def rc(s):
    return s.translate(COMP)[::-1]


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("loops") / "libloops.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-shared", "-fPIC", "-pthread", "-o", so, SRC])
    l = ctypes.CDLL(so)
    u64p, u8p = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint8)
    l.loops_packed.restype = ctypes.c_int
    l.loops_packed.argtypes = [u64p, u64p, u8p, ctypes.c_uint64, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint64, u64p, u64p, u64p, u8p, ctypes.c_uint64, ctypes.c_char_p,
                               ctypes.c_uint64]
    l.loops_string.restype = ctypes.c_int
    l.loops_string.argtypes = [u64p, u64p, u8p, ctypes.c_uint64, ctypes.c_uint, u64p, u64p, u64p, u8p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_uint64]
    return l


def loop_set(circles, k):
    """canonical k-mer -> InOutMask of the (k+1)-mers of the circular sequences (both strands); None unless every k-mer is a non-junction one"""
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    masks = {}

    def add(x, bit_if_canonical, bit_if_not):
        r = rc(x)
        if x <= r:
            masks[x] = masks.get(x, 0) | bit_if_canonical
        else:
            masks[r] = masks.get(r, 0) | bit_if_not

    for s in circles:
        G = len(s)
        ext = s * ((k + 1) // G + 2)
        for p in range(G):
            e = ext[p:p + k + 1]
            c, b = code[e[k]], code[e[0]]
            add(e[:k], 1 << c, 1 << (4 + (3 - c)))      # out extension of the prefix k-mer
            add(e[1:], 1 << (4 + b), 1 << (3 - b))      # in extension of the suffix k-mer
    for m in masks.values():
        o, i = m & 15, m >> 4
        if o & (o - 1) or i & (i - 1) or not o or not i:
            return None
    return masks


def run(lib, masks, k, rng, threads):
    kmers = list(masks)
    rng.shuffle(kmers)  # the k-mer file is in hash order: any order may come
    n, nw = len(kmers), (k + 31) // 32
    packed = np.zeros((n, nw), np.uint64)
    for t, x in enumerate(kmers):
        for j, ch in enumerate(x):
            packed[t, j >> 5] |= np.uint64("ACGT".index(ch)) << np.uint64((j & 31) << 1)
    ranks = np.sort(rng.choice(10 * n + 10, size=n, replace=False)).astype(np.uint64)
    mk = np.array([masks[x] for x in kmers], np.uint8)
    u64p, u8p = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint8)
    out = []
    for which in ("packed", "string"):
        cap = 2 * n + 2
        lens, starts, ends = (np.zeros(cap, np.uint64) for _ in range(3))
        selfs = np.zeros(cap, np.uint8)
        text = ctypes.create_string_buffer(4 * (n + 2) * 2 + 2 * cap * (k + 1))
        args = [packed.ctypes.data_as(u64p), ranks.ctypes.data_as(u64p), mk.ctypes.data_as(u8p), n, k]
        tail = [lens.ctypes.data_as(u64p), starts.ctypes.data_as(u64p), ends.ctypes.data_as(u64p), selfs.ctypes.data_as(u8p), cap, text, len(text)]
        nl = lib.loops_packed(*args, threads, 4, *tail) if which == "packed" else lib.loops_string(*args, *tail)
        assert nl >= 0, (which, nl)
        seqs, at = [], 0
        for i in range(nl):
            seqs.append(text.raw[at:at + int(lens[i])].decode())
            at += int(lens[i])
        out.append((seqs, starts[:nl].tolist(), ends[:nl].tolist(), selfs[:nl].tolist()))
    return out


def circles_for(kind, k, rng):
    rnd = lambda n: "".join(rng.choice(list("ACGT"), n))
    if kind == "many":
        return [rnd(int(rng.integers(1, 4 * k + 40))) for _ in range(int(rng.integers(1, 12)))]
    if kind == "tiny":
        return [rnd(int(rng.integers(1, 6))) for _ in range(int(rng.integers(1, 4)))]
    if kind == "hairpin":  # X + RC(X) closed to a circle is its own reverse complement
        out = []
        for _ in range(int(rng.integers(1, 4))):
            x = rnd(int(rng.integers(1, 3 * k + 10)))
            out.append(x + rc(x))
        return out + ([rnd(3 * k)] if rng.random() < 0.5 else [])
    if kind == "simple":
        return [rng.choice(["A", "AT", "AC", "ACGT", "AATT", "ACG"])]
    raise AssertionError(kind)


@pytest.mark.parametrize("k", [3, 4, 5, 6, 21, 31, 32, 33, 55, 63, 64, 65, 77, 127, 128])
@pytest.mark.parametrize("kind", ["many", "tiny", "hairpin", "simple"])
def test_packed_collector_equals_the_string_collector(lib, k, kind):
    if kind == "hairpin" and k % 2 == 0:
        pytest.skip("a hairpin turns around in a palindromic k-mer when k is even: no perfect loop (SPAdes takes odd k only)")
    rng = np.random.default_rng(k * 100 + len(kind))
    done = with_split = with_self = 0
    for _ in range(400):
        if done >= 12:
            break
        masks = loop_set(circles_for(kind, k, rng), k)
        if not masks:
            continue  # some k-mer recurs with another neighbour: not a set of perfect loops
        if k % 2 == 0 and kind != "simple" and any(x == rc(x) for x in masks):
            continue  # even k (SPAdes takes odd k only): a palindromic k-mer joins the two strands of a circle, which is then no cycle
        a, b = run(lib, masks, k, rng, threads=int(rng.integers(1, 5)))
        assert a == b
        assert sum(len(s) - k for s in a[0]) >= len(masks) and len(a[0]) >= 1
        with_split += any(len(s) == k + 1 for s in a[0])
        with_self += any(a[3])
        done += 1
    assert done >= (3 if kind == "simple" or k < 6 else 8), done
    if kind == "hairpin" and k % 2 == 1 and k >= 5:
        assert with_split, "no hairpin was split at a palindromic (k+1)-mer"


def test_a_set_that_is_not_closed_is_reported(lib):
    k = 21
    rng = np.random.default_rng(5)
    masks = loop_set(["".join(rng.choice(list("ACGT"), 200))], k)
    masks.pop(next(iter(masks)))  # a k-mer of the loop is missing
    kmers = list(masks)
    n, nw = len(kmers), 1
    packed = np.zeros((n, nw), np.uint64)
    for t, x in enumerate(kmers):
        for j, ch in enumerate(x):
            packed[t, 0] |= np.uint64("ACGT".index(ch)) << np.uint64(j << 1)
    ranks = np.arange(n, dtype=np.uint64)
    mk = np.array([masks[x] for x in kmers], np.uint8)
    u64p, u8p = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint8)
    z = np.zeros(2 * n + 2, np.uint64)
    z8 = np.zeros(2 * n + 2, np.uint8)
    text = ctypes.create_string_buffer(1 << 16)
    rcv = lib.loops_packed(packed.ctypes.data_as(u64p), ranks.ctypes.data_as(u64p), mk.ctypes.data_as(u8p), n, k, 2, 4, z.ctypes.data_as(u64p), z.ctypes.data_as(u64p),
                           z.ctypes.data_as(u64p), z8.ctypes.data_as(u8p), len(z), text, len(text))
    assert rcv < 0
