"""Out-of-core merge: the planner that cuts ONE bucket's slices of the spilled runs by key range when the bucket alone exceeds what the
HBM budget can merge (spades_amd/csrc/smx_spill_split.hpp — host-only code, compiled here with g++ as it is). Properties the device merge
relies on: parts are disjoint key intervals in ascending order, copies of a key share their part, every part fits, nothing is lost."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "host_shims", "spill_split_shim.cpp")
HDR = os.path.join(ROOT, "spades_amd", "csrc", "smx_spill_split.hpp")


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("split") / "libsplit.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-shared", "-fPIC", "-o", so, SRC])
    l = ctypes.CDLL(so)
    l.spill_split_plan.restype = ctypes.c_int
    l.spill_split_plan.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint, ctypes.c_uint, ctypes.c_uint64,
                                   ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint]
    return l


def _plan(lib, runs, nw, max_part, cap=1 << 16):
    R = len(runs)
    ptrs = (ctypes.c_void_p * R)(*[r.ctypes.data if len(r) else None for r in runs])
    ns = (ctypes.c_uint64 * R)(*[len(r) for r in runs])
    out = np.zeros((cap, R), np.uint64)
    n = lib.spill_split_plan(ptrs, ns, R, nw, max_part, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), cap)
    assert n >= 2
    return out[:n].astype(np.int64)


def _sorted_unique(a):
    """records (n, nw) in the library's order: word 0 most significant"""
    if len(a) == 0:
        return a
    a = np.unique(a, axis=0)  # lexicographic by columns = word 0 first
    return np.ascontiguousarray(a)


def _runs(rng, nruns, nw, n, kind):
    runs = []
    pool = rng.integers(0, 1 << 63, size=(max(n, 1) * 2, nw), dtype=np.uint64)
    if kind == "dupes":  # most keys in every run (30x data: the true k-mers recur in every batch)
        pool = pool[: max(n, 1)]
    if kind == "skew":  # word 0 nearly constant: the cut has to look at the later words
        pool[:, 0] = pool[:, 0] % 3
    if kind == "narrow":  # few distinct keys at all
        pool = pool[:5]
    for r in range(nruns):
        m = int(rng.integers(0, n + 1)) if kind != "dupes" else n
        idx = rng.integers(0, len(pool), size=m)
        runs.append(_sorted_unique(pool[idx]))
    return runs


@pytest.mark.parametrize("kind", ["plain", "dupes", "skew", "narrow"])
@pytest.mark.parametrize("nw", [1, 2, 4])
@pytest.mark.parametrize("nruns,n,max_part", [(1, 1000, 100), (3, 5000, 700), (8, 2000, 64), (5, 300, 10_000), (6, 50, 1), (4, 0, 10)])
def test_parts_are_disjoint_key_intervals_that_fit(lib, kind, nw, nruns, n, max_part):
    rng = np.random.default_rng(nw * 1000 + nruns * 10 + len(kind))
    runs = _runs(rng, nruns, nw, n, kind)
    cuts = _plan(lib, runs, nw, max_part)
    R = len(runs)
    assert (cuts[0] == 0).all() and (cuts[-1] == [len(r) for r in runs]).all()
    assert (np.diff(cuts, axis=0) >= 0).all()
    sizes = np.diff(cuts, axis=0).sum(axis=1)
    assert (sizes <= max(max_part, R)).all(), (sizes.max(), max_part)
    if sizes.sum() > 4 * max_part and max_part >= 16 * R:  # joined leaves: no confetti
        assert len(sizes) <= 2 * (sizes.sum() // max_part + 1)
    merged = []
    prev_last = None
    for p in range(len(cuts) - 1):
        part = [runs[r][cuts[p][r]:cuts[p + 1][r]] for r in range(R)]
        cat = np.concatenate(part) if part else np.zeros((0, nw), np.uint64)
        u = _sorted_unique(cat)
        if len(u):
            if prev_last is not None:
                assert tuple(prev_last) < tuple(u[0]), "parts overlap or are out of order"
            prev_last = u[-1]
        merged.append(u)
    got = np.concatenate(merged) if merged else np.zeros((0, nw), np.uint64)
    want = _sorted_unique(np.concatenate(runs)) if sum(len(r) for r in runs) else np.zeros((0, nw), np.uint64)
    assert got.shape == want.shape and (got == want).all()
