"""The sink behind smx_write_final_kmers and the device GFA writer (spades_amd/csrc/smx_file_sink.hpp, host-only, compiled here with g++ as
it is): a tmpfs output of known size is mapped and filled by several threads (pwrite on tmpfs allocates under the inode lock: one
writer thread was 80 % of spades-kmercount-mi355x's wall time), everything else goes through pwrite. Whatever the mode, the file must
hold the bytes."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "host_shims", "file_sink_shim.cpp")
SHM = "/dev/shm"


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("sink") / "libsink.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-shared", "-fPIC", "-pthread", "-o", so, SRC])
    l = ctypes.CDLL(so)
    l.file_sink_write.restype = ctypes.c_int
    l.file_sink_write.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64,
                                  ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint64, ctypes.c_int]
    return l


def _write(lib, path, head, data, block, rng, rdwr=1):
    nblocks = (len(data) + block - 1) // block
    order = rng.permutation(nblocks).astype(np.uint64)  # (the two-strand writer sends buckets, i.e. offsets, in any order)
    return lib.file_sink_write(path.encode(), head, len(head), data.tobytes(), len(data), block, order.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), nblocks, rdwr)


def _is_tmpfs(d):
    try:
        with open("/proc/mounts") as f:
            return any(l.split()[1] == d and l.split()[2] == "tmpfs" for l in f)
    except OSError:
        return False


@pytest.mark.parametrize("force,threads", [(None, None), ("0", None), ("1", "1"), ("1", "3"), ("1", "8")])
@pytest.mark.parametrize("where", ["tmp", "shm"])
def test_the_file_holds_the_bytes_in_every_mode(lib, tmp_path, monkeypatch, where, force, threads):
    if where == "shm" and not (os.path.isdir(SHM) and os.access(SHM, os.W_OK)):
        pytest.skip("no /dev/shm")
    monkeypatch.delenv("SMX_WRITE_MMAP", raising=False)
    monkeypatch.delenv("SMX_WRITE_THREADS", raising=False)
    if force is not None:
        monkeypatch.setenv("SMX_WRITE_MMAP", force)
    if threads is not None:
        monkeypatch.setenv("SMX_WRITE_THREADS", threads)
    rng = np.random.default_rng(3)
    path = os.path.join(SHM, f"smx_sink_test_{os.getpid()}.bin") if where == "shm" else str(tmp_path / "out.bin")
    try:
        for n, block in ((0, 4096), (1, 4096), (5_000_001, 1 << 20), (9 << 20, 3 << 20)):
            head = b"H\tsp:Z:test\n" if n % 2 else b""
            data = rng.integers(0, 256, size=n, dtype=np.uint8)
            mode = _write(lib, path, head, data, block, rng)
            assert mode >= 0
            if force == "0" or force is None:  # (round 5: pwrite unless asked)
                assert mode == 0
            if force == "1" and n + len(head) > 0:
                assert mode == 1
            with open(path, "rb") as f:
                got = f.read()
            assert got == head + data.tobytes()
    finally:
        if os.path.exists(path):
            os.unlink(path)


def test_a_large_tmpfs_output_is_mapped_by_itself_and_a_write_only_descriptor_falls_back(lib, monkeypatch):
    if not (_is_tmpfs(SHM) and os.access(SHM, os.W_OK)):
        pytest.skip("/dev/shm is not a tmpfs here")
    monkeypatch.setenv("SMX_WRITE_MMAP", "-1")  # "decide by the file system" (round 5: opt-in; with nothing set every output goes through pwrite)
    rng = np.random.default_rng(4)
    n = (64 << 20) + 12345
    data = rng.integers(0, 256, size=n, dtype=np.uint8)
    path = os.path.join(SHM, f"smx_sink_big_{os.getpid()}.bin")
    try:
        for rdwr, expect in ((1, 1), (0, 0)):
            mode = _write(lib, path, b"", data, 16 << 20, rng, rdwr)
            assert mode == expect
            with open(path, "rb") as f:
                assert f.read() == data.tobytes()
    finally:
        if os.path.exists(path):
            os.unlink(path)
