"""oracle/oracle.py — ctypes view of the CPU restatement (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (spades_amd/) never does.
"""
import ctypes as C
import os
import subprocess
from typing import List, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libsmx_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "smx_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libsmx_oracle.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        u64p = C.POINTER(C.c_uint64)
        _lib.orc_words.restype = C.c_uint
        _lib.orc_xxh3_64.restype = C.c_uint64
        _lib.orc_xxh3_64.argtypes = [C.c_void_p, C.c_size_t]
        _lib.orc_bucket.restype = C.c_uint64
        _lib.orc_bucket.argtypes = [u64p, C.c_uint, C.c_uint64]
        _lib.orc_from_string.argtypes = [u64p, C.c_uint, C.c_char_p]
        _lib.orc_to_string.argtypes = [u64p, C.c_uint, C.c_char_p]
        _lib.orc_shl.argtypes = [u64p, C.c_uint, C.c_uint]
        _lib.orc_rc.argtypes = [u64p, C.c_uint, u64p]
        _lib.orc_is_minimal.argtypes = [u64p, C.c_uint]
        _lib.orc_less_nucl.argtypes = [u64p, u64p, C.c_uint]
        _lib.orc_count.restype = C.c_int64
        _lib.orc_count.argtypes = [C.c_char, C.c_uint, C.c_uint, C.c_char_p, u64p, C.c_uint64,
                                   C.POINTER(u64p), u64p]
        _lib.orc_free.argtypes = [C.c_void_p]
        _lib.orc_longest_valid.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    return _lib


def words(K: int) -> int:
    return (K + 31) // 32


def kmer_from_string(s: str) -> np.ndarray:
    w = (C.c_uint64 * 4)()
    lib().orc_from_string(w, len(s), s.encode())
    return np.array(w[: words(len(s))], dtype=np.uint64)


def _w4(rec: Sequence[int]):
    w = (C.c_uint64 * 4)()
    for i, v in enumerate(rec):
        w[i] = int(v)
    return w


def kmer_to_string(rec: Sequence[int], K: int) -> str:
    buf = C.create_string_buffer(K + 1)
    lib().orc_to_string(_w4(rec), K, buf)
    return buf.value.decode()


def rc(rec: Sequence[int], K: int) -> np.ndarray:
    out = (C.c_uint64 * 4)()
    lib().orc_rc(_w4(rec), K, out)
    return np.array(out[: words(K)], dtype=np.uint64)


def is_minimal(rec: Sequence[int], K: int) -> bool:
    return bool(lib().orc_is_minimal(_w4(rec), K))


def xxh3_64(data: bytes) -> int:
    return int(lib().orc_xxh3_64(data, len(data)))


def bucket(rec: Sequence[int], K: int, num_buckets: int) -> int:
    return int(lib().orc_bucket(_w4(rec), K, num_buckets))


def longest_valid(s: str) -> Tuple[int, int]:
    a, b = C.c_size_t(), C.c_size_t()
    lib().orc_longest_valid(s.encode(), len(s), C.byref(a), C.byref(b))
    return a.value, b.value


def concat_reads(reads: Sequence[str]) -> Tuple[bytes, np.ndarray]:
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    if len(reads):
        off[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
    return "".join(reads).encode(), off


def count(reads: Sequence[str], K: int, mode: str = "A", num_buckets: int = 16) -> Tuple[np.ndarray, np.ndarray]:
    """-> (records [n, words(K)] uint64 in final_kmers order, bucket_sizes [num_buckets])."""
    bases, off = concat_reads(reads)
    return count_raw(bases, off, K, mode, num_buckets)


def count_raw(bases: bytes, off: np.ndarray, K: int, mode: str = "A", num_buckets: int = 16):
    off = np.ascontiguousarray(off, dtype=np.uint64)
    out = C.POINTER(C.c_uint64)()
    sizes = np.zeros(num_buckets, dtype=np.uint64)
    n = lib().orc_count(mode.encode(), K, num_buckets, bases, off.ctypes.data_as(C.POINTER(C.c_uint64)),
                        len(off) - 1, C.byref(out), sizes.ctypes.data_as(C.POINTER(C.c_uint64)))
    nw = words(K)
    rec = np.ctypeslib.as_array(out, shape=(max(n, 1) * nw,))[: n * nw].copy().reshape(n, nw)
    lib().orc_free(out)
    return rec, sizes
