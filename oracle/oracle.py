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
    srcs = [os.path.join(_HERE, f) for f in ("smx_oracle.c", "smx_oracle_graph.c", "smx_oracle.h")]
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libsmx_oracle.so"])
    return _LIB


_lib = None


class OrcGraph(C.Structure):
    _fields_ = [("n_kpomers", C.c_uint64), ("n_kmers", C.c_uint64), ("kmers", C.POINTER(C.c_uint64)),
                ("masks", C.POINTER(C.c_uint8)), ("n_unitigs", C.c_uint64), ("n_loops", C.c_uint64),
                ("unitig_off", C.POINTER(C.c_uint64)), ("unitig_seq", C.c_char_p), ("n_vertices", C.c_uint64),
                ("n_links", C.c_uint64), ("gfa", C.c_char_p), ("gfa_len", C.c_uint64)]


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
        _lib.orc_build_graph.restype = C.POINTER(OrcGraph)
        _lib.orc_build_graph.argtypes = [C.c_uint, C.c_uint, C.c_char_p, u64p, C.c_uint64, C.c_char_p]
        _lib.orc_graph_free.argtypes = [C.POINTER(OrcGraph)]
        _lib.orc_build_graph_cov.restype = C.POINTER(OrcGraph)
        _lib.orc_build_graph_cov.argtypes = [C.c_uint, C.c_uint, C.c_char_p, u64p, C.c_uint64, C.c_char_p, C.c_int]
        _lib.orc_build_graph_ex.restype = C.POINTER(OrcGraph)
        _lib.orc_build_graph_ex.argtypes = [C.c_uint, C.c_uint, C.c_char_p, u64p, C.c_uint64, C.c_char_p, C.c_int, C.c_int, C.c_int]
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


def build_graph(reads: Sequence[str], k: int, num_buckets: int, flavour_version: str = "SPAdes-4.3.0-dev", coverage: bool = False,
                sort_edges: bool = False, keep_loops: bool = True, early_tip_bound: int = 0, early_at: bool = False, coverage_reads: int = 0) -> dict:
    """spades-gbuilder restated: -> dict(kmers, masks, unitigs (list of str, reference order), n_loops, gfa (str), ...)."""
    bases, off = concat_reads(reads)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    lib().orc_set_early_tip_bound(C.c_uint64(early_tip_bound))
    lib().orc_set_early_at_remover(1 if early_at else 0)
    lib().orc_set_coverage_reads(C.c_uint64(coverage_reads))
    g = lib().orc_build_graph_ex(k, num_buckets, bases, off.ctypes.data_as(C.POINTER(C.c_uint64)), len(off) - 1,
                                 flavour_version.encode(), 1 if coverage else 0, 1 if sort_edges else 0, 1 if keep_loops else 0)
    gc = g.contents
    nw = words(k)
    nk = gc.n_kmers
    kmers = np.ctypeslib.as_array(gc.kmers, shape=(max(nk, 1) * nw,))[: nk * nw].copy().reshape(nk, nw)
    masks = np.ctypeslib.as_array(gc.masks, shape=(max(nk, 1),))[:nk].copy()
    nu = gc.n_unitigs
    uoff = np.ctypeslib.as_array(gc.unitig_off, shape=(nu + 1,)).copy()
    seq = gc.unitig_seq or b""
    unitigs = [seq[int(uoff[i]):int(uoff[i + 1])].decode() for i in range(nu)]
    res = dict(n_kpomers=int(gc.n_kpomers), kmers=kmers, masks=masks, unitigs=unitigs, n_loops=int(gc.n_loops),
               n_vertices=int(gc.n_vertices), n_links=int(gc.n_links), gfa=(gc.gfa or b"").decode())
    lib().orc_graph_free(g)
    lib().orc_set_early_tip_bound(C.c_uint64(0))
    lib().orc_set_early_at_remover(0)
    lib().orc_set_coverage_reads(C.c_uint64(0))
    return res
