"""Host-side mirror of the reference's k-mer counting interface, backed by libspades_mi355x.so.

Names and argument meaning follow the reference so that its tests read the same:
  ReadKMerSplitter   ~ ParallelSortingSplitter (projects/spades_tools/kmercount.cpp:48-122, mode 'A') and
                       DeBruijnReadKMerSplitter<..., StoringTypeFilter<InvertableStoring>>
                       (common/kmer_index/kmer_mph/kmer_splitters.hpp:94-136, mode 'B')
  KMerDiskCounter    ~ kmers::KMerDiskCounter<RtSeq>   (kmer_mph/kmer_index_builder.hpp:284-431)
  KMerDiskStorage    ~ kmers::KMerDiskStorage<RtSeq>   (kmer_mph/kmer_index_builder.hpp:48-256)
The "disk" in the names is historical: buckets live in HBM; files are written only on request
(final_kmers(), bucket_file()) in the reference's exact byte layout.
"""
import ctypes as C
import os
from typing import Iterable, List, Optional, Sequence

import numpy as np

from . import _lib


class SmxError(RuntimeError):
    """Raised with the reference's exit code (common/utils/logger/error_codes.hpp:14-20) in .code."""

    def __init__(self, code: int, msg: str):
        super().__init__(f"[{code}] {msg}")
        self.code = code


def _chk(ctx, rc):
    if rc != 0:
        raise SmxError(rc, (_lib.load().smx_last_error(ctx) or b"").decode())


class Context:
    """Owns one smx_ctx (one GPU)."""

    def __init__(self, device: int = 0, hbm_budget: int = 0):
        lib = _lib.load()
        h = C.c_void_p()
        rc = lib.smx_create(C.byref(h), device, hbm_budget)
        if rc != 0:
            raise SmxError(rc, "smx_create failed: no usable MI355X / HIP device (spades_amd has no CPU fallback)")
        self._h = h
        self.lib = lib

    def reads_info(self):
        n, b = C.c_uint64(), C.c_uint64()
        _chk(self._h, self.lib.smx_reads_info(self._h, C.byref(n), C.byref(b)))
        return n.value, b.value

    def set_option(self, key: str, value: int):
        _chk(self._h, self.lib.smx_set_option(self._h, key.encode(), int(value)))
        if key == "async_upload":
            self.async_upload = int(value) > 0  # (push_back_packed keeps its host arrays alive only then)

    def graph_clear(self):
        """smx_graph_clear: the graph of the last smx_build_graph gives its HBM back; reads stay resident"""
        _chk(self._h, self.lib.smx_graph_clear(self._h))

    def bucket_sizes(self, num_buckets: int):
        """smx_bucket_sizes of the context's count-result view (after a construction on the default route: makes the sorted k-mer file first)"""
        sizes = (C.c_uint64 * num_buckets)()
        _chk(self._h, self.lib.smx_bucket_sizes(self._h, sizes))
        return [int(v) for v in sizes]

    def prewarm(self, bottom_bytes: int, top_bytes: int) -> None:
        """smx_prewarm: the arena's helper thread maps that much physical memory while the caller goes on"""
        _chk(self._h, self.lib.smx_prewarm(self._h, bottom_bytes, top_bytes))

    def arena_free_bytes(self) -> int:
        """smx_arena_free_bytes: HBM the context holds but does not use right now (invisible to hipMemGetInfo: the arena only grows)"""
        n = C.c_size_t()
        _chk(self._h, self.lib.smx_arena_free_bytes(self._h, C.byref(n)))
        return int(n.value)

    def trim(self) -> int:
        """smx_trim: unused device memory back to the device; returns the bytes (0 with the default arena: it only grows, include/smx.h)"""
        n = C.c_size_t()
        _chk(self._h, self.lib.smx_trim(self._h, C.byref(n)))
        return int(n.value)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.smx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def timings(self):
        names = (C.c_char_p * 128)()
        ms = (C.c_float * 128)()
        n = self.lib.smx_last_timings(self._h, names, ms, 128)
        return [(names[i].decode(), float(ms[i])) for i in range(min(n, 128))]


class ReadKMerSplitter:
    """Collects the reads whose K-mers are to be counted.

    mode 'A' = every K-mer of read and RC(read) (spades-kmercount); mode 'B' = only K-mers with
    IsMinimal() (construction). K-mers are never materialised on the host.
    """

    def __init__(self, K: int, mode: str = "A", ctx: Optional[Context] = None):
        if mode not in ("A", "B"):
            raise SmxError(_lib.INVALID_PARAMETER, f"bad mode {mode!r}")
        self.K_ = int(K)
        self.mode = mode
        self.ctx = ctx or Context()

    def K(self) -> int:
        return self.K_

    def kmer_size(self) -> int:  # KMerSplitter::kmer_size, kmer_splitter.hpp:40-42
        return 8 * ((self.K_ + 31) // 32)

    def push_back_reads(self, reads: Sequence[str]):
        """ASCII reads (may contain N / lower case): the reference's longest-valid rule is applied."""
        off = np.zeros(len(reads) + 1, dtype=np.uint64)
        if len(reads):
            off[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
        bases = "".join(reads).encode()
        self.push_back_ascii(bases, off)

    def push_back_ascii(self, bases: bytes, offsets: np.ndarray):
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        _chk(self.ctx._h, self.ctx.lib.smx_submit_reads_ascii(
            self.ctx._h, bases, offsets.ctypes.data_as(C.POINTER(C.c_uint64)), len(offsets) - 1))

    def push_back_packed(self, words: np.ndarray, start: np.ndarray, length: np.ndarray):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        start = np.ascontiguousarray(start, dtype=np.uint64)
        length = np.ascontiguousarray(length, dtype=np.uint32)
        # option "async_upload": the call returns before the copy is done — the arrays stay referenced until the reads are cleared
        # (a synchronous submission has copied them when it returns: nothing is kept, however often the caller submits)
        if getattr(self.ctx, "async_upload", False):
            self._keep = getattr(self, "_keep", []) + [(words, start, length)]
        _chk(self.ctx._h, self.ctx.lib.smx_submit_reads_packed(
            self.ctx._h, words.ctypes.data_as(C.POINTER(C.c_uint64)), len(words),
            start.ctypes.data_as(C.POINTER(C.c_uint64)), length.ctypes.data_as(C.POINTER(C.c_uint32)), len(start)))

    def push_back_fastq_text(self, text: bytes, is_final: bool = True) -> int:
        """Uncompressed 4-line FASTQ bytes, parsed on the device; returns the number of bytes consumed (complete records)."""
        n, used = C.c_uint64(), C.c_uint64()
        _chk(self.ctx._h, self.ctx.lib.smx_submit_fastq_text(self.ctx._h, text, len(text), 1 if is_final else 0, C.byref(n), C.byref(used)))
        return used.value

    def push_back_binary(self, seq_path: str):
        """SPAdes binary reads (<prefix>.seq of io::ReadConverter::ConvertToBinary)."""
        _chk(self.ctx._h, self.ctx.lib.smx_submit_reads_binary(self.ctx._h, seq_path.encode()))

    def push_back_device(self, d_words: int, n_words: int, d_start: int, d_len: int, n_reads: int):
        """HBM-resident packed reads (raw device addresses, e.g. torch tensor .data_ptr())."""
        _chk(self.ctx._h, self.ctx.lib.smx_submit_reads_device(self.ctx._h, d_words, n_words, d_start, d_len, n_reads))

    def clear(self):
        _chk(self.ctx._h, self.ctx.lib.smx_reads_clear(self.ctx._h))
        self._keep = []


class KMerDiskStorage:
    """Result of a count: num_buckets sorted-unique runs resident in HBM."""

    def __init__(self, ctx: Context, K: int, num_buckets: int, workdir: Optional[str]):
        self.ctx, self.k_, self.nb, self.workdir = ctx, K, num_buckets, workdir
        n, nw, inst = C.c_uint64(), C.c_uint(), C.c_uint64()
        _chk(ctx._h, ctx.lib.smx_count_info(ctx._h, C.byref(n), C.byref(nw), C.byref(inst)))
        self._n, self._nw, self._inst = n.value, nw.value, inst.value
        sizes = np.zeros(num_buckets, dtype=np.uint64)
        _chk(ctx._h, ctx.lib.smx_bucket_sizes(ctx._h, sizes.ctypes.data_as(C.POINTER(C.c_uint64))))
        self._sizes = sizes
        self._final = None

    def k(self) -> int:
        return self.k_

    def num_buckets(self) -> int:
        return self.nb

    def total_kmers(self) -> int:  # kmer_index_builder.hpp:139-150
        return int(self._n)

    def kmer_instances(self) -> int:
        return int(self._inst)

    def bucket_size(self, i: int) -> int:  # kmer_index_builder.hpp:176-178
        return int(self._sizes[i])

    def bucket_sizes(self) -> np.ndarray:
        return self._sizes.copy()

    def bucket(self, i: int) -> np.ndarray:
        """Records of bucket i as [n, words] uint64 (the bytes of the reference's kmers_XXXXXX.<i>)."""
        out = np.empty((self.bucket_size(i), self._nw), dtype=np.uint64)
        _chk(self.ctx._h, self.ctx.lib.smx_copy_bucket(self.ctx._h, i, out.ctypes.data_as(C.c_void_p)))
        return out

    def records(self) -> np.ndarray:
        """All buckets concatenated (the bytes of final_kmers) as [n, words] uint64."""
        out = np.empty((self._n, self._nw), dtype=np.uint64)
        _chk(self.ctx._h, self.ctx.lib.smx_copy_final_kmers(self.ctx._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def device_ptr(self) -> int:
        return int(self.ctx.lib.smx_device_kmers(self.ctx._h) or 0)

    def bucket_to_device(self, i: int, d_dst: int) -> None:
        """bucket i into caller-owned HBM of bucket_size(i) records (a result held as two strands is merged into the block)"""
        _chk(self.ctx._h, self.ctx.lib.smx_copy_bucket_device(self.ctx._h, i, C.c_void_p(d_dst)))

    def merge(self):  # KMerDiskStorage::merge, kmer_index_builder.hpp:190-203
        if self.workdir is None:
            raise SmxError(_lib.INVALID_PARAMETER, "merge() needs a workdir")
        path = os.path.join(self.workdir, "final_kmers")
        _chk(self.ctx._h, self.ctx.lib.smx_write_final_kmers(self.ctx._h, path.encode()))
        self._final = path

    def final_kmers(self) -> str:
        if self._final is None:
            raise SmxError(_lib.INVALID_PARAMETER, "k-mers were not merged yet")  # VERIFY_MSG, kmer_index_builder.hpp:172
        return self._final


class KMerDiskCounter:
    def __init__(self, workdir: Optional[str], splitter: ReadKMerSplitter):
        self.workdir, self.splitter = workdir, splitter

    def k(self) -> int:
        return self.splitter.K()

    def kmer_size(self) -> int:
        return self.splitter.kmer_size()

    def Count(self, num_buckets: int, num_threads: int = 1) -> KMerDiskStorage:
        """num_threads is accepted for signature parity; it does not influence the result
        (the reference's result is thread-independent too: SURVEY.md finding 3)."""
        ctx = self.splitter.ctx
        mode = _lib.MODE_ALL if self.splitter.mode == "A" else _lib.MODE_CANONICAL
        _chk(ctx._h, ctx.lib.smx_count(ctx._h, self.splitter.K(), mode, int(num_buckets)))
        return KMerDiskStorage(ctx, self.splitter.K(), int(num_buckets), self.workdir)

    def CountAll(self, num_buckets: int, num_threads: int = 1, merge: bool = True) -> KMerDiskStorage:
        """Count + merge (kmer_index_builder.hpp:306-332). With a workdir the destination is known before the count starts (smx_count_to_file): a
        count that goes out of core streams its merged bucket ranges into <workdir>/final_kmers as the reference's merge does instead of holding
        the merged result in host memory; the records of such a result are in the file, not in the storage object."""
        if merge and self.workdir is not None:
            ctx = self.splitter.ctx
            mode = _lib.MODE_ALL if self.splitter.mode == "A" else _lib.MODE_CANONICAL
            path = os.path.join(self.workdir, "final_kmers")
            _chk(ctx._h, ctx.lib.smx_count_to_file(ctx._h, self.splitter.K(), mode, int(num_buckets), path.encode()))
            st = KMerDiskStorage(ctx, self.splitter.K(), int(num_buckets), self.workdir)
            st._final = path
            return st
        st = self.Count(num_buckets, num_threads)
        if merge:
            st.merge()
        return st
