"""spades_amd — MI355X-native k-mer counting / de Bruijn construction hot path of SPAdes.

The product is libspades_mi355x.so (hand-written gfx950 HIP kernels behind the C ABI of
include/smx.h). This package is the thin Python host side: a ctypes binding (`_lib`), a mirror of
the reference's counter interface (`kmercount`), and read packing / synthetic data (`reads`).
There is NO CPU fallback: without the built library or without a GPU every compute call raises.
"""
from . import _lib  # noqa: F401
from .kmercount import KMerDiskCounter, KMerDiskStorage, ReadKMerSplitter, SmxError  # noqa: F401

__all__ = ["KMerDiskCounter", "KMerDiskStorage", "ReadKMerSplitter", "SmxError"]
