"""Host-side mirror of the standalone graph builder (projects/spades_tools/gbuilder.cpp:112-245) over
libspades_mi355x.so: reads -> extension index -> unbranching paths + perfect loops -> graph -> GFA / unitig FASTA.

`threads` plays the role of the reference's -t ONLY as an input of the output order (bucket count 10*threads,
common/kmer_index/extension_index/kmer_extension_index_builder.hpp:75); it does not control parallelism here.
"""
import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from .kmercount import Context, ReadKMerSplitter, SmxError, _chk


class GraphBuilder:
    def __init__(self, k: int, threads: int = 1, ctx: Optional[Context] = None):
        self.k, self.threads = int(k), int(threads)
        self.reads = ReadKMerSplitter(self.k + 1, "B", ctx)
        self.ctx = self.reads.ctx
        self._info = None

    # -- input (same entry points as the counter's splitter) --
    def push_back_reads(self, reads: Sequence[str]):
        self.reads.push_back_reads(reads)

    def push_back_packed(self, words, start, length):
        self.reads.push_back_packed(words, start, length)

    def push_back_device(self, d_words, n_words, d_start, d_len, n_reads):
        self.reads.push_back_device(d_words, n_words, d_start, d_len, n_reads)

    # -- steps 1-3 of gbuilder.cpp --
    def build(self):
        h = self.ctx._h
        _chk(h, self.ctx.lib.smx_build_graph(h, self.k, 10 * self.threads))
        info = (C.c_uint64 * 8)()
        _chk(h, self.ctx.lib.smx_graph_info(h, info))
        self._info = dict(n_kpomers=info[0], n_kmers=info[1], n_unitigs=info[2], n_loops=info[3], n_vertices=info[4],
                          n_links=info[5], unitig_bases=info[6], words=info[7])
        return self._info

    def adopt(self, info: dict):
        """take over a graph built on this context by spades_amd.dist.sharded_build_graph"""
        self._info = dict(info)
        return self._info

    def info(self):
        info = (C.c_uint64 * 8)()
        _chk(self.ctx._h, self.ctx.lib.smx_graph_info(self.ctx._h, info))
        self._info.update(n_links=info[5])
        return self._info

    def fingerprint(self):
        """order-sensitive checksums of the device-resident graph arrays (smx_graph_fingerprint): k-mer file, masks, packed unitigs,
        unitig lengths, start / end nodes, link records, vertex starts — (sum, position-weighted sum) each"""
        out = (C.c_uint64 * 16)()
        _chk(self.ctx._h, self.ctx.lib.smx_graph_fingerprint(self.ctx._h, out))
        return [int(v) for v in out]

    def fingerprint_portable(self):
        """the part of fingerprint() that does not depend on how the k-mers are numbered (smx_graph_fingerprint_portable): packed
        unitigs, lengths, self-conjugate flags, link structure by vertex order — equal between the construction routes"""
        out = (C.c_uint64 * 8)()
        _chk(self.ctx._h, self.ctx.lib.smx_graph_fingerprint_portable(self.ctx._h, out))
        return [int(v) for v in out]

    def route_stats(self):
        """smx_graph_route_stats: which construction route the last build took and its partition / junction figures"""
        st = (C.c_uint64 * 8)()
        _chk(self.ctx._h, self.ctx.lib.smx_graph_route_stats(self.ctx._h, st))
        return dict(route=("pm", "ext", "kpo")[min(int(st[0]), 2)], kmers_in_chunks=int(st[1]), kmers_of_cut_partitions=int(st[2]), chunks=int(st[3]),
                    junction_kmers=int(st[4]), start_de_edges=int(st[5]), superkmer_slots=int(st[6]), folded_instances=int(st[7]))

    def tip_stats(self):
        """(k-mers isolated, tips removed) by the early tip clipper and (A/T edges, A/T tip k-mers) by the early A/T remover of the
        last build (options early_tip_bound, early_at_remover)."""
        st = (C.c_uint64 * 4)()
        _chk(self.ctx._h, self.ctx.lib.smx_graph_tip_stats(self.ctx._h, st))
        return int(st[0]), int(st[1]), int(st[2]), int(st[3])

    def kmers(self):
        n, nw = self._info["n_kmers"], self._info["words"]
        rec = np.empty((n, nw), dtype=np.uint64)
        masks = np.empty(n, dtype=np.uint8)
        _chk(self.ctx._h, self.ctx.lib.smx_graph_copy_kmers(self.ctx._h, rec.ctypes.data_as(C.c_void_p), masks.ctypes.data_as(C.c_void_p)))
        return rec, masks

    def unitigs(self) -> List[str]:
        n = self._info["n_unitigs"]
        off = np.zeros(n + 1, dtype=np.uint64)
        seq = np.empty(max(self._info["unitig_bases"], 1), dtype=np.uint8)
        _chk(self.ctx._h, self.ctx.lib.smx_graph_copy_unitigs(self.ctx._h, off.ctypes.data_as(C.POINTER(C.c_uint64)), seq.ctypes.data_as(C.c_void_p)))
        b = seq.tobytes()
        return [b[int(off[i]):int(off[i + 1])].decode() for i in range(n)]

    def fill_coverage(self):
        """-c: CoverageHashMapBuilder + FillCoverageAndFlankingFromPHM (gbuilder.cpp:208-219)."""
        _chk(self.ctx._h, self.ctx.lib.smx_graph_fill_coverage(self.ctx._h))

    def raw_coverage(self) -> np.ndarray:
        out = np.zeros(self._info["n_unitigs"], dtype=np.uint32)
        _chk(self.ctx._h, self.ctx.lib.smx_graph_copy_coverage(self.ctx._h, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def flanking_coverage(self):
        """(flank of every canonical edge, flank of its conjugate): raw counts over the first / last `flank_range` (k+1)-mers."""
        a = np.zeros(self._info["n_unitigs"], dtype=np.uint32)
        b = np.zeros(self._info["n_unitigs"], dtype=np.uint32)
        _chk(self.ctx._h, self.ctx.lib.smx_graph_copy_flanking(self.ctx._h, a.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                                b.ctypes.data_as(C.POINTER(C.c_uint32))))
        return a, b

    # -- step 4: outputs --
    def write_gfa(self, path: str, flavour_version: str = "SPAdes-4.3.0-dev"):
        _chk(self.ctx._h, self.ctx.lib.smx_graph_write_gfa(self.ctx._h, path.encode(), flavour_version.encode()))

    def write_fastg(self, path: str):
        _chk(self.ctx._h, self.ctx.lib.smx_graph_write_fastg(self.ctx._h, path.encode()))

    def write_spades(self, basename: str):
        """--spades: <basename>.grseq + <basename>.cvr (io::binary::BasicGraphIO::Save)."""
        _chk(self.ctx._h, self.ctx.lib.smx_graph_write_spades(self.ctx._h, basename.encode()))

    def write_unitigs(self, path: str):
        _chk(self.ctx._h, self.ctx.lib.smx_graph_write_unitigs(self.ctx._h, path.encode()))
