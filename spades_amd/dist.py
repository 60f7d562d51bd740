"""Multi-GPU sharding of the counting path: one process per GPU, bucket-range owners,
ONE all-to-all of k-mer records over RCCL/xGMI, no other data-path collective.

  local reads --extract+XXH3--> records grouped by owner rank --all_to_all_single--> owner: sort+unique

Rank r owns buckets [first(r), first(r+1)), first(r) = ceil(r*B/world) (bucket = mulhi(XXH3, B) is monotone in
the hash, so ownership is a contiguous hash range); the final file is the concatenation of the ranks' outputs.
The engine is pluggable so that the orchestration is testable on CPU with gloo (tests inject a CPU engine
built on the test oracle); the product engine is GpuEngine (libspades_mi355x.so), nothing else.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib
from .kmercount import Context, _chk


XCHG_LIMIT = 1 << 27  # int64 elements (1 GiB) per pair and round


def rank_first_bucket(num_buckets: int, world: int, rank: int) -> int:
    return (rank * num_buckets + world - 1) // world


class GpuEngine:
    def __init__(self, ctx: Context, mode: str):
        self.ctx = ctx
        self.mode = _lib.MODE_ALL if mode == "A" else _lib.MODE_CANONICAL

    def alloc(self, n_words: int, dev):
        return torch.empty(max(n_words, 1), dtype=torch.int64, device=dev)

    def extract_count(self, K: int) -> int:
        n = C.c_uint64()
        _chk(self.ctx._h, self.ctx.lib.smx_extract_count(self.ctx._h, K, self.mode, C.byref(n)))
        return n.value

    def extract_partition(self, K: int, nb: int, world: int, buf: torch.Tensor, capacity: int):
        counts = (C.c_uint64 * world)()
        _chk(self.ctx._h, self.ctx.lib.smx_extract_partition(self.ctx._h, K, self.mode, nb, world, buf.data_ptr(), capacity, counts))
        return [int(c) for c in counts]

    def count_records(self, K: int, nb: int, buf: torch.Tensor, n: int):
        h = self.ctx._h
        _chk(h, self.ctx.lib.smx_count_records(h, K, nb, buf.data_ptr(), n))
        nrec, nw, inst = C.c_uint64(), C.c_uint(), C.c_uint64()
        _chk(h, self.ctx.lib.smx_count_info(h, C.byref(nrec), C.byref(nw), C.byref(inst)))
        sizes = (C.c_uint64 * nb)()
        _chk(h, self.ctx.lib.smx_bucket_sizes(h, sizes))
        return {"distinct": nrec.value, "instances": inst.value, "bucket_sizes": [int(s) for s in sizes],
                "device_ptr": int(self.ctx.lib.smx_device_kmers(h) or 0)}


    # -- construction (the compact structure is gathered and the lookup replicated) --
    def result_tensor(self, n_words: int, dev):
        t = torch.empty(max(n_words, 1), dtype=torch.int64, device=dev)
        _chk(self.ctx._h, self.ctx.lib.smx_copy_kmers_device(self.ctx._h, t.data_ptr()))
        return t

    def build_graph_from_records(self, k: int, nb: int, buf: torch.Tensor, n: int):
        h = self.ctx._h
        _chk(h, self.ctx.lib.smx_build_graph_from_records(h, k, nb, buf.data_ptr(), n))
        info = (C.c_uint64 * 8)()
        _chk(h, self.ctx.lib.smx_graph_info(h, info))
        return dict(n_kpomers=info[0], n_kmers=info[1], n_unitigs=info[2], n_loops=info[3], n_vertices=info[4],
                    unitig_bases=info[6], words=info[7])

    def local_raw_coverage(self, n_unitigs: int) -> torch.Tensor:
        h = self.ctx._h
        _chk(h, self.ctx.lib.smx_graph_fill_coverage(h))
        out = torch.zeros(max(n_unitigs, 1), dtype=torch.int32)
        _chk(h, self.ctx.lib.smx_graph_copy_coverage(h, C.cast(out.data_ptr(), C.POINTER(C.c_uint32))))
        return out[:n_unitigs]

    def set_raw_coverage(self, cov: torch.Tensor):
        cov = cov.contiguous()
        _chk(self.ctx._h, self.ctx.lib.smx_graph_set_coverage(self.ctx._h, C.cast(cov.data_ptr(), C.POINTER(C.c_uint32)), cov.numel()))


def sharded_count(engine, K: int, nb: int, rank: int, world: int, dev):
    """One step of the sharded path on this rank. Returns the owner-side result dict of the engine
    (+ 'sent'/'received' record counts). Collective: every rank must call it."""
    nw = (K + 31) // 32
    n_local = engine.extract_count(K)
    send = engine.alloc(n_local * nw, dev)
    counts = engine.extract_partition(K, nb, world, send, n_local)
    n_sent = sum(counts)  # < n_local when the engine pre-dedupes its shard before the exchange
    cnt_t = torch.tensor(counts, dtype=torch.int64, device=dev)
    rcv_t = torch.empty_like(cnt_t)
    dist.all_to_all_single(rcv_t, cnt_t)
    rcounts = [int(c) for c in rcv_t.tolist()]
    n_recv = sum(rcounts)
    recv = engine.alloc(n_recv * nw, dev)
    # The exchange proper. Splits are capped at XCHG_LIMIT elements per (pair, round): one all_to_all_single of a
    # 30 GB buffer (3.8 G int64 elements) silently truncates on this stack (measured: tail left untouched), so large
    # segments go in several rounds of views (no staging copies); every pair still moves each record exactly once.
    soff = [0]
    for c in counts:
        soff.append(soff[-1] + c * nw)
    roff = [0]
    for c in rcounts:
        roff.append(roff[-1] + c * nw)
    mx = torch.tensor([max(counts) * nw if counts else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    rounds = max(1, -(-int(mx.item()) // XCHG_LIMIT))
    if rounds == 1:
        dist.all_to_all_single(recv[:n_recv * nw], send[:n_sent * nw],
                               output_split_sizes=[c * nw for c in rcounts], input_split_sizes=[c * nw for c in counts])
    else:
        # grouped point-to-point rounds on views (ncclSend/ncclRecv pairs under one group on RCCL): every pair has its own
        # xGMI link, there is no ring to serialise on, and no staging copy is needed
        for r in range(rounds):
            ops = []
            for p in range(world):
                a = min(soff[p] + r * XCHG_LIMIT, soff[p + 1])
                b = min(a + XCHG_LIMIT, soff[p + 1])
                c = min(roff[p] + r * XCHG_LIMIT, roff[p + 1])
                d = min(c + XCHG_LIMIT, roff[p + 1])
                if p == rank:
                    recv[c:d].copy_(send[a:b])
                    continue
                if b > a:
                    ops.append(dist.P2POp(dist.isend, send[a:b], p))
                if d > c:
                    ops.append(dist.P2POp(dist.irecv, recv[c:d], p))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
    res = engine.count_records(K, nb, recv, n_recv)
    res["sent"], res["received"] = n_sent, n_recv
    res["instances"] = n_local  # k-mer instances extracted from this rank's reads
    return res


def _bcast_chunks(t: torch.Tensor, a: int, b: int, src: int):
    """broadcast t[a:b] from rank src in rounds of XCHG_LIMIT elements (same size cap as the exchange above)"""
    while a < b:
        e = min(a + XCHG_LIMIT, b)
        dist.broadcast(t[a:e], src=src)
        a = e


def sharded_build_graph(engine, k: int, threads: int, rank: int, world: int, dev, coverage: bool = False):
    """Construction on `world` ranks (collective). Counting of the canonical (k+1)-mers is sharded exactly like
    sharded_count; the owners' sorted-unique arrays are then all-gathered (bucket-major, so the concatenation in rank
    order IS the (k+1)-mer file) and every rank derives the same k-mer file, masks, unitigs and links from it — the
    replicated-lookup variant of SURVEY.md §8e (unitig walks cross owners at every step). Coverage (-c) stays sharded:
    each rank counts its own reads against the replicated (k+1)-mer file, raw edge coverages are all-reduced (SUM mod 2^32).
    Every rank ends with the same graph; rank 0 normally writes it. Returns the engine's graph info dict."""
    K1, nb = k + 1, 10 * threads
    nw = (K1 + 31) // 32
    res = sharded_count(engine, K1, nb, rank, world, dev)
    mine = torch.tensor([res["distinct"]], dtype=torch.int64, device=dev)
    every = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    counts = [int(c.item()) for c in every]
    off = [0]
    for c in counts:
        off.append(off[-1] + c * nw)
    full = engine.alloc(off[-1], dev)
    if counts[rank]:
        full[off[rank]:off[rank + 1]].copy_(engine.result_tensor(counts[rank] * nw, dev)[:counts[rank] * nw])
    for r in range(world):
        _bcast_chunks(full, off[r], off[r + 1], r)
    info = engine.build_graph_from_records(k, nb, full, sum(counts))
    del full
    if coverage:
        cov = engine.local_raw_coverage(info["n_unitigs"]).to(torch.int64) & 0xFFFFFFFF
        cov = cov.to(dev)
        dist.all_reduce(cov, op=dist.ReduceOp.SUM)
        cov = (cov & 0xFFFFFFFF).to("cpu")
        cov = torch.where(cov >= 2 ** 31, cov - 2 ** 32, cov).to(torch.int32)  # uint32 bit pattern
        engine.set_raw_coverage(cov)
    info["kpomers_per_rank"] = counts
    return info
