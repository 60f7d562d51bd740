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


class CollectiveFailure(RuntimeError):
    """A local step failed on SOME rank: raised on EVERY rank (code = the worst code any rank saw), so that nobody is left waiting
    in the next collective. .code carries the reference's exit code (68 = memory limit exceeded, ...)."""

    def __init__(self, code: int, what: str, cause: Exception = None):
        super().__init__(f"{what}: " + (str(cause) if cause is not None else f"another rank failed with code {code}"))
        self.code = code
        self.memory_only = code == _lib.MEMORY_LIMIT_EXCEEDED  # (_guarded sets it from what ALL ranks reported)


def _staged(t: torch.Tensor) -> bool:
    """ranks that share a GPU, or a box without RCCL between them, run on gloo: it moves HBM tensors only for broadcast / all-reduce, so
    the all-to-all and the all-gather go through host memory there (tests/test_dist_gpu.py: two processes on one MI355X)"""
    return t.is_cuda and dist.get_backend() == "gloo"


def _all_to_all_single(out: torch.Tensor, inp: torch.Tensor, output_split_sizes=None, input_split_sizes=None):
    if _staged(inp):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes)


def _all_gather(outs, t: torch.Tensor):
    if _staged(t):
        o = [torch.empty(x.shape, dtype=x.dtype) for x in outs]
        dist.all_gather(o, t.cpu())
        for x, y in zip(outs, o):
            x.copy_(y)
    else:
        dist.all_gather(outs, t)


def _error_code(err: Exception) -> int:
    """the reference's exit code of a local failure: the library's own (.code), 68 for an allocator that ran out of HBM on the torch side
    (torch.cuda.OutOfMemoryError carries no .code), 1 for anything else"""
    code = getattr(err, "code", None)
    if isinstance(code, int) and code:
        return code
    oom = getattr(getattr(torch, "cuda", None), "OutOfMemoryError", None)
    if (oom is not None and isinstance(err, oom)) or isinstance(err, MemoryError):
        return _lib.MEMORY_LIMIT_EXCEEDED
    return 1


def _guarded(dev, what: str, fn, *args):
    """run a rank-local step between two collectives; every rank then learns (one tiny all-reduce) whether all of them got through.
    Two figures travel: the worst code any rank saw, and whether EVERY failure was a memory limit — only then may a caller fall back
    to a leaner route; a genuine error on some other rank (invalid parameter, an exception) must not be swallowed by a rank's 68."""
    err, res = None, None
    try:
        res = fn(*args)
    except Exception as e:  # noqa: BLE001 — whatever it is, the other ranks must hear of it
        err = e
    code = 0 if err is None else _error_code(err)
    other = 1 if (code and code != _lib.MEMORY_LIMIT_EXCEEDED) else 0
    t = torch.tensor([code, other], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    worst, any_other = int(t[0].item()), int(t[1].item())
    if worst:
        f = CollectiveFailure(worst, what, err)
        f.memory_only = not any_other  # every rank that failed ran out of memory: the one condition a fallback may handle
        raise f
    return res


def rank_first_bucket(num_buckets: int, world: int, rank: int) -> int:
    return (rank * num_buckets + world - 1) // world


class _DevView:
    """int64 view of library-owned HBM for torch.as_tensor (no copy, no ownership)"""

    def __init__(self, p: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (p, False), "version": 2}


class GpuEngine:
    def __init__(self, ctx: Context, mode: str):
        self.ctx = ctx
        self.mode = _lib.MODE_ALL if mode == "A" else _lib.MODE_CANONICAL

    def alloc(self, n_words: int, dev):
        return torch.empty(max(n_words, 1), dtype=torch.int64, device=dev)

    def trim(self) -> int:
        return self.ctx.trim()

    def extract_count(self, K: int) -> int:
        n = C.c_uint64()
        _chk(self.ctx._h, self.ctx.lib.smx_extract_count(self.ctx._h, K, self.mode, C.byref(n)))
        return n.value

    def extract_partition(self, K: int, nb: int, world: int, buf: torch.Tensor, capacity: int):
        counts = (C.c_uint64 * world)()
        _chk(self.ctx._h, self.ctx.lib.smx_extract_partition(self.ctx._h, K, self.mode, nb, world, buf.data_ptr(), capacity, counts))
        return [int(c) for c in counts]

    def extract_partition_owned(self, K: int, nb: int, world: int, dev):
        """records of this rank grouped by owner, in a buffer the library sized after its local pre-dedupe; returned as a tensor view"""
        counts = (C.c_uint64 * world)()
        ptr = C.c_void_p()
        _chk(self.ctx._h, self.ctx.lib.smx_extract_partition_owned(self.ctx._h, K, self.mode, nb, world, C.byref(ptr), counts))
        counts = [int(c) for c in counts]
        n_words = sum(counts) * ((K + 31) // 32)
        if n_words == 0:
            return torch.empty(1, dtype=torch.int64, device=dev), counts
        return torch.as_tensor(_DevView(ptr.value, n_words), device=dev), counts

    def alloc_recv(self, n_words: int, dev):
        """receive side of the exchange in the library's own HBM pool (consumed by count_records), as a tensor view"""
        if n_words == 0:
            return torch.empty(1, dtype=torch.int64, device=dev)
        ptr = C.c_void_p()
        _chk(self.ctx._h, self.ctx.lib.smx_exchange_buffer(self.ctx._h, n_words, C.byref(ptr)))
        return torch.as_tensor(_DevView(ptr.value, n_words), device=dev)

    def extract_release(self):
        _chk(self.ctx._h, self.ctx.lib.smx_extract_release(self.ctx._h))

    def state(self, n: int, dtype, dev):
        """a long-lived working array of the CALLER inside the library's device arena (smx_pool_alloc), as a tensor view: (tensor of n elements, handle for
        state_free). The arena only grows: after the library's big steps a framework allocator finds the device full although half of the arena is free —
        the walk state of distributed_walks (9-10 B per oriented node of the shard) lives where that room is. Not initialised."""
        item = torch.empty(0, dtype=dtype).element_size()
        words = (max(n, 1) * item + 7) // 8
        ptr = C.c_void_p()
        _chk(self.ctx._h, self.ctx.lib.smx_pool_alloc(self.ctx._h, words * 8, C.byref(ptr)))
        t = torch.as_tensor(_DevView(ptr.value, words), device=dev)
        if dtype != torch.int64:
            t = t.view(torch.uint8)
            if dtype != torch.uint8:
                t = t.view(dtype)
        return t[:n], ptr.value

    def state_free(self, handle):
        if handle:
            _chk(self.ctx._h, self.ctx.lib.smx_pool_free(self.ctx._h, handle))

    def arena_free_bytes(self) -> int:
        return self.ctx.arena_free_bytes()

    def exchange_release(self):
        """give back a receive buffer of the library's pool that no count consumed (a route abandoned between two collectives)"""
        _chk(self.ctx._h, self.ctx.lib.smx_exchange_release(self.ctx._h))

    def graph_clear(self):
        """drop whatever shard / graph state the context holds (smx_graph_clear)"""
        _chk(self.ctx._h, self.ctx.lib.smx_graph_clear(self.ctx._h))

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

    # -- sharded construction (owner-side mask fill) --
    def shard_updates(self, k: int, nb: int, world: int, buf: torch.Tensor, capacity: int):
        counts = (C.c_uint64 * world)()
        _chk(self.ctx._h, self.ctx.lib.smx_graph_shard_updates(self.ctx._h, k, nb, world, buf.data_ptr(), capacity, counts))
        return [int(c) for c in counts]

    def shard_build(self, k: int, nb: int, world: int, rank: int, buf: torch.Tensor, n: int):
        h = self.ctx._h
        _chk(h, self.ctx.lib.smx_graph_shard_build(h, k, nb, world, rank, buf.data_ptr(), n))
        nk = C.c_uint64()
        sizes = (C.c_uint64 * nb)()
        _chk(h, self.ctx.lib.smx_graph_shard_info(h, C.byref(nk), sizes))
        return nk.value, [int(x) for x in sizes]

    # -- the same shard by ONE exchange: k-mers travel with the InOutMask byte the sender's reads give them --
    def ext_supported(self, k: int) -> bool:
        return bool(self.ctx.lib.smx_kmers_with_masks_supported(k))

    def extract_kmers_ext_owned(self, k: int, nb: int, world: int, dev):
        counts = (C.c_uint64 * world)()
        ptr = C.c_void_p()
        _chk(self.ctx._h, self.ctx.lib.smx_extract_kmers_ext_owned(self.ctx._h, k, nb, world, C.byref(ptr), counts))
        counts = [int(c) for c in counts]
        n_words = sum(counts) * ((k + 31) // 32)
        if n_words == 0:
            return torch.empty(1, dtype=torch.int64, device=dev), counts
        return torch.as_tensor(_DevView(ptr.value, n_words), device=dev), counts

    def shard_from_ext(self, k: int, nb: int, world: int, rank: int, buf: torch.Tensor, n: int):
        """-> (k-mers of the shard, their bucket sizes, extension bits set, palindromic (k+1)-mers among them)"""
        h = self.ctx._h
        _chk(h, self.ctx.lib.smx_graph_shard_from_ext(h, k, nb, world, rank, buf.data_ptr(), n))
        nk = C.c_uint64()
        sizes = (C.c_uint64 * nb)()
        st = (C.c_uint64 * 2)()
        _chk(h, self.ctx.lib.smx_graph_shard_info(h, C.byref(nk), sizes))
        _chk(h, self.ctx.lib.smx_graph_shard_ext_stats(h, st))
        return nk.value, [int(x) for x in sizes], int(st[0]), int(st[1])

    def shard_copy(self, kmers: torch.Tensor, masks: torch.Tensor):
        _chk(self.ctx._h, self.ctx.lib.smx_graph_shard_copy(self.ctx._h, kmers.data_ptr(), masks.data_ptr()))

    def alloc_bytes(self, n: int, dev):
        return torch.empty(max(n, 1), dtype=torch.uint8, device=dev)

    def build_graph_from_kmers(self, k: int, nb: int, kmers: torch.Tensor, masks: torch.Tensor, n: int, bucket_sizes, n_kpomers: int):
        h = self.ctx._h
        bs = (C.c_uint64 * nb)(*bucket_sizes)
        _chk(h, self.ctx.lib.smx_build_graph_from_kmers(h, k, nb, kmers.data_ptr(), masks.data_ptr(), n, bs, n_kpomers))
        info = (C.c_uint64 * 8)()
        _chk(h, self.ctx.lib.smx_graph_info(h, info))
        return dict(n_kpomers=info[0], n_kmers=info[1], n_unitigs=info[2], n_loops=info[3], n_vertices=info[4],
                    unitig_bases=info[6], words=info[7])

    # -- distributed walks: the k-mer-specific steps, on this rank's shard --
    def walk_counts(self):
        a, b = C.c_uint64(), C.c_uint64()
        _chk(self.ctx._h, self.ctx.lib.smx_shard_walk_counts(self.ctx._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def walk_requests(self, starts: bool, k: int, world: int, dev, first: int = 0, n_items: int = -1):
        """-> (canonical successor k-mers grouped by owner, their tags in the same order, records per owner); first / n_items: only the
        oriented nodes (or start de-edges) [first, first + n_items) ask — the caller walks a large shard range by range"""
        n_chain, n_start = self.walk_counts()
        if n_items < 0:
            n = n_start if starts else n_chain
        else:
            n = n_items  # (at most one request per item)
        recs = torch.empty(max(n * ((k + 31) // 32), 1), dtype=torch.int64, device=dev)
        tags = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
        counts = (C.c_uint64 * world)()
        _sync(dev)  # the caching allocator may hand out a block that queued torch kernels still read: the library writes from its own stream
        if n_items < 0:
            _chk(self.ctx._h, self.ctx.lib.smx_shard_walk_requests(self.ctx._h, 1 if starts else 0, world, recs.data_ptr(),
                                                                  C.cast(tags.data_ptr(), C.POINTER(C.c_uint64)), counts))
        else:
            _chk(self.ctx._h, self.ctx.lib.smx_shard_walk_requests_range(self.ctx._h, 1 if starts else 0, world, int(first), int(n_items), recs.data_ptr(),
                                                                        C.cast(tags.data_ptr(), C.POINTER(C.c_uint64)), counts))
        counts = [int(c) for c in counts]
        return recs, tags[:sum(counts)], counts

    def walk_starts(self, dev):
        _, n = self.walk_counts()
        t = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
        _chk(self.ctx._h, self.ctx.lib.smx_shard_walk_starts(self.ctx._h, C.cast(t.data_ptr(), C.POINTER(C.c_uint64))))
        return t[:n]

    def shard_lookup(self, recs: torch.Tensor, n: int, dev):
        out = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
        _sync(dev)
        _chk(self.ctx._h, self.ctx.lib.smx_shard_lookup(self.ctx._h, recs.data_ptr(), n, C.cast(out.data_ptr(), C.POINTER(C.c_uint64))))
        return out[:n]

    def shard_gather_kmers(self, local_ranks: torch.Tensor, k: int, dev):
        n = local_ranks.numel()
        km = torch.empty(max(n * ((k + 31) // 32), 1), dtype=torch.int64, device=dev)
        mk = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        _sync(dev)
        _chk(self.ctx._h, self.ctx.lib.smx_shard_gather_kmers(self.ctx._h, C.cast(local_ranks.data_ptr(), C.POINTER(C.c_uint64)), n, km.data_ptr(),
                                                             C.cast(mk.data_ptr(), C.POINTER(C.c_uint8))))
        return km, mk

    def shard_unitigs(self, first_rank: int, steps: torch.Tensor, last: torch.Tensor, boff: torch.Tensor, bases: torch.Tensor, dev):
        """-> this rank's kept unitigs: (packed words, lengths, start nodes, end nodes, self-conjugate flags)"""
        u64 = C.POINTER(C.c_uint64)
        nk, nwd = C.c_uint64(), C.c_uint64()
        _sync(dev)
        _chk(self.ctx._h, self.ctx.lib.smx_shard_unitigs(self.ctx._h, first_rank, C.cast(steps.data_ptr(), u64), C.cast(last.data_ptr(), u64),
                                                        C.cast(boff.data_ptr(), u64), C.cast(bases.data_ptr(), C.POINTER(C.c_uint8)), C.byref(nk), C.byref(nwd)))
        ne, nwords = nk.value, nwd.value
        words = torch.empty(max(nwords, 1), dtype=torch.int64, device=dev)
        ln, st, en = (torch.empty(max(ne, 1), dtype=torch.int64, device=dev) for _ in range(3))
        sf = torch.empty(max(ne, 1), dtype=torch.uint8, device=dev)
        _sync(dev)
        _chk(self.ctx._h, self.ctx.lib.smx_shard_unitigs_copy(self.ctx._h, C.cast(words.data_ptr(), u64), C.cast(ln.data_ptr(), u64), C.cast(st.data_ptr(), u64),
                                                             C.cast(en.data_ptr(), u64), C.cast(sf.data_ptr(), C.POINTER(C.c_uint8))))
        return words[:nwords], ln[:ne], st[:ne], en[:ne], sf[:ne]

    def build_graph_from_unitigs(self, k: int, nb: int, n_kmers: int, n_kpomers: int, words, n_words: int, ln, st, en, sf, ne: int, loop_ranks, loop_kmers, loop_masks):
        """loop_*: host numpy arrays (uint64 global ranks in file order, uint64 k-mer words, uint8 masks)"""
        h = self.ctx._h
        u64 = C.POINTER(C.c_uint64)
        nl = int(len(loop_ranks))
        _chk(h, self.ctx.lib.smx_build_graph_from_unitigs(
            h, k, nb, n_kmers, n_kpomers, C.cast(words.data_ptr(), u64), n_words, C.cast(ln.data_ptr(), u64), C.cast(st.data_ptr(), u64),
            C.cast(en.data_ptr(), u64), C.cast(sf.data_ptr(), C.POINTER(C.c_uint8)), ne,
            loop_ranks.ctypes.data_as(u64) if nl else None, loop_kmers.ctypes.data_as(u64) if nl else None,
            loop_masks.ctypes.data_as(C.POINTER(C.c_uint8)) if nl else None, nl))
        info = (C.c_uint64 * 8)()
        _chk(h, self.ctx.lib.smx_graph_info(h, info))
        return dict(n_kpomers=info[0], n_kmers=info[1], n_unitigs=info[2], n_loops=info[3], n_vertices=info[4],
                    unitig_bases=info[6], words=info[7])

    def set_kpomers(self, buf: torch.Tensor, n: int, bucket_sizes):
        nb = len(bucket_sizes)
        bs = (C.c_uint64 * nb)(*bucket_sizes)
        _chk(self.ctx._h, self.ctx.lib.smx_graph_set_kpomers(self.ctx._h, buf.data_ptr(), n, bs))

    def local_raw_coverage(self, n_unitigs: int) -> torch.Tensor:
        h = self.ctx._h
        _chk(h, self.ctx.lib.smx_graph_fill_coverage(h))
        out = torch.zeros(max(n_unitigs, 1), dtype=torch.int32)
        _chk(h, self.ctx.lib.smx_graph_copy_coverage(h, C.cast(out.data_ptr(), C.POINTER(C.c_uint32))))
        return out[:n_unitigs]

    def set_raw_coverage(self, cov: torch.Tensor):
        cov = cov.contiguous()
        _chk(self.ctx._h, self.ctx.lib.smx_graph_set_coverage(self.ctx._h, C.cast(cov.data_ptr(), C.POINTER(C.c_uint32)), cov.numel()))


def _a2a(send: torch.Tensor, counts, rank: int, world: int, dev, alloc=None):
    """ONE all-to-all of a 1-D tensor: counts[p] ELEMENTS go to rank p. Returns (recv, elements received from every rank).
    Splits are capped at XCHG_LIMIT elements per (pair, round): one all_to_all_single of a 30 GB buffer (3.8 G int64 elements)
    silently truncates on this stack (measured: tail left untouched), so large segments go in several rounds of views (no staging
    copies); every pair still moves each element exactly once."""
    n_sent = sum(counts)
    cnt_t = torch.tensor(counts, dtype=torch.int64, device=dev)
    rcv_t = torch.empty_like(cnt_t)
    _all_to_all_single(rcv_t, cnt_t)
    rcounts = [int(c) for c in rcv_t.tolist()]
    n_recv = sum(rcounts)
    recv = alloc(n_recv) if alloc is not None else torch.empty(max(n_recv, 1), dtype=send.dtype, device=dev)
    soff = [0]
    for c in counts:
        soff.append(soff[-1] + c)
    roff = [0]
    for c in rcounts:
        roff.append(roff[-1] + c)
    mx = torch.tensor([max(counts) if counts else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    rounds = max(1, -(-int(mx.item()) // XCHG_LIMIT))
    if rounds == 1:
        _all_to_all_single(recv[:n_recv], send[:n_sent], output_split_sizes=rcounts, input_split_sizes=list(counts))
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
    if dev.type == "cuda":  # the library runs on its own stream: what was received must have landed before it reads it
        torch.cuda.current_stream(dev).synchronize()
    return recv, rcounts


def _exchange(engine, send: torch.Tensor, counts, wpr: int, rank: int, world: int, dev, pool: bool = False):
    """ONE all-to-all of records of `wpr` int64 words: counts[p] records go to rank p. Returns (recv tensor, records received)."""
    # pool: the receive buffer comes from the engine's own HBM pool and is consumed by the count that follows
    alloc = engine.alloc_recv if pool and hasattr(engine, "alloc_recv") else engine.alloc
    recv, rcounts = _a2a(send, [c * wpr for c in counts], rank, world, dev, alloc=lambda n: _guarded(dev, "receive buffer", alloc, n, dev))
    return recv, sum(rcounts) // wpr


def sharded_count(engine, K: int, nb: int, rank: int, world: int, dev):
    """One step of the sharded path on this rank. Returns the owner-side result dict of the engine
    (+ 'sent'/'received' record counts, + 'phase_ms': wall time of extract / exchange / owner count on this rank — every phase
    already ends in a synchronisation, nothing is added for the timing). Collective: every rank must call it."""
    import time
    nw = (K + 31) // 32
    t0 = time.perf_counter()

    def local_extract():
        n_local = engine.extract_count(K)
        if hasattr(engine, "extract_partition_owned"):
            # the library sizes the send buffer itself: what its local pre-dedupe leaves, not one record per window instance
            send, counts = engine.extract_partition_owned(K, nb, world, dev)
        else:
            send = engine.alloc(n_local * nw, dev)
            counts = engine.extract_partition(K, nb, world, send, n_local)
        return n_local, send, counts

    n_local, send, counts = _guarded(dev, "extract + partition by owner", local_extract)
    n_sent = sum(counts)  # < n_local when the engine pre-dedupes its shard before the exchange
    t1 = time.perf_counter()
    recv, n_recv = _exchange(engine, send, counts, nw, rank, world, dev, pool=True)
    t2 = time.perf_counter()
    del send
    if hasattr(engine, "extract_release"):
        engine.extract_release()  # room for the owner-side count
    res = _guarded(dev, "owner-side count", engine.count_records, K, nb, recv, n_recv)
    res["phase_ms"] = {"extract": (t1 - t0) * 1e3, "exchange": (t2 - t1) * 1e3, "owner_count": (time.perf_counter() - t2) * 1e3}
    res["sent"], res["received"] = n_sent, n_recv
    res["instances"] = n_local  # k-mer instances extracted from this rank's reads
    return res


def _bcast_chunks(t: torch.Tensor, a: int, b: int, src: int):
    """broadcast t[a:b] from rank src in rounds of XCHG_LIMIT elements (same size cap as the exchange above)"""
    while a < b:
        e = min(a + XCHG_LIMIT, b)
        dist.broadcast(t[a:e], src=src)
        a = e


def _gather_shards(engine, mine: torch.Tensor, n_mine: int, per_rank, unit: int, rank: int, world: int, dev, alloc):
    """all ranks end with the concatenation (rank order) of every rank's `n * unit` elements; mine = this rank's part"""
    off = [0]
    for c in per_rank:
        off.append(off[-1] + c * unit)
    full = alloc(off[-1], dev)
    if n_mine:
        full[off[rank]:off[rank + 1]].copy_(mine[:n_mine * unit])
    for r in range(world):
        _bcast_chunks(full, off[r], off[r + 1], r)
    return full


def _sync(dev):
    if dev.type == "cuda":  # the library runs on its own stream: torch's work on its inputs must be over
        torch.cuda.current_stream(dev).synchronize()


def _by_owner(owner: torch.Tensor, world: int):
    """-> (stable order that groups by owner, elements per owner). The keys are ranks: sorted as 16-bit integers (two radix passes
    instead of the eight of an int64 sort), counted from the sorted keys."""
    skey, order = torch.sort(owner.to(torch.int16), stable=True)
    edges = torch.searchsorted(skey, torch.arange(world + 1, dtype=torch.int16, device=owner.device))
    return order, [int(c) for c in (edges[1:] - edges[:-1]).tolist()]


def _remote_rows(targets: torch.Tensor, owner: torch.Tensor, table: torch.Tensor, my_base: int, rank: int, world: int, dev):
    """table[:, targets - base of the owner] from the ranks that own them (table: one row per field, one column per local node): one
    all-to-all of the indices, one of the fields (both in the order of the requests, so nothing but indices and fields travels).
    -> (len(targets), fields)"""
    w = table.shape[0]
    order, counts = _by_owner(owner, world)
    q, rcounts = _a2a(targets[order].contiguous(), counts, rank, world, dev)
    nq = sum(rcounts)
    rows = table[:, q[:nq] - my_base].t().reshape(-1).contiguous()
    back, _ = _a2a(rows, [c * w for c in rcounts], rank, world, dev)
    out = torch.empty((targets.numel(), w), dtype=table.dtype, device=dev)
    out[order] = back[:targets.numel() * w].reshape(-1, w)
    return out


def _ragged(off: torch.Tensor, ln: torch.Tensor, dev):
    """indices off[i] .. off[i] + ln[i] of every i, concatenated"""
    tot = int(ln.sum().item()) if ln.numel() else 0
    if tot == 0:
        return torch.empty(0, dtype=torch.int64, device=dev)
    seg = torch.repeat_interleave(torch.arange(ln.numel(), device=dev), ln)
    start = torch.cumsum(ln, 0) - ln
    return off[seg] + (torch.arange(tot, device=dev) - start[seg])


WALK_HOP_BITS = 24         # bits of a node's packed word that count the steps to its pointer (a chain of 2^24 k-mers or more is refused)
WALK_CHUNK = 1 << 26       # local nodes per exchange of the doubling / of the chain nucleotides (bounds the temporaries: ~40 B per node)
WALK_START_CHUNK = 1 << 22  # start de-edges per fetch of their chains


def _rounds_of(n_local: int, chunk: int, dev) -> int:
    """chunks the rank with the most elements needs: every rank runs that many (collective) rounds"""
    t = torch.tensor([-(-n_local // chunk)], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def distributed_walks(engine, k: int, rank: int, world: int, dev, kmers_per_rank):
    """Unitigs of a graph whose k-mer file stays sharded (collective). Every rank holds its bucket range of
    {k-mer file, InOutMask bytes}; a walk of the reference (debruijn_graph_constructor.hpp:264-273) would change rank at every step, so
    nothing is walked:
      1. every oriented non-junction k-mer learns its successor node and whether that is a junction k-mer (ONE lookup exchange: the
         canonical successor k-mer travels to its owner, a node id comes back), every start de-edge its first node (a second one);
      2. the chains of non-junction k-mers are ranked by pointer doubling — per round one exchange of node ids and one of (pointer,
         hops, last chain k-mer, end node), ceil(log2(longest chain)) rounds at most; what never finishes lies on perfect loops.
         (Rows may be read in the state of this round or of the one before — a pointer only ever moves ahead along its chain, the
         hops with it — so the rounds run in chunks of WALK_CHUNK nodes and the temporaries stay bounded.)
      3. a chain k-mer x is hops(x^1) steps behind the head tail(x^1)^1 of its chain (the reverse strand went through the same
         doubling): it sends its outgoing nucleotide there, and the owner of the head lays the chain's nucleotides out in order;
      4. the owner of a start de-edge fetches length, end node and nucleotides of the chain behind it and assembles, keeps or drops the
         unitig exactly as the single-GPU route does (engine.shard_unitigs).
    Returns this rank's kept unitigs (k-mer-file order of their start k-mers: concatenated in rank order they are the reference's edge
    list), the local ranks of its k-mers on perfect loops, and the number of doubling rounds."""
    import os
    import time
    nw = (k + 31) // 32
    first = [0]
    for c in kmers_per_rank:
        first.append(first[-1] + int(c))
    n_mine = int(kmers_per_rank[rank])
    base = 2 * first[rank]
    bounds = torch.tensor([2 * f for f in first[1:]], dtype=torch.int64, device=dev)
    t_last = [time.perf_counter()]

    def mark(what):  # SMX_DEBUG: wall time of every phase on rank 0
        if os.environ.get("SMX_DEBUG") and rank == 0:
            _sync(dev)
            now = time.perf_counter()
            print(f"[dist] walks: {what} {1e3 * (now - t_last[0]):.0f} ms", flush=True)
            t_last[0] = now

    def owner_of(nodes):
        return torch.bucketize(nodes, bounds, right=True)

    def lookup(starts: bool, first_item: int = 0, n_items: int = -1):
        """-> (tags of this rank's requests, node each one leads to, is that a junction k-mer) in the order the library grouped them;
        first_item / n_items: the requests of that range of oriented nodes (start de-edges) only (collective: every rank its own range)"""
        if n_items < 0:
            recs, tags, counts = _guarded(dev, "successor requests of the shard", engine.walk_requests, starts, k, world, dev)
        else:
            recs, tags, counts = _guarded(dev, "successor requests of the shard", engine.walk_requests, starts, k, world, dev, first_item, n_items)
        recv, rcounts = _a2a(recs, [c * nw for c in counts], rank, world, dev)
        del recs
        n_recv = sum(rcounts) // nw
        reply = _guarded(dev, "lookup in the shard", engine.shard_lookup, recv, n_recv, dev)
        del recv
        back, _ = _a2a(reply.contiguous(), [c // nw for c in rcounts], rank, world, dev)
        del reply
        n = sum(counts)
        back = back[:n]

        def check():
            if n and bool((back < 0).any().item()):
                raise RuntimeError(f"{int((back < 0).sum().item())} successor k-mers are in no shard: the k-mer file and the masks disagree")
        _guarded(dev, "successor lookups", check)
        junc = (back & 1).to(torch.bool)
        back >>= 1
        o = 0
        for p_, c in enumerate(counts):  # local rank at the owner -> global rank
            back[o:o + c] += first[p_]
            o += c
        back <<= 1
        back |= (tags >> 2) & 1
        return tags, back, junc

    # 1. successors of the chain k-mers, first nodes of the start de-edges. State of a local oriented node: ONE packed word + one byte
    #    (round 3 kept four int64 rows + two more int64 arrays per node: 110 B per owned k-mer; this is 18 B):
    #      word  F << 63 | T << 62 | id << 24 | hops      F: the end of the chain is known; T: this node IS the end (its successor is a
    #            open:      id = pointer, hops = steps to it    junction k-mer); id: 38 bits (2.7e11 nodes), hops: 24 bits (a chain of
    #            tail (T):  id = the junction node behind it    16.7 M k-mers or more is refused)
    #            finished:  id = the tail of its chain, hops = steps to the tail
    #      byte  bit 0 chain k-mer (non-junction), bit 1 its successor is a junction k-mer, bits 2-3 its outgoing nucleotide
    n2 = 2 * n_mine
    HB = WALK_HOP_BITS  # (24; tests make it small: chains at the limit and loops whose hop counts saturate, on inputs of a few thousand reads)
    FBIT, TBIT, IDM, HM = -(1 << 63), 1 << 62, (1 << (62 - HB)) - 1, (1 << HB) - 1
    if 2 * first[-1] > IDM:
        raise ValueError(f"{first[-1]} k-mers: node ids beyond {62 - HB} bits")
    # (the arrays of the size of the shard's node set live in the library's arena where the engine offers that — GpuEngine.state: after the sharded
    # count the arena holds ~2x the shard and gives nothing back, a 4.3 G-k-mer shard left torch 0 bytes of a 288 GB device, round 5 — and are torch's
    # own on the CPU doubles)
    _state = getattr(engine, "state", None)
    _handles = {}

    def big(name, n, dtype, zero):
        if _state is None:
            return torch.zeros(n, dtype=dtype, device=dev) if zero else torch.empty(n, dtype=dtype, device=dev)
        t, h = _state(n, dtype, dev)
        _handles[name] = h
        return t.zero_() if zero else t

    def free_big(*names):  # (the tensor views of what is freed must be gone)
        for nm in names:
            h = _handles.pop(nm, None)
            if h:
                engine.state_free(h)

    word = big("word", n2, torch.int64, True)
    flag = big("flag", n2, torch.uint8, True)
    # (range by range: the requests of WALK_CHUNK oriented nodes at a time — a k-mer record out and a node id back per request; all at once
    # the exchange buffers of a shard were 48 B per oriented node, the peak of the whole construction)
    for c in range(_rounds_of(n2, WALK_CHUNK, dev)):
        a = min(c * WALK_CHUNK, n2)
        tags, node, junc = lookup(False, a, min(WALK_CHUNK, n2 - a))
        xl = tags >> 4
        word[xl] = torch.where(junc, node << HB | (FBIT | TBIT), node << HB | 1)  # tails know their end node; the others: pointer, one step
        flag[xl] = (1 | (junc.to(torch.int64) << 1) | ((tags & 3) << 2)).to(torch.uint8)
        del node, junc, tags, xl
    n_cand = int(engine.walk_counts()[1])
    c_first = torch.empty(n_cand, dtype=torch.int64, device=dev)
    c_fj = torch.empty(n_cand, dtype=torch.bool, device=dev)
    for c in range(_rounds_of(n_cand, WALK_CHUNK, dev)):
        a = min(c * WALK_CHUNK, n_cand)
        ctags, cfirst, cjunc = lookup(True, a, min(WALK_CHUNK, n_cand - a))
        ci = ctags >> 4
        c_first[ci] = cfirst
        c_fj[ci] = cjunc
        del ctags, cfirst, cjunc, ci

    def is_open(w, f):
        return ((f & 1) != 0) & (w >= 0)

    mark("successor lookups")
    # 2. pointer doubling over the chains
    node_rounds = _rounds_of(n2, WALK_CHUNK, dev)
    prev, rounds = -1, 0
    too_long = False
    def chunks_of_nodes():
        # (every pass over the node array goes chunk by chunk: an elementwise expression over all 2 x |shard| words makes temporaries of that size —
        # 40 GiB each at the 2.7 G k-mers of a 62.5 M-read share, where the first run of this path at that size ran out of memory, round 5)
        for c_ in range(node_rounds):
            yield min(c_ * WALK_CHUNK, n2), min((c_ + 1) * WALK_CHUNK, n2)

    while True:
        tot = torch.zeros(1, dtype=torch.int64, device=dev)
        for a, b in chunks_of_nodes():
            tot += is_open(word[a:b], flag[a:b]).sum()
        dist.all_reduce(tot)
        tot = int(tot.item())
        if tot == 0 or tot == prev:  # every round ends at least one k-mer of every open chain: what is left runs in circles
            break
        prev = tot
        rounds += 1
        mark(f"round {rounds}: {tot} open")
        for c in range(node_rounds):
            a, b = min(c * WALK_CHUNK, n2), min((c + 1) * WALK_CHUNK, n2)
            act = is_open(word[a:b], flag[a:b]).nonzero().squeeze(1) + a
            mine_w = word[act]
            tg = (mine_w >> HB) & IDM
            wp = _remote_rows(tg, owner_of(tg), word.unsqueeze(0), base, rank, world, dev)[:, 0]
            p_tail = (wp & TBIT) != 0                 # the target ends its chain: it is the tail, no step is added
            p_fin = wp < 0
            hops = (mine_w & HM) + torch.where(p_tail, torch.zeros_like(wp), wp & HM)
            # Only a node that FINISHES this round has a chain length to overflow. A node on a perfect loop never finishes and its hop count
            # doubles every round (2^r after r rounds): once an ordinary chain needs ~24 rounds, every plasmid in the input tripped the
            # check although no real chain was that long. Open nodes saturate at HM instead — sticky: a node that finishes with
            # HM or more hops is refused, so a saturated count can never pass for a real one.
            too_long = too_long or (act.numel() > 0 and bool((p_fin & (hops >= HM)).any().item()))
            hops = hops.clamp_(max=HM)
            nid = torch.where(p_tail, tg, (wp >> HB) & IDM)
            word[act] = torch.where(p_fin, torch.full_like(wp, FBIT), torch.zeros_like(wp)) | (nid << HB) | (hops & HM)
            del act, mine_w, tg, wp, p_tail, p_fin, hops, nid

    def check_hops():
        if too_long:
            raise RuntimeError(f"a chain of 2^{HB} k-mers or more: beyond the packed hop count of the distributed walks")
    _guarded(dev, "chain lengths", check_hops)
    left = [is_open(word[a:b], flag[a:b]).nonzero().squeeze(1) + a for a, b in chunks_of_nodes()]
    left = torch.cat(left) if left else torch.empty(0, dtype=torch.int64, device=dev)
    loop_local = torch.unique(left >> 1) if left.numel() else torch.empty(0, dtype=torch.int64, device=dev)
    del left
    def done_of(a, b):  # finished chain k-mers among the nodes [a, b) (computed where it is needed: one byte per node less to hold)
        return ((flag[a:b] & 1) != 0) & (word[a:b] < 0)

    mark("doubling")
    # 3. every chain k-mer to the head of its chain. A node is a head when the reverse strand's node of its k-mer is a tail; the heads'
    #    bookkeeping (chain length, offset of its nucleotides, end node) is kept per HEAD (hidx: their local nodes, ascending), not per node
    hidx = []
    pair_chunk = max(2, WALK_CHUNK // 2 * 2)  # (whole k-mers per chunk: the two nodes of a k-mer are looked at together)
    for a in range(0, n2, pair_chunk):
        b = min(a + pair_chunk, n2)
        rev_tail = ((word[a:b] & TBIT) != 0).view(-1, 2).flip(1).reshape(-1)
        hidx.append((done_of(a, b) & rev_tail).nonzero().squeeze(1) + a)
        del rev_tail
    hidx = torch.cat(hidx) if hidx else torch.empty(0, dtype=torch.int64, device=dev)
    hw = word[hidx]
    hlen = torch.where((hw & TBIT) != 0, torch.zeros_like(hw), hw & HM) + 1  # k-mers of the chain (a head that is its own tail: 1)
    del hw
    hoff = torch.cumsum(hlen, 0) - hlen
    hend = torch.full_like(hlen, -1)
    total = int(hlen.sum().item()) if hidx.numel() else 0
    bases = big("bases", max(total, 1), torch.uint8, True)
    n_heads = hidx.numel()

    def head_slot(local_nodes):
        """ordinal among this rank's heads of local nodes that must be heads (-> slots, all found)"""
        if n_heads == 0:
            return torch.zeros_like(local_nodes), local_nodes.numel() == 0
        slot = torch.searchsorted(hidx, local_nodes).clamp_(max=n_heads - 1)
        return slot, bool((hidx[slot] == local_nodes).all().item())

    def tail_of(nodes, w):  # the tail of the chain of finished nodes (w = their words)
        return torch.where((w & TBIT) != 0, nodes, (w >> HB) & IDM)

    n_got_all, bad_head = 0, False
    for c in range(node_rounds):
        a, b = min(c * WALK_CHUNK, n2), min((c + 1) * WALK_CHUNK, n2)
        xs = done_of(a, b).nonzero().squeeze(1) + a
        xr = xs ^ 1
        wr = word[xr]
        head = tail_of(xr + base, wr) ^ 1
        steps_back = torch.where((wr & TBIT) != 0, torch.zeros_like(wr), wr & HM)  # x is that many k-mers behind the head of its chain
        payload = (steps_back << 2) | ((flag[xs].to(torch.int64) >> 2) & 3)
        # the tails also tell the head which junction node ends the chain (payload: end node << 2 | 3 marks it: no nucleotide code 3 + huge)
        ws = word[xs]
        tl = ((ws & TBIT) != 0).nonzero().squeeze(1)
        head = torch.cat([head, head[tl]])
        payload = torch.cat([payload, -(((ws[tl] >> HB) & IDM) + 1)])  # negative: "the end node of your chain is -(payload) - 1"
        del wr, steps_back, ws, tl
        order, counts = _by_owner(owner_of(head), world)
        msg = torch.stack([head[order], payload[order]], 1).reshape(-1).contiguous()
        del xs, xr, head, payload, order
        got, rcounts = _a2a(msg, [2 * c_ for c_ in counts], rank, world, dev)
        del msg
        n_got = sum(rcounts) // 2
        got = got[:2 * n_got].reshape(-1, 2)
        if n_got:
            slot, ok_ = head_slot(got[:, 0] - base)
            bad_head = bad_head or not ok_
            is_end = got[:, 1] < 0
            e_sl = slot[is_end]
            hend[e_sl] = -got[:, 1][is_end] - 1
            n_sl, n_pl = slot[~is_end], got[:, 1][~is_end]
            pos = hoff[n_sl] + (n_pl >> 2)
            pos.clamp_(0, max(total, 1) - 1)
            bases[pos] = (n_pl & 3).to(torch.uint8)
            n_got_all += int(n_pl.numel())
            del slot, is_end, e_sl, n_sl, n_pl, pos
        del got

    def placed():
        if bad_head:
            raise RuntimeError("a chain nucleotide arrived at a k-mer that heads no chain")
        if n_got_all != total:
            raise RuntimeError(f"{n_got_all} chain nucleotides arrived for chains of {total} k-mers")
        if n_heads and bool((hend < 0).any().item()):
            raise RuntimeError("a chain whose end node never reached its head")
    _guarded(dev, "chain nucleotides at the heads", placed)
    del flag, word
    free_big("word", "flag")  # (the chain nucleotides stay until the chains have been fetched)

    mark("chain nucleotides to the heads")
    # 4. the chains behind this rank's start de-edges
    q = (~c_fj).nonzero().squeeze(1)
    steps = big("steps", max(n_cand, 1), torch.int64, True)[:n_cand]
    last = big("last", max(n_cand, 1), torch.int64, False)[:n_cand]
    last.copy_(c_first)
    boff = big("boff", n_cand + 1, torch.int64, True)
    pieces, have, headless = [], 0, False
    for c in range(_rounds_of(q.numel(), WALK_START_CHUNK, dev)):
        qc = q[c * WALK_START_CHUNK:(c + 1) * WALK_START_CHUNK]
        tq = c_first[qc]
        order, counts = _by_owner(owner_of(tq), world)
        asks, rcounts = _a2a(tq[order].contiguous(), counts, rank, world, dev)
        n_asks = sum(rcounts)
        slot, ok_ = head_slot(asks[:n_asks] - base)
        headless = headless or not ok_
        a_len, a_end, a_off = hlen[slot], hend[slot], hoff[slot]
        if not ok_:  # (reported below, on every rank; nothing may be indexed with a wrong slot's length meanwhile)
            a_len = torch.zeros_like(a_len)
        rows, _ = _a2a(torch.stack([a_len, a_end], 1).reshape(-1).contiguous(), [2 * c_ for c_ in rcounts], rank, world, dev)
        seg_of = torch.repeat_interleave(torch.arange(world, device=dev), torch.tensor(rcounts, dtype=torch.int64, device=dev))
        per_rank = torch.zeros(world, dtype=torch.int64, device=dev)
        if n_asks:
            per_rank.index_add_(0, seg_of, a_len)
        flat = bases[_ragged(a_off, a_len, dev)] if n_asks else torch.empty(0, dtype=torch.uint8, device=dev)
        mine, rc2 = _a2a(flat.contiguous(), [int(v) for v in per_rank.tolist()], rank, world, dev)
        rows = rows[:2 * qc.numel()].reshape(-1, 2)
        qo = qc[order]
        steps[qo] = rows[:, 0]
        last[qo] = rows[:, 1]
        boff[qo] = have + torch.cumsum(rows[:, 0], 0) - rows[:, 0]
        headless = headless or (qc.numel() > 0 and bool((rows[:, 0] <= 0).any().item()))
        pieces.append(mine[:sum(rc2)])
        have += sum(rc2)
        del asks, slot, a_len, a_end, a_off, rows, flat, mine

    def check_chains():
        if headless:
            raise RuntimeError("a start de-edge leads to a k-mer that heads no chain")
    _guarded(dev, "chains behind the start de-edges", check_chains)
    del hidx, hlen, hoff, hend, bases
    free_big("bases")
    # the fetched chains in one array (piece after piece into place: a torch.cat would hold them twice)
    my_bases = big("my_bases", max(have, 1), torch.uint8, have == 0)
    at = 0
    while pieces:
        pc = pieces.pop(0)
        my_bases[at:at + pc.numel()] = pc
        at += pc.numel()
        del pc
    del pieces
    # (no torch.cuda.empty_cache() here: VRAM that one allocator has just released is not safe for the next one to take at once on this
    # stack — arena_trim in csrc/smx_ctx.hpp has the measurements)
    _sync(dev)
    mark("chains of the start de-edges")
    if os.environ.get("SMX_DEBUG") and n_cand:
        print(f"[dist] walks: rank {rank}: {n_cand} start de-edges, steps max {int(steps.max().item())} sum {int(steps.sum().item())}, "
              f"{my_bases.numel()} nucleotides fetched", flush=True)
    unitigs = _guarded(dev, "unitigs of the shard", engine.shard_unitigs, first[rank], steps, last, boff, my_bases, dev)
    del steps, last, boff, my_bases
    free_big("steps", "last", "boff", "my_bases")
    mark("unitigs")
    return unitigs, loop_local, rounds


def sharded_build_graph(engine, k: int, threads: int, rank: int, world: int, dev, coverage: bool = False, route: str = "auto", walks: str = "gathered"):
    """Construction on `world` ranks (collective), owner-side masks. Two routes to the owner's shard of
    {k-mer file, InOutMask bytes}:
      route "ext" (taken by "auto" where the k-mer record has 8 spare bits): every rank extracts the canonical k-mers of ITS reads, each
         with the InOutMask byte those reads give it (the (k+1)-mers around every instance), and ONE all-to-all delivers them to the
         owners of the k-mers, which sort them and OR the bytes of the copies (smx_extract_kmers_ext_owned / smx_graph_shard_from_ext);
      route "kpomers" (the reference's order of work):
      1. sharded count of the canonical (k+1)-mers (sharded_count): every rank owns a bucket range of that file;
      2. every rank turns its shard into extension updates (canonical k-mer, InOutMask bit), grouped by the owner of the K-MER, and
         a second all-to-all delivers them: the mask fill never sees more than the rank's own shard;
      3. the owners sort/unique their k-mers and OR the bits into the mask bytes.
    Then, either way:
      4. the compact structure {k-mer file, masks} (bucket-major, so rank order IS file order) is gathered and every rank derives the
         same unitigs and link records from it (unitig walks cross owners at every step).
    walks = "distributed" ("auto": when the gathered structure would take more than two thirds of some rank's free HBM) replaces step 4 for
    graphs whose gathered structure does not fit one GPU: the k-mer file stays sharded, the
    unitigs come out of distributed_walks() (lookup exchanges + pointer doubling), and only the UNITIGS (2 bits per nucleotide + 25 B per
    unitig, a few % of the k-mer file) and the k-mers of perfect loops are gathered; every rank then derives link records and vertices
    from them (engine.build_graph_from_unitigs). The graph is the same, bit for bit; it has no k-mer file on any rank.
    No rank ever holds the whole (k+1)-mer file, coverage (-c) included: its counters are keyed by (k+1)-mer, so the pass runs shard
    by shard — the owner broadcasts its bucket range, every rank counts its own reads against that shard alone and sums what the
    unitigs' (k+1)-mers of the shard collected; the partial raw edge coverages are all-reduced (SUM mod 2^32). Every rank ends with
    the same graph; rank 0 normally writes it. Returns the engine's graph info."""
    K1, nb = k + 1, 10 * threads
    nw = (K1 + 31) // 32
    ext = route != "kpomers" and hasattr(engine, "shard_from_ext") and engine.ext_supported(k)
    if route == "ext" and not ext:
        raise ValueError(f"k={k}: the one-exchange route needs 8 spare bits in the k-mer record")
    n_kpo, kpo_sizes, kpo_mine, bits, pals = 0, [0] * nb, None, 0, 0

    def count_kpomers():
        res = sharded_count(engine, K1, nb, rank, world, dev)
        return res["distinct"], res["bucket_sizes"], (engine.result_tensor(res["distinct"] * nw, dev) if coverage else None)

    if coverage or not ext:
        n_kpo, kpo_sizes, kpo_mine = count_kpomers()  # (the count result is consumed by the next steps)
    if ext:
        try:
            send, counts = _guarded(dev, "k-mers with extension bytes, grouped by owner", engine.extract_kmers_ext_owned, k, nb, world, dev)
            recv, n_recv = _exchange(engine, send, counts, nw, rank, world, dev, pool=True)
            del send
            if hasattr(engine, "extract_release"):
                engine.extract_release()
            n_kmers, ksizes, bits, pals = _guarded(dev, "owner-side shard from the extension records", engine.shard_from_ext, k, nb, world, rank, recv, n_recv)
            del recv
        except CollectiveFailure as e:
            # one rank's distinct k-mers did not fit where the one-exchange route needs them: EVERY rank heard of it (same exception
            # everywhere) and all take the other route together — the single-GPU library falls back the same way (SMX_ROUTE_NA)
            if not getattr(e, "memory_only", False) or route == "ext":
                raise
            ext = False
            send = recv = None
            # a rank has just run out of HBM: everything the abandoned route left behind goes before the retry — the ranks whose
            # shard_from_ext succeeded hold a shard (k-mers, masks, rank directory), and an unconsumed receive buffer may be resident
            for rel in ("extract_release", "exchange_release", "graph_clear"):
                if hasattr(engine, rel):
                    getattr(engine, rel)()
            # the (k+1)-mer route starts from the count result IN THE CONTEXT: also when -c had counted before, that result went when the
            # abandoned route extracted its k-mers (smx_extract_kmers_ext_owned drops it for room), so it is counted again
            n_kpo, kpo_sizes, kpo_mine = count_kpomers()
    if not ext:
        # 2. extension updates -> owners of the k-mers
        upd = _guarded(dev, "update buffer", engine.alloc, 2 * n_kpo * (nw + 1), dev)
        ucounts = _guarded(dev, "extension updates grouped by owner", engine.shard_updates, k, nb, world, upd, 2 * n_kpo)
        recv, n_recv = _exchange(engine, upd, ucounts, nw + 1, rank, world, dev)
        del upd
        # 3. owner side
        n_kmers, ksizes = _guarded(dev, "owner-side k-mer shard + masks", engine.shard_build, k, nb, world, rank, recv, n_recv)
        del recv
    # 4. gather {k-mers, masks}
    me = torch.tensor([n_kmers, n_kpo, bits, pals] + ksizes + kpo_sizes, dtype=torch.int64, device=dev)
    every = [torch.empty_like(me) for _ in range(world)]
    _all_gather(every, me)
    every = [e.tolist() for e in every]
    kmers_per_rank = [int(e[0]) for e in every]
    kpo_per_rank = [int(e[1]) for e in every]
    g_ksizes = [sum(int(e[4 + b]) for e in every) for b in range(nb)]
    g_psizes = [sum(int(e[4 + nb + b]) for e in every) for b in range(nb)]
    if ext:  # every non-palindromic (k+1)-mer set two extension bits somewhere in the graph, a palindromic one a single bit
        tot = sum(int(e[2]) + int(e[3]) for e in every)
        if tot % 2:
            raise RuntimeError("odd number of extension bits over all shards")
        n_kpo_all = tot // 2
        if coverage and n_kpo_all != sum(kpo_per_rank):
            raise RuntimeError(f"the masks ({n_kpo_all} (k+1)-mers) and the (k+1)-mer count ({sum(kpo_per_rank)}) disagree")
    else:
        n_kpo_all = sum(kpo_per_rank)
    if walks == "auto":
        # the gathered structure costs every rank 8 * words + 1 B per k-mer of the WHOLE graph twice over (torch's gathered copy + the
        # library's), its successor table 16 B more: distributed walks as soon as that leaves some rank less than a third of its free HBM
        need = sum(kmers_per_rank) * (2 * (8 * nw + 1) + 16)
        short = 0
        if dev.type == "cuda":
            # free HBM as this rank's library sees it: the device's figure + what the library's own arena holds unused (it only grows:
            # the room a finished step left inside it is invisible to the device-level number)
            free_b = torch.cuda.mem_get_info(dev)[0] + (engine.arena_free_bytes() if hasattr(engine, "arena_free_bytes") else 0)
            short = 1 if need > (2 * free_b) // 3 else 0
        t_short = torch.tensor([short], dtype=torch.int64, device=dev)
        dist.all_reduce(t_short, op=dist.ReduceOp.MAX)
        walks = "distributed" if int(t_short.item()) else "gathered"
    if walks == "distributed":
        nwk = (k + 31) // 32
        (u_words, u_len, u_st, u_en, u_sf), loop_local, rounds = distributed_walks(engine, k, rank, world, dev, kmers_per_rank)
        first_mine = sum(kmers_per_rank[:rank])
        loop_local = loop_local.contiguous()
        _sync(dev)
        l_k, l_m = _guarded(dev, "k-mers on perfect loops", engine.shard_gather_kmers, loop_local, k, dev)
        me2 = torch.tensor([u_len.numel(), u_words.numel(), loop_local.numel()], dtype=torch.int64, device=dev)
        every2 = [torch.empty_like(me2) for _ in range(world)]
        _all_gather(every2, me2)
        every2 = [e.tolist() for e in every2]
        ne_r, nwd_r, nl_r = ([int(e[i]) for e in every2] for i in range(3))
        g_words = _gather_shards(engine, u_words, u_words.numel(), nwd_r, 1, rank, world, dev, engine.alloc)
        g_len = _gather_shards(engine, u_len, u_len.numel(), ne_r, 1, rank, world, dev, engine.alloc)
        g_st = _gather_shards(engine, u_st, u_st.numel(), ne_r, 1, rank, world, dev, engine.alloc)
        g_en = _gather_shards(engine, u_en, u_en.numel(), ne_r, 1, rank, world, dev, engine.alloc)
        g_sf = _gather_shards(engine, u_sf, u_sf.numel(), ne_r, 1, rank, world, dev, engine.alloc_bytes)
        g_lr = _gather_shards(engine, (loop_local + first_mine).contiguous(), loop_local.numel(), nl_r, 1, rank, world, dev, engine.alloc)
        g_lk = _gather_shards(engine, l_k, loop_local.numel(), nl_r, nwk, rank, world, dev, engine.alloc)
        g_lm = _gather_shards(engine, l_m, loop_local.numel(), nl_r, 1, rank, world, dev, engine.alloc_bytes)
        del u_words, u_len, u_st, u_en, u_sf, l_k, l_m
        _sync(dev)
        import numpy as np
        nl = sum(nl_r)
        h_lr = g_lr[:nl].cpu().numpy().view(np.uint64).copy()
        h_lk = g_lk[:nl * nwk].cpu().numpy().view(np.uint64).copy()
        h_lm = g_lm[:nl].cpu().numpy().copy()
        info = _guarded(dev, "graph from the gathered unitigs", engine.build_graph_from_unitigs, k, nb, sum(kmers_per_rank), n_kpo_all, g_words, sum(nwd_r),
                        g_len, g_st, g_en, g_sf, sum(ne_r), h_lr, h_lk, h_lm)
        info["walk_rounds"] = rounds
        info["unitigs_per_rank"] = ne_r
        del g_words, g_len, g_st, g_en, g_sf
    elif walks != "gathered":
        raise ValueError(f"walks = {walks!r}: 'gathered', 'distributed' or 'auto'")
    else:
        info = None

    def my_shard():
        if hasattr(engine, "trim"):
            engine.trim()  # (a no-op with the default arena: smx_trim in include/smx.h)
        a, b = engine.alloc(n_kmers * nw, dev), engine.alloc_bytes(n_kmers, dev)
        engine.shard_copy(a, b)
        return a, b

    if info is None:
        my_k, my_m = _guarded(dev, "copy of the owner-side shard", my_shard)
        full_k = _gather_shards(engine, my_k, n_kmers, kmers_per_rank, nw, rank, world, dev, engine.alloc)
        full_m = _gather_shards(engine, my_m, n_kmers, kmers_per_rank, 1, rank, world, dev, engine.alloc_bytes)
        del my_k, my_m
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()
        info = _guarded(dev, "graph from the gathered k-mers + masks", engine.build_graph_from_kmers, k, nb, full_k, full_m, sum(kmers_per_rank), g_ksizes, n_kpo_all)
        del full_k, full_m
    if coverage:
        # The counters of the coverage pass are keyed by (k+1)-mer, and no rank may hold that whole file (16 B per (k+1)-mer of the WHOLE
        # graph: 8 x what a rank owns at world 8). So the pass runs shard by shard: the owner of a bucket range broadcasts its shard, every
        # rank installs it as a (k+1)-mer file that holds those buckets only (the lookups of the coverage kernels miss everything else and
        # skip it), counts ITS OWN reads against it and sums what the unitigs' (k+1)-mers of that shard collected. An edge's coverage is a
        # sum over its (k+1)-mers and over the ranks' reads: the partial sums add up (mod 2^32, as the counters do). Never more than one
        # shard of the file next to the graph — at the price of one pass over the reads and the unitigs per shard.
        n_e = info["n_unitigs"]
        cov = torch.zeros(max(n_e, 1), dtype=torch.int64, device=dev)
        for s_ in range(world):
            n_s = kpo_per_rank[s_]
            if n_s == 0:
                continue
            b0, b1 = rank_first_bucket(nb, world, s_), rank_first_bucket(nb, world, s_ + 1)
            sizes_s = [g_psizes[b] if b0 <= b < b1 else 0 for b in range(nb)]
            if sum(sizes_s) != n_s:
                raise RuntimeError(f"rank {s_} owns {n_s} (k+1)-mers, its buckets hold {sum(sizes_s)}")
            shard = kpo_mine if s_ == rank else engine.alloc(n_s * nw, dev)
            _bcast_chunks(shard, 0, n_s * nw, s_)
            if dev.type == "cuda":
                torch.cuda.current_stream(dev).synchronize()
            engine.set_kpomers(shard, n_s, sizes_s)
            if s_ != rank:
                del shard
            part = engine.local_raw_coverage(n_e).to(torch.int64) & 0xFFFFFFFF
            cov[:n_e] += part.to(dev)
        cov &= 0xFFFFFFFF
        dist.all_reduce(cov, op=dist.ReduceOp.SUM)
        cov = cov[:n_e]
        cov = (cov & 0xFFFFFFFF).to("cpu")
        cov = torch.where(cov >= 2 ** 31, cov - 2 ** 32, cov).to(torch.int32)  # uint32 bit pattern
        engine.set_raw_coverage(cov)
    info["route"] = "ext" if ext else "kpomers"
    info["kpomers_per_rank"] = kpo_per_rank
    info["kmers_per_rank"] = kmers_per_rank
    info["walks"] = walks
    return info
