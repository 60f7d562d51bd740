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


class _DevBytes:
    """uint8 view of library-owned HBM (the segments of smx_shard_walks' exchanges: elements of 1, 8, 16 or 8 * words bytes)"""

    def __init__(self, p: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (p, False), "version": 2}


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

    def shard_walks(self, k: int, rank: int, world: int, dev, kmers_per_rank):
        """smx_shard_walks: the whole of the distributed walks inside the library, on the device; this method lends it the three collectives
        (torch.distributed on views of the library's own buffers — counts, segments of bytes, a few words) and fetches what it leaves.
        -> ((packed unitig words, lengths, start nodes, end nodes, self-conjugate flags), local ranks of the k-mers on perfect loops, doubling rounds)"""
        u64 = C.POINTER(C.c_uint64)
        XC = C.CFUNCTYPE(C.c_int, C.c_void_p, u64, u64)
        A2A = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, u64, C.c_void_p, u64, C.c_uint)
        AR = C.CFUNCTYPE(C.c_int, C.c_void_p, u64, C.c_uint, C.c_int)
        failure = []

        def guard(fn):
            def g(*a):
                try:
                    fn(*a)
                    return 0
                except Exception as e:  # noqa: BLE001 — a Python exception must not unwind through the C frames of the library
                    failure.append(e)
                    return 1
            return g

        def exchange_counts(_u, send_c, recv_c):
            got = _exchange_counts([int(send_c[p]) & 0x7FFFFFFFFFFFFFFF if int(send_c[p]) != 0xFFFFFFFFFFFFFFFF else -1 for p in range(world)], dev)
            for p in range(world):
                recv_c[p] = got[p] & 0xFFFFFFFFFFFFFFFF  # (-1: the poison of a failed rank, all ones again)

        def alltoallv(_u, d_send, send_c, d_recv, recv_c, unit):
            sc, rc = [int(send_c[p]) * unit for p in range(world)], [int(recv_c[p]) * unit for p in range(world)]
            send = torch.as_tensor(_DevBytes(d_send, max(sum(sc), 1)), device=dev)
            recv = torch.as_tensor(_DevBytes(d_recv, max(sum(rc), 1)), device=dev)
            _p2p_known(send, sc, recv, rc, rank, world, dev, XCHG_LIMIT * 8)

        def allreduce(_u, vals, n, op):
            t = torch.tensor([int(vals[i]) for i in range(n)], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)
            for i, v in enumerate(t.tolist()):
                vals[i] = int(v)

        class Coll(C.Structure):
            _fields_ = [("user", C.c_void_p), ("rank", C.c_uint), ("world", C.c_uint), ("exchange_counts", XC), ("alltoallv", A2A), ("allreduce_u64", AR)]

        coll = Coll(None, rank, world, XC(guard(exchange_counts)), A2A(guard(alltoallv)), AR(guard(allreduce)))
        per = (C.c_uint64 * world)(*[int(c) for c in kmers_per_rank])
        info = (C.c_uint64 * 4)()
        _sync(dev)
        fn = self.ctx.lib.smx_shard_walks
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, u64, C.POINTER(Coll), u64]
        rc = fn(self.ctx._h, per, C.byref(coll), info)
        if failure:
            raise failure[0]
        _chk(self.ctx._h, rc)
        ne, nwords, nloop, rounds = (int(v) for v in info)
        words = torch.empty(max(nwords, 1), dtype=torch.int64, device=dev)
        ln, st, en = (torch.empty(max(ne, 1), dtype=torch.int64, device=dev) for _ in range(3))
        sf = torch.empty(max(ne, 1), dtype=torch.uint8, device=dev)
        loops = torch.empty(max(nloop, 1), dtype=torch.int64, device=dev)
        _sync(dev)
        _chk(self.ctx._h, self.ctx.lib.smx_shard_unitigs_copy(self.ctx._h, C.cast(words.data_ptr(), u64), C.cast(ln.data_ptr(), u64), C.cast(st.data_ptr(), u64),
                                                             C.cast(en.data_ptr(), u64), C.cast(sf.data_ptr(), C.POINTER(C.c_uint8))))
        _chk(self.ctx._h, self.ctx.lib.smx_shard_walk_loops(self.ctx._h, C.cast(loops.data_ptr(), u64)))
        return (words[:nwords], ln[:ne], st[:ne], en[:ne], sf[:ne]), loops[:nloop], rounds

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


def _a2a_known(send: torch.Tensor, counts, recv: torch.Tensor, rcounts, rank: int, world: int, dev, limit: int = None):
    """the data of ONE all-to-all of a 1-D tensor when both sides' counts are known: counts[p] ELEMENTS of send go to rank p, rcounts[p] arrive
    from it. Splits are capped at `limit` elements per (pair, round) (1 GiB: one all_to_all_single of a 30 GB buffer silently truncates on
    this stack — measured: tail left untouched), so large segments go in several rounds of views (no staging copies); every pair still
    moves each element exactly once."""
    n_sent, n_recv = sum(counts), sum(rcounts)
    limit = limit or max(1, XCHG_LIMIT * 8 // max(send.element_size(), 1))
    soff = [0]
    for c in counts:
        soff.append(soff[-1] + c)
    roff = [0]
    for c in rcounts:
        roff.append(roff[-1] + c)
    mx = torch.tensor([max(counts) if counts else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    rounds = max(1, -(-int(mx.item()) // limit))
    if rounds == 1:
        _all_to_all_single(recv[:n_recv], send[:n_sent], output_split_sizes=list(rcounts), input_split_sizes=list(counts))
    else:
        # grouped point-to-point rounds on views (ncclSend/ncclRecv pairs under one group on RCCL): every pair has its own
        # xGMI link, there is no ring to serialise on, and no staging copy is needed
        for r in range(rounds):
            ops = []
            for p in range(world):
                a = min(soff[p] + r * limit, soff[p + 1])
                b = min(a + limit, soff[p + 1])
                c = min(roff[p] + r * limit, roff[p + 1])
                d = min(c + limit, roff[p + 1])
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


def _p2p_known(send: torch.Tensor, counts, recv: torch.Tensor, rcounts, rank: int, world: int, dev, limit: int):
    """the same movement by grouped point-to-point operations only: both ends of a pair know that pair's size and cut it into the same pieces of
    `limit` elements, so no rank needs to agree with all the others on a number of rounds (no all-reduce per exchange — smx_shard_walks makes
    thousands of small ones), and the segment that stays on this rank is a device copy, as in the C++ host (tools/gbuilder_mgpu.hpp)."""
    soff, roff = [0], [0]
    for c in counts:
        soff.append(soff[-1] + c)
    for c in rcounts:
        roff.append(roff[-1] + c)
    if counts[rank]:
        recv[roff[rank]:roff[rank + 1]].copy_(send[soff[rank]:soff[rank + 1]])
    staged = _staged(send)
    r = 0
    while True:
        ops, keep = [], []
        for p in range(world):
            if p == rank:
                continue
            a, c = min(soff[p] + r * limit, soff[p + 1]), min(roff[p] + r * limit, roff[p + 1])
            b, d = min(a + limit, soff[p + 1]), min(c + limit, roff[p + 1])
            if b > a:
                ops.append(dist.P2POp(dist.isend, send[a:b].cpu() if staged else send[a:b], p))
            if d > c:
                if staged:  # (ranks that share a GPU run on gloo: through host memory)
                    h = torch.empty(d - c, dtype=recv.dtype)
                    keep.append((c, d, h))
                    ops.append(dist.P2POp(dist.irecv, h, p))
                else:
                    ops.append(dist.P2POp(dist.irecv, recv[c:d], p))
        if not ops:
            break
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        for c, d, h in keep:
            recv[c:d].copy_(h)
        r += 1
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()


def _exchange_counts(counts, dev):
    cnt_t = torch.tensor(list(counts), dtype=torch.int64, device=dev)
    rcv_t = torch.empty_like(cnt_t)
    _all_to_all_single(rcv_t, cnt_t)
    return [int(c) for c in rcv_t.tolist()]


def _a2a(send: torch.Tensor, counts, rank: int, world: int, dev, alloc=None):
    """ONE all-to-all of a 1-D tensor: counts[p] ELEMENTS go to rank p. Returns (recv, elements received from every rank)."""
    rcounts = _exchange_counts(counts, dev)
    n_recv = sum(rcounts)
    recv = alloc(n_recv) if alloc is not None else torch.empty(max(n_recv, 1), dtype=send.dtype, device=dev)
    _a2a_known(send, counts, recv, rcounts, rank, world, dev, limit=XCHG_LIMIT)
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


def distributed_walks(engine, k: int, rank: int, world: int, dev, kmers_per_rank):
    """Unitigs of a graph whose k-mer file stays sharded (collective). Every rank holds its bucket range of
    {k-mer file, InOutMask bytes}; a walk of the reference (debruijn_graph_constructor.hpp:264-273) would change rank at every step, so
    nothing is walked:
      1. every oriented non-junction k-mer learns its successor node and whether that is a junction k-mer (ONE lookup exchange: the
         canonical successor k-mer travels to its owner, a node id comes back), every start de-edge its first node (a second one);
      2. the chains of non-junction k-mers are ranked by pointer doubling — per round one exchange of node ids and one of node words,
         ceil(log2(longest chain)) rounds at most; what never finishes lies on perfect loops;
      3. a chain k-mer x is hops(x^1) steps behind the head tail(x^1)^1 of its chain (the reverse strand went through the same
         doubling): it sends its outgoing nucleotide there, and the owner of the head lays the chain's nucleotides out in order;
      4. the owner of a start de-edge fetches length, end node and nucleotides of the chain behind it and assembles, keeps or drops the
         unitig exactly as the single-GPU route does.
    ALL of it runs inside the library, on the device (smx_shard_walks: csrc/smx_dwalk.hpp, kernels in csrc/smx_dwalk.hip); this host only
    lends it three collectives (counts, all-to-all of device segments, all-reduce of a few words) — no tensor code of a framework between the
    first exchange and the gather of the unitigs (rounds 3-5 did steps 1-4's plumbing with torch sort / searchsorted / index kernels:
    19.8 s for one rank's share of BASELINE config 4). The C++ host drives the same entry point over grouped ncclSend / ncclRecv
    (tools/gbuilder_mgpu.hpp).
    Returns this rank's kept unitigs (k-mer-file order of their start k-mers: concatenated in rank order they are the reference's edge
    list), the local ranks of its k-mers on perfect loops, and the number of doubling rounds."""
    return engine.shard_walks(k, rank, world, dev, kmers_per_rank)


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
