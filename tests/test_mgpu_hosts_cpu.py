"""CPU, world 2-3, gloo: the C++ multi-GPU hosts of the CLI tools (spades_amd/tools/gbuilder_mgpu.hpp, kmercount_mgpu.hpp), compiled as
they are into tests/mgpu_shim (hip* = host memory; nccl* and the product's C ABI forwarded to this file, which answers with gloo
collectives and with the oracle-backed engine doubles of test_dist_cpu.py). Under test: the hosts' own logic at world size > 1 — who
reads which part of the input, offsets / counts / rounds of the exchanges, the order of the gathered shards, bucket-size and (k+1)-mer
bookkeeping, the coverage sum, who writes. (On the GPU box the same hosts run with one rank against the real library: test_zz_cli_rccl_gpu.py.)"""
import ctypes
import os
import subprocess
import sys
import traceback

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, free_port, read_lines

SHIM_DIR = os.path.join(ROOT, "tests", "mgpu_shim")
SHIM = os.path.join(SHIM_DIR, "libmgpu_shim.so")
HAVE_HEADERS = os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h") and os.path.exists("/opt/rocm/include/rccl/rccl.h")
pytestmark = pytest.mark.skipif(not HAVE_HEADERS, reason="the hosts include the HIP and RCCL headers (declarations only)")


def _build_shim():
    srcs = [os.path.join(SHIM_DIR, "shim.cpp")] + [os.path.join(ROOT, "spades_amd", "tools", f) for f in
                                                   ("gbuilder_mgpu.hpp", "kmercount_mgpu.hpp", "read_input.hpp", "read_share.hpp", "fastq_split.hpp", "bgzf_reader.hpp", "rank_watchdog.hpp")]
    if os.path.exists(SHIM) and all(os.path.getmtime(SHIM) >= os.path.getmtime(s) for s in srcs):
        return
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", SHIM + ".tmp", srcs[0],
                           "-lz", "-pthread"])
    os.replace(SHIM + ".tmp", SHIM)


def _u64(ptr, n):
    return np.ctypeslib.as_array((ctypes.c_uint64 * max(n, 1)).from_address(ptr))[:n]


def _u32(ptr, n):
    return np.ctypeslib.as_array((ctypes.c_uint32 * max(n, 1)).from_address(ptr))[:n]


def _u8(ptr, n):
    return np.ctypeslib.as_array((ctypes.c_uint8 * max(n, 1)).from_address(ptr))[:n]


def _i64_tensor(ptr, n):
    """torch view (int64) of n words at ptr: what the engine doubles read and write"""
    return torch.from_numpy(_u64(ptr, n).view(np.int64)) if n else torch.empty(0, dtype=torch.int64)


class ShimRank:
    """Answers the forwarded calls of ONE rank. smx_*: an engine double fed with the reads the host submitted; nccl*: gloo."""

    def __init__(self, rank, world, all_reads, mode):
        self.rank, self.world, self.all_reads, self.mode = rank, world, all_reads, mode
        self.reads, self.eng, self.keep, self.ops = [], None, [], None
        self.count, self.shard, self.info, self.cov_local, self.written = None, None, None, None, None
        self.calls, self.p2p_bytes, self.p2p_ops, self.error = [], 0, 0, ""

    def engine(self):
        if self.eng is None:
            from test_dist_cpu import OracleEngine, OracleWalkEngine
            self.eng = OracleWalkEngine(self.reads, self.all_reads) if self.mode == "B" else OracleEngine(self.reads, "A")
        return self.eng

    # ---- dispatch ----
    def __call__(self, name, a):
        name = name.decode()
        self.calls.append(name)
        inj = os.environ.get("SHIM_FAIL", "")  # "rank:function:code[,...]": that call fails once on that rank, as the library would
        for item in [x for x in inj.split(",") if x]:
            r, fn, code = item.split(":")
            if int(r) == self.rank and fn == name and (fn, "done") not in self.keep:
                self.keep.append((fn, "done"))
                return int(code)
        try:
            return int(getattr(self, "f_" + name)(*[int(a[i]) for i in range(10)]) or 0)
        except Exception:  # noqa: BLE001 — report through the C ABI's error code, as the library would
            self.error = traceback.format_exc()
            sys.stderr.write(f"[rank {self.rank}] {name}:\n{self.error}\n")
            return 70

    # ---- RCCL over gloo ----
    def f_ncclCommInitRank(self, world, rank, *_):
        assert (world, rank) == (self.world, self.rank)

    def f_ncclGroupStart(self, *_):
        assert self.ops is None
        self.ops = []

    def f_ncclSend(self, buf, nbytes, peer, *_):
        assert self.ops is not None and peer != self.rank and nbytes > 0  # (the hosts copy their own segment)
        self.ops.append(dist.P2POp(dist.isend, torch.from_numpy(_u8(buf, nbytes)), peer))
        self.p2p_bytes += nbytes
        self.p2p_ops += 1

    def f_ncclRecv(self, buf, nbytes, peer, *_):
        assert self.ops is not None and peer != self.rank and nbytes > 0
        self.ops.append(dist.P2POp(dist.irecv, torch.from_numpy(_u8(buf, nbytes)), peer))

    def f_ncclGroupEnd(self, *_):
        ops, self.ops = self.ops, None
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    def f_ncclAllGather(self, src, dst, nbytes, *_):
        mine = torch.from_numpy(_u8(src, nbytes).copy())
        outs = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
        dist.all_gather(outs, mine)
        _u8(dst, nbytes * self.world)[:] = torch.cat(outs).numpy()

    def f_ncclAllReduceU64(self, src, dst, count, is_max, *_):
        t = torch.from_numpy(_u64(src, count).astype(np.int64))  # (values below 2^62, or all ones = -1 under max: the order is kept)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if is_max else dist.ReduceOp.SUM)
        _u64(dst, count)[:] = t.numpy().view(np.uint64)

    def f_ncclAllReduceU32Sum(self, src, dst, count, *_):
        t = torch.from_numpy(_u32(src, count).astype(np.int64))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        _u32(dst, count)[:] = (t.numpy() & 0xFFFFFFFF).astype(np.uint32)

    # ---- input ----
    def f_smx_create(self, device, *_):
        assert device == self.rank

    def f_smx_submit_fastq_text(self, text, n, is_final, p_reads, p_used, *_):
        data = bytes(_u8(text, n))
        lines = data.split(b"\n")
        complete = len(lines) - 1  # lines that end in a newline
        recs = complete // 4
        used = sum(len(x) + 1 for x in lines[:4 * recs])
        got = [lines[4 * i + 1].decode() for i in range(recs)]
        for i in range(recs):
            assert lines[4 * i][:1] == b"@" and lines[4 * i + 2][:1] == b"+" and len(lines[4 * i + 1]) == len(lines[4 * i + 3]), "a chunk must start at a record"
        if is_final and used < n:  # the last record may lack its newline
            rest = data[used:].split(b"\n")
            assert len(rest) == 4 and rest[0][:1] == b"@" and rest[2][:1] == b"+" and len(rest[1]) == len(rest[3])
            got.append(rest[1].decode())
            used = n
        self.reads += got
        _u64(p_reads, 1)[0] = len(got)
        _u64(p_used, 1)[0] = used

    def f_smx_submit_reads_ascii(self, bases, off, n, *_):
        o = _u64(off, n + 1)
        b = bytes(_u8(bases, int(o[n]))) if n else b""
        self.reads += [b[int(o[i]):int(o[i + 1])].decode() for i in range(n)]

    # ---- count ----
    def f_smx_kmers_with_masks_supported(self, k, *_):
        return 1 if (k >= 21 and 2 * k + 8 <= 64 * ((k + 31) // 32)) else 0

    def f_smx_extract_partition_owned(self, K, mode, nb, world, p_ptr, p_counts, *_):
        assert mode == (1 if self.mode == "B" else 0) and world == self.world
        eng = self.engine()
        n, nw = eng.extract_count(K), (K + 31) // 32
        buf = torch.zeros(max(n * nw, 1), dtype=torch.int64)
        counts = eng.extract_partition(K, nb, world, buf, n)
        self.keep.append(buf)
        _u64(p_ptr, 1)[0] = buf.data_ptr()
        _u64(p_counts, world)[:] = counts

    def f_smx_extract_release(self, *_):
        pass

    def f_smx_exchange_release(self, *_):
        pass

    def f_smx_graph_clear(self, *_):
        self.shard = None

    def f_smx_exchange_buffer(self, n_words, p_ptr, *_):
        buf = torch.full((max(n_words, 1),), -1, dtype=torch.int64)  # (words the exchange does not fill would show up as k-mers)
        self.keep.append(buf)
        _u64(p_ptr, 1)[0] = buf.data_ptr()

    def f_smx_count_records(self, K, nb, d, n, *_):
        nw = (K + 31) // 32
        self.count = dict(self.engine().count_records(K, nb, _i64_tensor(d, n * nw), n), K=K, nb=nb, nw=nw)

    def f_smx_count_info(self, p_n, p_wpr, p_inst, *_):
        if p_n:
            _u64(p_n, 1)[0] = self.count["distinct"]
        if p_wpr:
            _u32(p_wpr, 1)[0] = self.count["nw"]
        if p_inst:
            _u64(p_inst, 1)[0] = self.count["instances"]

    def f_smx_bucket_sizes(self, p_sizes, *_):
        _u64(p_sizes, self.count["nb"])[:] = self.count["bucket_sizes"]

    def f_smx_copy_bucket(self, b, dst, *_):
        off = np.concatenate([[0], np.cumsum(self.count["bucket_sizes"])])
        rows = self.eng.result[int(off[b]):int(off[b + 1])]
        _u64(dst, rows.size)[:] = rows.reshape(-1)

    def f_smx_copy_kmers_device(self, d, *_):
        _u64(d, self.eng.result.size)[:] = self.eng.result.reshape(-1)

    # ---- owner-side shard ----
    def f_smx_extract_kmers_ext_owned(self, k, nb, world, p_ptr, p_counts, *_):
        t, counts = self.engine().extract_kmers_ext_owned(k, nb, world, None)
        t = t.contiguous()
        self.keep.append(t)
        _u64(p_ptr, 1)[0] = t.data_ptr()
        _u64(p_counts, world)[:] = counts

    def f_smx_graph_shard_from_ext(self, k, nb, world, rank, d, n, *_):
        assert (world, rank) == (self.world, self.rank)
        nk, sizes, bits, pals = self.engine().shard_from_ext(k, nb, world, rank, _i64_tensor(d, n * ((k + 31) // 32)), n)
        self.shard = dict(n=nk, sizes=sizes, bits=bits, pals=pals, nb=nb)
        self.kk, self.eng.nb, self.shard_n = k, nb, nk

    def f_smx_graph_shard_ext_stats(self, p_st, *_):
        _u64(p_st, 2)[:] = [self.shard["bits"], self.shard["pals"]]

    def f_smx_graph_shard_updates(self, k, nb, world, d, cap, p_counts, *_):
        counts = self.engine().shard_updates(k, nb, world, _i64_tensor(d, cap * ((k + 31) // 32 + 1)), cap)
        _u64(p_counts, world)[:] = counts

    def f_smx_graph_shard_build(self, k, nb, world, rank, d, n, *_):
        assert (world, rank) == (self.world, self.rank)
        nk, sizes = self.engine().shard_build(k, nb, world, rank, _i64_tensor(d, n * ((k + 31) // 32 + 1)), n)
        self.shard = dict(n=nk, sizes=sizes, bits=0, pals=0, nb=nb)
        self.kk, self.eng.nb, self.shard_n = k, nb, nk

    def f_smx_graph_shard_info(self, p_n, p_sizes, *_):
        _u64(p_n, 1)[0] = self.shard["n"]
        if p_sizes:
            _u64(p_sizes, self.shard["nb"])[:] = self.shard["sizes"]

    def f_smx_graph_shard_copy(self, dk, dm, *_):
        e = self.eng
        e.shard_copy(_i64_tensor(dk, e.shard_kmers.size), torch.from_numpy(_u8(dm, e.shard_masks.size)))

    # ---- graph ----
    def f_smx_build_graph_from_kmers(self, k, nb, dk, dm, n, p_sizes, n_kpo, *_):
        nw = (k + 31) // 32
        self.info = self.engine().build_graph_from_kmers(k, nb, _i64_tensor(dk, n * nw), torch.from_numpy(_u8(dm, n)), n, [int(x) for x in _u64(p_sizes, nb)], n_kpo)

    # ---- distributed walks: smx_shard_walks answered by the torch restatement, its data moved by the HOST's own collectives ----
    def f_smx_shard_walks(self, p_per, user, rank, world, f_counts, f_a2a, f_allreduce, p_info, *_):
        import dwalk_torch_double as dbl
        assert (rank, world) == (self.rank, self.world)
        u64p = ctypes.POINTER(ctypes.c_uint64)
        xc = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, u64p, u64p)(f_counts)
        a2av = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, u64p, ctypes.c_void_p, u64p, ctypes.c_uint)(f_a2a)
        ar = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, u64p, ctypes.c_uint, ctypes.c_int)(f_allreduce)
        per = [int(x) for x in _u64(p_per, world)]
        eng = self.engine()
        assert per[rank] == len(eng.shard_kmers)
        self.walk_a2a = 0

        def a2a(send, counts, rank_, world_, dev, alloc=None):
            sc, rc = (ctypes.c_uint64 * world)(*[int(c) for c in counts]), (ctypes.c_uint64 * world)()
            assert xc(user, sc, rc) == 0
            rcounts = [int(c) for c in rc]
            send = send.contiguous()
            recv = torch.empty(max(sum(rcounts), 1), dtype=send.dtype)
            assert a2av(user, send.data_ptr(), sc, recv.data_ptr(), rc, send.element_size()) == 0
            self.walk_a2a += 1
            return recv, rcounts

        def all_reduce(t, op=dist.ReduceOp.SUM):
            n = t.numel()
            vals = (ctypes.c_uint64 * n)(*[int(v) for v in t.reshape(-1).tolist()])
            assert ar(user, vals, n, 1 if op == dist.ReduceOp.MAX else 0) == 0
            t.copy_(torch.tensor([int(v) for v in vals], dtype=t.dtype).reshape(t.shape))

        if os.environ.get("SHIM_WALK_CHUNK"):
            dbl.WALK_CHUNK, dbl.WALK_START_CHUNK = [int(v) for v in os.environ["SHIM_WALK_CHUNK"].split(",")]
        self.walk = dbl.torch_walks(eng, self.kk, rank, world, torch.device("cpu"), per, a2a=a2a, all_reduce=all_reduce)
        (w, ln, st, en, sf), loops, rounds = self.walk
        _u64(p_info, 4)[:] = [ln.numel(), w.numel(), loops.numel(), rounds]

    def f_smx_shard_unitigs_copy(self, pw, pl, ps, pe, pf, *_):
        (w, ln, st, en, sf), _loops, _r = self.walk
        for p_, t in ((pw, w), (pl, ln), (ps, st), (pe, en)):
            _u64(p_, t.numel())[:] = t.numpy().view(np.uint64)
        _u8(pf, sf.numel())[:] = sf.numpy()

    def f_smx_shard_walk_loops(self, p, *_):
        loops = self.walk[1]
        _u64(p, loops.numel())[:] = loops.numpy().view(np.uint64)

    def f_smx_shard_gather_kmers(self, p_ranks, n, dk, dm, *_):
        km, mk = self.eng.shard_gather_kmers(torch.from_numpy(_u64(p_ranks, n).astype(np.int64)), self.kk, None)
        if n:
            nw = (self.kk + 31) // 32
            _u64(dk, n * nw)[:] = km.numpy().view(np.uint64)[:n * nw]
            _u8(dm, n)[:] = mk.numpy()[:n]

    def f_smx_build_graph_from_unitigs(self, p_args, *_):
        a = [int(v) for v in np.ctypeslib.as_array((ctypes.c_longlong * 15).from_address(p_args))]
        k, nb, n_kmers, n_kpo, pw, n_words, pl, ps, pe, pf, ne, plr, plk, plm, nl = a
        nw = (k + 31) // 32
        self.info = self.eng.build_graph_from_unitigs(k, nb, n_kmers, n_kpo, _i64_tensor(pw, n_words), n_words, _i64_tensor(pl, ne), _i64_tensor(ps, ne), _i64_tensor(pe, ne),
                                                      torch.from_numpy(_u8(pf, ne)), ne, _u64(plr, nl).copy(), _u64(plk, nl * nw).copy(), _u8(plm, nl).copy())

    def f_smx_graph_info(self, p_info, *_):
        i = self.info
        _u64(p_info, 8)[:] = [i["n_kpomers"], i["n_kmers"], i["n_unitigs"], i["n_loops"], i["n_vertices"], 0, i["unitig_bases"], i["words"]]

    def f_smx_graph_set_kpomers(self, d, n, p_sizes, *_):
        nw = (self.eng.k + 1 + 31) // 32
        assert sum(int(x) for x in _u64(p_sizes, self.count["nb"])) == n
        self.eng.set_kpomers(_i64_tensor(d, n * nw), n, None)

    def f_smx_graph_fill_coverage(self, *_):
        self.cov_local = self.eng.local_raw_coverage(self.info["n_unitigs"]).numpy().astype(np.uint32)

    def f_smx_graph_copy_coverage(self, p_raw, *_):
        _u32(p_raw, len(self.cov_local))[:] = self.cov_local

    def f_smx_graph_set_coverage(self, p_raw, n, *_):
        assert n == self.info["n_unitigs"]
        self.eng.set_raw_coverage(torch.from_numpy(_u32(p_raw, n).astype(np.int64)))

    def f_smx_graph_write(self, path, kind, *_):
        self.written = (ctypes.string_at(path).decode(), kind)
        with open(self.written[0], "w") as f:
            f.write(self.eng.g["gfa"])


CB = ctypes.CFUNCTYPE(ctypes.c_longlong, ctypes.c_char_p, ctypes.POINTER(ctypes.c_longlong))


def _worker(rank, world, port, tool, args, env, all_reads, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), **env)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = ctypes.CDLL(SHIM)
    me = ShimRank(rank, world, all_reads, "B" if tool == "gbuilder" else "A")
    cb = CB(me)
    lib.mgpu_shim_set_callback(cb)
    if tool == "gbuilder":
        k, threads, coverage, files, out = args
        rc = lib.mgpu_shim_run_gbuilder(rank, world, k, threads, int(coverage), 1, out.encode(), "\n".join(files).encode(), (out + ".id").encode())
        res = dict(rank=rank, rc=rc, reads=me.reads, written=me.written, built=me.info is not None,
                   cov=me.eng.cov.tobytes() if getattr(me.eng, "cov", None) is not None else None, p2p_ops=me.p2p_ops, p2p_bytes=me.p2p_bytes,
                   shard=getattr(me, "shard_n", None), kpo=me.count["distinct"] if me.count else None, calls=me.calls, error=me.error)
    else:
        K, workdir, files = args
        rc = lib.mgpu_shim_run_kmercount(rank, world, K, workdir.encode(), "\n".join(files).encode())
        res = dict(rank=rank, rc=rc, reads=me.reads, p2p_ops=me.p2p_ops, owned=me.count["distinct"] if me.count else None, error=me.error)
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def _run(world, tool, args, env, all_reads, port_base):
    _build_shim()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()  # (port_base: kept in the signature for the callers)
    procs = [ctx.Process(target=_worker, args=(r, world, port, tool, args, env, all_reads, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    got, t0 = [], time.time()
    try:
        while len(got) < world:
            try:
                got.append(q.get(timeout=2))
            except queue.Empty:
                dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
                assert not dead, f"a rank died with exit code {dead} (the others would wait for it)"
                assert time.time() - t0 < 300, "the ranks did not finish"
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    return sorted(got, key=lambda d: d["rank"])


def _write_fastq(path, reads, final_newline=True):
    text = "".join(f"@r{i} x\n{r}\n+\n{'@' * len(r)}\n" for i, r in enumerate(reads))  # quality lines that begin with '@'
    with open(path, "w") as f:
        f.write(text if final_newline else text[:-1])


@pytest.mark.parametrize("world,k,coverage,fmt,env", [
    (2, 21, False, "fq", {"SMX_MGPU_ROUND_WORDS": "96", "SMX_MGPU_CHUNK": "1500"}),
    (2, 21, True, "fq", {"SMX_MGPU_ROUND_WORDS": "4096"}),
    (3, 33, True, "fa", {"SMX_MGPU_ROUND_WORDS": "200"}),
    (3, 21, False, "fq", {"SMX_MGPU_KPOMERS": "1", "SMX_MGPU_ROUND_WORDS": "128", "SMX_MGPU_PARTS": "2"}),
    (2, 31, True, "fa", {}),  # (a k without room for the extension byte: the route by the sharded (k+1)-mer count)
    (3, 21, False, "fq.gz", {"SMX_MGPU_CHUNK": "4096", "SMX_IO_THREADS": "3"}),
    (2, 33, True, "fq.gz", {"TEST_BGZF_BLOCK": "65280", "SMX_MGPU_PARTS": "2"}),
    # round 6: the k-mer file stays sharded — smx_shard_walks with the HOST's collectives (counts by all-gather, grouped ncclSend / ncclRecv in
    # several rounds, all-reduce), unitigs + loop k-mers gathered, smx_build_graph_from_unitigs; -c shard by shard on a graph without a k-mer file
    (2, 21, False, "fq", {"SMX_MGPU_WALKS": "distributed", "SMX_MGPU_ROUND_WORDS": "16", "SHIM_WALK_CHUNK": "97,13"}),
    (3, 21, True, "fq", {"SMX_MGPU_WALKS": "distributed", "SMX_MGPU_ROUND_WORDS": "64", "SHIM_WALK_CHUNK": "301,29"}),
    (3, 33, False, "fa", {"SMX_MGPU_WALKS": "distributed"}),
    (2, 31, True, "fa", {"SMX_MGPU_WALKS": "distributed"}),  # (the shard by the (k+1)-mer route, then the walks)
    (2, 21, False, "fq", {"SMX_MGPU_ASSUME_FREE_BYTES": "1000"}),  # the gathered structure "does not fit": all ranks take the walks by themselves
])
def test_gbuilder_host_world_n(tmp_path, world, k, coverage, fmt, env):
    from oracle import oracle
    reads = [r for r in read_lines("reads_small.txt")[:120] if r]
    inp = str(tmp_path / ("r." + fmt))
    if fmt == "fq":
        _write_fastq(inp, reads, final_newline=(k != 21 or coverage))
    elif fmt == "fq.gz":  # BGZF: every rank inflates only the blocks under its range of the text
        from test_bgzf_cpu import bgzf_bytes
        _write_fastq(inp[:-3], reads)
        with open(inp, "wb") as f:
            f.write(bgzf_bytes(open(inp[:-3], "rb").read(), block=int(env.get("TEST_BGZF_BLOCK", "700"))))
        os.remove(inp[:-3])
    else:
        with open(inp, "w") as f:
            for i, r in enumerate(reads):
                f.write(f">r{i}\n{r}\n")
    out = str(tmp_path / "g.gfa")
    got = _run(world, "gbuilder", (k, 1, coverage, [inp], out), env, reads, 36500 + 7 * world + k)
    assert all(g["rc"] == 0 for g in got), [g["error"] for g in got]
    # the input: every read on exactly one rank, every rank has some
    assert sorted(r for g in got for r in g["reads"]) == sorted(reads) and all(g["reads"] for g in got)
    if fmt in ("fq", "fq.gz"):  # byte ranges of the text (for BGZF: only the blocks under the range are inflated), never the read-dealing host parser
        assert all("smx_submit_fastq_text" in g["calls"] and "smx_submit_reads_ascii" not in g["calls"] for g in got)
    else:
        assert all("smx_submit_reads_ascii" in g["calls"] for g in got)
    # who builds and who writes: rank 0, and with -c every rank (its coverage pass needs the graph); the engine double asserted inside
    # build_graph_from_kmers that the gathered structure IS the reference's k-mer file and mask bytes and that the (k+1)-mer count fits
    walks = env.get("SMX_MGPU_WALKS") == "distributed" or "SMX_MGPU_ASSUME_FREE_BYTES" in env
    assert all(("smx_shard_walks" in g["calls"]) == walks and ("smx_build_graph_from_kmers" in g["calls"]) == (not walks and g["built"]) for g in got)
    if walks:  # only unitigs and loop k-mers travelled to the builders: the engine double asserted that they ARE the reference's edge list and loops
        assert all(("smx_build_graph_from_unitigs" in g["calls"]) == g["built"] and "smx_graph_shard_copy" not in g["calls"] for g in got)
    assert [g["built"] for g in got] == [True] + [coverage] * (world - 1)
    assert [g["written"] is not None for g in got] == [True] + [False] * (world - 1)
    ref = oracle.build_graph(reads, k, 10, coverage=True)
    assert open(out).read() == ref["gfa"]
    assert sum(g["shard"] for g in got) == len(ref["kmers"]) and all(g["shard"] > 0 for g in got)
    ext = (k >= 21 and 2 * k + 8 <= 64 * ((k + 31) // 32)) and "SMX_MGPU_KPOMERS" not in env
    assert all(("smx_extract_kmers_ext_owned" in g["calls"]) == ext and ("smx_graph_shard_updates" in g["calls"]) == (not ext) for g in got)
    if coverage or not ext:
        assert sum(g["kpo"] for g in got) == ref["n_kpomers"]
    if coverage:
        kc = np.array([int(ln.split("KC:i:")[1]) for ln in ref["gfa"].splitlines() if ln.startswith("S\t")], dtype=np.uint32)
        for g in got:
            assert (np.frombuffer(g["cov"], dtype=np.uint32) == kc).all()  # summed over the ranks' own reads == reference KC tags
    if "SMX_MGPU_ROUND_WORDS" in env and int(env["SMX_MGPU_ROUND_WORDS"]) < 1000:
        assert all(g["p2p_ops"] > 2 * (world - 1) for g in got)  # the exchanges really went in several rounds


@pytest.mark.parametrize("world,K,nfiles,env", [(2, 21, 3, {"SMX_MGPU_ROUND_WORDS": "64"}), (3, 33, 4, {}), (3, 21, 2, {"SMX_MGPU_ROUND_WORDS": "512"}), (2, 33, 1, {}), (3, 21, 1, {"TEST_FIRST_BGZF": "1"})])
def test_kmercount_host_world_n(tmp_path, world, K, nfiles, env):
    """at least one file per rank: whole files round-robin; fewer (R1 / R2 on more GPUs): every file is cut among all ranks"""
    from oracle import oracle
    reads = [r for r in read_lines("reads_tiny.txt") if r]
    files = []
    for i in range(nfiles):
        p = str(tmp_path / (f"f{i}.fq" if i != 1 else "f1.fa"))
        if i == 1:
            with open(p, "w") as f:
                for j, r in enumerate(reads[i::nfiles]):
                    f.write(f">s{j}\n{r}\n")
        elif i == 2:  # a BGZF-compressed FASTQ (what BCL Convert writes): inflated block-parallel by the tools' reader
            from test_bgzf_cpu import bgzf_bytes
            _write_fastq(p, reads[i::nfiles])
            with open(p + ".gz", "wb") as f:
                f.write(bgzf_bytes(open(p, "rb").read(), block=300))
            os.remove(p)
            p += ".gz"
        else:
            _write_fastq(p, reads[i::nfiles])
            if i == 0 and env.get("TEST_FIRST_BGZF"):
                from test_bgzf_cpu import bgzf_bytes
                with open(p + ".gz", "wb") as f:
                    f.write(bgzf_bytes(open(p, "rb").read(), block=500))
                os.remove(p)
                p += ".gz"
        files.append(p)
    got = _run(world, "kmercount", (K, str(tmp_path), files), env, reads, 38500 + 11 * world + K)
    assert all(g["rc"] == 0 for g in got), [g["error"] for g in got]
    assert sorted(r for g in got for r in g["reads"]) == sorted(reads) and all(g["reads"] for g in got)
    ref, _ = oracle.count(reads, K, "A", 16)
    assert open(tmp_path / "final_kmers", "rb").read() == ref.tobytes()  # the ranks' bucket ranges, each written at its offset
    assert sum(g["owned"] for g in got) == len(ref) and all(g["owned"] > 0 for g in got)


def test_gbuilder_host_refuses_a_graph_that_does_not_fit(tmp_path):
    """the gathered structure would not fit where the graph is built and the distributed walks are forbidden (SMX_MGPU_WALKS=gathered: the
    behaviour of rounds 4-5): EVERY rank leaves with the reference's memory-limit code (68), none waits in a collective, nothing is written"""
    reads = [r for r in read_lines("reads_small.txt")[:120] if r]
    inp = str(tmp_path / "r.fq")
    _write_fastq(inp, reads)
    out = str(tmp_path / "g.gfa")
    got = _run(2, "gbuilder", (21, 1, False, [inp], out), {"SMX_MGPU_ASSUME_FREE_BYTES": "1000", "SMX_MGPU_WALKS": "gathered"}, reads, 40500)
    assert [g["rc"] for g in got] == [68, 68] and not os.path.exists(out)
    assert all("smx_build_graph_from_kmers" not in g["calls"] and "smx_shard_walks" not in g["calls"] for g in got)


@pytest.mark.parametrize("coverage,inject,expect", [
    (False, "1:smx_graph_shard_from_ext:68", "fallback"),   # the memory limit on ONE rank: all ranks take the (k+1)-mer route together
    (True, "0:smx_extract_kmers_ext_owned:68", "fallback"),  # ... also before the exchange, and with the (k+1)-mers already counted for -c
    (False, "1:smx_graph_shard_from_ext:68,0:smx_graph_shard_from_ext:67", "fail67"),  # a genuine error next to it is not swallowed
])
def test_gbuilder_host_falls_back_on_all_ranks_together(tmp_path, coverage, inject, expect):
    from oracle import oracle
    reads = [r for r in read_lines("reads_small.txt")[:120] if r]
    inp = str(tmp_path / "r.fq")
    _write_fastq(inp, reads)
    out = str(tmp_path / "g.gfa")
    got = _run(2, "gbuilder", (21, 1, coverage, [inp], out), {"SHIM_FAIL": inject}, reads, 41500 + len(inject))
    if expect == "fallback":
        assert all(g["rc"] == 0 for g in got), [g["error"] for g in got]
        assert all("smx_graph_shard_updates" in g["calls"] and "smx_graph_clear" in g["calls"] for g in got)
        assert open(out).read() == oracle.build_graph(reads, 21, 10, coverage=True)["gfa"]
    else:
        assert [g["rc"] for g in got] == [67, 67] and not os.path.exists(out)
