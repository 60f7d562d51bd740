"""CPU, world_size 2, gloo: the sharded orchestration (owner ranges, one all-to-all, owner-side merge) with a CPU
engine built on the test oracle. The GPU engine shares every line of spades_amd.dist.sharded_count."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import free_port, read_lines

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """Test double for spades_amd.dist.GpuEngine: same contract, CPU tensors, oracle arithmetic."""

    def __init__(self, reads, mode):
        self.reads, self.mode = reads, mode
        self._recs = {}

    def _instances(self, K):
        if K not in self._recs:
            from oracle import oracle
            out = []
            for r in self.reads:  # every instance, duplicates kept: count each read separately and weight by multiplicity
                a, b = oracle.longest_valid(r)
                s = r[a:b].upper()
                rc = s[::-1].translate(str.maketrans("ACGT", "TGCA"))
                for t in (s, rc):
                    for j in range(len(t) - K + 1):
                        rec = oracle.kmer_from_string(t[j:j + K])
                        if self.mode == "B" and not oracle.is_minimal(rec, K):
                            continue
                        out.append(rec)
            self._recs[K] = np.array(out, dtype=np.uint64).reshape(-1, (K + 31) // 32)
        return self._recs[K]

    def alloc(self, n_words, dev):
        return torch.empty(max(n_words, 1), dtype=torch.int64)

    def extract_count(self, K):
        return len(self._instances(K))

    def extract_partition(self, K, nb, world, buf, capacity):
        from oracle import oracle
        rec = self._instances(K)
        owner = np.array([oracle.bucket(r, K, nb) * world // nb for r in rec], dtype=np.int64)
        order = np.argsort(owner, kind="stable")
        flat = rec[order].reshape(-1).view(np.int64)
        buf[:len(flat)] = torch.from_numpy(flat.copy())
        return [int((owner == r).sum()) for r in range(world)]

    def count_records(self, K, nb, buf, n):
        from oracle import oracle
        nw = (K + 31) // 32
        rec = buf[:n * nw].numpy().view(np.uint64).reshape(n, nw)
        keyed = sorted({(oracle.bucket(r, K, nb),) + tuple(int(x) for x in r) for r in rec})
        sizes = [0] * nb
        for k in keyed:
            sizes[k[0]] += 1
        self.result = np.array([k[1:] for k in keyed], dtype=np.uint64).reshape(-1, nw)
        return {"distinct": len(keyed), "instances": n, "bucket_sizes": sizes, "device_ptr": 0}


def _worker(rank, world, port, K, mode, nb, q, limit=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spades_amd import dist as smx_dist
    if limit:
        smx_dist.XCHG_LIMIT = limit  # force the multi-round point-to-point exchange
    reads = read_lines("reads_small.txt")[:120]
    eng = OracleEngine(reads[rank::world], mode)
    res = smx_dist.sharded_count(eng, K, nb, rank, world, torch.device("cpu"))
    q.put((rank, res["bucket_sizes"], eng.result.tobytes(), res["sent"], res["received"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("K,mode,nb,limit,world", [(21, "A", 16, None, 2), (33, "B", 10, None, 2), (21, "A", 16, 1000, 2),
                                                   (21, "A", 16, 700, 3), (33, "B", 10, None, 4)])
def test_sharded_count_gloo(K, mode, nb, limit, world):
    """world 2-4 (uneven bucket ranges at 3 ranks, 10 buckets over 4 ranks), one-shot all-to-all and forced point-to-point rounds"""
    from oracle import oracle
    from spades_amd.dist import rank_first_bucket
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, K, mode, nb, q, limit)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    reads = read_lines("reads_small.txt")[:120]
    ref, sizes = oracle.count(reads, K, mode, nb)
    # concatenating the owners' outputs in rank order gives the reference's final_kmers bytes
    assert b"".join(g[2] for g in got) == ref.tobytes()
    tot = np.sum([g[1] for g in got], axis=0)
    assert (tot == sizes).all()
    for rank, bs, _, _, _ in got:  # each rank holds only the buckets it owns
        lo, hi = rank_first_bucket(nb, world, rank), rank_first_bucket(nb, world, rank + 1)
        assert all(b == 0 for i, b in enumerate(bs) if not (lo <= i < hi))
    assert sum(g[3] for g in got) == sum(g[4] for g in got)  # every record sent is received exactly once


class OracleGraphEngine(OracleEngine):
    """adds the construction half of the GpuEngine contract (owner-side mask fill, gathered compact structure, sharded coverage)"""

    def __init__(self, reads, all_reads):
        super().__init__(reads, "B")
        self.all_reads = all_reads

    def result_tensor(self, n_words, dev):
        return torch.from_numpy(self.result.reshape(-1).view(np.int64).copy())

    # -- sharded construction (owner-side mask fill): the same contract as GpuEngine, oracle arithmetic --
    def alloc_bytes(self, n, dev):
        return torch.empty(max(n, 1), dtype=torch.uint8)

    def shard_updates(self, k, nb, world, buf, capacity):
        """extension updates of this rank's (k+1)-mer shard (self.result), grouped by the owner of the k-mer"""
        from oracle import oracle
        tr = str.maketrans("ACGT", "TGCA")
        nwk = (k + 31) // 32
        upd = []
        for rec in self.result:
            x = oracle.kmer_to_string(rec, k + 1)
            pn, nn = "ACGT".index(x[0]), "ACGT".index(x[k])
            for km, fwd_bit, rc_bit in ((x[:k], nn, 7 - nn), (x[1:], pn + 4, 3 - pn)):
                r = km[::-1].translate(tr)
                canon, bit = (km, fwd_bit) if km <= r else (r, rc_bit)
                w = oracle.kmer_from_string(canon)
                owner = oracle.bucket(w, k, nb) * world // nb
                upd.append((owner, tuple(int(v) for v in w) + (bit,)))
        upd.sort(key=lambda u: u[0])
        flat = np.array([u[1] for u in upd], dtype=np.uint64).reshape(-1)
        assert len(upd) <= capacity
        buf[:len(flat)] = torch.from_numpy(flat.view(np.int64).copy())
        return [sum(1 for u in upd if u[0] == r) for r in range(world)]

    def shard_build(self, k, nb, world, rank, buf, n):
        from oracle import oracle
        nwk = (k + 31) // 32
        rec = buf[:n * (nwk + 1)].numpy().view(np.uint64).reshape(n, nwk + 1)
        masks = {}
        for r in rec:
            key = (oracle.bucket(r[:nwk], k, nb),) + tuple(int(v) for v in r[:nwk])
            masks[key] = masks.get(key, 0) | (1 << int(r[nwk]))
        keys = sorted(masks)
        lo, hi = (rank * nb + world - 1) // world, ((rank + 1) * nb + world - 1) // world
        assert all(lo <= key[0] < hi for key in keys)  # only k-mers of the rank's own buckets arrive
        self.shard_kmers = np.array([key[1:] for key in keys], dtype=np.uint64).reshape(-1, nwk)
        self.shard_masks = np.array([masks[key] for key in keys], dtype=np.uint8)
        self.shard_updates_seen = n
        sizes = [0] * nb
        for key in keys:
            sizes[key[0]] += 1
        return len(keys), sizes

    # -- the one-exchange route: k-mers travel with the InOutMask byte the sender's reads give them (EXT layout: last word =
    #    k-mer bits << 8 | byte) --
    def ext_supported(self, k):
        nw = (k + 31) // 32
        return k >= 21 and 2 * k + 8 <= 64 * nw

    def extract_kmers_ext_owned(self, k, nb, world, dev):
        from oracle import oracle
        g = oracle.build_graph(self.reads, k, nb)  # k-mers and masks of THIS rank's reads
        self.__dict__.pop("result", None)  # (as the library: the extraction drops the context's previous count result for room)
        nwk = (k + 31) // 32
        rec = np.array(g["kmers"], dtype=np.uint64).reshape(-1, nwk).copy()
        owner = np.array([oracle.bucket(r, k, nb) * world // nb for r in rec], dtype=np.int64)
        rec[:, nwk - 1] = (rec[:, nwk - 1] << np.uint64(8)) | np.asarray(g["masks"], dtype=np.uint64)
        order = np.argsort(owner, kind="stable")
        self.ext_sent = len(rec)
        flat = rec[order].reshape(-1).view(np.int64)
        return torch.from_numpy(flat.copy()) if len(flat) else torch.empty(1, dtype=torch.int64), [int((owner == r).sum()) for r in range(world)]

    def shard_from_ext(self, k, nb, world, rank, buf, n):
        from oracle import oracle
        tr = str.maketrans("ACGT", "TGCA")
        nwk = (k + 31) // 32
        rec = buf[:n * nwk].numpy().view(np.uint64).reshape(n, nwk).copy()
        byte = (rec[:, nwk - 1] & np.uint64(0xFF)).astype(np.uint8)
        rec[:, nwk - 1] >>= np.uint64(8)
        masks = {}
        for r, m in zip(rec, byte):
            key = (oracle.bucket(r, k, nb),) + tuple(int(v) for v in r)
            masks[key] = masks.get(key, 0) | int(m)
        keys = sorted(masks)
        lo, hi = (rank * nb + world - 1) // world, ((rank + 1) * nb + world - 1) // world
        assert all(lo <= key[0] < hi for key in keys)  # only k-mers of the rank's own buckets arrive
        self.shard_kmers = np.array([key[1:] for key in keys], dtype=np.uint64).reshape(-1, nwk)
        self.shard_masks = np.array([masks[key] for key in keys], dtype=np.uint8)
        self.shard_updates_seen = n
        sizes = [0] * nb
        bits = pals = 0
        for key in keys:
            sizes[key[0]] += 1
            m = masks[key]
            x = oracle.kmer_to_string(np.array(key[1:], dtype=np.uint64), k)
            for c in range(4):
                for e, on in ((x + "ACGT"[c], m >> c & 1), ("ACGT"[c] + x, m >> (4 + c) & 1)):
                    if on:
                        bits += 1
                        pals += e == e[::-1].translate(tr)
        return len(keys), sizes, bits, pals

    def shard_copy(self, kmers, masks):
        kmers[:self.shard_kmers.size] = torch.from_numpy(self.shard_kmers.reshape(-1).view(np.int64).copy())
        masks[:self.shard_masks.size] = torch.from_numpy(self.shard_masks.copy())

    def build_graph_from_kmers(self, k, nb, kmers, masks, n, bucket_sizes, n_kpomers):
        from oracle import oracle
        nw = (k + 31) // 32
        self.k = k
        self.g = oracle.build_graph(self.all_reads, k, nb, coverage=True)
        # the gathered compact structure must be the reference's k-mer file and its InOutMask bytes
        self.gathered_kmers = kmers[:n * nw].numpy().view(np.uint64).reshape(n, nw).copy()
        self.gathered_masks = masks[:n].numpy().copy()
        assert self.gathered_kmers.tobytes() == self.g["kmers"].tobytes()
        assert (self.gathered_masks == self.g["masks"]).all()
        assert sum(bucket_sizes) == n and n_kpomers == self.g["n_kpomers"]
        return dict(n_kpomers=self.g["n_kpomers"], n_kmers=len(self.g["kmers"]), n_unitigs=len(self.g["unitigs"]),
                    n_loops=self.g["n_loops"], n_vertices=self.g["n_vertices"], unitig_bases=0, words=nw)

    def set_kpomers(self, buf, n, bucket_sizes):
        """a (k+1)-mer file that may hold some buckets only (dist.py installs one owner's shard at a time for -c; the C++ host the whole
        file): the lookups of the coverage pass miss what is not in it, like the library's"""
        nw = (self.k + 1 + 31) // 32
        assert bucket_sizes is None or sum(int(x) for x in bucket_sizes) == n  # (the C++ hosts' shim forwards no sizes)
        part = buf[:n * nw].numpy().view(np.uint64).reshape(n, nw).copy()
        self.kpo_parts = getattr(self, "kpo_parts", []) + [part]
        self.gathered = np.concatenate(self.kpo_parts)  # (in rank order the shards ARE the file: the tests compare it with the reference's)
        self.kpo_max_resident = max(getattr(self, "kpo_max_resident", 0), n)
        K1 = self.k + 1
        self.kpo_now = {"".join("ACGT"[(int(rec[j >> 5]) >> ((j & 31) << 1)) & 3] for j in range(K1)) for rec in part}

    def local_raw_coverage(self, n_unitigs):
        # (k+1)-mer instances of this rank's reads (+RC), per unitig; a (k+1)-mer and its RC are the same canonical key
        from collections import Counter
        from oracle import oracle
        K1 = self.k + 1
        tr = str.maketrans("ACGT", "TGCA")
        c = Counter()
        for r in self.reads:
            a, b = oracle.longest_valid(r)
            s = r[a:b].upper()
            for j in range(len(s) - K1 + 1):  # read + RC stream, minimal orientation counted: once per occurrence, palindromes twice
                x = s[j:j + K1]
                y = x[::-1].translate(tr)
                if min(x, y) not in self.kpo_now:  # (not in the installed file: the rank lookup misses, nothing is counted)
                    continue
                c[min(x, y)] += 2 if x == y else 1
        out = []
        for u in self.g["unitigs"]:
            out.append(sum(c[min(u[j:j + K1], u[j:j + K1][::-1].translate(tr))] for j in range(len(u) - K1 + 1)))
        return torch.tensor(out, dtype=torch.int64).to(torch.int32)

    def set_raw_coverage(self, cov):
        self.cov = cov.numpy().astype(np.uint32)


def _graph_worker(rank, world, port, k, threads, q, route="kpomers", coverage=True):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spades_amd import dist as smx_dist
    smx_dist.XCHG_LIMIT = 700  # several broadcast rounds per owner
    reads = read_lines("reads_small.txt")[:120]
    eng = OracleGraphEngine(reads[rank::world], reads)
    info = smx_dist.sharded_build_graph(eng, k, threads, rank, world, torch.device("cpu"), coverage=coverage, route=route, walks="auto" if route == "ext" else "gathered")
    assert info["route"] == route and info["walks"] == "gathered"  # ("auto" has room for the gathered structure here)
    if coverage and world > 1:  # -c: one owner's shard of the (k+1)-mer file at a time, never the whole file on a rank
        assert 0 < eng.kpo_max_resident < sum(info["kpomers_per_rank"]) and len(eng.kpo_parts) == sum(1 for c in info["kpomers_per_rank"] if c)
    q.put((rank, eng.gathered.tobytes() if coverage else b"", eng.cov.tobytes() if coverage else b"", info["kpomers_per_rank"], eng.g["gfa"],
           len(eng.result) if hasattr(eng, "result") else 0, eng.shard_updates_seen, info["kmers_per_rank"], getattr(eng, "ext_sent", 0)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_build_graph_world2_gloo():
    from oracle import oracle
    world, k, threads = 2, 21, 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_graph_worker, args=(r, world, port, k, threads, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    reads = read_lines("reads_small.txt")[:120]
    ref, _ = oracle.count(reads, k + 1, "B", 10 * threads)
    kc = np.array([int(l.split("KC:i:")[1]) for l in got[0][4].splitlines() if l.startswith("S\t")], dtype=np.uint32)
    for rank, gathered, cov, per_rank, _, n_shard, n_upd, kmers_per_rank, _ in got:
        assert gathered == ref.tobytes()  # (for -c only) every rank holds the reference's (k+1)-mer file
        assert sum(per_rank) == len(ref)
        assert n_shard == per_rank[rank] < len(ref)  # the mask fill of a rank started from ITS shard of the (k+1)-mer file ...
        assert (np.frombuffer(cov, dtype=np.uint32) == kc).all()  # all-reduced sharded coverage == reference KC tags
    assert sum(g[6] for g in got) == 2 * len(ref)  # ... and every (k+1)-mer sent exactly two extension updates
    assert all(0 < c for c in got[0][7])  # both ranks own a part of the k-mer file (its assembly is asserted inside the engine)


@pytest.mark.parametrize("k,coverage,world", [(21, False, 2), (21, True, 2), (33, False, 3)])
def test_sharded_build_graph_one_exchange_gloo(k, coverage, world):
    """route "ext": every rank sends the canonical k-mers of ITS reads with the InOutMask byte those reads give them; the owners OR
    the bytes. The engine asserts (build_graph_from_kmers) that the gathered structure is the reference's k-mer file and masks and
    that the (k+1)-mer count derived from the mask bits is the reference's; -c adds the sharded (k+1)-mer count for the coverage pass."""
    from oracle import oracle
    threads = 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_graph_worker, args=(r, world, port, k, threads, q, "ext", coverage)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    reads = read_lines("reads_small.txt")[:120]
    g = oracle.build_graph(reads, k, 10 * threads, coverage=True)
    assert all(x[4] == g["gfa"] for x in got)
    assert sum(got[0][7]) == len(g["kmers"]) and all(0 < c for c in got[0][7])  # the owners' shards add up to the k-mer file
    assert sum(x[6] for x in got) == sum(x[8] for x in got)                      # every record sent arrived at exactly one owner
    assert sum(x[8] for x in got) > len(g["kmers"])                              # (k-mers shared by the ranks' reads travel once per rank)
    if coverage:
        ref, _ = oracle.count(reads, k + 1, "B", 10 * threads)
        kc = np.array([int(l.split("KC:i:")[1]) for l in got[0][4].splitlines() if l.startswith("S\t")], dtype=np.uint32)
        for x in got:
            assert x[1] == ref.tobytes() and (np.frombuffer(x[2], dtype=np.uint32) == kc).all()


def test_one_exchange_route_is_refused_where_the_record_has_no_room():
    """route "ext" asked for a k whose k-mer record has no 8 spare bits: an error before anything is exchanged ("auto" takes the
    (k+1)-mer route there)"""
    from spades_amd import dist as smx_dist
    eng = OracleGraphEngine(["ACGT" * 40], ["ACGT" * 40])
    assert not eng.ext_supported(31) and not eng.ext_supported(63) and eng.ext_supported(21) and eng.ext_supported(55)
    with pytest.raises(ValueError):
        smx_dist.sharded_build_graph(eng, 31, 1, 0, 1, torch.device("cpu"), route="ext")


class _FailingEngine(OracleGraphEngine):
    """a rank whose local step fails between two collectives (what a GPU rank does when ITS distinct k-mers overflow its HBM plan)"""

    def __init__(self, reads, all_reads, fail_in, code):
        super().__init__(reads, all_reads)
        self.fail_in, self.code = fail_in, code

    def _maybe(self, what):
        if what == self.fail_in:
            from spades_amd.kmercount import SmxError
            raise SmxError(self.code, f"injected failure in {what}")

    def shard_from_ext(self, *a):
        self._maybe("shard_from_ext")
        return super().shard_from_ext(*a)

    def count_records(self, *a):
        self._maybe("count_records")
        return super().count_records(*a)


def _failing_worker(rank, world, port, k, fail_rank, fail_in, code, route, q, other_code=0, coverage=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spades_amd import dist as smx_dist
    reads = read_lines("reads_small.txt")[:120]
    eng = _FailingEngine(reads[rank::world], reads, fail_in if (rank == fail_rank or other_code) else None, code if rank == fail_rank else other_code)
    try:
        info = smx_dist.sharded_build_graph(eng, k, 1, rank, world, torch.device("cpu"), coverage=coverage, route=route)
        q.put((rank, "ok", info["route"], eng.g["gfa"]))
    except smx_dist.CollectiveFailure as e:
        q.put((rank, "failed", e.code, ""))
    dist.barrier()  # nobody is stuck in a collective the failing rank never entered
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_in,code,route,expect", [
    ("shard_from_ext", 68, "auto", "fallback"),   # memory limit on ONE rank: all ranks take the (k+1)-mer route together
    ("shard_from_ext", 70, "auto", "all_fail"),   # any other error: every rank raises
    ("count_records", 68, "kpomers", "all_fail"),  # no further route to fall back to
    ("shard_from_ext", 68, "auto", "mixed"),      # a memory limit on one rank AND a genuine error (67) on the other: no fallback may swallow the 67
    ("shard_from_ext", 68, "auto", "fallback_cov"),  # the fallback with -c: the (k+1)-mer count made for -c went with the abandoned route and is made again
])
def test_a_failing_rank_does_not_leave_the_others_waiting(fail_in, code, route, expect):
    from oracle import oracle
    world, k = 2, 21
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, k, 1, fail_in, code, route, q, 67 if expect == "mixed" else 0, expect == "fallback_cov"))
             for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if expect in ("fallback", "fallback_cov"):
        g = oracle.build_graph(read_lines("reads_small.txt")[:120], k, 10, coverage=True)  # (the test engine always asks for the tags)
        assert all(x[1] == "ok" and x[2] == "kpomers" and x[3] == g["gfa"] for x in got)
    else:
        assert all(x[1] == "failed" and x[2] == code for x in got)


# ---- distributed walks: sharded_build_graph(walks="distributed") on a CPU double of smx_shard_walks (tests/dwalk_torch_double.py over
# string-level primitives); the library's own implementation of that call runs at world 2-3 on the SIMT stand-in: tests/test_dist_gpu.py -------
_TR = str.maketrans("ACGT", "TGCA")


def _rcs(s):
    return s[::-1].translate(_TR)


class OracleWalkEngine(OracleGraphEngine):
    """the k-mer-specific steps of the distributed walks (GpuEngine: smx_shard_walk_requests, smx_shard_lookup, smx_shard_unitigs,
    smx_build_graph_from_unitigs) restated on strings; everything between them is spades_amd.dist's own code, the code under test"""

    @staticmethod
    def _brev8(m):
        return int(f"{m:08b}"[::-1], 2)

    @staticmethod
    def _junction(m):
        o, i = m & 15, m >> 4
        return not (o in (1, 2, 4, 8) and i in (1, 2, 4, 8))

    def _prepare(self, k):
        from oracle import oracle
        if getattr(self, "_wk", None) == k:
            return
        self._wk = k
        self.strs = [oracle.kmer_to_string(r, k) for r in self.shard_kmers]
        self.index = {s: i for i, s in enumerate(self.strs)}
        self.cands = []
        for r, m in enumerate(self.shard_masks):
            m = int(m)
            if not self._junction(m):
                continue
            for c in range(4):
                if m >> c & 1:
                    self.cands.append((r, 0, c))
            mi = self._brev8(m)
            for c in range(4):
                if mi >> c & 1:
                    self.cands.append((r, 1, c))

    def _node_str(self, node):
        s = self.strs[node >> 1]
        return _rcs(s) if node & 1 else s

    def shard_walks(self, k, rank, world, dev, kmers_per_rank):
        """smx_shard_walks on this double: the algorithm restated on torch tensors (tests/dwalk_torch_double.py) between the string-level primitives
        below — what the library does on the device behind that one call"""
        import dwalk_torch_double as dbl
        return dbl.torch_walks(self, k, rank, world, dev, kmers_per_rank)

    def walk_counts(self):
        """(requests of the chain k-mers, start de-edges) of this shard: smx_shard_walk_counts"""
        return 2 * sum(1 for m in self.shard_masks if not self._junction(int(m))), len(self.cands)

    def walk_requests(self, starts, k, world, dev, first=0, n_items=-1):
        from oracle import oracle
        self._prepare(k)
        nb = self.nb
        items = []
        if starts:
            for i, (r, side, c) in enumerate(self.cands):
                items.append((i, 2 * r + side, c))
        else:
            for r, m in enumerate(self.shard_masks):
                m = int(m)
                if self._junction(m):
                    continue
                items.append((2 * r, 2 * r, (m & 15).bit_length() - 1))
                items.append((2 * r + 1, 2 * r + 1, (self._brev8(m) & 15).bit_length() - 1))
        if n_items >= 0:  # (smx_shard_walk_requests_range: the items of that range only)
            items = [t for t in items if first <= t[0] < first + n_items]
        req = []
        for item, node, c in items:
            y = self._node_str(node)[1:] + "ACGT"[c]
            ry = _rcs(y)
            yo = 0 if y <= ry else 1
            w = oracle.kmer_from_string(y if yo == 0 else ry)
            owner = oracle.bucket(w, k, nb) * world // nb
            req.append((owner, tuple(int(v) for v in w), (item << 4) | (8 if starts else 0) | (yo << 2) | c))
        req.sort(key=lambda t: t[0])
        nw = (k + 31) // 32
        recs = np.array([t[1] for t in req], dtype=np.uint64).reshape(-1, nw)
        tags = np.array([t[2] for t in req], dtype=np.int64)
        counts = [sum(1 for t in req if t[0] == p) for p in range(world)]
        rt = torch.from_numpy(recs.reshape(-1).view(np.int64).copy()) if len(req) else torch.empty(1, dtype=torch.int64)
        return rt, torch.from_numpy(tags), counts

    def shard_lookup(self, recs, n, dev):
        from oracle import oracle
        k = self._wk
        nw = (k + 31) // 32
        rec = recs[:n * nw].numpy().view(np.uint64).reshape(n, nw)
        out = []
        for r in rec:
            i = self.index.get(oracle.kmer_to_string(r, k))
            out.append(-1 if i is None else (i << 1) | (1 if self._junction(int(self.shard_masks[i])) else 0))
        return torch.tensor(out, dtype=torch.int64)

    def shard_gather_kmers(self, local_ranks, k, dev):
        idx = local_ranks.numpy()
        km = self.shard_kmers[idx].reshape(-1).view(np.int64).copy() if len(idx) else np.zeros(1, dtype=np.int64)
        mk = self.shard_masks[idx].copy() if len(idx) else np.zeros(1, dtype=np.uint8)
        return torch.from_numpy(km), torch.from_numpy(mk)

    def shard_unitigs(self, first_rank, steps, last, boff, bases, dev):
        k = self._wk
        words, ln, st, en, sf = [], [], [], [], []
        for i, (r, side, c) in enumerate(self.cands):
            n = int(steps[i])
            b = bases[int(boff[i]):int(boff[i]) + n].numpy()
            s = self._node_str(2 * r + side) + "ACGT"[c] + "".join("ACGT"[int(x)] for x in b)
            rs = _rcs(s)
            if s < rs:
                continue
            w = [0] * ((len(s) + 31) // 32)
            for t, ch in enumerate(s):
                w[t >> 5] |= "ACGT".index(ch) << ((t & 31) << 1)
            words += w
            ln.append(len(s))
            st.append(2 * (first_rank + r) + side)
            en.append(int(last[i]))
            sf.append(1 if s == rs else 0)
        self.my_unitigs = len(ln)
        t64 = lambda a: torch.from_numpy(np.array(a, dtype=np.uint64).view(np.int64).copy()) if a else torch.empty(0, dtype=torch.int64)
        return t64(words), t64(ln), t64(st), t64(en), torch.tensor(sf, dtype=torch.uint8)

    def build_graph_from_unitigs(self, k, nb, n_kmers, n_kpomers, words, n_words, ln, st, en, sf, ne, loop_ranks, loop_kmers, loop_masks):
        from oracle import oracle
        self.k = k
        g = self.g = oracle.build_graph(self.all_reads, k, nb, coverage=True)
        w = words[:n_words].numpy().view(np.uint64)
        seqs, o = [], 0
        for i in range(ne):
            n = int(ln[i])
            seqs.append("".join("ACGT"[(int(w[o + (t >> 5)]) >> ((t & 31) << 1)) & 3] for t in range(n)))
            o += (n + 31) // 32
        assert o == n_words
        npaths = len(g["unitigs"]) - g["n_loops"]
        # the ranks' kept unitigs, concatenated in rank order, are the reference's edge list (before the perfect loops)
        assert seqs == g["unitigs"][:npaths]
        rank_of = {oracle.kmer_to_string(r, k): i for i, r in enumerate(g["kmers"])}

        def node(x):
            rx = _rcs(x)
            return 2 * rank_of[min(x, rx)] + (0 if x <= rx else 1)
        assert [int(v) for v in st[:ne]] == [node(s[:k]) for s in seqs]
        assert [int(v) for v in en[:ne]] == [node(s[-k:]) for s in seqs]
        assert [int(v) for v in sf[:ne]] == [1 if s == _rcs(s) else 0 for s in seqs]
        # k-mers no chain reached == the k-mers of the reference's perfect loops, in file order with their global ranks and masks
        want = sorted({rank_of[min(u[j:j + k], _rcs(u[j:j + k]))] for u in g["unitigs"][npaths:] for j in range(len(u) - k + 1)})
        assert [int(v) for v in loop_ranks] == want
        nw = (k + 31) // 32
        assert loop_kmers.reshape(-1, nw).tobytes() == g["kmers"][want].tobytes() if want else len(loop_kmers) == 0
        assert (np.asarray(loop_masks) == g["masks"][want]).all()
        assert n_kmers == len(g["kmers"]) and n_kpomers == g["n_kpomers"]
        self.n_loop_kmers = len(want)
        return dict(n_kpomers=g["n_kpomers"], n_kmers=len(g["kmers"]), n_unitigs=len(g["unitigs"]), n_loops=g["n_loops"], n_vertices=g["n_vertices"],
                    unitig_bases=0, words=nw)


def _walk_worker(rank, world, port, k, threads, q, reads_file, nreads, coverage, limit, hop_bits=0):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from spades_amd import dist as smx_dist
    smx_dist.XCHG_LIMIT = limit
    import dwalk_torch_double as dbl
    if hop_bits:
        dbl.WALK_HOP_BITS = hop_bits
    if limit < 1000:  # several chunks per doubling round / per fetch of the chains, uneven over the ranks
        dbl.WALK_CHUNK, dbl.WALK_START_CHUNK = 97, 13
    reads = read_lines(reads_file)[:nreads]
    eng = OracleWalkEngine(reads[rank::world], reads)
    eng.nb = 10 * threads
    info = smx_dist.sharded_build_graph(eng, k, threads, rank, world, torch.device("cpu"), coverage=coverage, route="ext", walks="distributed")
    q.put((rank, info["walk_rounds"], info["unitigs_per_rank"], eng.my_unitigs, eng.n_loop_kmers, info["n_unitigs"], info["n_loops"],
           eng.cov.tobytes() if coverage else b"", eng.g["gfa"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("k,world,reads_file,nreads,coverage,limit", [(21, 2, "reads_small.txt", 120, False, 1 << 27), (21, 3, "reads_small.txt", 120, True, 500),
                                                                      (33, 2, "reads_small.txt", 80, False, 1 << 27), (21, 2, "reads_loop.txt", 10 ** 6, False, 1 << 27),
                                                                      (21, 4, "reads_mixed.txt", 10 ** 6, False, 300)])
def test_distributed_walks_gloo(k, world, reads_file, nreads, coverage, limit):
    """the k-mer file stays sharded; lookups, pointer doubling, chain nucleotides and the gather of the unitigs go over gloo. The engine
    asserts that the concatenated unitigs, their end nodes and the loop k-mers are the reference's (build_graph_from_unitigs above)."""
    threads = 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_walk_worker, args=(r, world, port, k, threads, q, reads_file, nreads, coverage, limit)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, rounds, per_rank, mine, nloopk, nu, nl, cov, gfa in got:
        assert per_rank[rank] == mine and sum(per_rank) == nu - nl
        assert rounds >= 1
    assert len({g[1] for g in got}) == 1  # every rank ran the same number of doubling rounds
    if reads_file == "reads_loop.txt":
        assert got[0][6] > 0 and got[0][4] > 0  # perfect loops: found as the k-mers the doubling never finishes
    if coverage:
        kc = np.array([int(l.split("KC:i:")[1]) for l in got[0][8].splitlines() if l.startswith("S\t")], dtype=np.uint32)
        for g in got:
            assert (np.frombuffer(g[7], dtype=np.uint32) == kc).all()


@pytest.mark.parametrize("reads_file", ["reads_mixed.txt"])
def test_loop_nodes_do_not_overflow_the_hop_count(reads_file):
    """a node on a perfect loop never finishes and its hop count doubles every round; with the hops packed into 24 bits every
    rank raised 'a chain of more than 2^24 k-mers' as soon as an ordinary chain needed ~24 rounds next to any plasmid. Here the hop field is
    made just wide enough for the longest real chain (WALK_HOP_BITS), so the loops' counts pass it in the last rounds: open nodes saturate, only a
    node that FINISHES with a saturated count is refused, and the graph is the reference's."""
    from oracle import oracle
    k, world = 21, 2
    reads = read_lines(reads_file)
    g = oracle.build_graph(reads, k, 10, coverage=False)
    assert g["n_loops"] > 0
    npaths = len(g["unitigs"]) - g["n_loops"]
    longest = max([len(u) - k + 1 for u in g["unitigs"][:npaths]] + [1])
    hop_bits = max(2, int(np.ceil(np.log2(longest + 2))))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_walk_worker, args=(r, world, port, k, 1, q, reads_file, 10 ** 6, False, 1 << 27, hop_bits)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][6] == g["n_loops"] and got[0][5] == len(g["unitigs"])
    assert (1 << got[0][1]) > (1 << hop_bits) - 1  # the loops' hop counts (2^rounds) did pass the field
