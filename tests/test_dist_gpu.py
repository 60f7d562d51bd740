"""GPU, one rank: the sharded path end-to-end on the real engine (extract_partition -> RCCL all_to_all_single ->
count_records) must give the same bytes as the single-GPU count. Multi-rank RCCL runs are the driver's; the
orchestration itself is covered at world_size 2 on CPU (tests/test_dist_cpu.py)."""
import os

import numpy as np
import pytest

from conftest import read_lines

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def test_sharded_path_single_rank_matches_direct_count():
    import torch
    import torch.distributed as dist
    from spades_amd import KMerDiskCounter, ReadKMerSplitter
    from spades_amd import dist as smx_dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(_free_port())
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        reads = [r for r in read_lines("reads_small.txt") if r]
        for K, mode, nb, pre in ((21, "A", 16, 0), (56, "B", 30, 0), (21, "A", 16, 1), (55, "A", 16, 1), (56, "B", 30, 1)):
            sp = ReadKMerSplitter(K, mode)
            sp.ctx.set_option("prededupe", pre)  # 1: the rank pre-dedupes before the exchange (fewer records sent, same result)
            sp.push_back_reads(reads)
            direct = KMerDiskCounter(None, sp).Count(nb)
            want, want_sizes = direct.records(), direct.bucket_sizes()
            eng = smx_dist.GpuEngine(sp.ctx, mode)
            res = smx_dist.sharded_count(eng, K, nb, 0, 1, dev)
            assert res["sent"] == res["received"] and res["instances"] == direct.kmer_instances()
            assert res["sent"] == direct.kmer_instances() if not pre else len(want) <= res["sent"] < direct.kmer_instances()
            assert res["distinct"] == len(want) and res["bucket_sizes"] == list(map(int, want_sizes))
            got = np.empty_like(want)
            import ctypes as C
            rc = sp.ctx.lib.smx_copy_final_kmers(sp.ctx._h, got.ctypes.data_as(C.c_void_p))
            assert rc == 0 and (got == want).all()
            sp.ctx.close()
    finally:
        dist.destroy_process_group()


def _golden(name):
    return open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)).read()


def test_sharded_build_graph_single_rank_nccl(tmp_path):
    """sharded construction (count -> gather -> replicated build -> all-reduced coverage) on the real engine, world 1"""
    import torch
    import torch.distributed as dist
    from spades_amd import dist as smx_dist
    from spades_amd.gbuilder import GraphBuilder
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(_free_port())
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        reads = [r for r in read_lines("reads_small.txt") if r]
        for k, t, route in ((21, 3, "kpomers"), (55, 1, "kpomers"), (21, 3, "ext"), (55, 1, "ext")):
            gb = GraphBuilder(k, t)
            gb.push_back_reads(reads)
            info = smx_dist.sharded_build_graph(smx_dist.GpuEngine(gb.ctx, "B"), k, t, 0, 1, dev, coverage=True, route=route)
            assert info["route"] == route
            gb.adopt(info)
            out = os.path.join(str(tmp_path), f"g{k}.gfa")
            gb.write_gfa(out)
            assert open(out).read() == _golden(f"graphcov_small_k{k}_t{t}.gfa")
            gb.ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k,t", [(21, 1), (21, 3), (55, 3)])
def test_two_rank_construction_through_the_c_abi(k, t, tmp_path):
    """The world-2 data flow of sharded_build_graph replayed in one process on one GPU (two contexts = two ranks, the
    collectives replaced by tensor copies): partition by owner -> owner-side count -> gathered (k+1)-mer file -> replicated
    build on BOTH ranks -> per-rank coverage summed. Both ranks must write the reference's `spades-gbuilder -c` bytes."""
    import ctypes as C
    import torch
    from spades_amd import dist as smx_dist
    from spades_amd.gbuilder import GraphBuilder
    dev = torch.device("cuda", 0)
    reads = [r for r in read_lines("reads_small.txt") if r]
    world, K1, nb = 2, k + 1, 10 * t
    nw = (K1 + 31) // 32
    gbs = [GraphBuilder(k, t) for _ in range(world)]
    engs = []
    for r, gb in enumerate(gbs):
        gb.push_back_reads(reads[r::world])
        engs.append(smx_dist.GpuEngine(gb.ctx, "B"))
    sends, counts = [], []
    for e in engs:
        n = e.extract_count(K1)
        buf = e.alloc(n * nw, dev)
        counts.append(e.extract_partition(K1, nb, world, buf, n))
        sends.append(buf)
    owned = []
    for r, e in enumerate(engs):  # "all-to-all": rank r receives segment r of every sender
        segs = []
        for s in range(world):
            a = sum(counts[s][:r]) * nw
            segs.append(sends[s][a:a + counts[s][r] * nw])
        recv = torch.cat(segs) if sum(x.numel() for x in segs) else e.alloc(0, dev)
        res = e.count_records(K1, nb, recv, sum(counts[s][r] for s in range(world)))
        owned.append(e.result_tensor(res["distinct"] * nw, dev)[:res["distinct"] * nw])
    full = torch.cat(owned)  # "all-gather" in rank order == the (k+1)-mer file
    n_full = full.numel() // nw
    covs = []
    for e in engs:
        info = e.build_graph_from_records(k, nb, full, n_full)
        covs.append(e.local_raw_coverage(info["n_unitigs"]).to(torch.int64) & 0xFFFFFFFF)
    total = (sum(covs) & 0xFFFFFFFF)
    total = torch.where(total >= 2 ** 31, total - 2 ** 32, total).to(torch.int32)
    want = _golden(f"graphcov_small_k{k}_t{t}.gfa")
    for r, (gb, e) in enumerate(zip(gbs, engs)):
        e.set_raw_coverage(total)
        gb.adopt(info)
        out = os.path.join(str(tmp_path), f"g{r}.gfa")
        gb.write_gfa(out)
        assert open(out).read() == want
        gb.ctx.close()


@pytest.mark.parametrize("k,t", [(21, 1), (21, 3), (55, 3)])
def test_two_rank_sharded_mask_fill_through_the_c_abi(k, t, tmp_path):
    """The world-2 data flow of sharded_build_graph (owner-side mask fill) replayed in one process on one GPU: two contexts = two
    ranks, the collectives replaced by tensor copies. (k+1)-mer shards -> extension updates grouped by k-mer owner -> owner-side
    k-mer shard + masks -> gathered compact structure -> graph on BOTH ranks; coverage through the gathered (k+1)-mer file.
    Both ranks must write the reference's `spades-gbuilder -c` bytes, and the gathered structure must be the single-GPU k-mer file."""
    import torch
    from spades_amd import dist as smx_dist
    from spades_amd.gbuilder import GraphBuilder
    dev = torch.device("cuda", 0)
    reads = [r for r in read_lines("reads_small.txt") if r]
    world, K1, nb = 2, k + 1, 10 * t
    nw = (K1 + 31) // 32
    gbs = [GraphBuilder(k, t) for _ in range(world)]
    engs = []
    for r, gb in enumerate(gbs):
        gb.push_back_reads(reads[r::world])
        engs.append(smx_dist.GpuEngine(gb.ctx, "B"))

    def all_to_all(sends, counts, wpr):
        out = []
        for r in range(world):
            segs = []
            for s_ in range(world):
                a = sum(counts[s_][:r]) * wpr
                segs.append(sends[s_][a:a + counts[s_][r] * wpr])
            n = sum(counts[s_][r] for s_ in range(world))
            out.append((torch.cat(segs) if n else engs[r].alloc(0, dev), n))
        return out

    sends, counts = [], []
    for e in engs:
        n = e.extract_count(K1)
        buf = e.alloc(n * nw, dev)
        counts.append(e.extract_partition(K1, nb, world, buf, n))
        sends.append(buf)
    kpo_shards, kpo_sizes, usends, ucounts = [], [], [], []
    for r, (recv, n) in enumerate(all_to_all(sends, counts, nw)):
        res = engs[r].count_records(K1, nb, recv, n)
        kpo_shards.append(engs[r].result_tensor(res["distinct"] * nw, dev)[:res["distinct"] * nw].clone())
        kpo_sizes.append(res["bucket_sizes"])
        ub = engs[r].alloc(2 * res["distinct"] * (nw + 1), dev)
        ucounts.append(engs[r].shard_updates(k, nb, world, ub, 2 * res["distinct"]))
        assert sum(ucounts[-1]) == 2 * res["distinct"]
        usends.append(ub)
    shards, ksizes = [], []
    for r, (recv, n) in enumerate(all_to_all(usends, ucounts, nw + 1)):
        nk, sz = engs[r].shard_build(k, nb, world, r, recv, n)
        km, mk = engs[r].alloc(nk * nw, dev), engs[r].alloc_bytes(nk, dev)
        engs[r].shard_copy(km, mk)
        shards.append((km[:nk * nw], mk[:nk], nk))
        ksizes.append(sz)
    full_k = torch.cat([s_[0] for s_ in shards])
    full_m = torch.cat([s_[1] for s_ in shards])
    n_k = sum(s_[2] for s_ in shards)
    g_ks = [sum(z[b] for z in ksizes) for b in range(nb)]
    g_ps = [sum(z[b] for z in kpo_sizes) for b in range(nb)]
    full_p = torch.cat(kpo_shards)
    # the gathered compact structure is the k-mer file + masks of a single-GPU build on all reads
    ref = GraphBuilder(k, t)
    ref.push_back_reads(reads)
    ref.build()
    rk, rm = ref.kmers()
    assert n_k == len(rk) and (full_k.cpu().numpy().view(np.uint64).reshape(-1, nw) == rk).all() and (full_m.cpu().numpy() == rm).all()
    ref.ctx.close()
    covs, infos = [], []
    for e in engs:
        info = e.build_graph_from_kmers(k, nb, full_k, full_m, n_k, g_ks, full_p.numel() // nw)
        e.set_kpomers(full_p, full_p.numel() // nw, g_ps)
        covs.append(e.local_raw_coverage(info["n_unitigs"]).to(torch.int64) & 0xFFFFFFFF)
        infos.append(info)
    total = (sum(covs) & 0xFFFFFFFF)
    total = torch.where(total >= 2 ** 31, total - 2 ** 32, total).to(torch.int32)
    want = _golden(f"graphcov_small_k{k}_t{t}.gfa")
    for r, (gb, e) in enumerate(zip(gbs, engs)):
        e.set_raw_coverage(total)
        gb.adopt(infos[r])
        out = os.path.join(str(tmp_path), f"s{r}.gfa")
        gb.write_gfa(out)
        assert open(out).read() == want
        gb.ctx.close()


@pytest.mark.parametrize("k,t", [(21, 1), (21, 3), (33, 1), (55, 3)])
def test_two_rank_one_exchange_through_the_c_abi(k, t, tmp_path):
    """Route "ext" of sharded_build_graph replayed in one process on one GPU (two contexts = two ranks, the all-to-all replaced by
    tensor copies): k-mers of each rank's reads with the InOutMask byte those reads give them -> owners -> bytes ORed -> gathered
    compact structure -> graph on BOTH ranks. The gathered structure must be the single-GPU k-mer file and masks, the (k+1)-mer
    count derived from the mask bits the single-GPU one, the GFA the single-GPU GFA."""
    import torch
    from spades_amd import dist as smx_dist
    from spades_amd.gbuilder import GraphBuilder
    dev = torch.device("cuda", 0)
    reads = [r for r in read_lines("reads_small.txt") if r]
    world, nb = 2, 10 * t
    nw = (k + 31) // 32
    gbs = [GraphBuilder(k, t) for _ in range(world)]
    engs = []
    for r, gb in enumerate(gbs):
        gb.push_back_reads(reads[r::world])
        engs.append(smx_dist.GpuEngine(gb.ctx, "B"))
        assert engs[-1].ext_supported(k)
    sends, counts = [], []
    for e in engs:
        view, c = e.extract_kmers_ext_owned(k, nb, world, dev)
        sends.append(view.clone())  # (the view is the library's buffer)
        counts.append(c)
        e.extract_release()
    shards, ksizes, bits, pals = [], [], 0, 0
    for r, e in enumerate(engs):  # "all-to-all": rank r receives segment r of every sender
        segs = []
        for s_ in range(world):
            a = sum(counts[s_][:r]) * nw
            segs.append(sends[s_][a:a + counts[s_][r] * nw])
        n = sum(counts[s_][r] for s_ in range(world))
        recv = e.alloc_recv(n * nw, dev)  # the library's exchange buffer: consumed by shard_from_ext
        recv[:n * nw].copy_(torch.cat(segs))
        nk, sz, b_, p_ = e.shard_from_ext(k, nb, world, r, recv, n)
        km, mk = e.alloc(nk * nw, dev), e.alloc_bytes(nk, dev)
        e.shard_copy(km, mk)
        shards.append((km[:nk * nw], mk[:nk], nk))
        ksizes.append(sz)
        bits += b_
        pals += p_
    full_k = torch.cat([s_[0] for s_ in shards])
    full_m = torch.cat([s_[1] for s_ in shards])
    n_k = sum(s_[2] for s_ in shards)
    g_ks = [sum(z[b] for z in ksizes) for b in range(nb)]
    ref = GraphBuilder(k, t)
    ref.push_back_reads(reads)
    ref.build()
    rk, rm = ref.kmers()
    assert n_k == len(rk) and (full_k.cpu().numpy().view(np.uint64).reshape(-1, nw) == rk).all() and (full_m.cpu().numpy() == rm).all()
    assert (bits + pals) % 2 == 0 and (bits + pals) // 2 == ref.info()["n_kpomers"]
    want = os.path.join(str(tmp_path), "ref.gfa")
    ref.write_gfa(want)
    ref.ctx.close()
    for r, (gb, e) in enumerate(zip(gbs, engs)):
        info = e.build_graph_from_kmers(k, nb, full_k, full_m, n_k, g_ks, (bits + pals) // 2)
        gb.adopt(info)
        out = os.path.join(str(tmp_path), f"g{r}.gfa")
        gb.write_gfa(out)
        assert open(out).read() == open(want).read()
        gb.ctx.close()


def test_distributed_walks_single_rank_nccl(tmp_path):
    """walks = "distributed"  on the real engine, world 1 over RCCL: successor requests -> lookups -> pointer doubling
    -> chain nucleotides -> smx_shard_unitigs -> smx_build_graph_from_unitigs, against the real tool's `-c` GFA (goldens), perfect loops
    included; the graph has no k-mer file afterwards and says so."""
    import torch
    import torch.distributed as dist
    from spades_amd import dist as smx_dist
    from spades_amd.gbuilder import GraphBuilder
    from spades_amd.kmercount import SmxError
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(_free_port())
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        for name, k, t, route in (("small", 21, 3, "ext"), ("small", 55, 1, "ext"), ("loop", 21, 1, "ext"), ("small", 21, 1, "kpomers"), ("polyA", 21, 1, "ext")):
            reads = [r for r in read_lines(f"reads_{name}.txt") if r]
            gb = GraphBuilder(k, t)
            gb.push_back_reads(reads)
            info = smx_dist.sharded_build_graph(smx_dist.GpuEngine(gb.ctx, "B"), k, t, 0, 1, dev, coverage=True, route=route, walks="distributed")
            assert info["route"] == route and info["walks"] == "distributed" and info["walk_rounds"] >= 1
            gb.adopt(info)
            out = os.path.join(str(tmp_path), f"g_{name}_{k}.gfa")
            gb.write_gfa(out)
            assert open(out).read() == _golden(f"graphcov_{name}_k{k}_t{t}.gfa"), (name, k, t, route)
            if name == "loop":
                assert info["n_loops"] > 0
            with pytest.raises(SmxError):
                gb.kmers()  # the k-mer file stayed sharded
            gb.ctx.close()
    finally:
        dist.destroy_process_group()


def _dwalk_rank(rank, world, port, k, t, seed, n_reads, genome, coverage, outdir, walks="distributed"):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import conftest  # (SMX_EMU=1: the spawned rank, too, drives the SIMT stand-in of the library — "device" memory is host memory then)
    from spades_amd import dist as smx_dist
    from spades_amd.gbuilder import GraphBuilder
    import synth
    if conftest.EMU:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # ranks that share a GPU cannot use RCCL: host-staged exchanges (dist._staged)
    try:
        codes = synth.synth_codes(seed, genome, n_reads)
        lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
        reads = [lut[c].tobytes().decode() for c in codes[rank::world]]
        gb = GraphBuilder(k, t)
        gb.push_back_reads(reads)
        if k == 21:  # several rounds per lookup range, per doubling round and per fetch of the chains (uneven over the ranks)
            gb.ctx.set_option("walk_chunk", 1 << 14)
            gb.ctx.set_option("walk_start_chunk", 1 << 10)
        info = smx_dist.sharded_build_graph(smx_dist.GpuEngine(gb.ctx, "B"), k, t, rank, world, dev, coverage=coverage, walks=walks)
        gb.adopt(info)
        gb.write_gfa(os.path.join(outdir, f"rank{rank}.gfa"))
        with open(os.path.join(outdir, f"rank{rank}.info"), "w") as f:
            f.write(repr({kk: v for kk, v in info.items() if isinstance(v, (int, str, list))}))
        gb.ctx.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k,t,n_reads,genome,coverage,world", [(55, 2, 20000, 100000, False, 2), (21, 1, 6000, 30000, True, 2), (77, 1, 10000, 60000, False, 3)])
def test_distributed_walks_ranks_sharing_one_gpu(k, t, n_reads, genome, coverage, world, tmp_path):
    """`world` processes, one context each on the same MI355X, every exchange of the distributed walks for real (gloo, staged through host
    memory): no rank ever holds more than its bucket range of the k-mer file, and every rank writes the single-GPU graph, byte for byte."""
    import torch.multiprocessing as mp
    import synth
    from spades_amd.gbuilder import GraphBuilder
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_dwalk_rank, args=(r, world, port, k, t, 4242 + k, n_reads, genome, coverage, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    codes = synth.synth_codes(4242 + k, genome, n_reads)
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    ref = GraphBuilder(k, t)
    ref.push_back_reads([lut[c].tobytes().decode() for c in codes])
    ref.build()
    if coverage:
        ref.fill_coverage()
    want = os.path.join(str(tmp_path), "ref.gfa")
    ref.write_gfa(want)
    n_kmers = ref.info()["n_kmers"]
    ref.ctx.close()
    for r in range(world):
        assert open(os.path.join(str(tmp_path), f"rank{r}.gfa")).read() == open(want).read()
        info = eval(open(os.path.join(str(tmp_path), f"rank{r}.info")).read())
        assert sum(info["kmers_per_rank"]) == n_kmers and max(info["kmers_per_rank"]) < n_kmers  # the file was sharded
        assert all(u > 0 for u in info["unitigs_per_rank"])


def _dwalk_failing_rank(rank, world, port, phase, outdir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import conftest
    from spades_amd import dist as smx_dist
    from spades_amd.gbuilder import GraphBuilder
    from spades_amd.kmercount import SmxError
    import synth
    if conftest.EMU:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        codes = synth.synth_codes(99, 30000, 6000)
        lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
        gb = GraphBuilder(21, 1)
        gb.push_back_reads([lut[c].tobytes().decode() for c in codes[rank::world]])
        gb.ctx.set_option("walk_chunk", 1 << 13)
        if rank == world - 1:
            gb.ctx.set_option("walk_fail_at", phase)  # (only this rank fails; the others must hear of it at their next collective)
        code, text = 0, ""
        try:
            smx_dist.sharded_build_graph(smx_dist.GpuEngine(gb.ctx, "B"), 21, 1, rank, world, dev, walks="distributed")
        except SmxError as e:
            code, text = e.code, str(e)
        with open(os.path.join(outdir, f"rank{rank}.err"), "w") as f:
            f.write(repr((code, text)))
        gb.ctx.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("phase", [1, 2, 4, 5])
def test_a_rank_that_fails_inside_the_distributed_walks_ends_them_on_every_rank(phase, tmp_path):
    """smx_shard_walks' failure protocol (csrc/smx_dwalk.hpp): a rank whose LOCAL step fails (test hook: the memory-limit code in the given phase
    on the last rank) carries poison into its next collective, so every rank comes back from the call — the failing one with its own code, the
    others with "another rank failed" — and nobody waits in a collective for a rank that is gone."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_dwalk_failing_rank, args=(r, world, port, phase, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0  # (not None: nobody hung)
    errs = [eval(open(os.path.join(str(tmp_path), f"rank{r}.err")).read()) for r in range(world)]
    assert errs[world - 1][0] == 68 and "test hook" in errs[world - 1][1]
    assert all(e[0] == 70 and "another rank failed" in e[1] for e in errs[:world - 1])


@pytest.mark.parametrize("k,t,n_reads,genome,world", [(21, 1, 6000, 30000, 2), (55, 2, 8000, 40000, 3)])
def test_gathered_construction_with_coverage_ranks_sharing_one_gpu(k, t, n_reads, genome, world, tmp_path):
    """the gathered route at world 2-3 with -c: the coverage pass runs shard by shard over the (k+1)-mer file (no rank installs more than one
    owner's bucket range at a time) and every rank ends with the single-process GFA, KC / DP tags included"""
    import torch.multiprocessing as mp
    import synth
    from spades_amd.gbuilder import GraphBuilder
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_dwalk_rank, args=(r, world, port, k, t, 777 + k, n_reads, genome, True, str(tmp_path), "gathered")) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    codes = synth.synth_codes(777 + k, genome, n_reads)
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    ref = GraphBuilder(k, t)
    ref.push_back_reads([lut[c].tobytes().decode() for c in codes])
    ref.build()
    ref.fill_coverage()
    want = os.path.join(str(tmp_path), "ref.gfa")
    ref.write_gfa(want)
    ref.ctx.close()
    for r in range(world):
        assert open(os.path.join(str(tmp_path), f"rank{r}.gfa")).read() == open(want).read()
        info = eval(open(os.path.join(str(tmp_path), f"rank{r}.info")).read())
        assert info["walks"] == "gathered" and max(info["kpomers_per_rank"]) < sum(info["kpomers_per_rank"])
