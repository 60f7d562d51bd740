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
        for k, t in ((21, 3), (55, 1)):
            gb = GraphBuilder(k, t)
            gb.push_back_reads(reads)
            info = smx_dist.sharded_build_graph(smx_dist.GpuEngine(gb.ctx, "B"), k, t, 0, 1, dev, coverage=True)
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
