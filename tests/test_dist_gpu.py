"""GPU, one rank: the sharded path end-to-end on the real engine (extract_partition -> RCCL all_to_all_single ->
count_records) must give the same bytes as the single-GPU count. Multi-rank RCCL runs are the driver's; the
orchestration itself is covered at world_size 2 on CPU (tests/test_dist_cpu.py)."""
import os

import numpy as np
import pytest

from conftest import read_lines

pytestmark = pytest.mark.gpu


def test_sharded_path_single_rank_matches_direct_count():
    import torch
    import torch.distributed as dist
    from spades_amd import KMerDiskCounter, ReadKMerSplitter
    from spades_amd import dist as smx_dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 1000))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        reads = [r for r in read_lines("reads_small.txt") if r]
        for K, mode, nb in ((21, "A", 16), (56, "B", 30)):
            sp = ReadKMerSplitter(K, mode)
            sp.push_back_reads(reads)
            direct = KMerDiskCounter(None, sp).Count(nb)
            want, want_sizes = direct.records(), direct.bucket_sizes()
            eng = smx_dist.GpuEngine(sp.ctx, mode)
            res = smx_dist.sharded_count(eng, K, nb, 0, 1, dev)
            assert res["sent"] == res["received"] == direct.kmer_instances()
            assert res["distinct"] == len(want) and res["bucket_sizes"] == list(map(int, want_sizes))
            got = np.empty_like(want)
            import ctypes as C
            rc = sp.ctx.lib.smx_copy_final_kmers(sp.ctx._h, got.ctypes.data_as(C.c_void_p))
            assert rc == 0 and (got == want).all()
            sp.ctx.close()
    finally:
        dist.destroy_process_group()
