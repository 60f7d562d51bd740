"""Distributed walks on ONE rank at size (world 1 over RCCL): step-by-step check of the shard primitives, then the whole
sharded_build_graph(walks="distributed") against the single-GPU build (smx_graph_fingerprint_portable) with wall times.
usage: python tools/dwalk_probe.py [n_reads=10e6] [genome=50e6] [k=55] [T=16] [--steps] [--no-reference | --reference-only]
--no-reference / --reference-only: the two builds in processes of their own (the fingerprints are printed and compared by the caller): the library's
arena only grows, so after a single-GPU build of a large input the same process has little HBM left for the walks' torch tensors."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from spades_amd import dist as smx_dist  # noqa: E402
from spades_amd.gbuilder import GraphBuilder  # noqa: E402


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    n_reads = int(float(pos[0])) // 32 * 32 if pos else 10_000_000
    genome = int(float(pos[1])) if len(pos) > 1 else 50_000_000
    k = int(pos[2]) if len(pos) > 2 else 55
    T = int(pos[3]) if len(pos) > 3 else 16
    steps = "--steps" in sys.argv
    no_ref, ref_only = "--no-reference" in sys.argv, "--reference-only" in sys.argv
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    words, start, ln, codes = bench.synth_reads_device(int(os.environ.get("DWALK_SEED", "1")), genome, n_reads, dev, n_rate=0.001)  # (DWALK_SEED=1000: the bench's batch)
    del codes
    torch.cuda.synchronize()
    gb = GraphBuilder(k, T)
    gb.push_back_device(words.data_ptr(), n_reads * bench.L // 32, start.data_ptr(), ln.data_ptr(), n_reads)
    fp0 = None
    if not no_ref:
        t0 = time.perf_counter()
        info0 = gb.build()
        t_single = time.perf_counter() - t0
        fp0 = gb.fingerprint_portable()
        print(f"single GPU: {t_single:.3f} s, {info0['n_kmers']} k-mers, {info0['n_unitigs']} unitigs, {info0['n_loops']} loops", flush=True)
        gb.ctx.graph_clear()
        if ref_only:
            print("fingerprint single:", fp0, flush=True)
            gb.ctx.close()
            dist.destroy_process_group()
            sys.exit(0)
    eng = smx_dist.GpuEngine(gb.ctx, "B")
    nb, nw = 10 * T, (k + 31) // 32
    if steps:
        def csum():
            nk = eng_n[0]
            a, b = eng.alloc(nk * nw, dev), eng.alloc_bytes(nk, dev)
            eng.shard_copy(a, b)
            torch.cuda.synchronize()
            return int(a[:nk * nw].sum().item()), int(b[:nk].to(torch.int64).sum().item())
        send, counts = eng.extract_kmers_ext_owned(k, nb, 1, dev)
        n = counts[0]
        recv = eng.alloc_recv(n * nw, dev)
        recv[:n * nw].copy_(send[:n * nw])
        torch.cuda.synchronize()
        eng.extract_release()
        nk, sizes, bits, pals = eng.shard_from_ext(k, nb, 1, 0, recv, n)
        eng_n = [nk]
        print("shard:", nk, "k-mers; checksum", csum(), flush=True)
        print("trim:", eng.trim(), "bytes; checksum", csum(), flush=True)
        print("walk counts:", eng.walk_counts(), "; checksum", csum(), flush=True)
        recs, tags, cnt = eng.walk_requests(False, k, 1, dev)
        print("requests:", cnt, "; checksum", csum(), flush=True)
        recv, _ = smx_dist._a2a(recs, [cnt[0] * nw], 0, 1, dev)
        print("exchange to itself equal:", bool(torch.equal(recv[:cnt[0] * nw], recs[:cnt[0] * nw])), flush=True)
        del recv
        reply = eng.shard_lookup(recs, cnt[0], dev)
        bad = reply < 0
        print("lookups failed:", int(bad.sum().item()), "of", cnt[0], flush=True)
        if bool(bad.any().item()):
            idx = bad.nonzero().squeeze(1)
            print("first failures at", idx[:8].tolist(), "last", idx[-4:].tolist(), flush=True)
        gb.ctx.graph_clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info = smx_dist.sharded_build_graph(eng, k, T, 0, 1, dev, walks="distributed")
    torch.cuda.synchronize()
    t_dw = time.perf_counter() - t0
    gb.adopt(info)
    fp1 = gb.fingerprint_portable()
    print("fingerprints:", fp0, fp1, flush=True)
    print("fingerprint walks:", fp1, flush=True)
    import hashlib
    print("fingerprint md5 (bench.py's graph_fingerprint):", hashlib.md5(b"".join(int(v).to_bytes(8, "little") for v in fp1)).hexdigest(), flush=True)
    print(f"distributed walks: {t_dw:.3f} s, {info['walk_rounds']} doubling rounds, {info['n_unitigs']} unitigs, {info['n_loops']} loops; "
          + (f"graph identical to the single-GPU build: {fp0 == fp1}" if fp0 is not None else "reference build in another process"), flush=True)
    print(f"torch peak memory: {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB; kmers_per_rank {info.get('kmers_per_rank')}", flush=True)
    gb.ctx.close()
    dist.destroy_process_group()
    sys.exit(0 if (fp0 is None or fp0 == fp1) else 1)


if __name__ == "__main__":
    main()
