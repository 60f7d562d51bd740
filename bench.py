#!/usr/bin/env python3
"""bench.py — headline benchmark: M reads/s k-mer-counted (k=55, PE150) on N x MI355X.

One step = one pass of the whole hot path (mark -> extract+XXH3 bucket -> MSD partition ->
leaf sort/unique -> compact) over one synthetic batch that is already resident in HBM.
N=1 : smx_count on the batch.  N>1 : every rank counts its own read shard, records are
redistributed by bucket owner with ONE RCCL all-to-all, owners sort/unique.
Rank 0 prints ONE JSON line (contract in the task statement). cpu_baseline (rank 0, N=1 only)
times the reference's own classes (oracle/_ref/ref_kmercount) on a bounded sample of the SAME reads.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402  (device memory, streams, torch.distributed: plumbing only)

L = 150
INSERT = 350


def synth_reads_device(seed, genome_len, n_reads, dev, err=0.01, genome_seed=2):
    """SURVEY.md §8d generator on the GPU: iid genome, PE150 pairs (read2 = RC of the far end), 1 % substitutions.
    Returns (words int64[n_words+8], start int64[n], len int32[n], codes uint8[n, L])."""
    assert n_reads % 32 == 0
    gg = torch.Generator(device=dev)
    gg.manual_seed(genome_seed)
    genome = torch.randint(0, 4, (genome_len,), dtype=torch.uint8, device=dev, generator=gg)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    n_pairs = n_reads // 2
    codes = torch.empty((n_reads, L), dtype=torch.uint8, device=dev)
    idx = torch.arange(L, device=dev)
    chunk = 1 << 19
    for c0 in range(0, n_pairs, chunk):
        c1 = min(n_pairs, c0 + chunk)
        p = torch.randint(0, genome_len - INSERT + 1, (c1 - c0,), device=dev, generator=g)
        codes[2 * c0:2 * c1:2] = genome[p[:, None] + idx[None, :]]
        codes[2 * c0 + 1:2 * c1:2] = 3 - genome[(p + INSERT - 1)[:, None] - idx[None, :]]
    for c0 in range(0, n_reads, 1 << 20):
        blk = codes[c0:c0 + (1 << 20)]
        e = torch.rand(blk.shape, device=dev, generator=g) < err
        sub = torch.randint(1, 4, blk.shape, dtype=torch.uint8, device=dev, generator=g)
        blk[:] = torch.where(e, (blk + sub) % 4, blk)
    flat = codes.reshape(-1, 32)
    shifts = (2 * torch.arange(32, device=dev, dtype=torch.int64))[None, :]
    words = torch.zeros(flat.shape[0] + 8, dtype=torch.int64, device=dev)
    for c0 in range(0, flat.shape[0], 1 << 22):
        part = (flat[c0:c0 + (1 << 22)].to(torch.int64) << shifts).sum(dim=1)
        words[c0:c0 + part.numel()] = part
    start = torch.arange(n_reads, device=dev, dtype=torch.int64) * L
    ln = torch.full((n_reads,), L, dtype=torch.int32, device=dev)
    return words, start, ln, codes


def cpu_baseline(codes_sample, K, mode, nb, gpu_result=None):
    """Reference classes (kind 'reference') when oracle/_ref exists, else the C port (kind 'port').
    gpu_result(n) -> (device int64 tensor [D, nw], bucket sizes) of the GPU count of the same first n reads: when given, the
    reference's output file is compared with it byte for byte (the checker role of oracle/_ref)."""
    import numpy as np
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    arr = lut[codes_sample.cpu().numpy()]
    n = arr.shape[0]
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_kmercount")
    if os.path.exists(ref):
        cores = max(1, min(os.cpu_count() or 1, 64))
        with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
            rf = os.path.join(td, "reads.txt")
            with open(rf, "wb") as f:
                f.write(b"\n".join(r.tobytes() for r in arr) + b"\n")
            t0 = time.time()
            subprocess.check_call([ref, mode, str(K), str(nb), "0", rf, os.path.join(td, "wd"), os.path.join(td, "out"), str(cores)],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            dt = time.time() - t0
            parity = None
            if gpu_result is not None:
                import torch
                rec = gpu_result(n).reshape(-1)  # int64 words on the device, the GPU's final_kmers of the same reads
                out = os.path.join(td, "out")
                same = os.path.getsize(out) == rec.numel() * 8
                CH = 1 << 25  # words per compared chunk (256 MiB)
                with open(out, "rb") as f:
                    for c0 in range(0, rec.numel(), CH):
                        if not same:
                            break
                        ref_chunk = torch.from_numpy(np.fromfile(f, dtype=np.int64, count=min(CH, rec.numel() - c0))).to(rec.device)
                        same = bool(torch.equal(ref_chunk, rec[c0:c0 + ref_chunk.numel()]))
                parity = {"bit_identical_to_reference_output": same, "compared_bytes": int(rec.numel() * 8)}
        res = {"value": round(n / dt / 1e6, 4), "unit": "M reads/s", "cores": cores, "kind": "reference",
               "sample": f"first {n} reads of the bench batch, oracle/_ref/ref_kmercount (reference KMerDiskCounter, tmpfs workdir), {dt:.1f} s"}
        if parity:
            res.update(parity)
        return res
    from oracle import oracle
    reads = [r.tobytes().decode() for r in arr[:50000]]
    t0 = time.time()
    oracle.count(reads, K, mode, nb)
    dt = time.time() - t0
    return {"value": round(len(reads) / dt / 1e6, 4), "unit": "M reads/s", "cores": 1, "kind": "port",
            "sample": f"first {len(reads)} reads of the bench batch, oracle/smx_oracle.c, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=float, default=10e6, help="reads per GPU (weak scaling)")
    ap.add_argument("--k", type=int, default=55)
    ap.add_argument("--mode", default="A", choices=["A", "B"])
    ap.add_argument("--buckets", type=int, default=16)
    ap.add_argument("--genome", type=float, default=50e6)
    ap.add_argument("--cpu-sample", type=float, default=4e6, help="reads timed on the host with the reference classes (~10-15 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--construct-reads", type=float, default=2e6,
                    help="extra (untimed for the headline): de Bruijn construction (k=55, -t 16, -c) on this many reads; 0 disables")
    ap.add_argument("--construct-sharded", action="store_true",
                    help="with N>1 (or --force-sharded): also time spades_amd.dist.sharded_build_graph on --construct-reads reads per rank")
    ap.add_argument("--force-sharded", action="store_true", help="exercise the N>1 code path (extract/all-to-all/owner count) at any world size")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from spades_amd import KMerDiskCounter, ReadKMerSplitter
    from spades_amd.kmercount import Context
    from spades_amd import dist as smx_dist

    n_reads = int(args.reads) // 32 * 32
    K, nb = args.k, args.buckets
    nw = (K + 31) // 32
    words, start, ln, codes = synth_reads_device(1000 + rank, int(args.genome), n_reads, dev)
    sample = codes[:int(min(args.cpu_sample, n_reads))].clone() if rank == 0 else None
    del codes
    torch.cuda.synchronize()

    ctx = Context(device=local_rank)
    sp = ReadKMerSplitter(K, args.mode, ctx)
    sp.push_back_device(words.data_ptr(), words.numel() - 8, start.data_ptr(), ln.data_ptr(), n_reads)
    counter = KMerDiskCounter(None, sp)
    engine = smx_dist.GpuEngine(ctx, args.mode) if sharded else None

    def step():
        if not sharded:
            return counter.Count(nb)
        return smx_dist.sharded_count(engine, K, nb, rank, world, dev)

    def sync():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
            torch.cuda.synchronize()

    # setup (not a step of the measurement): let the library's device arena and torch's caching allocator reach their steady-state
    # footprint — the first passes hipMalloc tens of GB, which stalls for tens of ms each
    for _ in range(2 if sharded else 1):
        st = step()
    for _ in range(args.warmup):
        st = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = step()
    sync()
    dt = time.perf_counter() - t0
    if sharded:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    total_reads = n_reads * world
    value = total_reads / (dt / args.steps) / 1e6

    # ---- roofline of the counting pipeline (this rank), SURVEY.md §8d: B_alg = N*L/4 + 2*I*W + D*W ----
    tm = ctx.timings()
    kernel_ms = sum(ms for _, ms in tm)
    inst = st.kmer_instances() if not sharded else st["instances"]
    distinct = st.total_kmers() if not sharded else st["distinct"]
    W = 8 * nw
    b_alg = n_reads * L / 4 + 2 * inst * W + distinct * W
    achieved = b_alg / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    stages = {}
    for name, ms in tm:  # a stage name repeats when the pipeline runs more than once (batches, the cut-key pass)
        stages[name] = stages.get(name, 0.0) + ms
    dom = max(stages.items(), key=lambda x: x[1]) if stages else ("", 0.0)
    # HBM traffic per step from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs, FETCH x2
    # gfx950 correction: profiles/r01/bench_k55A_10M_pmc_hbm_traffic.csv). Only valid for the workload it was taken on.
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01", "bench_k55A_10M_pmc_hbm_traffic.csv")
    if not sharded and K == 55 and args.mode == "A" and n_reads == 10_000_000 and nb == 16 and os.path.exists(pmc):
        for line in open(pmc):
            if line.startswith("TOTAL"):
                f = line.strip().split(",")
                traffic = round(float(f[3]) + float(f[5]), 1)
    # per-stage view: algorithmic GB a stage has to move / its measured time. Behind the pre-dedupe stage the sort
    # pipeline sees Dc = D/2 canonical records in and n2 = D (mode A) or D (mode B: Dc = D) records through its levels.
    n2 = float(distinct)
    dc = n2 / 2 if args.mode == "A" else n2
    slots_b = inst / (2 if args.mode == "A" else 1) * 1.3  # ~1.3 B per canonical instance in super-k-mer slots (k=55)
    alg = {"skm_count": n_reads * L / 4, "skm_scatter": n_reads * L / 4 + slots_b, "skm_dedupe": slots_b + dc * W,
           "l1_hist": dc * W, "l1_scatter": dc * W + n2 * W, "l2_hist": n2 * W, "l2_scatter": 2 * n2 * W, "l3_hist": n2 * W,
           "l3_scatter": 2 * n2 * W, "sort_unique": 2 * n2 * W}
    stage_roofline = {}
    if "skm_dedupe" in stages:
        for name, b in alg.items():
            if stages.get(name, 0) > 0.05:
                stage_roofline[name] = {"alg_GB": round(b / 1e9, 2), "ms": round(stages[name], 3),
                                        "TBps": round(b / (stages[name] * 1e-3) / 1e12, 2), "frac": round(b / (stages[name] * 1e-3) / 8e12, 3)}
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 4), "traffic": traffic, "traffic_unit": "GB per step (PMC, profiles/r01)",
                "kernel": "smx_count pipeline (sum of stage kernels, HIP events on the library stream)",
                "algorithmic_bytes_per_step": int(b_alg), "kernel_ms_per_step": round(kernel_ms, 3),
                "dominant_stage": dom[0], "dominant_stage_ms": round(dom[1], 3),
                "stages_ms": {n: round(ms, 3) for n, ms in stages.items()}, "stage_roofline": stage_roofline}

    out = {
        "metric": "M reads/sec k-mer-counted (k=55, PE150)" if K == 55 else f"M reads/sec k-mer-counted (k={K}, PE150)",
        "value": round(value, 3), "unit": "M reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"synthetic {n_reads / 1e6:g} M PE150 reads per GPU (genome {args.genome / 1e6:g} Mbp iid, 1% subst.), "
                               f"k={K}, mode {args.mode} ({'spades-kmercount: all k-mers of read+RC' if args.mode == 'A' else 'construction: canonical (k)-mers'}), "
                               f"{nb} buckets, inputs resident in HBM",
                   "reads_per_gpu": n_reads, "k": K, "mode": args.mode, "num_buckets": nb,
                   "kmer_instances": int(inst), "distinct_kmers": int(distinct),
                   "parallelism": "1 GPU" if not sharded else f"{world} GPU(s), bucket-range owners, one RCCL all-to-all"},
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.force_sharded:
        def gpu_result(n):  # GPU count of the first n reads of the batch (they are the first n*L bases of the stream)
            class Wrap:
                def __init__(self, ptr, shape):
                    self.__cuda_array_interface__ = {"shape": shape, "typestr": "<i8", "data": (ptr, False), "version": 2}
            sp.clear()
            sp.push_back_device(words.data_ptr(), n * L // 32, start.data_ptr(), ln.data_ptr(), n)
            stn = counter.Count(nb)
            return torch.as_tensor(Wrap(stn.device_ptr(), (stn.total_kmers(), nw)), device=dev)
        out["cpu_baseline"] = cpu_baseline(sample, K, args.mode, nb, gpu_result if sample.shape[0] % 32 == 0 else None)
    if rank == 0 and world == 1 and args.construct_reads > 0 and not args.force_sharded:
        # BASELINE.json config 3 in small: count + construct + coverage on the first reads of the same batch (reported, not the metric)
        from spades_amd.gbuilder import GraphBuilder
        nc = int(min(args.construct_reads, n_reads)) // 32 * 32
        gb = GraphBuilder(55, 16, ctx)
        sp.clear()
        gb.push_back_device(words.data_ptr(), nc * L // 32, start.data_ptr(), ln.data_ptr(), nc)
        gb.build()  # warm-up (arena growth)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        info = gb.build()
        t1 = time.perf_counter()
        gb.fill_coverage()
        t2 = time.perf_counter()
        out["construct"] = {"workload": f"first {nc} reads, k=55, 160 buckets (-t 16): canonical 56-mers -> 55-mers -> masks -> unitigs -> links",
                            "reads": nc, "build_s": round(t1 - t0, 4), "coverage_s": round(t2 - t1, 4),
                            "M_reads_per_s": round(nc / (t1 - t0) / 1e6, 3), "n_kpomers": int(info["n_kpomers"]),
                            "n_kmers": int(info["n_kmers"]), "n_unitigs": int(info["n_unitigs"]), "n_vertices": int(info["n_vertices"])}
    if sharded and args.construct_sharded and args.construct_reads > 0:
        # multi-GPU construction (collective): sharded (k+1)-mer count -> gather -> replicated build -> all-reduced coverage
        nc = int(min(args.construct_reads, n_reads)) // 32 * 32
        sp.clear()
        sp.push_back_device(words.data_ptr(), nc * L // 32, start.data_ptr(), ln.data_ptr(), nc)
        geng = smx_dist.GpuEngine(ctx, "B")
        smx_dist.sharded_build_graph(geng, 55, 16, rank, world, dev, coverage=True)  # warm-up
        sync()
        t0 = time.perf_counter()
        info = smx_dist.sharded_build_graph(geng, 55, 16, rank, world, dev, coverage=True)
        sync()
        t1 = time.perf_counter()
        out["construct_sharded"] = {"workload": f"{nc} reads per rank x {world} ranks, k=55, 160 buckets, -c; replicated graph on every rank",
                                    "seconds": round(t1 - t0, 4), "M_reads_per_s": round(nc * world / (t1 - t0) / 1e6, 3),
                                    "n_kpomers": int(info["n_kpomers"]), "n_unitigs": int(info["n_unitigs"])}
    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
