"""Distributed walks with `world` processes sharing ONE GPU (gloo, exchanges staged through host memory) at a size beyond the tests:
every rank's GFA must be the single-GPU GFA, byte for byte. usage: python tools/dwalk_ranks_probe.py [world=2] [n_reads=1e6] [genome=5e6] [k=55] [T=2]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import torch.multiprocessing as mp
    import synth
    import test_dist_gpu as tdg
    from spades_amd.gbuilder import GraphBuilder
    pos = sys.argv[1:]
    world = int(pos[0]) if pos else 2
    n_reads = int(float(pos[1])) // 2 * 2 if len(pos) > 1 else 1_000_000
    genome = int(float(pos[2])) if len(pos) > 2 else 5_000_000
    k = int(pos[3]) if len(pos) > 3 else 55
    T = int(pos[4]) if len(pos) > 4 else 2
    out = tempfile.mkdtemp()
    ctx = mp.get_context("spawn")
    port = tdg._free_port()
    t0 = time.time()
    procs = [ctx.Process(target=tdg._dwalk_rank, args=(r, world, port, k, T, 99, n_reads, genome, False, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=900)
    print(f"{world} ranks done in {time.time() - t0:.1f} s (read synthesis on the host included), exit codes {[p.exitcode for p in procs]}", flush=True)
    codes = synth.synth_codes(99, genome, n_reads)
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    ref = GraphBuilder(k, T)
    ref.push_back_reads([lut[c].tobytes().decode() for c in codes])
    info = ref.build()
    want = os.path.join(out, "ref.gfa")
    ref.write_gfa(want)
    ref.ctx.close()
    ok = all(p.exitcode == 0 for p in procs)
    for r in range(world):
        same = ok and open(os.path.join(out, f"rank{r}.gfa"), "rb").read() == open(want, "rb").read()
        ok = ok and same
        print(f"rank {r}: GFA identical to the single-GPU build ({info['n_kmers']} k-mers, {info['n_unitigs']} unitigs): {same}; "
              f"{open(os.path.join(out, f'rank{r}.info')).read()[:300] if os.path.exists(os.path.join(out, f'rank{r}.info')) else ''}", flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
