#!/usr/bin/env python3
"""tests/golden/make_golden_scale.py — parity at size ("Next round" item 1c): md5s of what the REAL reference tools
write for a seeded >= 2 M-read set, generated in the build container (binaries under $SMX_REF_BIN, built from /root/reference by
the survey's cmake recipe). The read set comes from tests/synth.py, so the GPU box regenerates the identical reads and only the
md5s travel (tests/golden/scale_*.json).
usage: make_golden_scale.py [n_reads=2000000] [genome_len=10000000] [seed=77] [k=55] [what=all|kmercount|gfa|allskew|gfaplasmids]
  gfaplasmids: error-free reads from genome_len / 5000 circular genomes of 5 kb (tests/synth.py: synth_codes_plasmids) — one perfect
  loop per plasmid; written as next_scale_*.json, which tests/test_scale_gpu.py only takes with SMX_SCALE_NEXT=1
  BASELINE config 2 (10 M PE150 reads, k=21, spades-kmercount):  make_golden_scale.py 1e7 5e7 1 21 kmercount"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402

REF_BIN = os.environ.get("SMX_REF_BIN", "/tmp/spades_build2/bin")


def md5_file(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
    g = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 77
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 55
    what = sys.argv[5] if len(sys.argv) > 5 else "all"
    threads = 16
    out = {"n_reads": n, "genome_len": g, "seed": seed, "k": k, "threads": threads, "err": 0.01, "n_rate": 0.001,
           # spades-gbuilder clamps -t to omp_get_max_threads() (gbuilder.cpp:154): the bucket count that fixes the unitig order is 10 x this
           "effective_threads": min(threads, int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)))}
    plasmids = what.endswith("plasmids")
    if plasmids:
        what = what[:-8] or "all"
        out.update(plasmids=True, plasmid_len=5000, err=0.0, n_rate=0.0)
    skew = what.endswith("skew")  # repeats, low complexity, log-normal abundances over 64 genomes (tests/synth.py: synth_codes_skewed)
    if skew:
        what = what[:-4] or "all"
        out["skew"] = True
    codes = synth.synth_codes_plasmids(seed, g, n) if plasmids else synth.synth_codes_skewed(seed, g, n) if skew else synth.synth_codes(seed, g, n)
    out["codes_md5"] = hashlib.md5(codes.tobytes()).hexdigest()
    with tempfile.TemporaryDirectory(dir=os.environ.get("SMX_GOLDEN_TMP", "/tmp")) as td:
        os.makedirs(os.path.join(td, "kc"))
        os.makedirs(os.path.join(td, "tmp"))
        fq = os.path.join(td, "reads.fq")
        synth.write_fastq(codes, fq)
        out["fastq_md5"] = md5_file(fq)
        if what != "gfa":  # "gfa": spades-gbuilder only (at 20 M reads the k-mer file of spades-kmercount alone is 50 GB)
            t0 = time.time()
            subprocess.check_call([os.path.join(REF_BIN, "spades-kmercount"), "-k", str(k), "-t", str(threads), "-w", os.path.join(td, "kc"), fq],
                                  stdout=subprocess.DEVNULL)
            out["kmercount_s"] = round(time.time() - t0, 1)
            fk = os.path.join(td, "kc", "final_kmers")
            out["final_kmers_md5"] = md5_file(fk)
            out["final_kmers_bytes"] = os.path.getsize(fk)
            os.remove(fk)
        for cov in ((False, True) if what in ("all", "gfa") else ()):
            gfa = os.path.join(td, "g.gfa")
            t0 = time.time()
            log = subprocess.check_output([os.path.join(REF_BIN, "spades-gbuilder"), fq, gfa, "-k", str(k), "-t", str(threads), "--gfa"] + (["-c"] if cov else []) +
                                          ["-tmp-dir", os.path.join(td, "tmp")]).decode(errors="replace")
            for line in log.splitlines():
                if "loops collected" in line:
                    out["perfect_loops"] = int(line.split("finished.")[1].split()[0])
            key = "gfa_cov" if cov else "gfa"
            out[key + "_s"] = round(time.time() - t0, 1)
            out[key + "_md5"] = md5_file(gfa)
            out[key + "_bytes"] = os.path.getsize(gfa)
            if not cov:
                ns = nl = 0
                with open(gfa, "rb") as f:
                    for line in f:
                        ns += line[:1] == b"S"
                        nl += line[:1] == b"L"
                out["gfa_S_lines"], out["gfa_L_lines"] = ns, nl
            os.remove(gfa)
    # (SMX_GOLDEN_NEXT=1: a golden made without a GPU run to compare with — tests/test_scale_gpu.py takes next_* only with SMX_SCALE_NEXT=1)
    name = os.path.join(HERE, ("next_" if plasmids or os.environ.get("SMX_GOLDEN_NEXT") else "") + f"scale_{n // 1000}k_g{g // 1000}k_s{seed}" + ("" if k == 55 else f"_k{k}") + ("_skew" if out.get("skew") else "") + ("_plasmids" if plasmids else "") + ".json")
    with open(name, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
