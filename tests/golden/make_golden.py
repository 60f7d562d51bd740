#!/usr/bin/env python3
"""tests/golden/make_golden.py — regenerates the golden fixtures in this directory.

Run in the build container (needs /root/reference): it drives the REAL reference code
  * oracle/_ref/ref_kmercount   (reference classes compiled in place by oracle/ref_recipe/Makefile)
  * spades-kmercount / spades-gbuilder binaries if $SMX_REF_BIN points at a build of the reference
    (the survey built them under /tmp/spades_build2/bin; optional)
on small seeded synthetic inputs and stores inputs + outputs here. The GPU box has no
/root/reference; tests only read the committed fixtures.
"""
import hashlib
import json
import os
import random
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_KC = os.path.join(ROOT, "oracle", "_ref", "ref_kmercount")
REF_BIN = os.environ.get("SMX_REF_BIN", "/tmp/spades_build2/bin")


def rc(s):
    return s[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))


def synth_reads(seed, genome_len, n_reads, read_len, err=0.01, n_rate=0.002, lower=0.0, repeats=0):
    """Seeded toy generator (independent of the product's generator on purpose)."""
    rnd = random.Random(seed)
    g = [rnd.choice("ACGT") for _ in range(genome_len)]
    for _ in range(repeats):  # copy a segment elsewhere -> branching graph
        ln = rnd.randint(30, 80)
        a = rnd.randrange(0, genome_len - ln)
        b = rnd.randrange(0, genome_len - ln)
        g[b:b + ln] = g[a:a + ln]
    g = "".join(g)
    reads = []
    for i in range(n_reads):
        ln = read_len if rnd.random() < 0.8 else rnd.randint(max(1, read_len // 4), read_len)
        p = rnd.randrange(0, genome_len - ln + 1)
        r = list(g[p:p + ln])
        for j in range(ln):
            x = rnd.random()
            if x < err:
                r[j] = rnd.choice([c for c in "ACGT" if c != r[j]])
            elif x < err + n_rate:
                r[j] = "N"
            if lower and rnd.random() < lower:
                r[j] = r[j].lower()
        r = "".join(r)
        if rnd.random() < 0.5:
            r = rc(r)
        reads.append(r)
    return g, reads


def md5(b):
    return hashlib.md5(b).hexdigest()


def run_ref_kmercount(mode, K, nb, reads, bufsize=0):
    with tempfile.TemporaryDirectory() as td:
        rf = os.path.join(td, "reads.txt")
        with open(rf, "w") as f:
            f.write("\n".join(reads) + "\n")
        out = os.path.join(td, "out")
        subprocess.check_call([REF_KC, mode, str(K), str(nb), str(bufsize), rf, os.path.join(td, "wd"), out],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        data = open(out, "rb").read()
        sizes = [int(x) for x in open(out + ".sizes")]
    return data, sizes


def main():
    if not os.path.exists(REF_KC):
        sys.exit("build oracle/_ref first: make -C oracle ref")
    manifest = {"generator": "tests/golden/make_golden.py", "cases": []}

    # ---- counting fixtures -------------------------------------------------------------
    datasets = {
        "tiny": synth_reads(11, 300, 40, 60, err=0.02, n_rate=0.01, lower=0.1),
        "small": synth_reads(12, 1500, 300, 100, err=0.01, n_rate=0.003, repeats=3),
        "polyA": (None, ["A" * 90, "T" * 70, "ACGT" * 20, "N" * 30, "", "ACG", "AAAAAAAAAANAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA"] * 3),
    }
    for name, (_, reads) in datasets.items():
        with open(os.path.join(HERE, f"reads_{name}.txt"), "w") as f:
            f.write("\n".join(reads) + "\n")
    ks = [5, 21, 31, 32, 33, 55, 64, 65, 77, 96, 97, 127]
    for name, (_, reads) in datasets.items():
        for mode, nb in (("A", 16), ("B", 10), ("B", 30), ("A", 1)):
            for K in ks:
                if name == "tiny" and K > 55:
                    continue
                data, sizes = run_ref_kmercount(mode, K, nb, reads)
                case = {"kind": "count", "reads": f"reads_{name}.txt", "mode": mode, "K": K, "num_buckets": nb,
                        "n_records": sum(sizes), "bucket_sizes": sizes, "md5": md5(data), "source": "oracle/_ref/ref_kmercount"}
                # keep full bytes for a few small cases so a failing test can show a diff
                if name == "tiny" and K in (5, 21, 33) and nb in (16, 10):
                    fn = f"final_kmers_{name}_{mode}{K}_b{nb}.bin"
                    open(os.path.join(HERE, fn), "wb").write(data)
                    case["file"] = fn
                manifest["cases"].append(case)
    # multi-dump path of the reference (tiny buffers force several sorted runs + loser-tree merge)
    _, reads = datasets["small"]
    for mode, nb, K in (("A", 16, 21), ("B", 30, 55)):
        data, sizes = run_ref_kmercount(mode, K, nb, reads * 8, bufsize=1)  # cell_size clamps to 16384 records
        ref = [c for c in manifest["cases"] if c["reads"] == "reads_small.txt" and c["mode"] == mode and c["K"] == K and c["num_buckets"] == nb][0]
        assert md5(data) == ref["md5"], "reference result must not depend on dump count"

    # ---- real spades-kmercount binary on the same reads (FASTQ front-end included) -------
    kc = os.path.join(REF_BIN, "spades-kmercount")
    if os.path.exists(kc):
        for name in ("tiny", "small", "polyA"):
            _, reads = datasets[name]
            with tempfile.TemporaryDirectory() as td:
                fq = os.path.join(td, "r.fq")
                with open(fq, "w") as f:
                    for i, r in enumerate(reads):
                        if not r:
                            continue  # FASTQ cannot carry an empty record portably
                        f.write(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n")
                for K in (21, 55):
                    wd = os.path.join(td, f"w{K}")
                    os.makedirs(wd)
                    subprocess.check_call([kc, "-k", str(K), "-t", "2", "-w", wd, fq], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                    data = open(os.path.join(wd, "final_kmers"), "rb").read()
                    ref = [c for c in manifest["cases"] if c["reads"] == f"reads_{name}.txt" and c["mode"] == "A" and c["K"] == K and c["num_buckets"] == 16][0]
                    assert md5(data) == ref["md5"], (name, K, "spades-kmercount binary disagrees with ref harness")
                    ref["also_verified_by"] = "spades-kmercount binary (survey build)"
    # ---- graph fixtures: the REAL spades-gbuilder binary (GFA depends on -t: SURVEY.md finding 3) -------
    gb = os.path.join(REF_BIN, "spades-gbuilder")
    if os.path.exists(gb):
        rnd = random.Random(77)
        circle = "".join(rnd.choice("ACGT") for _ in range(400))            # perfect loop: circular genome, exact reads
        loop_reads = [(circle + circle)[p:p + 90] for p in range(0, 400, 7)]
        gsets = dict(datasets)
        gsets["loop"] = (None, loop_reads)
        gsets["mixed"] = (None, datasets["small"][1] + loop_reads)
        for name in ("loop", "mixed"):
            with open(os.path.join(HERE, f"reads_{name}.txt"), "w") as f:
                f.write("\n".join(gsets[name][1]) + "\n")
        for name, ks, ts in (("tiny", (5, 21), (1, 2)), ("small", (5, 21, 33, 55), (1, 3)), ("polyA", (21,), (1,)),
                             ("loop", (21, 55), (1, 2)), ("mixed", (21, 33), (1, 4))):
            reads = [r for r in gsets[name][1] if r]
            with tempfile.TemporaryDirectory() as td:
                fq = os.path.join(td, "r.fq")
                with open(fq, "w") as f:
                    for i, r in enumerate(reads):
                        f.write(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n")
                for K in ks:
                    for T in ts:
                        out = os.path.join(td, f"g_{K}_{T}.gfa")
                        subprocess.check_call([gb, fq, out, "-k", str(K), "-t", str(T), "--gfa", "-tmp-dir", os.path.join(td, f"t{K}_{T}")],
                                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                        txt = open(out).read()
                        fn = f"graph_{name}_k{K}_t{T}.gfa"
                        keep = len(txt) < 60000
                        if keep:
                            open(os.path.join(HERE, fn), "w").write(txt)
                        manifest["cases"].append({"kind": "graph", "reads": f"reads_{name}.txt", "K": K, "threads": T, "num_buckets": 10 * T,
                                                  "md5": md5(txt.encode()), "n_segments": txt.count("\nS\t"), "n_links": txt.count("\nL\t"),
                                                  "file": fn if keep else None, "source": "spades-gbuilder binary (survey build), --gfa"})
        # coverage (-c): DP:f / KC:i tags (CoverageHashMapBuilder + FillCoverageAndFlankingFromPHM)
        for name, ks, ts in (("tiny", (21,), (1,)), ("small", (21, 55), (1, 3)), ("loop", (21,), (1,)), ("polyA", (21,), (1,))):
            reads = [r for r in gsets[name][1] if r]
            with tempfile.TemporaryDirectory() as td:
                fq = os.path.join(td, "r.fq")
                with open(fq, "w") as f:
                    for i, r in enumerate(reads):
                        f.write(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n")
                for K in ks:
                    for T in ts:
                        out = os.path.join(td, f"gc_{K}_{T}.gfa")
                        subprocess.check_call([gb, fq, out, "-k", str(K), "-t", str(T), "-c", "--gfa", "-tmp-dir", os.path.join(td, f"tc{K}_{T}")],
                                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                        txt = open(out).read()
                        fn = f"graphcov_{name}_k{K}_t{T}.gfa"
                        open(os.path.join(HERE, fn), "w").write(txt)
                        manifest["cases"].append({"kind": "graph_cov", "reads": f"reads_{name}.txt", "K": K, "threads": T, "num_buckets": 10 * T,
                                                  "md5": md5(txt.encode()), "file": fn, "source": "spades-gbuilder binary (survey build), -c --gfa"})
        # --spades: SPAdes internal graph format (.grseq + .cvr), io/binary/graph.hpp:27-74, io/binary/coverage.hpp
        for name, K, T, cov in (("tiny", 21, 1, False), ("small", 21, 1, False), ("small", 21, 3, True), ("small", 55, 1, True),
                                ("loop", 21, 1, True), ("polyA", 21, 1, False), ("mixed", 33, 4, True)):
            reads = [r for r in gsets[name][1] if r]
            with tempfile.TemporaryDirectory() as td:
                fq = os.path.join(td, "r.fq")
                with open(fq, "w") as f:
                    for i, r in enumerate(reads):
                        f.write(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n")
                out = os.path.join(td, "sp")
                subprocess.check_call([gb, fq, out, "-k", str(K), "-t", str(T), "--spades", "-tmp-dir", os.path.join(td, "t")] + (["-c"] if cov else []),
                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                base = f"spades_{name}_k{K}_t{T}{'_c' if cov else ''}"
                for ext in (".grseq", ".cvr"):
                    open(os.path.join(HERE, base + ext), "wb").write(open(out + ext, "rb").read())
                manifest["cases"].append({"kind": "graph_spades", "reads": f"reads_{name}.txt", "K": K, "threads": T, "num_buckets": 10 * T,
                                          "coverage": cov, "base": base, "source": "spades-gbuilder binary (survey build), --spades"})
        # binary reads written by the reference's converter (kept in -tmp-dir): single_0.seq of reads_small / reads_tiny
        import shutil
        for name in ("small", "tiny"):
            reads = [r for r in gsets[name][1] if r]
            with tempfile.TemporaryDirectory() as td:
                fq = os.path.join(td, "r.fq")
                with open(fq, "w") as f:
                    for i, r in enumerate(reads):
                        f.write(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n")
                subprocess.check_call([gb, fq, os.path.join(td, "o.gfa"), "-k", "21", "-t", "1", "--gfa", "-tmp-dir", os.path.join(td, "t")],
                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                shutil.copy(os.path.join(td, "t", "single_0.seq"), os.path.join(HERE, f"binreads_{name}_single_0.seq"))
                manifest["cases"].append({"kind": "binary_reads", "reads": f"reads_{name}.txt", "file": f"binreads_{name}_single_0.seq",
                                          "source": "ReadConverter::ConvertToBinary via spades-gbuilder -tmp-dir (survey build)"})
        # --fastg (io/graph/fastg_writer.cpp)
        for name, K, T, cov in (("tiny", 21, 1, False), ("small", 21, 3, True), ("loop", 21, 1, True), ("polyA", 21, 1, False)):
            reads = [r for r in gsets[name][1] if r]
            with tempfile.TemporaryDirectory() as td:
                fq = os.path.join(td, "r.fq")
                with open(fq, "w") as f:
                    for i, r in enumerate(reads):
                        f.write(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n")
                out = os.path.join(td, "g.fastg")
                subprocess.check_call([gb, fq, out, "-k", str(K), "-t", str(T), "--fastg", "-tmp-dir", os.path.join(td, "t")] + (["-c"] if cov else []),
                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                fn = f"fastg_{name}_k{K}_t{T}{'_c' if cov else ''}.fastg"
                open(os.path.join(HERE, fn), "w").write(open(out).read())
                manifest["cases"].append({"kind": "graph_fastg", "reads": f"reads_{name}.txt", "K": K, "threads": T, "num_buckets": 10 * T,
                                          "coverage": cov, "file": fn, "source": "spades-gbuilder binary (survey build), --fastg"})
    json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1)
    print(f"{len(manifest['cases'])} cases written")


if __name__ == "__main__":
    main()
