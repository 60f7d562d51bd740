#!/usr/bin/env python3
"""integration/make_spades_case.py — test assets for the spades-core run with the MI355X Construction stage (build container only).

Runs the REFERENCE pipeline (its own spades.py and the spades-core of its own build tree, CPU) on two small datasets and keeps,
under integration/_build/spades_case/<name>/ (untracked; travels to the GPU box like the other built files):
    reads_1.fq.gz reads_2.fq.gz      the input
    run/...                          what spades.py generated for spades-core: K21/configs/*, dataset.info, input_dataset.yaml
    expected/...                     what the reference spades-core wrote: contigs, scaffolds, graphs
tests/test_integration_gpu.py recreates the run directory at the same absolute path on the GPU box, runs
integration/_build/spades-core-gpu on the SAME config and compares every expected file byte for byte.
spades.py is used from a scratch copy of the pipeline scripts (its developer mode looks for ../../../../bin next to itself and the
reference tree is read-only); nothing of the reference enters the repository."""
import gzip
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import synth  # noqa: E402

REF = os.environ.get("SMX_REF", "/root/reference")
BUILD = os.environ.get("SPADES_BUILD", "/tmp/spades_build2")
SCRATCH = "/tmp/smx_spades_root"
CASES = "/tmp/smx_spades_case"
KEEP = ["K21/final_contigs.fasta", "K21/scaffolds.fasta", "K21/before_rr.fasta", "K21/assembly_graph.fastg",
        "K21/assembly_graph_with_scaffolds.gfa", "K21/assembly_graph_after_simplification.gfa", "K21/final_contigs.paths"]


def scratch_root():
    shutil.rmtree(SCRATCH, ignore_errors=True)
    os.makedirs(os.path.join(SCRATCH, "src/projects/spades"))
    shutil.copytree(os.path.join(REF, "src/projects/spades/pipeline"), os.path.join(SCRATCH, "src/projects/spades/pipeline"))
    os.symlink(os.path.join(REF, "src/projects/spades/configs"), os.path.join(SCRATCH, "src/projects/spades/configs"))
    for p in ("hammer", "ionhammer", "corrector"):
        os.symlink(os.path.join(REF, "src/projects", p), os.path.join(SCRATCH, "src/projects", p))
    os.symlink(os.path.join(REF, "ext"), os.path.join(SCRATCH, "ext"))
    shutil.copy(os.path.join(REF, "VERSION"), os.path.join(SCRATCH, "VERSION"))
    os.makedirs(os.path.join(SCRATCH, "bin"))
    os.symlink(os.path.join(BUILD, "bin/spades-core"), os.path.join(SCRATCH, "bin/spades-core"))
    for b in ("spades-hammer", "spades-ionhammer", "spades-corrector-core", "spades-bwa"):  # --only-assembler never runs them
        with open(os.path.join(SCRATCH, "bin", b), "w") as f:
            f.write("#!/bin/sh\nexit 1\n")
        os.chmod(os.path.join(SCRATCH, "bin", b), 0o755)


def write_pairs(codes, p1, p2):
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    with gzip.open(p1, "wb") as f1, gzip.open(p2, "wb") as f2:
        for i in range(0, codes.shape[0], 2):
            a, b = lut[codes[i]].tobytes(), lut[codes[i + 1]].tobytes()
            f1.write(b"@p%d/1\n%s\n+\n%s\n" % (i // 2, a, b"I" * len(a)))
            f2.write(b"@p%d/2\n%s\n+\n%s\n" % (i // 2, b, b"I" * len(b)))


def make(name, r1, r2, ks=("21",)):
    case = os.path.join(CASES, name)
    shutil.rmtree(case, ignore_errors=True)
    os.makedirs(case)
    shutil.copy(r1, os.path.join(case, "reads_1.fq.gz"))
    shutil.copy(r2, os.path.join(case, "reads_2.fq.gz"))
    run = os.path.join(case, "run")
    subprocess.check_call([sys.executable, os.path.join(SCRATCH, "src/projects/spades/pipeline/spades.py"), "--only-assembler", "-k", ",".join(ks),
                           "-1", os.path.join(case, "reads_1.fq.gz"), "-2", os.path.join(case, "reads_2.fq.gz"), "-o", run, "-t", "4"],
                          stdout=subprocess.DEVNULL)
    out = os.path.join(HERE, "_build", "spades_case", name)
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(os.path.join(out, "run"))
    for f in ("reads_1.fq.gz", "reads_2.fq.gz"):
        shutil.copy(os.path.join(case, f), os.path.join(out, f))
    for f in ("dataset.info", "input_dataset.yaml"):
        shutil.copy(os.path.join(run, f), os.path.join(out, "run", f))
    for kk in ks:
        os.makedirs(os.path.join(out, f"expected/K{kk}"))
        os.makedirs(os.path.join(out, f"run/K{kk}"))
        shutil.copytree(os.path.join(run, f"K{kk}/configs"), os.path.join(out, f"run/K{kk}/configs"))
        for f in KEEP:
            f = f.replace("K21", f"K{kk}")
            if os.path.exists(os.path.join(run, f)):
                shutil.copy(os.path.join(run, f), os.path.join(out, "expected", f))
    tmp_dir = [l.split()[1] for l in open(os.path.join(run, "K21/configs/config.info")) if l.startswith("tmp_dir")][0]
    with open(os.path.join(out, "case.txt"), "w") as f:
        f.write(f"case_dir {case}\ntmp_dir {tmp_dir}\nks {','.join(ks)}\n")
    print(name, "->", out, [os.path.getsize(os.path.join(out, "expected", k)) for k in KEEP if os.path.exists(os.path.join(out, "expected", k))])


if __name__ == "__main__":
    scratch_root()
    d = os.path.join(REF, "src/projects/spades/test_dataset")
    make("ecoli_1K", os.path.join(d, "ecoli_1K_1.fq.gz"), os.path.join(d, "ecoli_1K_2.fq.gz"))  # BASELINE config 1: spades.py --test data
    os.makedirs(CASES, exist_ok=True)
    codes = synth.synth_codes(5, 60_000, 16_000, err=0.01, n_rate=0.001)  # 60 kbp genome at 40x: tips, bubbles, a few contigs
    p1, p2 = os.path.join(CASES, "s1.fq.gz"), os.path.join(CASES, "s2.fq.gz")
    write_pairs(codes, p1, p2)
    make("synth_60k", p1, p2)
    make("synth_60k_k21_33", p1, p2, ks=("21", "33"))  # the K33 iteration takes the K21 contigs as an extra stream (construction.cpp:108-117)
