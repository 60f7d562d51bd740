"""One-rank launches of spades-gbuilder-mi355x --gpus 1 in every variant the tests use, each under a short timeout, with the tool's
phase marks: which launch (if any) does not finish, and where it stands."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import GOLDEN, load_manifest, read_lines  # noqa: E402
GB = os.path.join(ROOT, "spades_amd", "tools", "spades-gbuilder-mi355x")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 80.0
t_start = time.time()
td = "/dev/shm/mgpu_diag"
os.makedirs(td, exist_ok=True)
reads = [r for r in read_lines("reads_small.txt") if r]
fa, fq = td + "/r.fa", td + "/r.fq"
with open(fa, "w") as f:
    for i, r in enumerate(reads):
        f.write(f">r{i}\n{r}\n")
with open(fq, "w") as f:
    for i, r in enumerate(reads):
        f.write(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n")
man = load_manifest()["cases"]
variants = [("fa self-rccl", fa, {"SMX_MGPU_SELF_RCCL": "1"}), ("fq parts3 chunk2048 copy", fq, {"SMX_MGPU_PARTS": "3", "SMX_MGPU_CHUNK": "2048"}),
            ("fq kpomers parts2 self-rccl", fq, {"SMX_MGPU_KPOMERS": "1", "SMX_MGPU_PARTS": "2", "SMX_MGPU_SELF_RCCL": "1"}),
            ("fq copy", fq, {}), ("fq kpomers copy", fq, {"SMX_MGPU_KPOMERS": "1"})]
n = 0
for rnd in range(3):
    for kind, cov in (("graph", []), ("graph_cov", ["-c"])):
        for c in [c for c in man if c["kind"] == kind and c["file"] and c["reads"] == "reads_small.txt" and c["K"] in (21, 55) and c["threads"] == 3]:
            for name, inp, env in variants:
                if time.time() - t_start > budget:
                    print("budget used", flush=True)
                    sys.exit(0)
                out = td + "/g.gfa"
                if os.path.exists(out):
                    os.remove(out)
                t0 = time.time()
                try:
                    r = subprocess.run([GB, inp, out, "-k", str(c["K"]), "-t", "3", "--gfa", "--gpus", "1"] + cov, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                                       env=dict(os.environ, SMX_DEBUG="1", **env), timeout=15)
                    ok = r.returncode == 0 and open(out).read() == open(os.path.join(GOLDEN, c["file"])).read()
                    print(f"#{n} k={c['K']} {cov} {name}: rc={r.returncode} same={ok} {time.time() - t0:.2f}s", flush=True)
                    if not ok:
                        print(r.stderr.decode(errors="replace")[-1500:], flush=True)
                except subprocess.TimeoutExpired as e:
                    print(f"#{n} k={c['K']} {cov} {name}: TIMEOUT after {time.time() - t0:.1f}s; stderr so far:", flush=True)
                    print((e.stderr or b"").decode(errors="replace")[-2500:], flush=True)
                n += 1
print("all done", n, flush=True)
