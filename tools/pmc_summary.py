#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_bench.sh into the two tables kept under profiles/:
  <dir>/kernel_stats.csv      the --stats table (as written by rocprofv3)
  <dir>/pmc_hbm_traffic.csv   HBM GB per kernel for ONE step: FETCH_SIZE (KB) x2 (gfx950 correction, MI355X_MICROARCH.md HBM
                               section; calibrated on a 4 GiB copy with tools/ubench) and WRITE_SIZE (KB)
The PMC runs execute setup + 1 step = S passes of the pipeline; per-step = total / S (S = dispatches of k_mark_windows)."""
import csv, glob, hashlib, os, shutil, sys
from collections import defaultdict

d = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def src_hash():
    """sha256 over the library sources: bench.py quotes a PMC table only while the kernels it was taken on are the ones that run"""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "spades_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "spades_amd", "csrc", "*.hpp")) +
                    [os.path.join(ROOT, "include", "smx.h")]):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def find(sub, suffix):
    f = glob.glob(os.path.join(d, sub, "**", "*" + suffix), recursive=True)
    return f[0] if f else None


ks = find("kt", "kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(d, "kernel_stats.csv"))
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = find(c, "counter_collection.csv")
    acc = defaultdict(lambda: [0, 0.0])
    if f:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != c:
                continue
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[name][0] += 1
            acc[name][1] += float(row["Counter_Value"])
    tot[c] = acc
names = sorted(set(tot["FETCH_SIZE"]) | set(tot["WRITE_SIZE"]), key=lambda n: -(tot["FETCH_SIZE"][n][1] * 2 + tot["WRITE_SIZE"][n][1]))
# passes of the hot path in the PMC run (setup pass + timed steps + the synchronous-upload step the stage times come from): one
# k_cand_tiles launch per construction
passes = max(1, (tot["FETCH_SIZE"].get("smx::k_cand_tiles") or tot["FETCH_SIZE"].get("smx::k_mark_windows", [1]))[0])
with open(os.path.join(d, "pmc_hbm_traffic.csv"), "w") as o:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from devcode_hash import device_code_hash
    o.write(f"# src_sha256={src_hash()} dev_sha256={device_code_hash()} kern_sha256=@KERN@ (library sources the counters were taken on; sha256 over .text + .rodata "
            "of the gfx950 code object they compile to; sha256 over the machine code + descriptors of the kernels of this table alone — "
            "tools/devcode_hash.py; bench.py quotes the table while any of the three is unchanged)\n")
    o.write("kernel,launches_per_step,FETCH_SIZE_KB(raw),fetch_GB(x2 gfx950 correction),WRITE_SIZE_KB,write_GB\n")
    tf = tw = 0.0
    for n in names:
        if not n.startswith("smx::"):
            continue
        nl, f = tot["FETCH_SIZE"][n]
        _, w = tot["WRITE_SIZE"][n]
        f, w = f / passes, w / passes
        fg, wg = f * 1024 * 2 / 1e9, w * 1024 / 1e9
        tf += fg
        tw += wg
        o.write(f"{n},{nl / passes:g},{f:.0f},{fg:.2f},{w:.0f},{wg:.2f}\n")
    o.write(f"TOTAL,,,{tf:.1f},,{tw:.1f}\n")
# the hash over the kernels the table names (they are known only now)
from devcode_hash import kernel_code_hash, pmc_table_kernels  # noqa: E402
_p = os.path.join(d, "pmc_hbm_traffic.csv")
_t = open(_p).read().replace("@KERN@", str(kernel_code_hash(pmc_table_kernels(_p))), 1)
open(_p, "w").write(_t)
print(f"passes={passes} fetch {tf:.1f} GB + write {tw:.1f} GB = {tf + tw:.1f} GB per step")
