#!/usr/bin/env python3
"""Per-kernel roofline table of the bench step from what a round recorded under profiles/rNN/: the bench line (stage times
from HIP events, the counts of the run), the rocprofv3 kernel averages and the PMC traffic table. Algorithmic bytes per kernel are the
formulas bench.py prices its `dominant_kernel` with (bench.py: `single`), evaluated on the recorded counts.
usage: kernel_roofline_table.py [profiles/r04]"""
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04")
b = json.load(open(os.path.join(d, "bench_config3.json")))
st = b["roofline"]["stages_ms"]
c = b["construct"]
rs = c["route_stats"]
n_reads, L, nw, W = b["config"]["reads_per_gpu"], 150, 2, 16
D0, nslots, nchunks, n_cand, ne, nbases = c["n_kmers"], rs["superkmer_slots"], rs["chunks"], rs["start_de_edges"], c["n_unitigs"], c["unitig_bases"]
fused = st.get("pm_tab", 99.0) < 0.25 * st.get("pm_remote", 1.0)  # round 6: the dedupe stage wrote the node table of the clean chunks (pm_fuse_tab)
pmc = {}
for line in open(os.path.join(d, "config3_pm_pmc_hbm_traffic.csv")):
    f = line.strip().rsplit(",", 5)
    if len(f) >= 6 and f[0] not in ("kernel", "TOTAL") and not f[0].startswith("#"):
        try:
            pmc[f[0]] = (float(f[3]), float(f[5]))
        except ValueError:
            pass
rows = [  # stage, kernel substring in the PMC table, algorithmic bytes, unit the bytes are per, what bounds it (SQ counters)
    ("kmers:mark_windows", "k_mark_windows", n_reads * 12 + n_reads * L / 8 * 2, "12 B (start, len) per read in, 2 bits per position out", "HBM (streaming)"),
    ("kmers:skm_count", "k_skm_scan", n_reads * L / 4 + n_reads * 12 + nslots * (8 * 2 * nw + 8) + nslots * 16, "37.5 B per read in; 56 B per super-k-mer out (staged slot, partition word, counter)", "VALU issue + counting atomics"),
    ("kmers:skm_scatter", "k_skm_permute", nslots * (2 * 8 * 2 * nw + 8 + 8), "80 B per super-k-mer (staged slot in, slot out, two words)", "requests: one line for the partition's offset, one filled for the 32-B slot (~48 G lines/s)"),
    ("kmers:skm_plan", "k_skm_plan", nslots * 8 * 2 * nw + nchunks * 16, "32 B per slot in, 16 B per chunk out", "latency of the slot loads"),
    ("kmers:skm_dedupe", "k_skm_dedupe2", nslots * 8 * 2 * nw + D0 * (W + 1 + (24 if fused else 4)) + nchunks * 4 * (512 if fused else 256),
     ("32 B per slot in; 41 B per distinct k-mer out (record, byte, two node entries, two jump words); 2 KB per chunk" if fused else "32 B per slot in; 21 B per distinct k-mer out; 1 KB per chunk"), "VALU issue + LDS round trips (hash inserts; the node table of the chunk)"),
    ("pm_tab", "k_pm_tab", (0 if fused else D0 * (1 + 4) + 2 * D0 * 8 + 2 * D0 * 4), ("the cut partitions' tail only (k_pm_tab_dirty): the clean chunks' table leaves the dedupe stage" if fused else "5 B per k-mer in, 24 B out (two node entries, two jump words)"), "HBM writes"),
    ("pm_remote", "k_pm_remote", 2 * D0 * 8 + 0.1 * 2 * D0 * (W + 8 + 4 + W + 8), "16 B per k-mer scanned + 52 B per successor outside its chunk (5 %)", "requests (~5 lines of 128 B per lookup at ~48 G lines/s)"),
    ("walk_len", "k_pm_walk_len", n_cand * (8 + 2 * W + 4 + (4 + 8) + W + 8 * 3 + 1), "97 B per start de-edge (junction record, group word + probed record, jump word + node entry, last record, results)", "requests (~5.5 lines per start de-edge)"),
    ("walk_write", "k_pm_walk_write", ne * (8 * 5 + W + W + 8 + 32) + nbases / 4, "112 B per kept path (bookkeeping, start and last record, place word, edge record) + 2 bits per base", "requests (~5 lines per kept path)"),
]
tot_ms = b["ms_per_step"]
print("| kernel | ms / step | % of step | algorithmic GB (per unit) | alg. GB/s (frac of 8 TB/s) | PMC fetch + write GB | moved GB/s | bound by |")
print("|---|---|---|---|---|---|---|---|")
for stage, key, alg, unit, bound in rows:
    ms = st[stage]
    tr = [v for k, v in pmc.items() if key in k]
    fe, wr = (sum(x[0] for x in tr), sum(x[1] for x in tr)) if tr else (0.0, 0.0)
    print(f"| `{key}` | {ms:.1f} | {100 * ms / tot_ms:.0f} | {alg / 1e9:.1f} ({unit}) | {alg / ms / 1e6:.0f} ({alg / ms / 1e6 / 8000:.2f}) | "
          f"{fe:.0f} + {wr:.0f} | {(fe + wr) / ms * 1e3:.0f} | {bound} |")
