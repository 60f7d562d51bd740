#!/usr/bin/env python3
"""one-screen summary of a bench.py JSON line"""
import json, sys
d = json.load(open(sys.argv[1]))
# rounds 1-4 reported the upload-inclusive rate as `value`; since round 5 that figure is pcie_inclusive.value (value_definition in the line)
like_r4 = (d.get("pcie_inclusive") or {}).get("value", d["value"] if d.get("config", {}).get("h2d_in_timed_region", True) else None)
print("comparable with BENCH_r01..r04 `value` (upload inside the step):", like_r4)
print("value", d["value"], "ms/step", d["ms_per_step"], d.get("step_breakdown_ms"), "pcie_inclusive", (d.get("pcie_inclusive") or {}).get("value"), (d.get("pcie_inclusive") or {}).get("ms_per_step"))
st = d["roofline"]["stages_ms"]
con = ("rank_dir", "fill_masks", "candidates", "walk_len", "keep", "walk_write", "succ")
agg = {}
for k, v in st.items():
    p = k.split(":")[0] if ":" in k else ("count" if k not in con else k)
    agg[p] = agg.get(p, 0) + v
print({k: round(v, 1) for k, v in agg.items()})
print("count:", {k: round(v, 1) for k, v in st.items() if ":" not in k and k not in con and v > 3})
print("kmers:", {k[6:]: round(v, 1) for k, v in st.items() if k.startswith("kmers:") and v > 3})
print("roofline count frac", d["roofline"]["frac"], "construct frac", d.get("construct", {}).get("roofline", {}).get("frac"))
cb = d.get("cpu_baseline")
if cb:
    print("cpu count", cb["value"], cb.get("bit_identical_to_reference_output"), "construct", cb.get("construct", {}).get("value"), cb.get("construct", {}).get("unitig_multiset_identical_to_reference"))
if "kmercount_mode" in d:
    print("kmercount mode", d["kmercount_mode"]["M_reads_per_s"], d["kmercount_mode"]["roofline_frac"])
