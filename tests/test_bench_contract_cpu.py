"""CPU: the bench line this round recorded on the GPU box (the latest profiles/rNN/bench_config3.json, printed by `python bench.py`) carries every
field of the driver's contract and its numbers agree with each other."""
import glob
import json
import os

from conftest import ROOT


def _latest():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_config3.json")))
    assert files
    return json.load(open(files[-1]))


def test_recorded_bench_line_has_the_contract_fields():
    d = _latest()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "M reads/s" and d["higher_is_better"] is True and d["n_gpus"] == 1 and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["algorithmic_bytes_per_step"] / (r["kernel_ms_per_step"] * 1e-3) / 1e9) < 1.0
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["bit_identical_to_reference_output"] is True
    # value = whole-job reads per second of the timed steps
    assert abs(d["value"] - d["config"]["reads_per_gpu"] / d["ms_per_step"] / 1e3) < 0.01 * d["value"]
    # round 5: `value` is measured on reads resident in HBM; the same step with the upload inside is reported beside it and is the slower one
    if "pcie_inclusive" in d:
        p = d["pcie_inclusive"]
        assert d["config"]["h2d_in_timed_region"] is False and p["unit"] == d["unit"]
        assert 0 < p["value"] < d["value"] and p["ms_per_step"] > d["ms_per_step"]
        assert abs(p["value"] - d["config"]["reads_per_gpu"] / p["ms_per_step"] / 1e3) < 0.01 * p["value"]
    # the dominant kernel is the longest single-kernel stage of the line and is priced on bytes of its own
    dk = d["dominant_kernel"]
    assert dk["ms"] == max(v for k, v in r["stages_ms"].items()) and 0 < dk["frac"] < 1


def test_the_pmc_table_bench_quotes_belongs_to_the_kernels_that_are_built():
    """roofline.traffic is read from the PMC passes recorded under profiles/ and quoted only while they describe the kernels that run: the
    library sources are the ones the table was taken on, or — after host-side edits — the machine code of the kernels is (tools/devcode_hash.py:
    .text + kernel descriptors of the gfx950 code object). A kernel edit makes bench.py print traffic null until tools/profile_bench.sh has
    been rerun; this test says which of the two holds."""
    import sys
    import pytest
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from devcode_hash import device_code_hash
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "config3_pm_pmc_hbm_traffic.csv")))
    assert files
    head = open(files[-1]).readline()
    assert "src_sha256=" in head and "dev_sha256=" in head
    dev = head.split("dev_sha256=")[1].split()[0]
    lib = os.path.join(ROOT, "spades_amd", "csrc", "libspades_mi355x.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    now = device_code_hash(lib)
    if now is None:
        pytest.skip("no llvm-objcopy / clang-offload-bundler here")
    if now == dev:
        return
    # other kernels were added or changed: the table still stands while the machine code of the kernels IT names is what it was taken on
    from devcode_hash import kernel_code_hash, pmc_table_kernels
    assert "kern_sha256=" in head
    kern = head.split("kern_sha256=")[1].split()[0]
    now_k = kernel_code_hash(pmc_table_kernels(files[-1]), lib)
    if now_k != kern:
        pytest.skip(f"kernels of the step changed since the PMC table was taken ({kern} -> {now_k}): bench.py prints traffic null until the PMC passes are rerun")
