"""GPU parity: HIP k-mer counting through the C ABI vs the oracle and the reference goldens. Bit-exact."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_manifest, read_lines

pytestmark = pytest.mark.gpu

CASES = [c for c in load_manifest()["cases"] if c["kind"] == "count"]


def _count(reads, K, mode, nb, opts=None):
    from spades_amd import KMerDiskCounter, ReadKMerSplitter
    sp = ReadKMerSplitter(K, mode)
    for k, v in (opts or {}).items():
        sp.ctx.set_option(k, v)
    sp.push_back_reads(reads)
    st = KMerDiskCounter(None, sp).Count(nb)
    rec, sizes = st.records(), st.bucket_sizes()
    sp.ctx.close()
    return rec, sizes


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c['reads'][6:-4]}-{c['mode']}{c['K']}-b{c['num_buckets']}")
def test_count_matches_reference_golden(case):
    reads = read_lines(case["reads"])
    rec, sizes = _count(reads, case["K"], case["mode"], case["num_buckets"])
    assert list(map(int, sizes)) == case["bucket_sizes"]
    assert hashlib.md5(rec.tobytes()).hexdigest() == case["md5"]
    if "file" in case:
        assert rec.tobytes() == open(os.path.join(GOLDEN, case["file"]), "rb").read()


def _synth(seed, glen, n, L, err=0.01, nrate=0.002):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, glen)
    reads = []
    for _ in range(n):
        p = int(rng.integers(0, glen - L + 1))
        r = g[p:p + L].copy()
        e = rng.random(L) < err
        r[e] = (r[e] + rng.integers(1, 4, int(e.sum()))) % 4
        s = np.array(list("ACGT"))[r]
        s[rng.random(L) < nrate] = "N"
        s = "".join(s)
        if rng.random() < 0.5:
            s = s[::-1].translate(str.maketrans("ACGT", "TGCA"))
        reads.append(s)
    return reads


@pytest.mark.parametrize("K,mode,nb", [(21, "A", 16), (55, "A", 16), (22, "B", 10), (56, "B", 80), (77, "A", 16),
                                       (78, "B", 30), (127, "A", 16), (128, "B", 20), (31, "A", 16), (32, "B", 16),
                                       (33, "A", 3), (64, "A", 1), (65, "B", 7), (5, "A", 16), (1, "A", 16)])
def test_count_vs_oracle_seeded(K, mode, nb):
    from oracle import oracle
    reads = _synth(100 + K, 20000, 3000, 150)
    ref, rs = oracle.count(reads, K, mode, nb)
    rec, sizes = _count(reads, K, mode, nb)
    assert (sizes == rs).all()
    assert rec.shape == ref.shape and (rec == ref).all()


@pytest.mark.parametrize("opts", [{"leaf_cap": 64}, {"leaf_cap": 8}, {"leaf_cap": 4}, {"leaf_cap": 64, "s1": 2, "s2": 3}, {"s1": 0, "s2": 0, "leaf_cap": 128},
                                  {"s1": 3, "s2": 0}, {"s1": 0, "s2": 6, "leaf_cap": 256}])
@pytest.mark.parametrize("K,mode,nb", [(21, "A", 16), (55, "B", 30), (77, "A", 16)])
def test_multilevel_and_oversized_bins(opts, K, mode, nb):
    """Forces tiny leaves / explicit splits so the level-2 pass and the merge path (oversized bins) run."""
    from oracle import oracle
    reads = _synth(7, 5000, 1500, 120) + ["A" * 150] * 40 + ["ACGT" * 30] * 10
    ref, rs = oracle.count(reads, K, mode, nb)
    rec, sizes = _count(reads, K, mode, nb, opts)
    assert (sizes == rs).all()
    assert rec.shape == ref.shape and (rec == ref).all()


@pytest.mark.parametrize("K,mode,nb", [(21, "A", 16), (55, "B", 30), (77, "A", 16), (127, "B", 10)])
def test_medium_leaves_take_the_lds_kernel(K, mode, nb):
    """few bins with 257..cap records each -> the workgroup-level LDS counting-split kernel."""
    from oracle import oracle
    reads = _synth(11, 3000, 260, 150)
    ref, rs = oracle.count(reads, K, mode, nb)
    rec, sizes = _count(reads, K, mode, nb, {"s1": 2, "s2": 0})
    assert (sizes == rs).all() and rec.shape == ref.shape and (rec == ref).all()


@pytest.mark.parametrize("K,mode,nb,batch", [(21, "A", 16, 100_000), (55, "A", 16, 40_000), (56, "B", 80, 30_000), (77, "B", 30, 200_000),
                                             (21, "A", 16, 1_000)])
def test_multi_batch_runs_are_merged(K, mode, nb, batch):
    """HBM-bounded batches (forced tiny here): per-range sorted-unique runs folded by the same pipeline (the reference's
    dump + loser-tree merge, kmer_splitter.hpp:123-170 / kmer_index_builder.hpp:346-430) -> same bytes as one batch."""
    from oracle import oracle
    reads = _synth(21, 6000, 1200, 150) + ["A" * 150] * 30
    if batch <= 1000:
        reads = reads[:60]
    ref, rs = oracle.count(reads, K, mode, nb)
    rec, sizes = _count(reads, K, mode, nb, {"batch_records": batch})
    assert (sizes == rs).all() and rec.shape == ref.shape and (rec == ref).all()


def test_edge_inputs():
    from oracle import oracle
    for reads in ([], [""], ["N" * 50], ["ACG"], ["A" * 21], ["A" * 20 + "N" + "C" * 25, "acgtacgtacgtacgtacgtacgtacgt"]):
        ref, rs = oracle.count(reads, 21, "A", 16)
        rec, sizes = _count(reads, 21, "A", 16)
        assert (sizes == rs).all() and rec.shape == ref.shape and (rec == ref).all()


def test_write_final_kmers_file(tmp_path):
    from spades_amd import KMerDiskCounter, ReadKMerSplitter
    case = [c for c in CASES if "file" in c and c["mode"] == "A" and c["K"] == 21][0]
    sp = ReadKMerSplitter(21, "A")
    sp.push_back_reads(read_lines(case["reads"]))
    st = KMerDiskCounter(str(tmp_path), sp).CountAll(16, 4, merge=True)
    assert open(st.final_kmers(), "rb").read() == open(os.path.join(GOLDEN, case["file"]), "rb").read()
    assert st.total_kmers() == case["n_records"]
    for b in range(16):
        assert st.bucket(b).shape[0] == case["bucket_sizes"][b]


def test_invalid_parameters():
    from spades_amd import KMerDiskCounter, ReadKMerSplitter, SmxError
    sp = ReadKMerSplitter(129, "A")
    sp.push_back_reads(["ACGT" * 40])
    with pytest.raises(SmxError) as e:
        KMerDiskCounter(None, sp).Count(16)
    assert e.value.code == 67  # InvalidParameter
    sp2 = ReadKMerSplitter(21, "A", sp.ctx)
    with pytest.raises(SmxError):
        KMerDiskCounter(None, sp2).Count(0)


def test_spades_binary_reads_input():
    """<prefix>.seq written by the reference's ReadConverter gives the same k-mer file as the ASCII reads it was made from."""
    from spades_amd import KMerDiskCounter, ReadKMerSplitter
    for c in [c for c in load_manifest()["cases"] if c["kind"] == "binary_reads"]:
        for K, mode, nb in ((21, "A", 16), (33, "A", 16), (32, "B", 10), (55, "B", 30)):
            gold = [g for g in CASES if g["reads"] == c["reads"] and g["K"] == K and g["mode"] == mode and g["num_buckets"] == nb][0]
            sp = ReadKMerSplitter(K, mode)
            sp.push_back_binary(os.path.join(GOLDEN, c["file"]))
            st = KMerDiskCounter(None, sp).Count(nb)
            assert hashlib.md5(st.records().tobytes()).hexdigest() == gold["md5"]
            sp.ctx.close()
    from spades_amd import SmxError
    sp = ReadKMerSplitter(21, "A")
    with pytest.raises(SmxError) as e:
        sp.push_back_binary("/nonexistent.seq")
    assert e.value.code == 65


def test_trim_and_one_parked_arena_per_device():
    """smx_trim gives the free physical memory of the context's arena back to the device and the context keeps working; destroyed
    contexts park at most ONE arena per device (a context of another budget class tears the parked one down)"""
    import torch
    from spades_amd import KMerDiskCounter, ReadKMerSplitter
    from spades_amd.kmercount import Context
    rng = np.random.default_rng(2)
    reads = ["".join(rng.choice(list("ACGT"), 150)) for _ in range(3000)]

    def count(ctx):
        sp = ReadKMerSplitter(33, "A", ctx)
        sp.push_back_reads(reads)
        st = KMerDiskCounter(None, sp).Count(16)
        return st.records().tobytes()

    free0 = torch.cuda.mem_get_info()[0]
    ctx = Context()
    a = count(ctx)
    got = ctx.trim()
    assert got >= 0
    assert count(ctx) == a          # what was unmapped is mapped again on demand
    ctx.close()
    for budget in (2 << 30, 0, 3 << 30, 0):  # alternating budget classes: never more than one parked arena
        c = Context(hbm_budget=budget) if budget else Context()
        assert count(c) == a
        c.close()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (8 << 30), "parked arenas pile up"


def test_pool_blocks_for_the_callers_own_arrays():
    """smx_pool_alloc / smx_pool_free (round 5): blocks of the context's device arena for a neighbour's working arrays (dist.distributed_walks keeps its per-node
    state there). Distinct, aligned, returned once; a block that cannot be had is the memory-limit code; what the caller forgets ends with the context; a
    count next to live blocks is still the oracle's."""
    import ctypes as C
    from oracle import oracle
    from spades_amd import KMerDiskCounter, ReadKMerSplitter
    from spades_amd.kmercount import Context
    ctx = Context(hbm_budget=1 << 30)
    lib, h = ctx.lib, ctx._h
    a, b = C.c_void_p(), C.c_void_p()
    assert lib.smx_pool_alloc(h, 1000, C.byref(a)) == 0 and lib.smx_pool_alloc(h, 5 << 20, C.byref(b)) == 0
    assert a.value and b.value and a.value != b.value and a.value % 256 == 0 and b.value % 256 == 0
    big = C.c_void_p(123)
    assert lib.smx_pool_alloc(h, 8 << 30, C.byref(big)) == 68 and not big.value  # beyond the context's budget
    reads = _synth(5, 3000, 300, 100)
    sp = ReadKMerSplitter(21, "A", ctx)
    sp.push_back_reads(reads)
    st = KMerDiskCounter(None, sp).Count(16)
    ref, rs = oracle.count(reads, 21, "A", 16)
    assert (st.bucket_sizes() == rs).all() and (st.records() == ref).all()
    assert lib.smx_pool_free(h, a) == 0
    assert lib.smx_pool_free(h, a) == 67  # not (any more) a block of the pool
    assert lib.smx_pool_free(h, None) == 0
    ctx.close()  # (b goes with the context)
