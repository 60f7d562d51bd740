"""GPU: the both-strands count (spades-kmercount, kmercount.cpp:48-122) held as two strands — the sorted canonical set and the sorted
set of its reverse complements, merged bucket by bucket (option two_strand: 1 = merged into one array at once, 2 = left in two strands,
every accessor merging on demand: the path of inputs whose direct expansion does not fit HBM) — against the oracle and the goldens."""
import hashlib

import numpy as np
import pytest

from conftest import load_manifest, read_lines
from test_count_gpu import _synth

pytestmark = pytest.mark.gpu


def _count_ts(reads, K, nb, ts, tmp_path=None, parts=0):
    from spades_amd import KMerDiskCounter, ReadKMerSplitter
    from spades_amd.kmercount import Context
    ctx = Context()
    ctx.set_option("prededupe", 1)
    ctx.set_option("two_strand", ts)
    if parts:
        ctx.set_option("two_strand_parts", parts)
    sp = ReadKMerSplitter(K, "A", ctx)
    sp.push_back_reads(reads)
    st = KMerDiskCounter(str(tmp_path) if tmp_path else None, sp).Count(nb)
    rec, sizes = st.records(), st.bucket_sizes()
    per_bucket = [st.bucket(b) for b in range(nb)]
    dptr = st.device_ptr()
    md5 = None
    if tmp_path is not None:
        st.merge()
        md5 = hashlib.md5(open(st.final_kmers(), "rb").read()).hexdigest()
    ctx.close()
    return rec, sizes, per_bucket, dptr, md5


@pytest.mark.parametrize("K,nb", [(21, 16), (32, 16), (55, 16), (64, 3), (77, 16), (96, 7), (127, 16), (128, 1)])
@pytest.mark.parametrize("ts", [1, 2])
def test_two_strand_count_equals_oracle(K, nb, ts, tmp_path):
    from oracle import oracle
    # both strands, N runs, low complexity, and (even K) k-mers that are their own reverse complement: (ACGT)^n, (AT)^n
    reads = _synth(300 + K, 20000, 2500, 150) + ["A" * 150] * 20 + ["ACGT" * 40] * 20 + ["AT" * 75] * 10 + ["GAATTC" * 25] * 10
    ref, rs = oracle.count(reads, K, "A", nb)
    rec, sizes, per_bucket, dptr, md5 = _count_ts(reads, K, nb, ts, tmp_path)
    assert (sizes == rs).all()
    assert rec.shape == ref.shape and (rec == ref).all()
    off = np.concatenate([[0], np.cumsum(rs.astype(np.int64))])
    for b in range(nb):
        assert (per_bucket[b] == ref[off[b]:off[b + 1]]).all()
    assert md5 == hashlib.md5(ref.tobytes()).hexdigest()
    assert (dptr != 0) == (ts == 1 or len(ref) == 0)  # two strands: no single resident array to point to


@pytest.mark.parametrize("K,nb,parts", [(21, 16, 2), (32, 16, 4), (55, 16, 3), (77, 5, 8), (128, 7, 2)])
def test_reverse_complements_sorted_in_bucket_ranges(K, nb, parts):
    """two_strand_parts: the reverse complements are sorted range by range of their buckets (what a count does whose three buffers of
    |C| records do not fit next to each other)"""
    from oracle import oracle
    reads = _synth(500 + K, 20000, 2500, 150) + ["ACGT" * 40] * 20 + ["AT" * 75] * 10
    ref, rs = oracle.count(reads, K, "A", nb)
    for ts in (1, 2):
        rec, sizes, per_bucket, _, _ = _count_ts(reads, K, nb, ts, parts=parts)
        assert (sizes == rs).all() and rec.shape == ref.shape and (rec == ref).all()


CASES = [c for c in load_manifest()["cases"] if c["kind"] == "count" and c["K"] >= 21 and c["mode"] == "A"]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c['reads'][6:-4]}-{c['mode']}{c['K']}-b{c['num_buckets']}")
def test_goldens_in_two_strands(case):
    rec, sizes, _, _, _ = _count_ts(read_lines(case["reads"]), case["K"], case["num_buckets"], 2)
    assert list(map(int, sizes)) == case["bucket_sizes"]
    assert hashlib.md5(rec.tobytes()).hexdigest() == case["md5"]


@pytest.mark.parametrize("K,nb", [(55, 16), (33, 5), (96, 16)])
@pytest.mark.parametrize("ts", [-1, 0, 1, 2])
@pytest.mark.parametrize("spill", [-1, 1])
@pytest.mark.parametrize("batch", [40000, 7000])
def test_two_strand_inside_position_batches(K, nb, ts, spill, batch):
    """a both-strands batch that comes back as a two-strand view (no room for the merged array, or two_strand = 2) has no single array
    for the batch loop of count_reads to fold or spill — round 4 took the null pointer for a run (error 70 on the MI355X under a budget, on the
    stand-in every batch but the last dropped with rc = 0). Its strands are runs of their own now. Every way into and out of the loop: the view left
    unmerged (2), merged when there is room (1, -1), never made (0) × folded on the device or spilled × few and many batches."""
    from oracle import oracle
    from spades_amd import KMerDiskCounter, ReadKMerSplitter
    from spades_amd.kmercount import Context
    reads = _synth(900 + K, 8000, 1200, 150) + ["ACGT" * 40] * 5 + ["AT" * 75] * 5
    ref, rs = oracle.count(reads, K, "A", nb)
    ctx = Context()
    ctx.set_option("prededupe", 1)
    ctx.set_option("skm_nkey_log2", 12)
    ctx.set_option("two_strand", ts)
    ctx.set_option("spill", spill)
    ctx.set_option("batch_records", batch)
    sp = ReadKMerSplitter(K, "A", ctx)
    sp.push_back_reads(reads)
    st = KMerDiskCounter(None, sp).Count(nb)
    rec, sizes = st.records(), st.bucket_sizes()
    assert (sizes == rs).all(), (int(sizes.sum()), int(rs.sum()))
    assert rec.shape == ref.shape and (rec == ref).all()
    off = np.concatenate([[0], np.cumsum(rs.astype(np.int64))])
    for b in (0, nb // 2, nb - 1):
        assert (st.bucket(b) == ref[off[b]:off[b + 1]]).all()
    ctx.close()
