"""GPU: FASTQ text cut into reads on the device (smx_submit_fastq_text) gives the same k-mer file as the host-parsed reads:
N runs (longest-run rule), CRLF, missing final newline, chunked text with a carried tail, lower case, empty sequences; anything
that is not strict 4-line FASTQ is refused without submitting a read."""
import hashlib

import numpy as np
import pytest

from conftest import load_manifest, read_lines
from test_count_gpu import _synth

pytestmark = pytest.mark.gpu


def _fastq(reads, eol="\n", last_newline=True):
    t = "".join(f"@read{i} some/description{eol}{r}{eol}+{eol}{'I' * len(r)}{eol}" for i, r in enumerate(reads))
    return (t if last_newline else t[:-len(eol)]).encode()


def _count_text(chunks_of, text, K, mode, nb):
    from spades_amd import KMerDiskCounter, ReadKMerSplitter
    sp = ReadKMerSplitter(K, mode)
    pos, buf = 0, b""
    while True:
        piece = text[pos:pos + chunks_of] if chunks_of else text[pos:]
        pos += len(piece)
        buf += piece
        final = pos >= len(text)
        used = sp.push_back_fastq_text(buf, is_final=final)
        buf = buf[used:]
        if final:
            assert buf == b""
            break
    st = KMerDiskCounter(None, sp).Count(nb)
    rec = st.records()
    sp.ctx.close()
    return rec


@pytest.mark.parametrize("eol,last_nl,chunk", [("\n", True, 0), ("\r\n", True, 0), ("\n", False, 0), ("\n", True, 5000), ("\r\n", False, 7777),
                                               ("\n", True, 1 << 20)])
def test_device_fastq_matches_host_parsed_reads(eol, last_nl, chunk):
    from oracle import oracle
    reads = _synth(3, 20000, 2000, 150) + ["acgtnACGT" * 10, "N" * 40, "A", "ACGTTGCA" * 30 + "N" + "C" * 100, "NNNNACGTACGTACGTACGTACGTAAA"]
    text = _fastq(reads, eol, last_nl)
    for K, mode, nb in ((21, "A", 16), (56, "B", 30)):
        ref, _ = oracle.count(reads, K, mode, nb)
        rec = _count_text(chunk, text, K, mode, nb)
        assert rec.shape == ref.shape and (rec == ref).all()


def test_device_fastq_matches_reference_golden():
    case = [c for c in load_manifest()["cases"] if c["kind"] == "count" and c["reads"] == "reads_small.txt" and c["K"] == 55 and c["mode"] == "A"
            and c["num_buckets"] == 16][0]
    text = _fastq([r for r in read_lines("reads_small.txt") if r])
    rec = _count_text(0, text, 55, "A", 16)
    assert hashlib.md5(rec.tobytes()).hexdigest() == case["md5"]


@pytest.mark.parametrize("text", [b">r1\nACGT\n>r2\nACGT\n", b"@r\nACGT\nACGT\n+\nIIII\nIIII\n@r2\nAC\n+\nII\n", b"@r\nACGT\n-\nIIII\n"])
def test_other_formats_are_refused_untouched(text):
    from spades_amd import ReadKMerSplitter, SmxError
    sp = ReadKMerSplitter(21, "A")
    with pytest.raises(SmxError) as e:
        sp.push_back_fastq_text(text)
    assert e.value.code == 64  # SMX_INVALID_INPUT_FORMAT
    assert sp.ctx.reads_info()[0] == 0
    sp.ctx.close()


def test_async_and_sync_submissions_can_be_mixed():
    """An asynchronous packed submission (its extent check runs on the copy stream long after the call returned) followed by a
    synchronous one (which releases the call's temporaries): the device block the asynchronous check writes to belongs to its read
    chunk, not to the temporaries — the count equals that of two synchronous submissions (smx_api.hip submit_packed_async)."""
    from spades_amd import KMerDiskCounter, ReadKMerSplitter
    from spades_amd.kmercount import Context
    rng = np.random.default_rng(11)
    L, K, nb = 150, 25, 16
    n_big = ((1 << 24) * 32) // L + 1000  # >= 2^24 words: the asynchronous path
    res = []
    for use_async in (True, False):
        ctx = Context()
        sp = ReadKMerSplitter(K, "B", ctx)
        for n, asyn in ((n_big, use_async), (5000, False), (n_big // 4, use_async)):
            words = rng.integers(0, 1 << 63, (n * L + 31) // 32 + 8, dtype=np.int64).view(np.uint64)
            start = (np.arange(n, dtype=np.uint64) * L)
            ln = np.full(n, L, dtype=np.uint32)
            ctx.set_option("async_upload", 1 if asyn else 0)
            sp.push_back_packed(words[:-8], start, ln)
        st = KMerDiskCounter(None, sp).Count(nb)
        res.append((st.total_kmers(), st.kmer_instances(), tuple(int(x) for x in st.bucket_sizes())))
        ctx.close()
        rng = np.random.default_rng(11)
    assert res[0] == res[1]
