"""CPU: the block-parallel BGZF reader of the CLI clones (spades_amd/tools/bgzf_reader.hpp): the text of a blocked-gzip file comes out in
order and complete for every buffer size and thread count, damaged and truncated files are refused, an ordinary gzip file is not taken for
BGZF (it keeps the zlib path). The header depends on zlib only and is compiled alone."""
import gzip
import os
import random
import struct
import subprocess
import zlib

from conftest import ROOT

DRIVER = r"""
#include "%s/spades_amd/tools/bgzf_reader.hpp"
#include <cstdio>
int main(int argc, char **argv) {
    if (argc < 5) return 2;
    const std::string path = argv[1];
    const size_t cap = (size_t)atol(argv[2]);
    if (!smxtool::BgzfReader::is_bgzf(path)) { printf("notbgzf\n"); return 0; }
    smxtool::BgzfReader r;
    if (!r.open(path, (unsigned)atoi(argv[3]))) return 3;
    std::vector<char> buf(cap);
    FILE *o = fopen(argv[4], "wb");
    for (;;) {
        const size_t n = r.read(buf.data(), cap);
        if (n == smxtool::BgzfReader::kError) { printf("error\n"); return 0; }
        if (!n) break;
        fwrite(buf.data(), 1, n, o);
    }
    fclose(o);
    printf("ok\n");
    return 0;
}
"""

DRIVER_TEXT = r"""
#include "%s/spades_amd/tools/bgzf_reader.hpp"
#include <cstdio>
// argv: file, then (offset, length, threads) triples: prints the size of the text, then the bytes of every range to stdout
int main(int argc, char **argv) {
    smxtool::BgzfText t;
    if (!t.open(argv[1])) { printf("error\n"); return 0; }
    printf("%%llu\n", (unsigned long long)t.size());
    for (int i = 2; i + 2 < argc; i += 3) {
        const unsigned long long off = strtoull(argv[i], nullptr, 10);
        const size_t n = (size_t)strtoull(argv[i + 1], nullptr, 10);
        std::vector<char> buf(n + 1);
        if (!t.read(off, n, buf.data(), (unsigned)atoi(argv[i + 2]))) { printf("bad\n"); return 0; }
        fwrite(buf.data(), 1, n, stdout);
    }
    return 0;
}
"""


def bgzf_bytes(data, block=60000, level=6, empty_block_inside=False):
    """`data` as a BGZF file (SAM/BAM specification 4.1): gzip members of <= 64 KiB with the 'BC' extra subfield, an empty one at the end"""
    chunks = [data[i:i + block] for i in range(0, len(data), block)]
    if empty_block_inside and len(chunks) > 2:
        chunks.insert(1, b"")
    out = []
    for c in chunks + [b""]:
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        d = co.compress(c) + co.flush()
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(d) + 25) + d + struct.pack("<II", zlib.crc32(c) & 0xFFFFFFFF, len(c)))
    return b"".join(out)


def _fastq(rng, n):
    return "".join(f"@r{i}\n{''.join(rng.choice('ACGT') for _ in range(150))}\n+\n{'I' * 150}\n" for i in range(n)).encode()


def test_bgzf_reader(tmp_path):
    src, exe = tmp_path / "drv.cpp", str(tmp_path / "drv")
    src.write_text(DRIVER % ROOT)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, str(src), "-lz", "-pthread"])
    rng = random.Random(2)
    gz, out = str(tmp_path / "t.gz"), str(tmp_path / "t.out")

    def run(cap, nt):
        return subprocess.check_output([exe, gz, str(cap), str(nt), out]).decode().strip()

    for n, block, empty in ((10, 60000, False), (3000, 60000, True), (8000, 65280, False), (8000, 1000, False)):
        data = _fastq(rng, n)
        with open(gz, "wb") as f:
            f.write(bgzf_bytes(data, block, empty_block_inside=empty))
        assert gzip.open(gz).read() == data  # (a valid multi-member gzip file for everybody else)
        for cap, nt in ((65536, 1), (70000, 3), (1 << 20, 8), (1 << 24, 16)):
            assert run(cap, nt) == "ok" and open(out, "rb").read() == data, (n, block, cap, nt)
    whole = bytearray(open(gz, "rb").read())
    whole[len(whole) // 2] ^= 0x55
    open(gz, "wb").write(whole)
    assert run(1 << 20, 4) == "error"      # a damaged block: CRC32 / inflate
    open(gz, "wb").write(bytes(whole[:len(whole) // 3]))
    assert run(1 << 20, 4) == "error"      # the file ends inside a block
    open(gz, "wb").write(gzip.compress(_fastq(rng, 100)))
    assert run(1 << 20, 4) == "notbgzf"    # ordinary gzip


def test_bgzf_random_access_text(tmp_path):
    """BgzfText: the block index and ranges of the text (what the ranks of the multi-GPU hosts read of one *.fastq.gz)"""
    src, exe = tmp_path / "drv.cpp", str(tmp_path / "drv")
    src.write_text(DRIVER_TEXT % ROOT)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, str(src), "-lz", "-pthread"])
    rng = random.Random(4)
    gz = str(tmp_path / "t.gz")
    for n, block, empty in ((2000, 777, True), (2000, 65280, False), (3, 60000, False)):
        data = _fastq(rng, n)
        with open(gz, "wb") as f:
            f.write(bgzf_bytes(data, block, empty_block_inside=empty))
        ranges = [(0, len(data), 4), (0, 0, 1), (len(data) - 1, 1, 1), (5, 1, 1)]
        for _ in range(30):
            a = rng.randrange(len(data))
            ranges.append((a, rng.randrange(min(len(data) - a, 200000) + 1), rng.choice((1, 3, 8))))
        out = subprocess.check_output([exe, gz] + [str(x) for r in ranges for x in r])
        head, _, body = out.partition(b"\n")
        assert int(head) == len(data)
        assert body == b"".join(data[a:a + ln] for a, ln, _ in ranges)
    out = subprocess.check_output([exe, gz, str(len(data)), "1", "1"])  # a range behind the text
    assert out.endswith(b"bad\n")
    whole = bytearray(open(gz, "rb").read())
    open(gz, "wb").write(bytes(whole[:len(whole) - 40]))  # the last blocks cut
    assert subprocess.check_output([exe, gz]).startswith(b"error")
