"""CPU: host-side pieces of the CLI clones that need no GPU — the dataset-YAML reader of spades_amd/tools/read_input.hpp (the subset of
YAML that spades.py writes for io::DataSet::load) compiled into a tiny driver."""
import os
import subprocess
import textwrap

from conftest import ROOT

DRIVER = r'''
#include "spades_amd/tools/read_input.hpp"
int main(int argc, char **argv) {
    std::vector<smxtool::DatasetLibrary> libs;
    if (!smxtool::load_dataset_yaml(argv[1], libs)) return 3;
    for (auto &l : libs) {
        printf("%s %d", l.type.c_str(), (int)l.graph_constructable());
        for (auto &f : l.files) printf(" %s", f.c_str());
        printf("\n");
    }
    return 0;
}
'''


def test_dataset_yaml_reader(tmp_path):
    src = tmp_path / "d.cpp"
    src.write_text(DRIVER)
    exe = str(tmp_path / "d")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", ROOT, str(src), "-lz", "-o", exe])
    y = tmp_path / "ds.yaml"
    y.write_text(textwrap.dedent('''\
        - orientation: "fr"
          type: "paired-end"
          right reads:
          - "/abs/r2.fastq.gz"
          - rel/r2b.fq
          left reads:
          - "/abs/r1.fastq.gz"   # comment
          - 'rel/r1b.fq'
          single reads: [ "/abs/s.fq", rel/s2.fq ]
        - type: "trusted-contigs"
          single reads:
          - "/abs/contigs.fa"
        - orientation: "fr"
          type: single
          interlaced reads:
            - "/abs/i.fq"
          merged reads:
            - /abs/m.fq
        '''))
    out = subprocess.check_output([exe, str(y)]).decode().splitlines()
    d = str(tmp_path)
    assert out == [f"paired-end 1 /abs/r2.fastq.gz {d}/rel/r2b.fq /abs/r1.fastq.gz {d}/rel/r1b.fq /abs/s.fq {d}/rel/s2.fq",
                   "trusted-contigs 0 /abs/contigs.fa",
                   "single 1 /abs/i.fq /abs/m.fq"]
    assert subprocess.call([exe, str(tmp_path / "missing.yaml")]) == 3


READER = r'''
#include "spades_amd/tools/read_input.hpp"
int main(int argc, char **argv) {  // file, chunk bytes: the file through read_plain in chunks, as submit_file fills its page-locked chunk
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 3;
    unsigned char head[2];
    if (fread(head, 1, 2, f) != 2) return 4;  // (submit_file looks at the first two bytes and rewinds)
    rewind(f);
    const size_t chunk = (size_t)atoll(argv[2]);
    std::vector<char> buf(chunk);
    for (;;) {
        size_t have = 0;
        bool eof = false;
        while (!eof && have < chunk) {
            const size_t got = smxtool::read_plain(f, buf.data() + have, chunk - have);
            have += got;
            if (got == 0) eof = true;
        }
        fwrite(buf.data(), 1, have, stdout);
        if (eof) break;
    }
    return 0;
}
'''


def test_plain_files_are_read_by_several_threads_without_losing_a_byte(tmp_path):
    """read_plain (spades_amd/tools/read_input.hpp): big reads of an uncompressed input are split over pread threads; SMX_IO_GRAIN makes a
    small file big. Chunk sizes that divide the file, that do not, and that exceed it; thread counts that do not divide the chunk."""
    import numpy as np
    src = tmp_path / "r.cpp"
    src.write_text(READER)
    exe = str(tmp_path / "r")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-I", ROOT, str(src), "-lz", "-o", exe])
    data = np.random.default_rng(1).integers(0, 256, size=1_000_003, dtype=np.uint8).tobytes()
    p = tmp_path / "in.bin"
    p.write_bytes(data)
    for grain, threads, chunk in ((1000, 5, 100_000), (1, 8, 333_333), (4096, 3, 2_000_000), (1 << 30, 4, 65536), (7, 2, 1_000_003)):
        env = dict(os.environ, SMX_IO_GRAIN=str(grain), SMX_IO_THREADS=str(threads))
        out = subprocess.check_output([exe, str(p), str(chunk)], env=env)
        assert out == data, (grain, threads, chunk)
