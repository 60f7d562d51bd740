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
