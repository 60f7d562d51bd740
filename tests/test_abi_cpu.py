"""CPU: the C-ABI library builds, loads and exports every symbol include/smx.h declares; host-side
logic that needs no GPU (error codes, parameter checks)."""
import os

import pytest

from spades_amd import _lib


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _lib.declared_symbols()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing


def test_error_codes_match_reference():
    # common/utils/logger/error_codes.hpp:14-20
    assert (_lib.INVALID_INPUT_FORMAT, _lib.INPUT_FILE_NOT_FOUND, _lib.IO_ERROR, _lib.INVALID_PARAMETER,
            _lib.MEMORY_LIMIT_EXCEEDED) == (64, 65, 66, 67, 68)


def test_rank_bucket_ownership_partitions_buckets():
    lib = _lib.load()
    for nb in (1, 16, 30, 160, 2560):
        for world in (1, 2, 3, 4, 8):
            firsts = [lib.smx_rank_first_bucket(nb, world, r) for r in range(world + 1)]
            assert firsts[0] == 0 and firsts[-1] == nb
            assert all(a <= b for a, b in zip(firsts, firsts[1:]))
            for b in range(nb):  # owner(b) = floor(b*world/nb) is the rank whose range holds b
                r = b * world // nb
                assert firsts[r] <= b < firsts[r + 1]


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from spades_amd import SmxError
    from spades_amd.kmercount import Context
    with pytest.raises(SmxError):
        Context()


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under spades_amd/ may import, link or execute it."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r"(^\s*(from|import)\s+oracle\b)|smx_oracle|libsmx_oracle|oracle/|orc_[a-z_]+\(", re.M)
    for dp, _, fs in os.walk(os.path.join(root, "spades_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                assert not pat.search(open(os.path.join(dp, f)).read()), (dp, f)


def test_product_never_touches_the_simt_emulator():
    """tests/simt_emu/ (the library's sources compiled by g++ against a fiber-based stand-in for the HIP runtime) is test infrastructure
    like the oracle: nothing under spades_amd/, no tool, bench.py and __graft_entry__.py never name it — the product has no CPU path."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r"simt_emu|libspades_emu|SMX_EMU|hip_emu|EMU_[A-Z_]+\(", re.M)
    files = [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]
    for top in ("spades_amd", "integration", "include"):
        for dp, _, fs in os.walk(os.path.join(root, top)):
            files += [os.path.join(dp, f) for f in fs if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h"))]
    for f in files:
        assert not pat.search(open(f).read()), f
