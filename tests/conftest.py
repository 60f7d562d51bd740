import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """The GPU tier runs with -x: one failure hides everything collected after it (round 4: a red 2 M-read test kept 84 cheap tests from running on the
    driver's box). So the order is by cost — the files of small cases first, the tests at size (2 M – 20 M reads) after them, the process-spawning
    RCCL hosts last — and stable otherwise."""
    def weight(it):
        f = os.path.basename(str(it.fspath))
        if f == "test_zz_cli_rccl_gpu.py":
            return 3
        if f == "test_scale_gpu.py" or "larger_than_the_hbm_budget" in it.name:
            return 2
        if f == "test_integration_gpu.py" and "spades_core" in it.name:
            return 1
        return 0
    items.sort(key=weight)  # (list.sort is stable)


# Tests of code that was written after a round's GPU time was spent: no GPU run stands behind them yet, and the tier runs with -x.
# SMX_NEXT=1 (or SMX_SCALE_NEXT=1, the older name) takes them in — the first thing to run in the next round; drop the mark once green.
NEXT = bool(os.environ.get("SMX_NEXT") or os.environ.get("SMX_SCALE_NEXT"))
needs_next = pytest.mark.skipif(not NEXT, reason="no GPU run behind this code yet: SMX_NEXT=1 takes it in")


# SMX_EMU=1: the tests drive tests/simt_emu/_build/libspades_emu.so instead of the HIP library — the library's own sources (host code and
# gfx950 kernels as written) compiled by g++ against a fiber-based SIMT stand-in for the HIP runtime (tests/simt_emu/). A way to run kernel
# LOGIC where there is no GPU (slow: small cases only); it proves nothing about the hardware and the product never sees it — only this
# test-side switch points the ctypes loader somewhere else.
EMU = bool(os.environ.get("SMX_EMU"))
if EMU:
    sys.path.insert(0, os.path.join(ROOT, "tests", "simt_emu"))
    import build_emu  # noqa: E402
    import spades_amd._lib as _smx_lib  # noqa: E402
    _smx_lib.LIB_PATH = build_emu.build()

    def _host_view(p, n):
        """dist.py hands library-owned "device" memory to torch through __cuda_array_interface__; under the stand-in that memory is host
        memory: a numpy view of it (torch.as_tensor takes it without a copy)"""
        import ctypes
        import numpy as np
        return np.ctypeslib.as_array((ctypes.c_int64 * n).from_address(p))

    def _host_bytes(p, n):
        import ctypes
        import numpy as np
        return np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(p))

    import spades_amd.dist as _smx_dist  # noqa: E402
    _smx_dist._DevView = _host_view
    _smx_dist._DevBytes = _host_bytes


def free_port():
    """a port the kernel hands out (a rendezvous port derived from the pid can collide with a parallel test run or a socket in TIME_WAIT)"""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def load_manifest():
    return json.load(open(os.path.join(GOLDEN, "manifest.json")))


def read_lines(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return [l.rstrip("\n") for l in f.read().split("\n")[:-1]]


@pytest.fixture(scope="session")
def manifest():
    return load_manifest()
