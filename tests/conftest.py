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


# Tests of code that was written after a round's GPU time was spent: no GPU run stands behind them yet, and the tier runs with -x.
# SMX_NEXT=1 (or SMX_SCALE_NEXT=1, the older name) takes them in — the first thing to run in the next round; drop the mark once green.
NEXT = bool(os.environ.get("SMX_NEXT") or os.environ.get("SMX_SCALE_NEXT"))
needs_next = pytest.mark.skipif(not NEXT, reason="no GPU run behind this code yet: SMX_NEXT=1 takes it in")


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
