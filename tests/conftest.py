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
