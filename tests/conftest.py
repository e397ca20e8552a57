import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
DEVTOOLS = os.path.join(ROOT, "tools", "dev")      # the developer scripts (gpu_*.py); four of them hold helpers the tests share
if DEVTOOLS not in sys.path:
    sys.path.insert(0, DEVTOOLS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The plain-C CPU oracle (oracle/liboracle.so), built on demand with gcc."""
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle")) if f.startswith("nhwo")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    from oracle.oraclepy import Oracle
    return Oracle(so)


@pytest.fixture(scope="session")
def ref():
    """The real reference encoder (oracle/_ref), if it was built in the container and travelled here."""
    so = os.path.join(ROOT, "oracle", "_ref", "libnhwref_enc.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    from oracle.harness import RefEncoder
    return RefEncoder(so)


@pytest.fixture(scope="session")
def manifest():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "manifest.json")) as f:
        return json.load(f)
