import base64
import json
import os
import sys

import numpy as np
import pytest

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")    # as the library sets it when it is loaded before HIP starts (engine.cpp)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle_py import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """The compiled reference (oracle/_ref); tests that need it skip when it was not built."""
    from oracle.oracle_py import Ref, have_ref
    if not have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    return Ref()


@pytest.fixture(scope="session")
def zlib_():
    import zpaq_amd
    zpaq_amd.lib()
    return zpaq_amd


@pytest.fixture(scope="session")
def gpu(zlib_):
    """Initialised engine on device 0; GPU tests fail (not skip) if it is unavailable."""
    zlib_.init(0)
    return zlib_


def gen_input(entry):
    """Rebuild the input of a golden entry."""
    from zpaq_amd import corpus
    kind = entry.get("kind") or entry.get("gen")
    n, seed = entry["n"], entry["seed"]
    if kind == "mixed":
        parts = [corpus.block(k, n // 4 + 1, seed + i) for i, k in enumerate(["text", "lcg", "zeros", "records"])]
        return np.concatenate(parts)[:n]
    return corpus.block(kind, n, seed)


def b64(entry):
    return base64.b64decode(entry["archive_b64"])
