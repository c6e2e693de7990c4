import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def _load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


def naf_cases():
    return _load("naf_cases.json")


def zstd_cases():
    return _load("zstd_cases.json")


def ref_cases():
    return _load("ref_cases.json")


def golden_bytes(*parts):
    with open(os.path.join(GOLDEN, *parts), "rb") as f:
        return f.read()
