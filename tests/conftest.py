"""pytest configuration: registers the `gpu` marker and makes the repo root importable.

-m "not gpu": oracle vs the reference's known answers, host logic, C-ABI symbol checks.
-m gpu      : parity tests proper — the HIP path through the C-ABI vs the oracle.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.build()
    return orc
