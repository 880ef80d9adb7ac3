import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from __graft_entry__ import load_package  # noqa: E402

load_package()

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _oracle_threads():
    """The CPU oracle's OpenMP loops are fastest at <= 16 threads (tools/oracle_scaling.py: 0.33 s/image at 16,
    2.9 s at 128 on the 256-thread GPU host): never let it default to every hardware thread."""
    import os

    from oracle import oracle

    oracle.set_num_threads(min(16, os.cpu_count() or 1))
    yield
