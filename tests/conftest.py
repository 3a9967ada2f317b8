import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ultra_amd import hwq  # noqa: E402

hwq.export_hw_queues()  # before the first HIP call: one hardware queue per launch lane (ULTRA_HW_QUEUES, clamped)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def ctx():
    """One libultra_hip context for the whole GPU session (fails loudly without a gfx950 device)."""
    from ultra_amd import lib
    c = lib.Context(0)
    yield c
    c.close()
