import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "big: a gpu test that holds 16 GB of HBM or more")


@pytest.fixture(scope="session")
def ctx():
    """A library context on device 0. GPU tests fail loudly (never skip, never fall back) when the HIP
    library is missing or no device is visible."""
    from minarrow_amd.host import Context

    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o

    o.lib()
    return o
