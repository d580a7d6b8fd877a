import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    from rodygs_amd import _lib
    return _lib.lib()


@pytest.fixture(scope="session", autouse=True)
def oracle_threads():
    """The oracle's tensors are a few hundred KB each: beyond ~16 intra-op threads torch's fork / join costs more than it buys
    (1 M / 1080p full frame: 15 s on 16 threads of the 256-core GPU box, 100 s on all of them).  Host-side setting only."""
    import torch
    if torch.get_num_threads() > 16:
        torch.set_num_threads(16)
    yield
