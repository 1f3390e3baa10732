import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def hip():
    """The ctypes binding; on the GPU box also asserts a gfx950 device is present."""
    import torch
    from pydreamer_amd import hip as h
    h.lib()
    if torch.cuda.is_available():
        h.call('dm_device_check')
    return h
