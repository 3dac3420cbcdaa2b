import os
import sys

import pytest

sys.dont_write_bytecode = True     # the needs_reference tests import files from /root/reference, which must stay untouched

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "needs_reference: needs /root/reference (authoring container only)")


def pytest_collection_modifyitems(config, items):
    ref = os.path.isdir("/root/reference/pointcept")
    skip_ref = pytest.mark.skip(reason="/root/reference not present (GPU box)")
    for item in items:
        if "needs_reference" in item.keywords and not ref:
            item.add_marker(skip_ref)


@pytest.fixture(scope="session")
def cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but torch.cuda.is_available() is False")
    return torch.device("cuda:0")
