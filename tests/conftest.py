import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def qlib():
    """The product library initialised on cuda:0 -- fails loudly when there is no device."""
    # torch first: libqemb_hip.so and torch both bring a HIP runtime (same SONAME).  When torch's is loaded and initialised first both
    # share it; the other way round torch.cuda reports no device afterwards and RCCL process groups cannot be created.
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    from quemb_amd import _lib

    return _lib.init(int(os.environ.get("LOCAL_RANK", "0")))
